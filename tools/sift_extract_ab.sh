#!/bin/bash
# gpurun: per-kernel time of the SIFT extraction batch path for the regular library and each librgbdfe_<tag>.so given
# (tools/build_variant.sh), plus the extraction parity tests on each.   tools/sift_extract_ab.sh [tag...]
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/sift_ab; mkdir -p $O
for T in base "$@"; do
  [ $T = base ] && unset RGBDFE_LIB || export RGBDFE_LIB=$R/rgbdslam_v2_amd/librgbdfe_$T.so
  python -m pytest tests/test_gpu_sift_extract.py -x -q 2>&1 | tail -1 | sed "s/^/$T tests: /"
  cd /tmp; rm -rf $O/$T
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$T -o trace -- python $R/tools/detect_workload.py sift_batch 640 480 0 32 4 > $O/$T.json 2> $O/$T.err
  cd $R
  python - $O/$T/trace_kernel_stats.csv $T <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
pick = [r for r in rows if "descriptor" in r["Name"] or "orientation" in r["Name"]]
print(sys.argv[2], "kernels us/frame: total %.1f" % (tot / 128e3), " ".join("%s %.2f" % (r["Name"].split("::")[-1].split("(")[0], float(r["TotalDurationNs"]) / 128e3) for r in pick))
PY
  find $O -name "*.db" -delete; find $O -name "*agent_info*" -delete; find $O -name "*kernel_trace.csv" -delete
done
