cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_orb.py -x -q -m gpu --timeout 150 2>&1 | grep -E "passed|failed|error" | tail -3
cd /tmp && export TMPDIR=/tmp
for lib in librgbdfe_v_pyr1.so librgbdfe_v_pyr2.so librgbdfe.so; do
O=$GRAFT_REPO_ROOT/gpurun_out/r05/pyr_$lib; rm -rf $O; mkdir -p $O
RGBDFE_LIB=$GRAFT_REPO_ROOT/rgbdslam_v2_amd/$lib rocprofv3 --kernel-trace --stats --output-format csv -d $O -o trace -- python $GRAFT_REPO_ROOT/tools/detect_workload.py orb 640 480 1000 7 24 > $O/run.log 2>&1
find $O -name "*.db" -delete
python - <<P
import csv, glob
f = glob.glob("$O/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
pyr = [float(r["TotalDurationNs"]) for r in rows if "orb_pyramid" in r["Name"]][0]
print("$lib: kernels total %.1f us per frame, orb_pyramid_kernel %.2f us per frame" % (tot / (7 * 24) / 1e3, pyr / (7 * 24) / 1e3))
P
done
