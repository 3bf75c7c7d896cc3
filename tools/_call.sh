cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_orb.py -x -q -m gpu --timeout 150 2>&1 | grep -E "passed|failed|error" | tail -3
RGBDFE_ORB_BRIEF=pool timeout 300 python -m pytest tests/test_gpu_orb.py -x -q -m gpu --timeout 150 2>&1 | grep -E "passed|failed|error" | tail -3
cd /tmp && export TMPDIR=/tmp
for mode in pool patch; do
O=$GRAFT_REPO_ROOT/gpurun_out/r05/brief_$mode; rm -rf $O; mkdir -p $O
RGBDFE_ORB_BRIEF=$mode rocprofv3 --kernel-trace --stats --output-format csv -d $O -o trace -- python $GRAFT_REPO_ROOT/tools/detect_workload.py orb 640 480 1000 7 24 > $O/run.log 2>&1
find $O -name "*.db" -delete
echo "== RGBDFE_ORB_BRIEF=$mode"
python - <<P
import csv, glob
f = glob.glob("$O/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("  kernels total %.1f us per frame" % (tot / (7 * 24) / 1e3))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:8]:
    print("  %-40s %7.2f us/frame (%s calls)" % (r["Name"].replace("rgbdfe::", "").split("(")[0][:40], float(r["TotalDurationNs"]) / (7 * 24) / 1e3, r["Calls"]))
P
done
cd $GRAFT_REPO_ROOT; for mode in pool patch; do echo -n "$mode: "; RGBDFE_ORB_BRIEF=$mode timeout 100 python tools/bench_detect_batch.py 640 480 1000 112 5 2>&1 | tail -1 | cut -c1-200; done
