# serial (non-overlapped) kernel trace of the spec-noise workload + the r1 workload: per-kernel durations
mkdir -p gpurun_out/r02f
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
for W in 0.01 0.002; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/r02f/trace_$W -o trace -- python $REPO/bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-extras --depth-noise $W > $REPO/gpurun_out/r02f/bench_$W.json 2> $REPO/gpurun_out/r02f/trace_$W.err
done
cd $REPO
for W in 0.01 0.002; do echo "== $W"; f=$(find gpurun_out/r02f/trace_$W -name "*kernel_stats.csv" | head -1); python3 -c "import csv,sys; [print(r[\"Name\"][:70].ljust(70), r[\"Calls\"], r[\"AverageNs\"], r[\"Percentage\"]) for r in csv.DictReader(open(sys.argv[1]))]" $f; done
find gpurun_out/r02f -name "*.db" -delete; find gpurun_out/r02f -name "*_agent_info.csv" -delete; find gpurun_out/r02f -name "*kernel_trace.csv" -size +20M -delete
du -sh gpurun_out/r02f
