# per-dispatch durations of one workload's serial steps (kernel trace), in launch order
mkdir -p gpurun_out/r02j
export TMPDIR=/tmp
REPO=$PWD
W=${1:-0.01}
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $REPO/gpurun_out/r02j/trace_$W -o trace -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --depth-noise $W > $REPO/gpurun_out/r02j/bench_$W.json 2> $REPO/gpurun_out/r02j/trace_$W.err
cd $REPO
python3 - <<PY
import csv, glob
f = glob.glob("gpurun_out/r02j/trace_$W/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
sel = [r for r in rows if "rgbdfe" in r["Kernel_Name"] and "expand" not in r["Kernel_Name"]]
for r in sel[-14:]:
    nm = r["Kernel_Name"].replace("rgbdfe::", "").replace("(anonymous namespace)::", "")[:48]
    print(nm.ljust(50), "grid", r.get("Grid_Size_X", r.get("Grid_Size", "?")), "dur_us %.1f" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
PY
find gpurun_out/r02j -name "*.db" -delete; find gpurun_out/r02j -name "*_agent_info.csv" -delete; find gpurun_out/r02j -name "*kernel_trace.csv" -delete
