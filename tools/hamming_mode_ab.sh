# GPU box: the driver line with Hamming mode 1 and mode 3 alternating, three times each (one call: boxes differ by a few %)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for i in 1 2 3; do for m in 1 3; do
timeout 120 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --hamming-mode $m > gpurun_out/ab_m${m}_$i.json 2> gpurun_out/ab_m${m}_$i.err
done; done
python - <<'P'
import json,glob
for f in sorted(glob.glob("gpurun_out/ab_m*_*.json")):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f, d["value"], d["ms_per_step"], d["repeats"]["values"], d["match_roofline"]["avg_launch_ms"], d["roofline"]["avg_launch_ms"], d["roofline"]["step_ms_serial"])
    except Exception as e: print(f,"ERR",e)
P
