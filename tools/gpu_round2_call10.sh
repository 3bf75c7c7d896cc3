mkdir -p gpurun_out/r02k
(timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > gpurun_out/r02k/tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02k/tests.log)
tail -4 gpurun_out/r02k/tests.log
timeout 300 python tools/fuzz_pairs.py phased 100 40 2>&1 | tail -1
timeout 300 python tools/fuzz_pairs.py latency 100 40 2>&1 | tail -1
timeout 300 python tools/fuzz_pairs.py one_wave 100 20 2>&1 | tail -1
timeout 600 python bench.py --steps 20 > gpurun_out/r02k/bench.json 2> gpurun_out/r02k/bench.err; python -c "
import json; d=json.load(open('gpurun_out/r02k/bench.json')); print(d['value'], d['ms_per_step'], d['timing']['serial_stage_ms'], d['match_roofline']['frac'], d['sift']['value'], d['detect']['640x480_orb1000']['value'], d['cpu_baseline']['value'])"
timeout 120 python tools/bench_live_latency.py 2>/dev/null
