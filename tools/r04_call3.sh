#!/bin/bash
# GPU box: the one-pass SIFT matcher -- parity tests, A/B timing against the two-pass form -- the thread stress, the end-to-end report
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_sift.py tests/test_gpu_flann.py tests/test_gpu_sift_e2e.py tests/test_gpu_bench_parity.py tests/test_gpu_multi.py -m gpu -q -p no:cacheprovider --timeout 300 -k "sift or flann or e2e" > gpurun_out/r04_gputests3.log 2>&1; echo "tests rc $?"; tail -5 gpurun_out/r04_gputests3.log
cat > /tmp/sift_ab.py <<'PY'
import json, sys, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import bench
seq, _, _ = bench.orb_workload(1)
r = bench.sift_subrecord(seq, 0)
print(json.dumps({k: r[k] for k in ("value", "ms_per_step", "serial_stage_ms", "roofline", "parity_check")}))
PY
for OP in 1 0; do echo "RGBDFE_SIFT_ONEPASS=$OP"; RGBDFE_SIFT_ONEPASS=$OP timeout 300 python /tmp/sift_ab.py 2>&1 | tail -1; done > gpurun_out/r04_sift_ab.log 2>&1; cat gpurun_out/r04_sift_ab.log | cut -c1-600
timeout 400 python tools/stress_threads.py 15 > gpurun_out/r04_stress_threads.log 2>&1; echo "stress rc $?"; tail -6 gpurun_out/r04_stress_threads.log
timeout 300 python tools/sift_e2e.py > gpurun_out/r04_sift_e2e.json 2> gpurun_out/r04_sift_e2e.err; echo "e2e rc $?"
