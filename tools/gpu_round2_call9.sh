mkdir -p gpurun_out/r02i
(timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > gpurun_out/r02i/tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02i/tests.log)
tail -8 gpurun_out/r02i/tests.log
timeout 300 python tools/check_huge_batch.py 2>&1 | grep -v amdgpu
timeout 300 python tools/fuzz_pairs.py latency 50 40 2>&1 | tail -1
timeout 300 python tools/fuzz_sift.py 2>&1 | tail -2
timeout 300 python tools/stress_determinism.py 2>&1 | tail -3
