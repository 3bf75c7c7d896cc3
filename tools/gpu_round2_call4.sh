mkdir -p gpurun_out/r02d
(timeout 1500 python -m pytest tests/test_gpu_g2o.py tests/test_gpu_pairs.py -m gpu -q --timeout 900 > gpurun_out/r02d/tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02d/tests.log)
tail -40 gpurun_out/r02d/tests.log
