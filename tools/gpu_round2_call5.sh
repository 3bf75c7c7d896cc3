mkdir -p gpurun_out/r02e
(timeout 1500 python -m pytest tests/test_gpu_pairs.py tests/test_gpu_sift.py tests/test_gpu_g2o.py tests/test_gpu_async.py tests/test_gpu_multi.py tests/test_gpu_flann.py -m gpu -q -x --timeout 900 > gpurun_out/r02e/tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02e/tests.log)
tail -15 gpurun_out/r02e/tests.log
timeout 300 python tools/fuzz_pairs.py latency 1 40 > gpurun_out/r02e/fuzz_latency.log 2>&1; tail -3 gpurun_out/r02e/fuzz_latency.log
timeout 300 python tools/fuzz_pairs.py phased 1 40 > gpurun_out/r02e/fuzz_phased.log 2>&1; tail -3 gpurun_out/r02e/fuzz_phased.log
timeout 300 python bench.py --steps 10 --no-extras --no-cpu-baseline > gpurun_out/r02e/bench_spec.json 2> gpurun_out/r02e/bench.err; cat gpurun_out/r02e/bench_spec.json | python -c "import sys,json; d=json.load(sys.stdin); print(d['value'], d['ms_per_step'], d['timing']['serial_stage_ms'], d['config']['mean_ransac_iterations'])"
timeout 300 python bench.py --steps 10 --depth-noise 0.002 --no-extras --no-cpu-baseline > gpurun_out/r02e/bench_r1noise.json 2>> gpurun_out/r02e/bench.err; cat gpurun_out/r02e/bench_r1noise.json | python -c "import sys,json; d=json.load(sys.stdin); print(d['value'], d['ms_per_step'], d['timing']['serial_stage_ms'], d['config']['mean_ransac_iterations'])"
timeout 300 python tools/bench_all_pairs.py 250 1000 > gpurun_out/r02e/allpairs_250.json 2>> gpurun_out/r02e/bench.err; cat gpurun_out/r02e/allpairs_250.json
timeout 120 python tools/bench_live_latency.py > gpurun_out/r02e/live.json 2>> gpurun_out/r02e/bench.err; cat gpurun_out/r02e/live.json
