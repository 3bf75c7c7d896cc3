mkdir -p gpurun_out/r02a
(timeout 1200 python -m pytest tests -m gpu -q -x --timeout 900 > gpurun_out/r02a/tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02a/tests.log)
tail -15 gpurun_out/r02a/tests.log
timeout 300 python tools/bench_hamming_modes.py > gpurun_out/r02a/hamming_modes.json 2> gpurun_out/r02a/hamming_modes.err; tail -3 gpurun_out/r02a/hamming_modes.err; cat gpurun_out/r02a/hamming_modes.json
timeout 600 python bench.py --steps 10 > gpurun_out/r02a/bench.json 2> gpurun_out/r02a/bench.err; tail -3 gpurun_out/r02a/bench.err; cat gpurun_out/r02a/bench.json
timeout 300 python bench.py --steps 10 --depth-noise 0.002 --no-extras --no-cpu-baseline > gpurun_out/r02a/bench_r1noise.json 2> gpurun_out/r02a/bench_r1noise.err; cat gpurun_out/r02a/bench_r1noise.json
timeout 300 python bench.py --steps 10 --depth-noise 0.002 --no-extras --no-cpu-baseline --hamming-mode 0 > gpurun_out/r02a/bench_r1noise_pop.json 2>&1; cat gpurun_out/r02a/bench_r1noise_pop.json
