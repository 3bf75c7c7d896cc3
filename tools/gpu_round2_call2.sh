mkdir -p gpurun_out/r02b
(timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > gpurun_out/r02b/tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02b/tests.log)
tail -40 gpurun_out/r02b/tests.log
timeout 300 python tools/bench_prefilter.py 90 1000 12 > gpurun_out/r02b/prefilter_90_1000.json 2> gpurun_out/r02b/prefilter.err; tail -3 gpurun_out/r02b/prefilter.err; cat gpurun_out/r02b/prefilter_90_1000.json
timeout 300 python tools/bench_prefilter.py 250 1000 12 > gpurun_out/r02b/prefilter_250_1000.json 2>> gpurun_out/r02b/prefilter.err; cat gpurun_out/r02b/prefilter_250_1000.json
timeout 300 python tools/bench_prefilter.py 64 4000 12 > gpurun_out/r02b/prefilter_64_4000.json 2>> gpurun_out/r02b/prefilter.err; cat gpurun_out/r02b/prefilter_64_4000.json
