mkdir -p gpurun_out/r02c
(timeout 1500 python -m pytest tests/test_gpu_flann.py tests/test_gpu_place.py tests/test_gpu_project3d.py tests/test_gpu_sift.py -m gpu -q --timeout 900 > gpurun_out/r02c/tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02c/tests.log)
tail -40 gpurun_out/r02c/tests.log
timeout 300 python tools/bench_prefilter.py 90 1000 12 > gpurun_out/r02c/prefilter_90_1000.json 2> gpurun_out/r02c/prefilter.err; tail -3 gpurun_out/r02c/prefilter.err; cat gpurun_out/r02c/prefilter_90_1000.json
timeout 300 python tools/bench_prefilter.py 250 1000 12 > gpurun_out/r02c/prefilter_250_1000.json 2>> gpurun_out/r02c/prefilter.err; cat gpurun_out/r02c/prefilter_250_1000.json
timeout 300 python tools/bench_prefilter.py 64 4000 12 > gpurun_out/r02c/prefilter_64_4000.json 2>> gpurun_out/r02c/prefilter.err; cat gpurun_out/r02c/prefilter_64_4000.json
