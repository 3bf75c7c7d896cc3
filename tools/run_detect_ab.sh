# gpurun: ORB / detect parity tests, then rgbdfe_detect_describe_batch timing at both sizes (+ kernel trace)
mkdir -p gpurun_out/r03f; export TMPDIR=/tmp; R=$PWD
python -m pytest tests/test_gpu_orb.py tests/test_gpu_cpp_host.py -x -q > gpurun_out/r03f/tests.log 2>&1; tail -3 gpurun_out/r03f/tests.log
python tools/bench_detect_batch.py 640 480 1000 56 5 2>/dev/null | tee gpurun_out/r03f/b640_56.json
python tools/bench_detect_batch.py 640 480 1000 112 5 2>/dev/null | tee gpurun_out/r03f/b640_112.json
RGBDFE_DETECT_TIMING=1 python tools/bench_detect_batch.py 1280 960 4000 56 4 2> gpurun_out/r03f/b1280.err | tee gpurun_out/r03f/b1280_28.json
tail -2 gpurun_out/r03f/b1280.err | cut -c1-500
cd /tmp
for cfg in "640 480 1000 56" "1280 960 4000 28"; do set -- $cfg
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03f/$1x$2 -o trace -- python $R/tools/bench_detect_batch.py $1 $2 $3 $4 4 > $R/gpurun_out/r03f/$1x$2.json 2> $R/gpurun_out/r03f/$1x$2.err
done
cd $R; find gpurun_out/r03f -name "*.db" -delete; find gpurun_out/r03f -name "*agent_info*" -delete; find gpurun_out/r03f -name "*kernel_trace.csv" -delete
