# gpurun: ORB / detect parity tests, then rgbdfe_detect_describe_batch timing at both sizes for the replay modes
mkdir -p gpurun_out/r03f; export TMPDIR=/tmp; R=$PWD
python -m pytest tests/test_gpu_orb.py tests/test_gpu_cpp_host.py -x -q > gpurun_out/r03f/tests.log 2>&1; tail -3 gpurun_out/r03f/tests.log
for mode in 1 2 0; do
  RGBDFE_SUPER_PARALLEL_REPLAY=$mode python tools/bench_detect_batch.py 640 480 1000 112 7 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('mode $mode 640x480 ', d['ms_per_frame_all'])"
  RGBDFE_SUPER_PARALLEL_REPLAY=$mode python tools/bench_detect_batch.py 1280 960 4000 56 5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('mode $mode 1280x960', d['ms_per_frame_all'])"
done
RGBDFE_DETECT_TIMING=1 python tools/bench_detect_batch.py 640 480 1000 112 3 2>&1 | grep "super-frame timing" | tail -1 | cut -c1-520
