# GPU box: the pipelined fp4 Hamming kernel (mode 3) -- parity tests, then stage times of modes 1 and 3 at 1000 and 4000
# keypoints for the build variants of tools/build_hamming_pipe_variants.sh (RGBDFE_LIB selects the library file).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export RGBDFE_BENCH_HAMMING_MODES=1,3
timeout 170 python -m pytest tests/test_gpu_hamming.py -q -x -p no:cacheprovider --timeout 120 -k "pipelined or resident" > gpurun_out/hm_pipe_tests.log 2>&1; echo "tests rc $?"; tail -3 gpurun_out/hm_pipe_tests.log
timeout 90 python tools/bench_hamming_modes.py > gpurun_out/hm_pipe_base.json 2> gpurun_out/hm_pipe_base.err; echo "base rc $?"
for v in $(ls rgbdslam_v2_amd/librgbdfe_hp_*.so 2>/dev/null); do
  n=$(basename $v .so)
  RGBDFE_LIB=$v RGBDFE_BENCH_HAMMING_MODES=3 timeout 90 python tools/bench_hamming_modes.py > gpurun_out/hm_pipe_$n.json 2> gpurun_out/hm_pipe_$n.err; echo "$n rc $?"
done
python - <<'P'
import json, glob
for f in sorted(glob.glob("gpurun_out/hm_pipe_*.json")):
    try:
        d = json.load(open(f))
    except Exception as e:
        print(f, "unreadable", e); continue
    print(f, {k: (v["hamming_ms"], v["bytes_equal_mode0"]) for k, v in d.items()})
P
