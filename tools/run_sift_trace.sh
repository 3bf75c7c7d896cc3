# gpurun: SIFT extraction parity tests, timing of single calls vs the batch entry point, kernel trace of the batch path
mkdir -p gpurun_out/r03s; export TMPDIR=/tmp; R=$PWD
python -m pytest tests/test_gpu_sift_extract.py -x -q > gpurun_out/r03s/tests.log 2>&1; grep -n "passed\|failed" gpurun_out/r03s/tests.log | tail -2; grep -n "^E " gpurun_out/r03s/tests.log | head -8
python - <<'PY'
import time, numpy as np, sys
sys.path.insert(0, ".")
from rgbdslam_v2_amd import synth
from rgbdslam_v2_amd.frontend import FrontEnd
seq = synth.make_image_sequence(n_frames=8, seed=1)
fe = FrontEnd(max_nodes=4, max_keypoints=64, max_pairs_per_batch=8)
run = [seq["gray"][i] for i in synth.forth_and_back(32, 8)]
fe.sift_detect_batch(run[:8]); fe.sift_detect(run[0], None)
for name, fn in (("single", lambda: [fe.sift_detect(g, None) for g in run]), ("batch", lambda: fe.sift_detect_batch(run)), ("batch, reused outputs", lambda: fe.sift_detect_batch(run, copy=False))):
    ts = []
    for _ in range(4):
        t0 = time.perf_counter(); fn(); ts.append((time.perf_counter() - t0) / len(run) * 1e3)
    print(name, "ms per frame", [round(t, 4) for t in sorted(ts)])
fe.close()
PY
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03s/t -o trace -- python $R/tools/detect_workload.py sift_batch 640 480 0 32 4 > $R/gpurun_out/r03s/run.json 2> $R/gpurun_out/r03s/run.err
cd $R; find gpurun_out/r03s -name "*.db" -delete; find gpurun_out/r03s -name "*agent_info*" -delete; find gpurun_out/r03s -name "*kernel_trace.csv" -delete
cat gpurun_out/r03s/run.json
