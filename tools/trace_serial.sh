#!/bin/bash
# Runs on the GPU box: per-dispatch timeline of ONE batch in flight (bench workload), from rocprofv3 --kernel-trace.
# Usage: tools/trace_serial.sh [bench args]   -> prints kernel, start offset, duration (us) for the last batch
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/trace_serial; rm -rf $O; mkdir -p $O
cat > /tmp/serial_run.py <<'PY'
import sys, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import numpy as np, torch
from rgbdslam_v2_amd import synth
from rgbdslam_v2_amd.frontend import FrontEnd
from rgbdslam_v2_amd._lib import RESULT_DTYPE
noise = float(os.environ.get("NOISE", "0.01"))
sift = os.environ.get("CONFIG", "orb") == "sift"      # CONFIG=sift: bench.py's configs[3] sub-record (100 frames, 2000 pairs)
F, N = (100 if sift else 200), 1000
SEED = 20260923
seq = synth.make_sequence(n_frames=200, n_kp=N, seed=SEED, depth_noise=noise)
pq, pt = synth.candidate_pairs(F, 20, seed=SEED)
fe = FrontEnd(max_nodes=F, max_keypoints=1024, max_pairs_per_batch=4096, seed=SEED)
sd = synth.sift_descriptors_like(seq["desc"][:F], seed=SEED) if sift else None
for f in range(F):
    if sift:
        fe.upload_sift_node(f, sd[f], seq["xyz1"][f])
    else:
        fe.upload_node(f, seq["desc"][f], seq["xyz1"][f])
buf = torch.zeros(len(pq) * RESULT_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
for i in range(4):
    fe.wait_ticket((fe.submit_sift_pair_list if sift else fe.submit_pair_list)(pq, pt, buf.data_ptr()), None)
    fe.synchronize()
PY
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O -o t -- python /tmp/serial_run.py > $O/run.log 2>&1)
python - <<'PY'
import csv, glob, os
f = glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/trace_serial/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# last batch = from the last hamming kernel on
first = ("hamming", "sift_top2", "sift_row_top2")   # the matcher launch that opens a batch
idx = max(i for i, r in enumerate(rows) if any(k in r["Kernel_Name"] for k in first) and "expand" not in r["Kernel_Name"])
t0 = int(rows[idx]["Start_Timestamp"])
for r in rows[idx:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%-60s start %8.1f us  dur %8.1f us  grid %s" % (r["Kernel_Name"][:60], (s - t0) / 1e3, (e - s) / 1e3, r.get("Grid_Size", "")))
PY
