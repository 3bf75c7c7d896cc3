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
F, N = 200, 1000
seq = synth.make_sequence(n_frames=F, n_kp=N, depth_noise=noise)
pq, pt = synth.candidate_pairs(F, 20)
fe = FrontEnd(max_nodes=F, max_keypoints=1024, max_pairs_per_batch=4096)
for f in range(F):
    fe.upload_node(f, seq["desc"][f], seq["xyz1"][f])
buf = torch.zeros(len(pq) * RESULT_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
for i in range(4):
    fe.wait_ticket(fe.submit_pair_list(pq, pt, buf.data_ptr()), None)
    fe.synchronize()
PY
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O -o t -- python /tmp/serial_run.py > $O/run.log 2>&1)
python - <<'PY'
import csv, glob, os
f = glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/trace_serial/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# last batch = from the last hamming kernel on
idx = max(i for i, r in enumerate(rows) if "hamming" in r["Kernel_Name"] and "expand" not in r["Kernel_Name"])
t0 = int(rows[idx]["Start_Timestamp"])
for r in rows[idx:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%-60s start %8.1f us  dur %8.1f us  grid %s" % (r["Kernel_Name"][:60], (s - t0) / 1e3, (e - s) / 1e3, r.get("Grid_Size", "")))
PY
