#!/bin/bash
# GPU box: SQ counters per dispatch of ONE bench-shaped batch in flight (the serial picture), last batch printed.
# Usage: NOISE=0.01 tools/pmc_serial.sh
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/pmc_serial; rm -rf $O; mkdir -p $O
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
for i in range(3):
    fe.wait_ticket(fe.submit_pair_list(pq, pt, buf.data_ptr()), None)
    fe.synchronize()
PY
SQ1="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
SQ2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"
(cd /tmp && rocprofv3 --pmc $SQ1 --output-format csv -d $O/sq -o sq -- python /tmp/serial_run.py > $O/run1.log 2>&1)
(cd /tmp && rocprofv3 --pmc $SQ2 --output-format csv -d $O/sq2 -o sq2 -- python /tmp/serial_run.py > $O/run2.log 2>&1)
python - <<'PY'
import csv, glob, os, collections
root = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc_serial"
def load(d):
    f = glob.glob(root + "/" + d + "/**/*counter_collection.csv", recursive=True)[0]
    disp = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        k = int(r["Dispatch_Id"])
        e = disp.setdefault(k, {"name": r["Kernel_Name"], "grid": int(r["Grid_Size"]), "vgpr": r["VGPR_Count"], "lds": r["LDS_Block_Size"],
                                "t": int(r["End_Timestamp"]) - int(r["Start_Timestamp"])})
        e[r["Counter_Name"]] = float(r["Counter_Value"])
    return disp
a, b = load("sq"), load("sq2")
keys = list(a.keys())
last = max(i for i, k in enumerate(keys) if "hamming_mfma" in a[k]["name"])
for k in keys[last:]:
    v = a[k]; w = b.get(k, {})
    n = v["name"].replace("rgbdfe::", "").replace("void ", "")[:28]
    gui = w.get("GRBM_GUI_ACTIVE", 0) / 8.0
    busy = 4 * w.get("SQ_ACTIVE_INST_VALU", 0) / (gui * 1024) if gui else 0
    occ = 4 * v.get("SQ_WAVE_CYCLES", 0) / (gui * 1024) if gui else 0
    print("%-28s grid %8d vgpr %3s lds %6s t %8.1f us VALU %7.2fM SALU %6.2fM LDS %6.2fM waves %6d | valu_busy %.2f waves/simd %.2f bankconf %.2fM wait_inst %.1fM" % (
        n, v["grid"], v["vgpr"], v["lds"], v["t"] / 1e3, v.get("SQ_INSTS_VALU", 0) / 1e6, v.get("SQ_INSTS_SALU", 0) / 1e6,
        v.get("SQ_INSTS_LDS", 0) / 1e6, int(v.get("SQ_WAVES", 0)), busy, occ, w.get("SQ_LDS_BANK_CONFLICT", 0) / 1e6, w.get("SQ_WAIT_INST_ANY", 0) / 1e6))
PY
