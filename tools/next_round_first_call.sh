#!/bin/bash
# Measurements prepared at the end of round 4 (the GPU budget was spent), meant as the FIRST gpurun call of the next round.
# CPU side first:   SIFT1_VARIANTS="base:-DRGBDFE_SIFT1_NV=11 d1:-DRGBDFE_SIFT1_DIAG=1 d2:-DRGBDFE_SIFT1_DIAG=2 d4:-DRGBDFE_SIFT1_DIAG=4" \
#                   bash tools/sweep_sift_onepass.sh build
# then:             gpurun --timeout 400 -- 'bash tools/next_round_first_call.sh'
#   1. what the one-pass SIFT kernel's MFMA + LDS + barrier stream costs without (part of) its digest (lead 2);
#   2. the clock the chip sustains in the pipelined fp4 Hamming kernel: GRBM_GUI_ACTIVE with kernel durations in the SAME pass
#      (DESIGN.md 4.1c quotes an estimate from two different runs);
#   3. instruction and LDS counts of orb_pyramid_kernel (instruction-issue bound by a static estimate only).
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/next
export SIFT1_NO_PARITY=1
SIFT1_VARIANTS="base:x d1:x d2:x d4:x" bash tools/sweep_sift_onepass.sh run > gpurun_out/next/sift_onepass_diag.log 2>&1; cat gpurun_out/next/sift_onepass_diag.log
cd /tmp && export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras"
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA --kernel-trace --stats --output-format csv \
  -d $GRAFT_REPO_ROOT/gpurun_out/next/hamming_clock -o clk -- $B > $GRAFT_REPO_ROOT/gpurun_out/next/hamming_clock.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --kernel-trace --stats --output-format csv \
  -d $GRAFT_REPO_ROOT/gpurun_out/next/pyramid_pmc -o pyr -- python $GRAFT_REPO_ROOT/tools/detect_workload.py orb 640 480 1000 7 8 \
  > $GRAFT_REPO_ROOT/gpurun_out/next/pyramid_pmc.log 2>&1
find $GRAFT_REPO_ROOT/gpurun_out/next -name "*.db" -delete
python - <<'P'
import csv, glob, os
root = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "next")
def rows(pat):
    out = []
    for f in glob.glob(os.path.join(root, pat), recursive=True):
        out += list(csv.DictReader(open(f)))
    return out
cc = rows("hamming_clock/**/*counter_collection.csv")
ham = [r for r in cc if "hamming_mfma_pipe" in r.get("Kernel_Name", "")]
by = {}
for r in ham:
    by.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
print("hamming_mfma_pipe_kernel counters (mean per launch):", {k: sum(v) / len(v) for k, v in by.items()})
st = [r for r in rows("hamming_clock/**/*kernel_stats.csv") if "hamming_mfma_pipe" in r["Name"]]
if st and "GRBM_GUI_ACTIVE" in by:
    ns = float(st[0]["AverageNs"]); g = sum(by["GRBM_GUI_ACTIVE"]) / len(by["GRBM_GUI_ACTIVE"])
    print("kernel %.1f us in this pass; GRBM_GUI_ACTIVE / 8 XCDs / time = %.2f GHz (if the counter is summed over XCDs)" % (ns / 1e3, g / 8 / ns))
pyr = [r for r in rows("pyramid_pmc/**/*counter_collection.csv") if "orb_pyramid" in r.get("Kernel_Name", "")]
by = {}
for r in pyr:
    by.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
print("orb_pyramid_kernel counters (mean per launch):", {k: sum(v) / len(v) for k, v in by.items()})
P
