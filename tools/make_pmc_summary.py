"""profiles/<tag>/summary.json (tools/summarize_prof.py: counters are already summed per batch) ->
profiles/r02_pmc_summary.json, the per-batch figures bench.py
reads for roofline.traffic and issue_roofline.  Usage: python tools/make_pmc_summary.py r02b [pairs_per_batch]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
d = json.load(open(os.path.join(ROOT, "profiles", tag, "summary.json")))
h, r = d["hamming_nn"], d["select_ransac"]
ISSUE = {"hamming": 1.10, "select_ransac": 1.66}  # mean ns per wave-instruction per SIMD of each kernel's mix (r01_ubench)
out = {
    "collected_with": "tools/profile_gpu.sh %s (rocprofv3 --kernel-trace --stats, then separate --pmc passes), bench.py "
                      "--steps 5 --warmup 1 --no-extras, configs[1] with depth noise 0.01 z^2" % tag,
    "workload": "configs[1]: %d pairs per batch, 1000 keypoints, 200 RANSAC iterations, depth noise 0.01 z^2" % pairs,
    "hamming": {
        "hamming_mode": 1, "pairs_per_batch": pairs, "kernel": "hamming_mfma_kernel<1,false>",
        "valu_wave_instructions_per_batch": h["SQ_INSTS_VALU_avg"],
        "mfma_instructions_per_batch": h["SQ_INSTS_MFMA_avg"],
        "mfma_busy_cycles_per_batch": h["SQ_VALU_MFMA_BUSY_CYCLES_avg"],
        "lds_instructions_per_batch": h["SQ_INSTS_LDS_avg"],
        "mean_issue_ns": ISSUE["hamming"], "valu_busy_frac": round(h["valu_busy_frac"], 4),
        "hbm_bytes_per_launch": h["hbm_bytes_per_launch"], "kernel_ns_in_profile": h["per_batch_ns"]},
    "select_ransac": {
        "pairs_per_batch": pairs,
        "kernels": "pair_prep_kernel + 4 x (select_ransac_kernel<1> + replay_walk_kernel) + select_ransac_kernel<2>",
        "valu_wave_instructions_per_batch": r["SQ_INSTS_VALU_avg"],
        "lds_instructions_per_batch": r["SQ_INSTS_LDS_avg"],
        "lds_bank_conflict_cycles_per_batch": r["SQ_LDS_BANK_CONFLICT_avg"],
        "mean_issue_ns": ISSUE["select_ransac"], "valu_busy_frac": round(r["valu_busy_frac"], 4),
        "hbm_bytes_per_launch": r["hbm_bytes_per_launch"],
        "stage_ns_in_profile": r["per_batch_ns"],
        "note": "per batch = summed over the stage's %d launches" % int(r["launches_per_batch"])},
}
json.dump(out, open(os.path.join(ROOT, "profiles", "r02_pmc_summary.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
