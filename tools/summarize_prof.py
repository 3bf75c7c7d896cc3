"""Summarise rocprofv3 outputs of tools/profile_gpu.sh: per-kernel average duration (kernel-trace
stats) and per-launch HBM traffic from the FETCH_SIZE / WRITE_SIZE PMC passes.

HBM bytes follow MI355X_MICROARCH.md section "HBM": FETCH_SIZE / WRITE_SIZE are reported in KiB
units (x1024); on gfx950 FETCH_SIZE reads exactly 1/2 of the bytes of a wide coalesced read, so
the read side is doubled ("corrected"); other access widths are uncalibrated -- both the raw and
the corrected numbers are kept."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

out_dir = sys.argv[1]
summary = {}


def find(pattern):
    return sorted(glob.glob(os.path.join(out_dir, pattern), recursive=True))


def short(name):
    if ("replay_walk_kernel" in name or "pair_prep_kernel" in name or "ransac_hyp_kernel" in name or
            "ransac_refine_kernel" in name):  # parts of the select+RANSAC stage of a batch
        return "select_ransac"
    if "hamming_mfma_kernel" in name or "hamming_mfma_pipe_kernel" in name:
        return "hamming_nn"
    if "sift_top2_fast" in name or "sift_row_top2_kernel" in name or "sift_top2_onepass" in name:  # the dot-product stage (one or two passes)
        return "sift_dot"
    if "sift_finish_kernel" in name:
        return "sift_finish"
    if "sift_sort_kernel" in name:
        return "sift_sort"
    for k in ("hamming_nn_kernel", "select_ransac_kernel", "project_to_3d_kernel"):
        if k in name:
            return k.replace("_kernel", "")
    return None


# kernel-trace stats.  select_ransac has several template instances (one wave per pair / record / replay): a batch's
# RANSAC stage is the sequence of those launches, so their totals are summed and divided by the number of batches
# (= hamming_nn launches) -> "per_batch_ns".
for f in find("trace/**/*kernel_stats.csv"):
    for row in csv.DictReader(open(f)):
        k = short(row.get("Name", ""))
        if k:
            e = summary.setdefault(k, {})
            e["calls"] = e.get("calls", 0) + int(row["Calls"])
            e["total_ns"] = e.get("total_ns", 0.0) + float(row["TotalDurationNs"])
            e["pct"] = e.get("pct", 0.0) + float(row["Percentage"])
            e.setdefault("instances", {})[re.sub(r"\(.*", "", row["Name"].replace("(anonymous namespace)::", "")).replace("void rgbdfe::", "")] = {"calls": int(row["Calls"]), "avg_ns": float(row["AverageNs"])}
    print("kernel stats:", f)
for k, e in summary.items():
    if "calls" in e:
        e["avg_ns"] = e["total_ns"] / e["calls"]
# batches of the run: one Hamming launch per ORB batch, one sift_finish launch per SIFT batch
batches = summary.get("hamming_nn", {}).get("calls") or summary.get("sift_finish", {}).get("calls") or None
if batches:
    for k, e in summary.items():
        if "total_ns" in e:
            e["per_batch_ns"] = e["total_ns"] / batches
            e["launches_per_batch"] = e["calls"] / batches

# per-dispatch durations from the kernel trace (cross-check of the stats)
for f in find("trace/**/*kernel_trace.csv"):
    dur = defaultdict(list)
    for row in csv.DictReader(open(f)):
        k = short(row.get("Kernel_Name", ""))
        if k:
            dur[k].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    nb = len(dur.get("hamming_nn", [])) or len(dur.get("sift_finish", [])) or None
    for k, v in dur.items():
        summary.setdefault(k, {})["trace_avg_ns"] = sum(v) / len(v)
        summary[k]["trace_min_ns"] = min(v)
        summary[k]["trace_max_ns"] = max(v)
        if nb:
            summary[k]["trace_per_batch_ns"] = sum(v) / nb


def pmc(pattern):
    """counter -> per kernel: (sum over dispatches, dispatches, batches in the same pass)"""
    acc = defaultdict(lambda: defaultdict(list))
    for f in find(pattern):
        for row in csv.DictReader(open(f)):
            k = short(row.get("Kernel_Name", ""))
            if k:
                acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    return acc


for name, pat in (("fetch", "pmc_fetch/**/*counter_collection.csv"),
                  ("write", "pmc_write/**/*counter_collection.csv"),
                  ("sq", "pmc_sq/**/*counter_collection.csv"),
                  ("sq2", "pmc_sq2/**/*counter_collection.csv"),
                  ("mfma", "pmc_mfma/**/*counter_collection.csv")):
    acc = pmc(pat)
    for k, ctrs in acc.items():
        for c, vals in ctrs.items():
            nb = len(acc.get("hamming_nn", {}).get(c, [])) or len(acc.get("sift_finish", {}).get(c, [])) or len(vals)
            # per batch: a batch's RANSAC stage may be several dispatches
            summary.setdefault(k, {})[c + "_avg"] = sum(vals) / nb
            summary[k][c + "_n"] = len(vals)

for k, s in summary.items():
    if "FETCH_SIZE_avg" in s and "WRITE_SIZE_avg" in s:
        s["hbm_read_bytes_raw"] = s["FETCH_SIZE_avg"] * 1024
        s["hbm_write_bytes_raw"] = s["WRITE_SIZE_avg"] * 1024
        s["hbm_bytes_per_launch"] = 2 * s["hbm_read_bytes_raw"] + s["hbm_write_bytes_raw"]
        s["note"] = "hbm_bytes_per_launch = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (gfx950 FETCH_SIZE half-count correction)"

    if "SQ_ACTIVE_INST_VALU_avg" in s and "GRBM_GUI_ACTIVE_avg" in s:
        # SQ_ACTIVE_INST_VALU counts quad-cycles summed over the 1024 SIMDs (MI355X_MICROARCH.md: SQ_* count
        # quad-cycles), GRBM_GUI_ACTIVE is summed over the 8 XCDs: fraction of SIMD time with a VALU instruction active
        s["valu_busy_frac"] = s["SQ_ACTIVE_INST_VALU_avg"] * 4.0 / (s["GRBM_GUI_ACTIVE_avg"] / 8.0 * 1024.0)

print(json.dumps(summary, indent=1, sort_keys=True))
json.dump(summary, open(os.path.join(out_dir, "summary.json"), "w"), indent=1, sort_keys=True)
