"""<tag> profile directories -> profiles/<tag>_pmc_summary.json, the static counter figures bench.py quotes
(roofline.traffic, issue_roofline, the sub-records' traffic / kernel time).

    python tools/make_pmc_summary.py r03 [--from gpurun_out|profiles]

Input: <from>/prof_<tag>/<workload>/ (gpurun_out) or profiles/<tag>/<workload>/ as tools/profile_round.sh leaves them:
orb / heavy / sift carry a summary.json of tools/summarize_prof.py (counters per batch); the frame-level workloads
(detect_*, sift_extract_*) are summed here over ALL kernels of the run and divided by the frames the run processed.
HBM bytes = 2 x FETCH_SIZE x 1024 + WRITE_SIZE x 1024 (MI355X_MICROARCH.md "HBM": KiB units, gfx950 FETCH_SIZE counts
half of a wide coalesced read)."""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src = sys.argv[sys.argv.index("--from") + 1] if "--from" in sys.argv else "profiles"
base = os.path.join(ROOT, "gpurun_out", "prof_" + tag) if src == "gpurun_out" else os.path.join(ROOT, "profiles", tag)
ISSUE = {"hamming": 1.10, "select_ransac": 1.66}  # mean ns per wave-instruction per SIMD of each kernel's mix (r01_ubench)
out = {"collected_with": "tools/profile_round.sh %s: rocprofv3 --kernel-trace --stats, then separate --pmc passes "
                         "(FETCH_SIZE / WRITE_SIZE / SQ_* / MFMA), one directory per workload under profiles/%s/" % (tag, tag)}


def pairs_of(d):
    try:
        line = [l for l in open(os.path.join(d, "run_trace.json")) if l.startswith("{")][-1]
        return int(json.loads(line)["config"]["pairs_per_gpu_per_step"])
    except Exception:
        return 4000


def orb_section(d, noise):
    s = json.load(open(os.path.join(d, "summary.json")))
    h, r = s["hamming_nn"], s["select_ransac"]
    pairs = pairs_of(d)
    piped = any("hamming_mfma_pipe_kernel" in k for k in h.get("instances", {}))   # mode 3 (the default since round 4)
    return {
        "pairs_per_batch": pairs, "depth_noise": noise,
        "hamming": {
            "hamming_mode": 3 if piped else 1, "pairs_per_batch": pairs,
            "kernel": "hamming_mfma_pipe_kernel" if piped else "hamming_mfma_kernel",
            "valu_wave_instructions_per_batch": h.get("SQ_INSTS_VALU_avg"),
            "mfma_instructions_per_batch": h.get("SQ_INSTS_MFMA_avg"),
            "mfma_busy_cycles_per_batch": h.get("SQ_VALU_MFMA_BUSY_CYCLES_avg"),
            "lds_instructions_per_batch": h.get("SQ_INSTS_LDS_avg"),
            "lds_bank_conflict_cycles_per_batch": h.get("SQ_LDS_BANK_CONFLICT_avg"),
            "mean_issue_ns": ISSUE["hamming"], "valu_busy_frac": round(h.get("valu_busy_frac", 0.0), 4),
            "hbm_bytes_per_launch": h.get("hbm_bytes_per_launch"), "kernel_ns_in_profile": h.get("per_batch_ns")},
        "select_ransac": {
            "pairs_per_batch": pairs, "kernels": sorted(r.get("instances", {}).keys()),
            "valu_wave_instructions_per_batch": r.get("SQ_INSTS_VALU_avg"),
            "lds_instructions_per_batch": r.get("SQ_INSTS_LDS_avg"),
            "lds_bank_conflict_cycles_per_batch": r.get("SQ_LDS_BANK_CONFLICT_avg"),
            "mean_issue_ns": ISSUE["select_ransac"], "valu_busy_frac": round(r.get("valu_busy_frac", 0.0), 4),
            "hbm_bytes_per_launch": r.get("hbm_bytes_per_launch"), "stage_ns_in_profile": r.get("per_batch_ns"),
            "note": "per batch = summed over the stage's %d launches" % int(r.get("launches_per_batch", 0))},
    }


def sift_section(d):
    s = json.load(open(os.path.join(d, "summary.json")))
    pairs = pairs_of(d)
    sec = {"pairs_per_batch": pairs}
    for k in ("sift_dot", "sift_finish", "sift_sort", "select_ransac"):
        if k in s:
            e = s[k]
            sec[k] = {"hbm_bytes_per_launch": e.get("hbm_bytes_per_launch"), "stage_ns_in_profile": e.get("per_batch_ns"),
                      "mfma_instructions_per_batch": e.get("SQ_INSTS_MFMA_avg"),
                      "mfma_busy_cycles_per_batch": e.get("SQ_VALU_MFMA_BUSY_CYCLES_avg"),
                      "launches_per_batch": e.get("launches_per_batch")}
    return sec


def frame_section(d):
    """all kernels of the run summed, per frame"""
    meta = None
    for f in ("run_trace.json", "run_fetch.json", "run_write.json"):
        try:
            meta = json.loads([l for l in open(os.path.join(d, f)) if l.startswith("{")][-1])
            break
        except Exception:
            continue
    frames = float(meta["frames"])
    tot_ns, kernels = 0.0, {}
    for f in glob.glob(os.path.join(d, "trace", "**", "*kernel_stats.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            tot_ns += float(row["TotalDurationNs"])
            name = row["Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("rgbdfe::", "")
            kernels[name] = {"calls_per_frame": round(int(row["Calls"]) / frames, 2), "avg_ns": float(row["AverageNs"]),
                             "ns_per_frame": round(float(row["TotalDurationNs"]) / frames, 1)}
    ctr = {}
    for p in ("fetch", "write"):
        tot = 0.0
        for f in glob.glob(os.path.join(d, "pmc_" + p, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                tot += float(row["Counter_Value"])
        ctr[p] = tot
    return {"frames_in_profile": int(frames), "kernel_ns_per_frame": round(tot_ns / frames, 1),
            "hbm_read_bytes_raw_per_frame": round(ctr["fetch"] * 1024 / frames),
            "hbm_write_bytes_raw_per_frame": round(ctr["write"] * 1024 / frames),
            "hbm_bytes_per_frame": round((2 * ctr["fetch"] + ctr["write"]) * 1024 / frames),
            "kernels": dict(sorted(kernels.items(), key=lambda kv: -kv[1]["ns_per_frame"]))}


def commit_of(d):
    try:
        return open(os.path.join(d, "commit")).read().strip()
    except Exception:
        return "unknown"


def serial_section(d):
    """kernel trace with ONE batch in flight: per-kernel averages and the stage sums of the last traced batches"""
    kern, n_batches = {}, 0
    for f in glob.glob(os.path.join(d, "trace", "**", "*kernel_stats.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            name = row["Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("rgbdfe::", "")
            kern[name] = {"calls": int(row["Calls"]), "avg_ns": float(row["AverageNs"]), "total_ns": float(row["TotalDurationNs"])}
            if ("hamming_mfma" in name and "expand" not in name) or "sift_top2" in name or "sift_row_top2" in name:
                n_batches = int(row["Calls"])
    nb = max(n_batches, 1)
    stage = sum(v["total_ns"] for k, v in kern.items() if any(t in k for t in ("pair_prep", "ransac_hyp", "ransac_refine", "replay_walk", "select_ransac")))
    ham = sum(v["total_ns"] for k, v in kern.items() if "hamming_mfma" in k and "expand" not in k)
    dot = sum(v["total_ns"] for k, v in kern.items() if "sift_top2" in k or "sift_row_top2" in k)
    return {"batches_in_trace": n_batches, "hamming_ms_per_batch": round(ham / nb / 1e6, 4),
            "sift_dot_ms_per_batch": round(dot / nb / 1e6, 4),
            "select_ransac_stage_ms_per_batch": round(stage / nb / 1e6, 4),
            "note": "one batch in flight; kernel durations under the tracer are a few percent longer than HIP-event spans",
            "kernels": {k: {"calls_per_batch": round(v["calls"] / nb, 2), "avg_ns": v["avg_ns"]} for k, v in sorted(kern.items(), key=lambda kv: -kv[1]["total_ns"]) if v["calls"] >= nb}}


if os.path.isdir(os.path.join(base, "orb")):
    out["orb"] = orb_section(os.path.join(base, "orb"), 0.01)
if os.path.isdir(os.path.join(base, "heavy")):
    out["ransac_heavy"] = orb_section(os.path.join(base, "heavy"), 0.002)
if os.path.isdir(os.path.join(base, "sift")):
    out["sift"] = sift_section(os.path.join(base, "sift"))
for d in sorted(glob.glob(os.path.join(base, "detect_*"))):
    out.setdefault("detect", {})[os.path.basename(d)[len("detect_"):]] = frame_section(d)
for d in sorted(glob.glob(os.path.join(base, "sift_extract_*"))):
    out.setdefault("sift_extract", {})[os.path.basename(d)[len("sift_extract_"):]] = frame_section(d)
for w in ("orb_serial", "heavy_serial", "sift_serial"):
    if os.path.isdir(os.path.join(base, w)):
        out[w] = serial_section(os.path.join(base, w))
# every section says which tree it was collected on; a summary is built from ONE tree unless --allow-mixed
stamps = {}
for name, d in [("orb", "orb"), ("ransac_heavy", "heavy"), ("sift", "sift"), ("orb_serial", "orb_serial"), ("heavy_serial", "heavy_serial"), ("sift_serial", "sift_serial")]:
    if name in out:
        out[name]["commit"] = stamps[name] = commit_of(os.path.join(base, d))
for grp, prefix in (("detect", "detect_"), ("sift_extract", "sift_extract_")):
    for key in out.get(grp, {}):
        out[grp][key]["commit"] = stamps[grp + "." + key] = commit_of(os.path.join(base, prefix + key))
out["commits"] = sorted(set(stamps.values()))
if len(out["commits"]) > 1 and "--allow-mixed" not in sys.argv:
    sys.exit("sections were collected on different trees: %s (pass --allow-mixed to build the summary anyway)" % stamps)
dst = os.path.join(ROOT, "gpurun_out" if src == "gpurun_out" else "profiles", "%s_pmc_summary.json" % tag)
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out, indent=1)[:6000])
print("wrote", dst)
