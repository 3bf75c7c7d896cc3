#!/bin/bash
# GPU box: kernel-trace stats + MFMA counters for `bench.py --config sift` (configs[3]).
set -u
TAG=${1:-r01_sift}
REPO=$PWD
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
BENCH="python $REPO/bench.py --config sift --steps 5 --warmup 1"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $BENCH > $OUT/bench_trace.json 2> $OUT/trace.err
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma -o mfma -- $BENCH > $OUT/bench_mfma.json 2> $OUT/mfma.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o fetch -- $BENCH > $OUT/bench_fetch.json 2> $OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o write -- $BENCH > $OUT/bench_write.json 2> $OUT/write.err
cd $REPO
python - <<PY
import csv, glob, json, collections, re
out = "$OUT"
res = collections.defaultdict(dict)
def short(name):
    name = name.replace("rgbdfe::", "").replace("void ", "")
    return re.sub(r"\\(.*$", "", name).strip()
def wanted(k):
    return any(t in k for t in ("sift", "select_ransac", "replay_walk"))
for f in glob.glob(out + "/trace/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        k = short(r["Name"])
        if wanted(k):
            res[k]["calls"] = int(r["Calls"]); res[k]["avg_ns"] = float(r["AverageNs"]); res[k]["pct"] = float(r["Percentage"])
for pat in ("pmc_mfma", "pmc_fetch", "pmc_write"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(out + "/" + pat + "/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if wanted(k):
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, c in acc.items():
        for n, v in c.items():
            res[k][n + "_avg"] = sum(v) / len(v)
for k, d in res.items():
    if "SQ_VALU_MFMA_BUSY_CYCLES_avg" in d and "GRBM_GUI_ACTIVE_avg" in d:
        # MFMA busy cycles are summed over the 1024 SIMDs; GRBM_GUI_ACTIVE counts device cycles of the dispatch
        d["mfma_util_est"] = d["SQ_VALU_MFMA_BUSY_CYCLES_avg"] / (d["GRBM_GUI_ACTIVE_avg"] * 1024)
    if "GRBM_GUI_ACTIVE_avg" in d and "avg_ns" in d:
        d["clock_ghz_est"] = d["GRBM_GUI_ACTIVE_avg"] / d["avg_ns"]
json.dump(res, open(out + "/summary.json", "w"), indent=1, sort_keys=True)
print(json.dumps(res, indent=1, sort_keys=True))
PY
find $OUT -name "*.db" -delete
