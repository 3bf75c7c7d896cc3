#!/bin/bash
# GPU box: kernel-trace stats + HBM counters for tools/bench_emm.py (environment measurement model).
set -u
TAG=${1:-r01_emm}
REPO=$PWD
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
BENCH="python $REPO/tools/bench_emm.py"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $BENCH > $OUT/bench_trace.json 2> $OUT/trace.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o fetch -- $BENCH > $OUT/bench_fetch.json 2> $OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o write -- $BENCH > $OUT/bench_write.json 2> $OUT/write.err
cd $REPO
python - <<PY
import csv, glob, json, collections
out = "$OUT"
res = collections.defaultdict(dict)
for f in glob.glob(out + "/trace/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        for k in ("emm_kernel", "create_cloud_kernel"):
            if k in r["Name"]:
                res[k]["calls"] = int(r["Calls"]); res[k]["avg_ns"] = float(r["AverageNs"]); res[k]["pct"] = float(r["Percentage"])
for pat in ("pmc_fetch", "pmc_write"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(out + "/" + pat + "/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            for k in ("emm_kernel", "create_cloud_kernel"):
                if k in r["Kernel_Name"]:
                    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, c in acc.items():
        for n, v in c.items():
            res[k][n + "_avg"] = sum(v) / len(v)
for k, d in res.items():
    if "FETCH_SIZE_avg" in d and "WRITE_SIZE_avg" in d:
        # MI355X_MICROARCH.md "HBM": FETCH_SIZE / WRITE_SIZE in KiB; gfx950 FETCH_SIZE counts half
        d["hbm_bytes_per_launch"] = 2 * d["FETCH_SIZE_avg"] * 1024 + d["WRITE_SIZE_avg"] * 1024
json.dump(res, open(out + "/summary.json", "w"), indent=1, sort_keys=True)
print(json.dumps(res, indent=1, sort_keys=True))
PY
find $OUT -name "*.db" -delete
find $OUT -name "*_agent_info.csv" -delete
