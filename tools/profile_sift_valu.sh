set -u
REPO=$PWD; OUT=$REPO/gpurun_out/prof_sift2; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/pmc -o p -- python $REPO/bench.py --config sift --steps 3 --warmup 1 > $OUT/b.json 2> $OUT/err.txt
cd $REPO
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/pmc/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        for k in ("sift_row_top2_kernel<false>", "sift_row_top2_kernel<true>"):
            if k in r["Kernel_Name"]:
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in acc.items():
    d = {n: sum(v)/len(v) for n, v in c.items()}
    simd_cycles = d["GRBM_GUI_ACTIVE"] / 8 * 1024
    print(k, {n: "%.3g" % v for n, v in d.items()})
    print("  valu_busy %.3f  any_busy %.3f  lds_busy %.3f  wave_cycles/simd_cycles %.2f" % (d["SQ_ACTIVE_INST_VALU"]*4/simd_cycles, d["SQ_ACTIVE_INST_ANY"]*4/simd_cycles, d.get("SQ_ACTIVE_INST_LDS",0)*4/simd_cycles, d["SQ_WAVE_CYCLES"]*4/simd_cycles))
PY
