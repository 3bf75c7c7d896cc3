#!/bin/bash
# GPU box (via gpurun): every counter figure bench.py quotes, one directory per workload under gpurun_out/prof_<tag>/:
#   orb      bench.py (configs[1], depth noise 0.01 z^2)      trace + FETCH/WRITE + SQ passes + MFMA pass
#   heavy    bench.py --depth-noise 0.002                     trace + FETCH/WRITE + SQ passes
#   sift     bench.py --config sift --frames 100              trace + FETCH/WRITE + MFMA pass
#   detect_640x480_orb1000, detect_1280x960_orb4000           trace + FETCH/WRITE   (tools/detect_workload.py orb)
#   sift_extract_640x480                                      trace + FETCH/WRITE   (tools/detect_workload.py sift_batch)
#   orb_serial, heavy_serial, sift_serial                     kernel trace with ONE batch in flight (tools/trace_serial.sh's
#                                                             run): the stage times bench.py's serial figures must agree with
# Counters are collected in their own runs (never together with a trace domain).  tools/make_pmc_summary.py <tag> turns the
# result into profiles/<tag>_pmc_summary.json.   Usage: tools/profile_round.sh <tag> [workloads...]
# Every workload directory gets a `commit` file = tools/_commit.txt (written on the build side right before the gpurun call:
# `git rev-parse HEAD`, "+dirty" when the tree differs): make_pmc_summary.py stamps each section with it and refuses to
# build a summary out of sections of different commits.
set -u
TAG=${1:-r03}
shift || true
WL=${*:-orb orb_serial heavy heavy_serial sift sift_serial detect_640x480_orb1000 detect_1280x960_orb4000 sift_extract_640x480}
REPO=$PWD
ROOT=$REPO/gpurun_out/prof_$TAG
export TMPDIR=/tmp
B="python $REPO/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras"
SQ1="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
SQ2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"
MF="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE"
run_passes() {  # <workload dir> <command> <passes...>
  local OUT=$ROOT/$1 CMD=$2; shift 2
  mkdir -p $OUT
  cp $REPO/tools/_commit.txt $OUT/commit 2>/dev/null || echo unknown > $OUT/commit
  cd /tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $CMD > $OUT/run_trace.json 2> $OUT/trace.err
  for P in "$@"; do
    case $P in
      fetch) C="FETCH_SIZE";; write) C="WRITE_SIZE";; sq) C=$SQ1;; sq2) C=$SQ2;; mfma) C=$MF;;
    esac
    rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$P -o $P -- $CMD > $OUT/run_$P.json 2> $OUT/$P.err
  done
  cd $REPO
}
for W in $WL; do
  case $W in
    orb)   run_passes orb "$B" fetch write sq sq2 mfma; python tools/summarize_prof.py $ROOT/orb > $ROOT/orb/summary.txt 2>&1;;
    orb_serial|heavy_serial|sift_serial)
      mkdir -p $ROOT/$W; cp $REPO/tools/_commit.txt $ROOT/$W/commit 2>/dev/null || echo unknown > $ROOT/$W/commit
      CONFIG=$([ $W = sift_serial ] && echo sift || echo orb) NOISE=$([ $W = heavy_serial ] && echo 0.002 || echo 0.01) GRAFT_REPO_ROOT=$REPO bash tools/trace_serial.sh > $ROOT/$W/last_batch.txt 2>&1
      rm -rf $ROOT/$W/trace; mkdir -p $ROOT/$W/trace; cp -r $REPO/gpurun_out/trace_serial/* $ROOT/$W/trace/ 2>/dev/null;;
    heavy) run_passes heavy "$B --depth-noise 0.002" fetch write sq sq2 mfma; python tools/summarize_prof.py $ROOT/heavy > $ROOT/heavy/summary.txt 2>&1;;
    sift)  run_passes sift "python $REPO/bench.py --config sift --frames 100 --steps 5 --warmup 1" fetch write mfma; python tools/summarize_prof.py $ROOT/sift > $ROOT/sift/summary.txt 2>&1;;
    detect_640x480_orb1000)  run_passes $W "python $REPO/tools/detect_workload.py orb 640 480 1000 112 2" fetch write;;
    detect_1280x960_orb4000) run_passes $W "python $REPO/tools/detect_workload.py orb 1280 960 4000 56 2" fetch write;;
    sift_extract_640x480)    run_passes $W "python $REPO/tools/detect_workload.py sift_batch 640 480 0 32 3" fetch write;;
  esac
done
python tools/make_pmc_summary.py $TAG --from gpurun_out > $ROOT/pmc_summary.txt 2>&1
tail -5 $ROOT/pmc_summary.txt
# keep the merge-back small: drop the databases and agent tables, keep stats, counters and summaries
find $ROOT -name "*.db" -delete
find $ROOT -name "*_agent_info.csv" -delete
du -sh $ROOT
