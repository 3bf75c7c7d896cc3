mkdir -p gpurun_out/r02g
for W in 0.01 0.002; do
 for PH in 4 2 1; do
  for CH in 28 64 100 200; do
   RGBDFE_PHASES=$PH timeout 120 python bench.py --steps 8 --warmup 2 --no-extras --no-cpu-baseline --depth-noise $W --ransac-path record_replay --chunk-iterations $CH > gpurun_out/r02g/b_${W}_${PH}_${CH}.json 2>/dev/null
   python - <<PY
import json
d=json.load(open("gpurun_out/r02g/b_${W}_${PH}_${CH}.json"))
print("noise $W phases $PH chunk $CH: value %.0f ms_per_step %.3f serial ransac %.3f" % (d["value"], d["ms_per_step"], d["timing"]["serial_stage_ms"]["select_ransac"]))
PY
  done
 done
done
