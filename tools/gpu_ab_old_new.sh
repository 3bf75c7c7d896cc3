for rep in 1 2; do
for L in old new; do
  if [ $L = new ]; then unset RGBDFE_LIB; else export RGBDFE_LIB=$PWD/rgbdslam_v2_amd/librgbdfe_$L.so; fi
  for W in 0.002 0.005 0.01; do
    python bench.py --steps 8 --warmup 2 --no-extras --no-cpu-baseline --depth-noise $W 2>/dev/null | python -c "
import sys,json; d=json.load(sys.stdin); print('$L noise $W: value %.0f step %.3f serial ransac %.3f match %.3f' % (d['value'], d['ms_per_step'], d['timing']['serial_stage_ms']['select_ransac'], d['timing']['serial_stage_ms']['match']))"
  done
done
done
