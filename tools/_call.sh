cd $GRAFT_REPO_ROOT
timeout 60 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
for lib in librgbdfe_v_prio0.so librgbdfe.so; do
  for noise in 0.01 0.002; do
  echo -n "$lib $noise: "
  RGBDFE_LIB=$GRAFT_REPO_ROOT/rgbdslam_v2_amd/$lib timeout 60 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --depth-noise $noise 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(json.dumps({'ms':d['ms_per_step'],'serial':d['timing']['serial_stage_ms']['select_ransac'],'parity':d['parity_check']['ok']}))"
  done
done
