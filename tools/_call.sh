cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/ -q -m gpu --timeout 200 -x 2>&1 | grep -E "passed|failed|error|Timeout|Error" | tail -6
for lib in librgbdfe.so librgbdfe_v_l3.so; do
  echo -n "$lib: "
  RGBDFE_LIB=$GRAFT_REPO_ROOT/rgbdslam_v2_amd/$lib timeout 100 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(json.dumps({'value':d['value'],'ms':d['ms_per_step'],'serial':d['timing']['serial_stage_ms']['select_ransac'],'parity':d['parity_check']['ok']}))"
done
