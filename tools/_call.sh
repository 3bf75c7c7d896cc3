cd $GRAFT_REPO_ROOT
timeout 200 python -m pytest tests/test_gpu_async.py -x -q -m gpu --timeout 100 2>&1 | grep -E "passed|failed|Error|error" | tail -5
timeout 200 python - <<'P'
import json, bench
seq, pq, pt = bench.orb_workload(1)
r = bench.host_io_subrecord(seq, pq, pt, 0, True)
print(json.dumps({k: r[k] for k in ("value", "ms_per_call", "pipelined")}))
P
