import json, os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import bench
r = bench.front_end_subrecord(0)
print(json.dumps({"value": r["value"], "parts": r["ms_per_frame_parts_last_run"], "parity": (r.get("parity_check") or {}).get("ok")}))
