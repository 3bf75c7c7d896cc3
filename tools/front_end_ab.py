"""GPU box: bench.py's front_end sub-record alone (detect + describe + resident nodes, then every node against its 20
predecessors): one JSON line.  For A/B runs of library variants or switches (RGBDFE_LIB, RGBDFE_MID_PLAN ...)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
r = bench.front_end_subrecord(0)
print(json.dumps({"value": r["value"], "parts": r["ms_per_frame_parts_last_run"], "parity": (r.get("parity_check") or {}).get("ok")}))
