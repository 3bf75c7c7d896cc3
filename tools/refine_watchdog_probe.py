"""GPU box, diagnostics build of the refinement kernel (csrc: hipcc ... -DRGBDFE_SPLIT_WATCHDOG -c ransac_split.hip, linked with the
other objects into librgbdfe_wd.so; picked with RGBDFE_LIB; RGBDFE_RANSAC_SPLIT=1 so that small batches take the split path): the many-threads tests in a loop until the refinement kernel's
watchdog reports a spin loop that never ended.  python tools/refine_watchdog_probe.py <seconds> <group|single|both>"""
import ctypes as C
import faulthandler
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from rgbdslam_v2_amd import synth, _lib  # noqa: E402
import test_gpu_multi as tm  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
which = sys.argv[2] if len(sys.argv) > 2 else "group"
seq = synth.make_sequence(n_frames=14, n_kp=500, n_world=2000, seed=21)
pq, pt = synth.candidate_pairs(14, per_frame=7, seed=21)
data = (seq, pq, pt)
fns = {"group": [tm.test_one_group_handle_from_many_threads], "single": [tm.test_one_context_from_many_threads],
       "both": [tm.test_one_context_from_many_threads, tm.test_one_group_handle_from_many_threads]}[which]
L = C.CDLL(_lib.LIB_PATH)
NAMES = ["site (1 = queue lock, 2 = idle wait for a unit, 3 = SVD request, 4 = rounds)", "block", "wave", "grid", "n_units", "qlock", "svd lock",
         "no_more_units"] + ["buf%d.%s" % (b, f) for b in range(3) for f in ("state", "next", "done", "n_items")] + \
        ["req[0..31]", "req[32..63]", "unit_counter", "phase_index"] + ["wave%d active | held<<8 | where<<16 (10 in lock, 11 unlocked, 12 refill done, 20/21 loading, 30 scoring, 40 svd wait, 50 closing, 99 exited)" % w for w in range(8)] + ["n_pairs", "n_shares", "share_iters"]


def check(tag):
    buf = (C.c_uint * 64)()
    if L.rgbdfe_debug_watchdog(buf, 0) != 0:
        print("watchdog read failed"); return False
    if buf[0] == 0:
        return False
    print("WATCHDOG (%s):" % tag)
    for i, n in enumerate(NAMES):
        print("  %-32s %d (0x%x)" % (n[:60], buf[i], buf[i]))
    return True


t0 = time.time()
n = 0
while time.time() - t0 < budget:
    for fn in fns:
        faulthandler.dump_traceback_later(40, exit=True)
        try:
            fn(data)
        except BaseException as e:  # noqa: BLE001
            print("iteration %d raised %s" % (n, repr(e)[:300]), flush=True)
            check("after the exception")
            sys.exit(3)
        faulthandler.cancel_dump_traceback_later()
        n += 1
        if check("iteration %d" % n):
            sys.exit(4)
print("WD_PROBE_DONE %d iterations, no watchdog in %.0f s" % (n, time.time() - t0), flush=True)
