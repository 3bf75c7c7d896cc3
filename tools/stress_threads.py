"""GPU box: tests/test_gpu_multi.py's many-threads tests in a loop (VERDICT r3 #7d: a hipGraph capture of one thread's batch
was invalidated by another thread's device-wide synchronisation -- create / first-use allocations now zero-fill on the
context's own streams and wait for those only).  Besides the tests' own assertions the loop watches the graph counters:
captures another thread invalidated (`capture_failures`) must stay 0 while a second context is created, SIFT slabs and
keypoint slabs are first used and detectors are prepared on other threads.

    python tools/stress_threads.py [rounds=20]
"""
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from rgbdslam_v2_amd import synth  # noqa: E402
from rgbdslam_v2_amd.frontend import FrontEnd  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
seq = synth.make_sequence(n_frames=14, n_kp=500, n_world=2000, seed=21)
pq, pt = synth.candidate_pairs(14, per_frame=7, seed=21)
t0 = time.time()
fails = 0
# the two tests as pytest runs them, `rounds` processes one after the other (a fresh HIP runtime each time)
import subprocess  # noqa: E402
for r in range(rounds):
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_multi.py"), "-q", "-x", "-p",
                          "no:cacheprovider", "-k", "many_threads", "--timeout", "120"], cwd=ROOT, capture_output=True, text=True)
    if out.returncode != 0:
        fails += 1
        print("round %d failed:\n%s" % (r, out.stdout[-1500:]), flush=True)
    if r % 5 == 4:
        print("  %d rounds, %d failures, %.1f s" % (r + 1, fails, time.time() - t0), flush=True)
print("many-threads tests: %d rounds x 2 tests, %d failures, %.1f s" % (rounds, fails, time.time() - t0), flush=True)

# captures of one context while other threads create contexts / first-use allocations
main = FrontEnd(device_id=0, max_nodes=24, max_keypoints=512, max_pairs_per_batch=64)
for f in range(14):
    main.upload_node(f, seq["desc"][f], seq["xyz1"][f])
ref = main.match_pair_list(pq, pt).tobytes()
stop = threading.Event()
side_errors = []


def churn():
    img = synth.make_image_sequence(n_frames=1, seed=3)
    sd = synth.sift_descriptors_like(seq["desc"][:2], seed=1)
    try:
        while not stop.is_set():
            fe = FrontEnd(device_id=0, max_nodes=4, max_keypoints=512, max_pairs_per_batch=8)        # rgbdfe_create
            fe.upload_sift_node(0, sd[0], seq["xyz1"][0])                                            # first SIFT use: slabs
            fe.detector_configure(max_keypoints=300)
            m = np.where(img["mask"][0] > 0, 255, 0).astype(np.uint8)
            fe.detect_describe(img["gray"][0], m, img["depth"][0], img["fx"], img["fy"], img["cx"], img["cy"])  # workspace
            fe.close()                                                                               # rgbdfe_destroy
    except Exception as e:  # noqa: BLE001
        side_errors.append(repr(e))


print("captures beside context churn ...", flush=True)
th = [threading.Thread(target=churn) for _ in range(2)]
for t in th:
    t.start()
wrong = 0
n_batches = 0
t1 = time.time()
while time.time() - t1 < 15.0:
    n = 5 + n_batches % 40                      # a new batch shape (= a capture) most of the time
    if main.match_pair_list(pq[:n], pt[:n]).tobytes() != ref[: n * 1744]:
        wrong += 1
    n_batches += 1
stop.set()
for t in th:
    t.join()
st = main.graph_stats()
main.close()
print("captures beside context churn: %d batches, %d wrong, side errors %s, graph stats %s" % (n_batches, wrong, side_errors, st), flush=True)
ok = fails == 0 and wrong == 0 and not side_errors and st["capture_failures"] == 0
print("STRESS_THREADS_OK" if ok else "STRESS_THREADS_FAILED")
sys.exit(0 if ok else 1)
