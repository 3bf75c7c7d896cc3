"""How the CPU oracle (bench.py's cpu_baseline, kind "port") scales with threads on this host: pairs/s of the
pair-parallel OpenMP run on a sample of the bench workload."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rgbdslam_v2_amd import synth
from oracle import pyoracle as po

F, N = 200, 1000
seq = synth.make_sequence(n_frames=F, n_kp=N)
pq, pt = synth.candidate_pairs(F, 20)
prm = po.default_params()
print("os.cpu_count", os.cpu_count(), "sched_getaffinity", len(os.sched_getaffinity(0)))
try:
    print("cgroup cpu.max:", open("/sys/fs/cgroup/cpu.max").read().strip())
except Exception as e:
    print("cgroup cpu.max: n/a", e)
for th in (1, 8, 16, 32, 64, 128, 256):
    n = min(len(pq), max(64, th * 6))
    t0 = time.perf_counter()
    po.match_pairs_mt(list(seq["desc"]), list(seq["xyz1"]), np.arange(F), pq[:n], pt[:n], prm, n_threads=th)
    dt = time.perf_counter() - t0
    print("threads %3d: %7.1f pairs/s (%d pairs, %.2f s)" % (th, n / dt, n, dt), flush=True)
