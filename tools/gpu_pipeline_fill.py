#!/usr/bin/env python3
"""Pipelined step API: wall time of K step calls between two host synchronisations, K = 1 .. 80 (what a short timed window pays for
filling and draining the pipeline).  Steady regime (600-step pre-roll), 4096 envs."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import humanoid_mujoco_amd as hb
m = hb.Model.load(os.path.join(ROOT, "humanoid_mujoco_amd", "assets", "humanoid27.hbm"))
N = 4096
segs = int(sys.argv[1]) if len(sys.argv) > 1 else 1
b = hb.Batch(m, N, 0)
T = 4000
ctrl = b.dev_alloc(T * N * m.nu * 4)
b.halton_ctrl_dev(T, 600, 0, ctrl)
b.reset(perturb=True); b.rollout_halton(600, 0, 0); b.sync()
b.pipeline(segs)
b.tune(fold=1)  # (this is about one launch per call)
stride = N * m.nu * 4
t = 0
for _ in range(50): b.step_dev(ctrl + (t % T) * stride); t += 1
b.sync()
print("segments in use: %d" % b.segments)
for K in (1, 2, 3, 5, 10, 20, 40, 80):
    best = 1e9
    for rep in range(7):
        b.sync()
        t0 = time.perf_counter()
        for _ in range(K): b.step_dev(ctrl + (t % T) * stride); t += 1
        b.sync()
        best = min(best, time.perf_counter() - t0)
    print("K = %3d: %7.1f us in all, %6.1f us per step" % (K, 1e6 * best, 1e6 * best / K), flush=True)
