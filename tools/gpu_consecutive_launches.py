#!/usr/bin/env python3
"""Consecutive 250-step launches of a running batch (4096 envs of the benchmark's steady regime): us per step of each.  The multi-step
two-envs-per-wave kernel pairs its envs by the order of their average cost over the previous launch(es): this is that prediction at work
(HB_TUNE_SCHEDULE = 0: env order, i.e. random pairs)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import humanoid_mujoco_amd as hb
m = hb.Model.load(os.path.join(ROOT, "humanoid_mujoco_amd", "assets", "humanoid27.hbm"))
N = 4096
for sched in (1, 0):
    b = hb.Batch(m, N, 0)
    b.tune(schedule=sched)
    b.reset(perturb=True)
    b.rollout_halton(600)
    b.sync()
    ts = []
    for w in range(12):
        t0 = time.perf_counter(); b.rollout_halton(250, 600 + 250 * w); b.sync(); ts.append(1e6 * (time.perf_counter() - t0) / 250)
    print("schedule %d: %s   mean of the last eight %.2f" % (sched, " ".join("%.1f" % t for t in ts), sum(ts[4:]) / 8), flush=True)
    b.close()
