#!/usr/bin/env python3
"""BASELINE configs[0] shape (one env, 1000 mj_step calls) on the device, and throughput against the batch size: the
latency of one wave stepping one env, and how many envs it takes to fill the chip."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import humanoid_mujoco_amd as hb
HBM = os.path.join(ROOT, "humanoid_mujoco_amd", "assets", "humanoid27.hbm")
for solver, name in ((0, "PGS/50"), (2, "Newton/100")):
    m = hb.Model.load(HBM)
    if solver == 2:
        m.set_opt(solver=2, iterations=100)
    for N in (1, 256, 2048, 4096, 8192, 16384, 32768):
        b = hb.Batch(m, N, 0)
        b.reset(perturb=True)
        b.rollout_halton(200); b.sync()
        T = 1000
        t0 = time.perf_counter()
        b.rollout_halton(T, t0=200); b.sync()
        dt = time.perf_counter() - t0
        print("%-10s %6d envs x %d steps as one rollout launch: %8.1f us per step, %.3e env-steps/s" % (name, N, T, 1e6 * dt / T, N * T / dt), flush=True)
        b.close()
