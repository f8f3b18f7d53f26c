#!/usr/bin/env python3
"""Per-step time of hb_rollout_halton(T) for several T: separates per-launch fixed cost from per-step cost."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import humanoid_mujoco_amd as hb
m = hb.Model.load(os.path.join(ROOT, "humanoid_mujoco_amd", "assets", "humanoid27.hbm"))
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
b = hb.Batch(m, N, 0)
b.reset(perturb=True)
b.rollout_halton(300); b.sync()
for T in (1, 2, 4, 8, 16, 64):
    reps = max(4, 256 // T)
    b.sync(); b.timer_start()
    for r in range(reps):
        b.rollout_halton(T, t0=300 + r * T)
    ms = b.timer_stop()
    print("T=%3d  launches=%3d  per-launch %.1f us  per-step %.1f us  -> %.2e env-steps/s" % (T, reps, 1e3 * ms / reps, 1e3 * ms / (reps * T), N * reps * T / (ms * 1e-3)))
