#!/usr/bin/env python3
"""Soak of the reference's own robot through the env adapter (staged step, realism, domain randomisation, auto-resets): many episodes
of random motor commands; reports throughput, episode ends, warning bits, row / contact maxima, portal searches and deferred steps."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import humanoid_mujoco_amd as hb
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
T = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
m = hb.Model.load(os.path.join(ROOT, "humanoid_mujoco_amd", "assets", "team_robot.hbm"))
env = hb.VecEnv(m, N, 0, team=True, realism=True, domain_randomization=True)
env.reset()
rng = np.random.default_rng(0)
acts = rng.uniform(-1, 1, (64, N, m.nu)).astype(np.float32)
ends = 0
mx_c = mx_e = mx_s = 0
t0 = time.perf_counter()
for t in range(T):
    obs, rew, term, trunc, info = env.step(acts[t % 64])
    ends += int((term | trunc).sum())
    if t % 500 == 499:
        nc, ne, _ = env.batch.counts()
        nw, ns = env.batch.collision_counts()
        mx_c, mx_e, mx_s = max(mx_c, int(nc.max())), max(mx_e, int(ne.max())), max(mx_s, int(ns.max()))
        assert np.isfinite(obs).all() and np.isfinite(rew).all()
        print("step %6d: %.3e env-steps/s, %d episodes ended, warnings %s, max ncon %d nefc %d searches %d" % (
            t + 1, N * (t + 1) / (time.perf_counter() - t0), ends, env.warning_counts(), mx_c, mx_e, mx_s), flush=True)
print("done: %d env-steps, %d episodes" % (N * T, ends))
