#!/usr/bin/env python3
"""Throughput of the reference's own robot (assets/team_robot.hbm: mesh hulls, condim 6, height-field floor, Newton, dt 0.002):
env-steps/s for the standing-reset and the standup-reset (lying) regimes, rollout launch and step API, plus the VecEnv loop."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import humanoid_mujoco_amd as hb
m = hb.Model.load(os.path.join(ROOT, "humanoid_mujoco_amd", "assets", "team_robot.hbm"))
N, T = 4096, 500
rng = np.random.default_rng(0)
for regime in ("standing reset", "standup reset (lying)"):
    b = hb.Batch(m, N, 0)
    b.reset(keyframe=0 if "lying" in regime else 1, perturb=True)
    ctrl_h = (0.3 * rng.uniform(-1, 1, (T, N, m.nu))).astype(np.float32)
    ctrl = b.dev_alloc(ctrl_h.nbytes)
    b.to_dev(ctrl, ctrl_h)
    b.rollout_dev(ctrl, 200); b.sync()
    t0 = time.perf_counter(); b.rollout_dev(ctrl, T); b.sync(); dt = time.perf_counter() - t0
    nc, ne, ni = b.counts()
    print("team robot, %s, rollout: %d envs x %d steps: %.1f us/step -> %.3e env-steps/s; mean ncon %.1f nefc %.1f (max %d) newton iterations %.2f; flagged %d"
          % (regime, N, T, 1e6 * dt / T, N * T / dt, nc.mean(), ne.mean(), ne.max(), ni.mean(), int((b.status() != 0).sum())))
    stride = N * m.nu * 4
    for pipe in (0, 1, 3, 4):
        b.pipeline(pipe)
        for t in range(20): b.step_dev(ctrl + t * stride)
        b.sync(); t0 = time.perf_counter()
        for t in range(T): b.step_dev(ctrl + t * stride)
        b.sync(); dt = time.perf_counter() - t0
        print("   step API%s: %.1f us/step -> %.3e env-steps/s" % ((" (pipelined, %d segments)" % (2 if pipe == 1 else pipe)) if pipe else "", 1e6 * dt / T, N * T / dt))
    b.pipeline(False)
    b.dev_free(ctrl); b.close()
env = hb.VecEnv(m, N, 0, team=True, realism=True, domain_randomization=True)
env.reset()
acts = (0.5 * rng.uniform(-1, 1, (50, N, m.nu))).astype(np.float32)
for t in range(10): env.step(acts[t])
t0 = time.perf_counter()
for t in range(200): env.step(acts[t % 50])
dt = time.perf_counter() - t0
print("VecEnv(team=True, realism, domain randomisation).step, host actions / observations every step: %.1f us/step -> %.3e env-steps/s" % (1e6 * dt / 200, N * 200 / dt))
# (hb_batch_pipeline on for this loop was measured too: 403 us per step against 382 - inside ONE step call the segments' kernels gain nothing on each other)
