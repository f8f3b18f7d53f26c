#!/usr/bin/env python3
"""BASELINE config 5 throughput: 8192 envs on the height-field terrain model, PGS exactly 50 sweeps."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import humanoid_mujoco_amd as hb
m = hb.Model.load(os.path.join(ROOT, "humanoid_mujoco_amd", "assets", "humanoid27_hfield.hbm"))
N, T = 8192, 300
b = hb.Batch(m, N, 0)
b.reset(perturb=True)
b.rollout_halton(200); b.sync()
ctrl = b.dev_alloc(T * N * m.nu * 4)
b.halton_ctrl_dev(T, 200, 0, ctrl)
b.sync(); t0 = time.perf_counter()
for t in range(T):
    b.step_dev(ctrl + t * N * m.nu * 4)
b.sync(); dt = time.perf_counter() - t0
nc, ne, ni = b.counts()
print("config 5 (step API): %d envs x %d steps: %.1f us/step -> %.3e env-steps/s; mean ncon %.1f nefc %.1f sweeps %.1f; warnings %d"
      % (N, T, 1e6 * dt / T, N * T / dt, nc.mean(), ne.mean(), ni.mean(), (b.status() != 0).sum()))
b.pipeline(True)
for t in range(20):
    b.step_dev(ctrl + t * N * m.nu * 4)
b.sync(); t0 = time.perf_counter()
for t in range(T):
    b.step_dev(ctrl + t * N * m.nu * 4)
b.sync(); dt = time.perf_counter() - t0
print("config 5 (step API, pipelined %d segments): %.1f us/step -> %.3e env-steps/s" % (b.segments, 1e6 * dt / T, N * T / dt))
b.pipeline(False)
b.sync(); t0 = time.perf_counter(); b.rollout_dev(ctrl, T); b.sync(); dt = time.perf_counter() - t0
print("config 5 (rollout) : %.1f us/step -> %.3e env-steps/s" % (1e6 * dt / T, N * T / dt))
