#!/usr/bin/env python3
"""What the physics alone costs in the closed loop's regime: the humanoids driven by the configs[3] policy for 300 steps, then the
pipelined step API with the policy's last controls frozen (no policy kernel), against the closed loop itself."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import humanoid_mujoco_amd as hb
hip = ctypes.CDLL("libamdhip64.so")
m = hb.Model.load(os.path.join(ROOT, "humanoid_mujoco_amd", "assets", "humanoid27.hbm"))
g = np.load(os.path.join(ROOT, "tests", "golden", "policy_mlp_seed0.npz"))
ws = [g["w%d" % i] for i in range(3)]; bs = [g["b%d" % i] for i in range(3)]
N = 4096
b = hb.Batch(m, N, 0)
b.set_policy_mlp(ws, bs)
b.reset(perturb=True)
b.pipeline(True)
b.rollout_policy(300); b.sync()
t0 = time.perf_counter(); b.rollout_policy(200); b.sync(); dt = time.perf_counter() - t0
nc, ne, ni = b.counts()
print("closed loop, %d segments: %.1f us/step; mean ncon %.1f nefc %.1f sweeps %.1f" % (b.segments, 1e6 * dt / 200, nc.mean(), ne.mean(), ni.mean()))
ctrl = np.ascontiguousarray(b.policy_eval(), dtype=np.float32)
d = b.dev_alloc(ctrl.nbytes)
assert hip.hipMemcpy(ctypes.c_void_p(d), ctrl.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(ctrl.nbytes), 1) == 0
for K in (20, 50, 100):
    b.sync(); t0 = time.perf_counter()
    for _ in range(K): b.step_dev(d)
    b.sync(); dt = time.perf_counter() - t0
    nc, ne, ni = b.counts()
    print("physics only, frozen controls, next %3d steps: %.1f us/step; mean ncon %.1f nefc %.1f sweeps %.1f" % (K, 1e6 * dt / K, nc.mean(), ne.mean(), ni.mean()))
