#!/usr/bin/env python3
"""hb_policy_kernel duration against the MLP shape (run under rocprofv3 --kernel-trace; durations in launch order)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import humanoid_mujoco_amd as hb
m = hb.Model.load(os.path.join(ROOT, "humanoid_mujoco_amd", "assets", "humanoid27.hbm"))
rng = np.random.default_rng(0)
for n_env in (4096, 256):
    b = hb.Batch(m, n_env, 0)
    b.reset(perturb=True)
    for hidden in ((), (32,), (256,), (256, 256), (256, 256, 256)):
        sizes = [m.nobs, *hidden, m.nu]
        ws = [rng.uniform(-0.1, 0.1, size=(a, c)).astype(np.float32) for a, c in zip(sizes[:-1], sizes[1:])]
        bs = [np.zeros(c, np.float32) for c in sizes[1:]]
        b.set_policy_mlp(ws, bs)
        for _ in range(5):
            b.policy_eval()
        print("n_env", n_env, "hidden", hidden, flush=True)
