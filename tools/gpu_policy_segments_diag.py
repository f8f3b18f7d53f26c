#!/usr/bin/env python3
"""closed-loop policy rollout: is the result the same for every number of pipeline segments (and from run to run)?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import humanoid_mujoco_amd as hb
m = hb.Model.load(os.path.join(ROOT, "humanoid_mujoco_amd", "assets", "humanoid27.hbm"))
g = np.load(os.path.join(ROOT, "tests", "golden", "policy_mlp_seed0.npz"))
ws = [g["w%d" % i] for i in range(3)]; bs = [g["b%d" % i] for i in range(3)]
n, T = 4096, 40
a = hb.Batch(m, n, 0)
a.reset(perturb=True); a.rollout_halton(150)
st0 = a.get_state(hb.STATE_INTEGRATION)
ref = None
for segs in (0, 2, 2, 3, 3, 4, 1):
    b = hb.Batch(m, n, 0)
    b.set_policy_mlp(ws, bs)
    b.set_state(hb.STATE_INTEGRATION, st0)
    b.pipeline(segs)
    b.rollout_policy(T)
    f = b.get_state(hb.STATE_INTEGRATION)
    if segs == 2 and ref is None: ref = f
    if ref is not None:
        d = np.abs(f - ref).max(1)
        print("segments %d (%d): envs differing from the first 2-segment run: %d, max %.3g, worst envs %s" % (segs, b.segments, (d > 0).sum(), d.max(), np.argsort(-d)[:6]))
    else:
        f0 = f
    b.close()
d = np.abs(ref - f0)[:, 1:1 + m.nq].max(1)
print("2 segments vs launch chain: max %.3g median %.3g; quantiles 0.9 0.99 0.999: %s; envs above 1e-3: %d" % (d.max(), np.median(d), np.quantile(d, [0.9, 0.99, 0.999]), (d > 1e-3).sum()))
