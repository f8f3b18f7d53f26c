#!/usr/bin/env python3
"""configs[4] (terrain humanoid, MuJoCo's prism scheme): capacity statistics over a rollout (diagnostic)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import humanoid_mujoco_amd as hb
m = hb.Model.load(os.path.join(ROOT, "humanoid_mujoco_amd", "assets", "humanoid27_hfield.hbm"))
n = 8192
b = hb.Batch(m, n, 0)
b.reset(perturb=True)
hist_c = np.zeros(64, int); hist_e = np.zeros(300, int)
for k in range(20):
    b.rollout_halton(50, 50 * k)
    nc, ne, ni = b.counts()
    hist_c += np.bincount(nc, minlength=64)[:64]; hist_e += np.bincount(ne, minlength=300)[:300]
    s = b.status()
    print("t %4d  ncon mean %.2f max %d  nefc mean %.1f max %d  flagged contactfull %d cnstrfull %d other %d" % (50 * (k + 1), nc.mean(), nc.max(), ne.mean(), ne.max(),
          int(((s & 2) != 0).sum()), int(((s & 4) != 0).sum()), int(((s & ~6) != 0).sum())))
print("nefc quantiles", [int(np.searchsorted(np.cumsum(hist_e) / hist_e.sum(), q)) for q in (0.5, 0.9, 0.99, 0.999)])
