"""Diagnostic: free-running two-lane vs full kernel, compare every step; report the first divergence."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import humanoid_mujoco_amd as hb
from oracle_lib import HUMANOID_HBM
m = hb.Model.load(HUMANOID_HBM)
n = 1000
pipelined = len(sys.argv) > 1 and sys.argv[1] == "p"
check_every = int(sys.argv[2]) if len(sys.argv) > 2 else 1
os.environ["HB_LANE_WINDOW"] = "8"
os.environ["HB_TWO_LANE"] = "0"; full = hb.Batch(m, n, 0)
os.environ["HB_TWO_LANE"] = "1"; two = hb.Batch(m, n, 0)
for b in (full, two):
    if pipelined: b.pipeline(True)
    b.reset(perturb=True)
    b.rollout(np.zeros((500, n, m.nu), np.float32))
bad = False
for t in range(400):
    full.rollout_halton(1, t0=t); two.rollout_halton(1, t0=t)
    if t % check_every == check_every - 1:
        a = full.get_state(hb.STATE_INTEGRATION); c = two.get_state(hb.STATE_INTEGRATION)
        if not np.array_equal(a, c):
            d = np.abs(a.astype(np.float64) - c.astype(np.float64)).max(1)
            envs = np.nonzero(d > 0)[0]
            lanes = two.lanes()
            nc, ne, ni = full.counts(); nc2, ne2, ni2 = two.counts()
            print("first divergence at step", t, ":", len(envs), "envs", envs[:20], "lanes", lanes[envs[:20]], "nefc full/two", ne[envs[:10]], ne2[envs[:10]],
                  "time full/two", a[envs[:5], 0], c[envs[:5], 0])
            bad = True
            break
print("pipelined", pipelined, "check_every", check_every, "->", "DIVERGED" if bad else "identical over 400 steps", "; slow now", int(two.lanes().sum()))
