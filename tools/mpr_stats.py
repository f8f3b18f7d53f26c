"""Narrowphase work per step of the reference's own robot, counted in the oracle (CPU, single thread): MPR tests, support calls,
climb rounds and neighbour evaluations per test.  Guides the device narrowphase: the device runs one test per lane,
so the slowest test of an env sets that env's collision time."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle_lib import Oracle  # noqa: E402

path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "humanoid_mujoco_amd", "assets", "team_robot.hbm")
key = int(sys.argv[2]) if len(sys.argv) > 2 else 0
o = Oracle(path)
o.L.om_mpr_stats.argtypes = [ctypes.POINTER(ctypes.c_longlong), ctypes.c_int]
st = (ctypes.c_longlong * 8)()
o.reset(key)
nu = len(o.ctrl)
rows = []
for t in range(400):
    o.ctrl[:] = 0.3 * np.sin(0.05 * t + np.arange(nu))
    o.L.om_mpr_stats(st, 1)
    o.step()
    o.L.om_mpr_stats(st, 1)
    rows.append(list(st)[:5] + [o.ncon] + list(st)[5:8])
r = np.array(rows, dtype=float)
print("per step: tests %.1f (max %d), mesh support calls %.1f, climb rounds %.1f, neighbour evaluations %.1f, mpr_support calls %.1f, contacts %.2f"
      % (r[:, 0].mean(), r[:, 0].max(), r[:, 1].mean(), r[:, 2].mean(), r[:, 3].mean(), r[:, 4].mean(), r[:, 5].mean()))
print("per test: mpr_support calls %.1f, climb rounds per mesh support %.2f, neighbours per round %.1f"
      % (r[:, 4].sum() / r[:, 0].sum(), r[:, 2].sum() / r[:, 1].sum(), r[:, 3].sum() / r[:, 2].sum()))
print("longest test of a step: mean %.1f support calls, max %d; tests over 16 support calls per step %.2f; hits per step %.2f"
      % (r[:, 6].mean(), r[:, 6].max(), r[:, 7].mean(), r[:, 8].mean()))
print("histogram of the longest test per step:", np.bincount(r[:, 6].astype(int)))
