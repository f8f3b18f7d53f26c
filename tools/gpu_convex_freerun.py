#!/usr/bin/env python3
"""Free-running device vs oracle on the team robot with identical random controls: where do they part? (diagnostic)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import humanoid_mujoco_amd as hb
from oracle_lib import Oracle
from test_gpu_convex import TEAM_HBM
m = hb.Model.load(TEAM_HBM)
n, T = 8, 1500
b = hb.Batch(m, n, 0)
b.reset(perturb=True)
st = b.get_state(hb.STATE_INTEGRATION, dtype=np.float64)
os_ = [Oracle(TEAM_HBM) for _ in range(n)]
for e, o in enumerate(os_):
    o.reset(); o.qpos[:] = st[e, 1:1 + m.nq]
rng = np.random.default_rng(5)
parted = [None] * n
for t in range(T):
    if t % 50 == 0: c = rng.uniform(-1, 1, (n, m.nu))
    b.step(c.astype(np.float32))
    q = b.qpos.astype(np.float64)
    nc, ne, ni = b.counts()
    for e, o in enumerate(os_):
        o.ctrl[:] = c[e]; o.step()
        d = np.abs(q[e] - o.qpos).max()
        if parted[e] is None and (d > 1e-3 or nc[e] != o.ncon):
            parted[e] = t
            print("env %d parts at step %d: |dq| %.2e ncon gpu %d ora %d nefc gpu %d ora %d root z gpu %.3f ora %.3f" % (e, t, d, nc[e], o.ncon, ne[e], o.nefc, q[e, 2], o.qpos[2]))
    if t % 250 == 249:
        print("t %d root z gpu %s" % (t, q[:, 2].round(3)), "ora", np.array([o.qpos[2] for o in os_]).round(3), "status", b.status())
