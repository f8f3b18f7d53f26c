#!/usr/bin/env python3
"""Per-contact comparison device vs oracle on team-robot states (diagnostic for the MPR path)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import humanoid_mujoco_amd as hb
from oracle_lib import Oracle
from test_gpu_convex import TEAM_HBM, _oracle_states

def init(o, e, rng):
    o.qpos[7:] += rng.uniform(-0.2, 0.2, o.nq - 7)
    o.qpos[2] += 0.02 * e
states, ctrls = _oracle_states(TEAM_HBM, envs=6, T=1500, every=60, seed=1, init=init)
m = hb.Model.load(TEAM_HBM)
o = Oracle(TEAM_HBM)
n = len(states)
b = hb.Batch(m, n, 0)
b.diag_enable(True)
b.set_state(hb.STATE_INTEGRATION, np.array(states))
b.forward(np.array(ctrls, dtype=np.float32))
con = b.contacts().astype(np.float64)
a = b.qacc().astype(np.float64)
nc, ne, ni = b.counts()
tot = bad = 0
for k in range(n):
    o.reset()
    o.qpos[:] = states[k][1:1 + m.nq]; o.qvel[:] = states[k][1 + m.nq:1 + m.nq + m.nv]; o.qacc_warmstart[:] = states[k][1 + m.nq + m.nv:]
    o.ctrl[:] = ctrls[k]
    o.forward()
    for i, c in enumerate(o.contacts()):
        dn = np.abs(con[k, i, 4:7] - c["frame"][0]).max()
        dd = abs(con[k, i, 0] - c["dist"])
        tot += 1
        if dn > 1e-3 or dd > 1e-5:
            bad += 1
            print("state %3d con %d geoms %d-%d  dist gpu %.6f ora %.6f  n gpu %s ora %s  dpos %.1e" % (k, i, c["geom1"], c["geom2"], con[k, i, 0], c["dist"], con[k, i, 4:7].round(4), c["frame"][0].round(4), np.abs(con[k, i, 1:4] - c["pos"]).max()))
    dq = np.abs(a[k] - o.qacc).max() / max(1, np.abs(o.qacc).max())
    if o.ncon and dq > 1e-3: print("   state %d ncon %d nefc %d qacc rel diff %.2e iters gpu %d ora %d" % (k, o.ncon, o.nefc, dq, ni[k], o.dint("solver_niter")))
print("contacts %d, disagreeing %d" % (tot, bad))
