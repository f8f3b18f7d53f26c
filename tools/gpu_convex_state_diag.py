#!/usr/bin/env python3
"""Device against oracle on ONE state of tests/test_gpu_convex.py::test_team_robot_one_step_parity_along_oracle_trajectories (argv[1]):
every contact of both sides."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import humanoid_mujoco_amd as hb
from oracle_lib import Oracle, load_state
import test_gpu_convex as T
K = int(sys.argv[1])
def init(o, e, rng):
    o.qpos[7:] += rng.uniform(-0.2, 0.2, o.nq - 7)
    if e % 2:
        o.qpos[0:3] = [0, 0, -0.6 + 0.1 * rng.uniform()]
        q = np.array([-0.5, -0.5, 0.5, 0.5]) + rng.uniform(-0.1, 0.1, 4)
        o.qpos[3:7] = q / np.linalg.norm(q)
    else:
        q = np.array([-0.7, 0, 0, 0.7]) + rng.uniform(-0.08, 0.08, 4)
        o.qpos[3:7] = q / np.linalg.norm(q)
states, ctrls = T._oracle_states(T.TEAM_HBM, envs=6, T=1200, every=40, seed=1, init=init)
calm = T._oracle_states(T.TEAM_HBM, envs=6, T=800, every=25, seed=2, init=init, ctrl_scale=0.15)
states += calm[0]; ctrls += calm[1]
m = hb.Model.load(T.TEAM_HBM)
o = Oracle(T.TEAM_HBM)
st = np.array(states).astype(np.float32).astype(np.float64)
ct = np.array(ctrls, dtype=np.float32).reshape(len(st), m.nu)
b = hb.Batch(m, len(st), 0)
b.diag_enable(True)
b.set_state(hb.STATE_INTEGRATION, st)
b.step(ct)
con = b.contacts().astype(np.float64)
nc, ne, ni = b.counts()
load_state(o, st[K], ct[K]); o.forward()
print("state %d: device ncon %d nefc %d, oracle ncon %d nefc %d" % (K, nc[K], ne[K], o.ncon, o.nefc))
for i in range(nc[K]):
    print("  dev %d: geoms %d-%d dim %d dist %.7f pos %s n %s" % (i, con[K, i, 14], con[K, i, 15], con[K, i, 13], con[K, i, 0], con[K, i, 1:4].round(6), con[K, i, 4:7].round(6)))
for i, c in enumerate(o.contacts()):
    print("  ora %d: geoms %d-%d dim %d dist %.7f pos %s n %s" % (i, c["geom1"], c["geom2"], c["dim"], c["dist"], c["pos"].round(6), c["frame"][0].round(6)))
