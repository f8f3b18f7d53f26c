#!/usr/bin/env python3
"""Find a bad-state event of the terrain humanoid on the device and replay the step before it through the oracle (diagnostic)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import humanoid_mujoco_amd as hb
from oracle_lib import Oracle
path = os.path.join(ROOT, "humanoid_mujoco_amd", "assets", "humanoid27_hfield.hbm")
m = hb.Model.load(path)
n = 4096
b = hb.Batch(m, n, 0)
b.diag_enable(True)
b.reset(perturb=True)
o = Oracle(path)
found = 0
for t in range(400):
    prev = b.get_state(hb.STATE_INTEGRATION, dtype=np.float64)
    b.rollout_halton(1, t)
    s = b.status()
    vmax = np.abs(b.qvel).max(axis=1)
    pmax = np.abs(prev[:, 1 + m.nq:1 + m.nq + m.nv]).max(axis=1)
    badenv = np.nonzero((vmax > 60) & (pmax < 40))[0]  # the step on which an env's velocities take off
    if len(badenv):
        for e in badenv[:3]:
            found += 1
            con = b.contacts()[e]; nc, ne, ni = b.counts()
            print("t %d env %d status %d: device ncon %d nefc %d iters %d qacc max %.3e" % (t, e, s[e], nc[e], ne[e], ni[e], np.abs(b.qacc()[e]).max()))
            o.reset(); o.L.om_data_set_time(o.d, prev[e, 0])
            o.qpos[:] = prev[e, 1:1 + m.nq]; o.qvel[:] = prev[e, 1 + m.nq:1 + m.nq + m.nv]; o.qacc_warmstart[:] = prev[e, 1 + m.nq + m.nv:]
            o.ctrl[:] = o.ctrl_env(t, e)
            o.forward()
            print("   oracle ncon %d nefc %d qacc max %.3e  |qvel| before %.2f; device |qvel| after %.2f" % (o.ncon, o.nefc, np.abs(o.qacc).max(), np.abs(o.qvel).max(), vmax[e]))
            fo = o.efc_force[:o.nefc]; fg = b.efc_force()[e][:ne[e]]
            print("   forces gpu max %.3e ora max %.3e" % (np.abs(fg).max() if len(fg) else 0, np.abs(fo).max() if len(fo) else 0))
            oc = o.contacts()
            for i in range(max(nc[e], o.ncon)):
                g = ("%.5f %s n %s g %d-%d" % (con[i, 0], con[i, 1:4].round(3), con[i, 4:7].round(3), con[i, 14], con[i, 15])) if i < nc[e] else "-"
                r = ("%.5f %s n %s g %d-%d" % (oc[i]["dist"], oc[i]["pos"].round(3), oc[i]["frame"][0].round(3), oc[i]["geom1"], oc[i]["geom2"])) if i < o.ncon else "-"
                print("     gpu", g, "| ora", r)
        mk = np.zeros(n, np.uint8); mk[badenv] = 1
        b.reset(mask=mk, perturb=True)
    if found >= 4: break
