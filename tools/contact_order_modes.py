#!/usr/bin/env python3
"""Body-pair-major against geom-major contact order (include/hb.h: hb_model_pair_order) on the 128 golden states: how many states emit
their contacts in another order, and what that does to one step of PGS / 50 (the benchmark configuration) and of Newton (converged).
CPU only (the fp64 oracle on the two orderings of the same model).  Output: profiles/r04_contact_order_modes.txt"""
import os, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import humanoid_mujoco_amd as hb
from oracle_lib import GOLDEN, HUMANOID_HBM, Oracle

g = dict(np.load(os.path.join(GOLDEN, "humanoid27_steps.npz")))
# ... and on the benchmark's steady regime: fallen humanoids (every 4th step of steps 300 .. 1000 of eight envs of the Halton workload)
if "--steady" in sys.argv:
    oo = Oracle()
    rec = {k: [] for k in ("env", "qpos", "qvel", "warm", "ctrl")}
    for e in range(8):
        oo.init_env(e)
        for t in range(1000):
            c = oo.ctrl_env(t, e)
            oo.ctrl[:] = c
            if t >= 300 and t % 4 == 0:
                rec["env"].append(e); rec["qpos"].append(oo.qpos.copy()); rec["qvel"].append(oo.qvel.copy()); rec["warm"].append(oo.qacc_warmstart.copy()); rec["ctrl"].append(c.copy())
            oo.step()
    g = {k: np.array(v) for k, v in rec.items()}
    print("steady regime: %d states" % len(g["env"]))
tmp = tempfile.mkdtemp()
paths = {}
for order in (0, 1):
    m = hb.Model.load(HUMANOID_HBM)
    assert m.pair_order(order) == order
    paths[order] = os.path.join(tmp, "order%d.hbm" % order)
    m.save(paths[order])
for solver, name in ((0, "PGS, 50 sweeps"), (2, "Newton, 100 iterations")):
    o = {k: Oracle(p) for k, p in paths.items()}
    for x in o.values():
        x.set_opt(solver=solver, iterations=50 if solver == 0 else 100)
    moved = changed = 0
    worst = dict(qacc=0.0, qvel=0.0, force=0.0)
    hist = []
    for k in range(len(g["env"])):
        out = {}
        for order, x in o.items():
            x.qpos[:] = g["qpos"][k]; x.qvel[:] = g["qvel"][k]; x.qacc_warmstart[:] = g["warm"][k]; x.ctrl[:] = g["ctrl"][k]
            x.forward()
            out[order] = (x.qacc.copy(), [(c["geom1"], c["geom2"]) for c in x.contacts()], x.ncon)
        assert sorted(out[0][1]) == sorted(out[1][1])  # the same contacts ...
        if out[0][1] != out[1][1]:  # ... in another order
            moved += 1
            d = np.abs(out[0][0] - out[1][0]).max() / max(1.0, np.abs(out[1][0]).max())
            hist.append(d)
            worst["qacc"] = max(worst["qacc"], d)
            changed += d > 1e-9
    hist = np.array(hist) if hist else np.zeros(1)
    print("%s: %d of %d states emit their contacts in another order (multi-geom bodies in self collision, or two bodies on the floor at once); "
          "%d of them step differently: relative qacc difference median %.1e, max %.1e" % (name, moved, len(g["env"]), changed, np.median(hist), worst["qacc"]))
