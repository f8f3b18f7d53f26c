#!/usr/bin/env python3
"""North-star drift bar on trajectories without contact events: the benchmark humanoid with contacts disabled (it falls and
flails under the Halton actions, joint limits active), 1000 free-running steps, GPU fp32 vs oracle fp64, both solvers."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import humanoid_mujoco_amd as hb
from oracle_lib import Oracle, HUMANOID_HBM
for solver, name in ((0, "PGS/50"), (2, "Newton/100")):
    m = hb.Model.load(HUMANOID_HBM)
    kw = dict(disableflags=16)
    if solver == 2:
        kw.update(solver=2, iterations=100)
    m.set_opt(**kw)
    envs = list(range(8))
    b = hb.Batch(m, len(envs), 0); b.reset(perturb=True)
    os_ = []
    for e in envs:
        o = Oracle(); o.set_opt(**kw); o.init_env(e); os_.append(o)
    worst = 0.0
    for t in range(1000):
        ctrl = np.stack([o.ctrl_env(t, e) for o, e in zip(os_, envs)]).astype(np.float32)
        b.step(ctrl)
        for o, c in zip(os_, ctrl):
            o.ctrl[:] = c; o.step()
        if t in (99, 249, 499, 999):
            q = b.qpos
            w = max(float((np.abs(q[i] - o.qpos) / np.maximum(1.0, np.abs(o.qpos))).max()) for i, o in enumerate(os_))
            rows = np.mean([o.nefc for o in os_])
            print("%-10s step %4d: max relative qpos drift over %d envs %.2e (mean limit rows now %.1f)" % (name, t + 1, len(envs), w, rows), flush=True)
