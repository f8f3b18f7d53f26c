#!/usr/bin/env python3
"""Free-running GPU vs oracle over the contact-free opening of the benchmark workload (the quantity
tests/test_gpu_parity.py::test_contact_free_drift_within_north_star_bar asserts), per step.  HB_LIB overrides the library."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import humanoid_mujoco_amd.engine as eng
if os.environ.get("HB_LIB"):
    eng.LIB_PATH = os.environ["HB_LIB"]
import humanoid_mujoco_amd as hb
from oracle_lib import Oracle, HUMANOID_HBM
solver = int(sys.argv[1]) if len(sys.argv) > 1 else 0
m = hb.Model.load(HUMANOID_HBM)
if solver == 2:
    m.set_opt(solver=2, iterations=100)
envs = list(range(16))
b = hb.Batch(m, len(envs), 0); b.diag_enable(True); b.reset(perturb=True)
os_ = []
for e in envs:
    o = Oracle()
    if solver == 2:
        o.set_opt(solver=2, iterations=100)
    o.init_env(e); os_.append(o)
worst = 0.0
for t in range(50):
    ctrl = np.stack([o.ctrl_env(t, e) for o, e in zip(os_, envs)]).astype(np.float32)
    b.step(ctrl)
    for o, c in zip(os_, ctrl):
        o.ctrl[:] = c; o.step()
    q = b.qpos; a = b.qacc()
    w = 0.0; wa = 0.0
    for i, o in enumerate(os_):
        if o.ncon == 0:
            w = max(w, float((np.abs(q[i] - o.qpos) / np.maximum(1.0, np.abs(o.qpos))).max()))
            wa = max(wa, float(np.abs(a[i] - o.qacc).max() / max(1.0, np.abs(o.qacc).max())))
    worst = max(worst, w)
    if t % 7 == 0 or t == 49:
        print("step %2d: contact-free rel qpos drift %.2e (max so far %.2e), one-step-ish rel qacc diff %.1e" % (t, w, worst, wa), flush=True)
