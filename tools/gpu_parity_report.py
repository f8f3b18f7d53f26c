#!/usr/bin/env python3
"""Per-sample one-step parity report: GPU (fp32) vs the fp64 oracle on the golden states.
usage: gpu_parity_report.py [newton]   (default: the benchmark configuration, PGS/50, against the golden vectors;
"newton": solver = Newton/100 on both sides, the oracle run here on the same states)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import humanoid_mujoco_amd as hb
from oracle_lib import GOLDEN, HUMANOID_HBM, Oracle

g = dict(np.load(os.path.join(GOLDEN, "humanoid27_steps.npz")))
n = len(g["env"])
m = hb.Model.load(HUMANOID_HBM)
if len(sys.argv) > 1 and sys.argv[1] == "newton":
    m.set_opt(solver=2, iterations=100)
    o = Oracle()
    o.set_opt(solver=2, iterations=100)
    for key in ("qpos1", "qvel1", "qacc", "efc_force", "niter"):
        g[key] = g[key].copy()
    for k in range(n):
        o.qpos[:] = g["qpos"][k]; o.qvel[:] = g["qvel"][k]; o.qacc_warmstart[:] = g["warm"][k]; o.ctrl[:] = g["ctrl"][k]
        o.forward()
        g["qacc"][k] = o.qacc; g["niter"][k] = o.dint("solver_niter")
        g["efc_force"][k] = 0; g["efc_force"][k, :o.nefc] = o.efc_force[:o.nefc]
        o.step()
        g["qpos1"][k] = o.qpos; g["qvel1"][k] = o.qvel
    print("solver: Newton/100 on both sides")
b = hb.Batch(m, n, 0)
b.diag_enable(True)
st = np.concatenate([g["time"][:, None], g["qpos"], g["qvel"], g["warm"]], axis=1)
b.set_state(hb.STATE_INTEGRATION, st)
b.step(g["ctrl"].astype(np.float32))
q, v, a, f = b.qpos.astype(float), b.qvel.astype(float), b.qacc().astype(float), b.efc_force().astype(float)
nc, ne, ni = b.counts()
print("%4s %4s %5s %5s %5s %6s %6s | %9s %9s %9s %9s | %9s %9s" % ("k", "env", "step", "ncon", "nefc", "it_gpu", "it_ref", "dqpos", "dqvel_rel", "dqacc_rel", "dforce_rel", "max|qacc|", "max|f|"))
rows = []
for k in range(n):
    dq = (np.abs(q[k] - g["qpos1"][k]) / np.maximum(1, np.abs(g["qpos1"][k]))).max()
    dv = np.abs(v[k] - g["qvel1"][k]).max() / max(1, np.abs(g["qvel1"][k]).max())
    da = np.abs(a[k] - g["qacc"][k]).max() / max(1, np.abs(g["qacc"][k]).max())
    df = np.abs(f[k] - g["efc_force"][k]).max() / max(1, np.abs(g["efc_force"][k]).max())
    rows.append((dq, dv, da, df))
    print("%4d %4d %5d %5d %5d %6d %6d | %9.2e %9.2e %9.2e %9.2e | %9.2e %9.2e" % (k, g["env"][k], g["step"][k], nc[k], ne[k], ni[k], g["niter"][k], dq, dv, da, df,
          np.abs(g["qacc"][k]).max(), np.abs(g["efc_force"][k]).max()))
r = np.array(rows)
print("max   dqpos %.2e dqvel_rel %.2e dqacc_rel %.2e dforce_rel %.2e" % tuple(r.max(0)))
print("median dqpos %.2e dqvel_rel %.2e dqacc_rel %.2e dforce_rel %.2e" % tuple(np.median(r, 0)))
print("count mismatch ncon %d nefc %d" % ((nc != g["ncon"]).sum(), (ne != g["nefc"]).sum()))
