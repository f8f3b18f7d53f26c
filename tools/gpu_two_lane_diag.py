"""Diagnostic: one teacher-forced step, two-lane vs full kernel; per-env differences."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import humanoid_mujoco_amd as hb
from oracle_lib import HUMANOID_HBM
m = hb.Model.load(HUMANOID_HBM)
n = 768
os.environ["HB_TWO_LANE"] = "0"; full = hb.Batch(m, n, 0)
os.environ["HB_TWO_LANE"] = "1"; two = hb.Batch(m, n, 0)
full.reset(perturb=True); full.rollout_halton(300)
for t in range(3):
    st = full.get_state(hb.STATE_INTEGRATION)
    two.set_state(hb.STATE_INTEGRATION, st)
    full.rollout_halton(1, t0=300 + t); two.rollout_halton(1, t0=300 + t)
    a = full.get_state(hb.STATE_INTEGRATION).astype(np.float64); b = two.get_state(hb.STATE_INTEGRATION).astype(np.float64)
    nq, nv = m.nq, m.nv
    dq = np.abs(a[:, 1:1+nq] - b[:, 1:1+nq]).max(1)
    dv = np.abs(a[:, 1+nq:1+nq+nv] - b[:, 1+nq:1+nq+nv]).max(1) / np.maximum(1, np.abs(a[:, 1+nq:1+nq+nv]).max(1))
    dw = np.abs(a[:, 1+nq+nv:] - b[:, 1+nq+nv:]).max(1) / np.maximum(1, np.abs(a[:, 1+nq+nv:]).max(1))
    lanes = two.lanes()
    nc, ne, ni = full.counts(); nc2, ne2, ni2 = two.counts()
    print("step", t, "envs differing", int(((dq > 0) | (dv > 0) | (dw > 0)).sum()), "slow", int(lanes.sum()),
          "| dq max %.2e dv rel max %.2e dw rel max %.2e" % (dq.max(), dv.max(), dw.max()))
    print("   rel dw percentiles 50/90/99:", np.percentile(dw, [50, 90, 99]), " counts equal:", np.array_equal(nc, nc2), np.array_equal(ne, ne2), np.array_equal(ni, ni2),
          "niter differing", int((ni != ni2).sum()))
    worst = np.argsort(-dw)[:5]
    for e in worst:
        print("   env %d nefc %d/%d niter %d/%d lane %d dq %.2e dv %.2e dw %.2e" % (e, ne[e], ne2[e], ni[e], ni2[e], lanes[e], dq[e], dv[e], dw[e]))
