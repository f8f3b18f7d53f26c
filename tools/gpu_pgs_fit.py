#!/usr/bin/env python3
"""Least-squares split of the PGS phase's cycles (diagnostic stamps build) into a fixed part, a per-sweep part and a
per-row-update part, over the envs of one step."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import humanoid_mujoco_amd.engine as eng
eng.LIB_PATH = os.environ.get("HB_STAMPS_LIB", os.path.join(ROOT, "build", "libhb_stamps.so"))
import humanoid_mujoco_amd as hb
L = eng.lib()
L.hb_get_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
N = 4096
m = hb.Model.load(os.path.join(ROOT, "humanoid_mujoco_amd", "assets", "humanoid27.hbm"))
b = hb.Batch(m, N, 0)
b.reset(perturb=True)
b.rollout_halton(400)
st = np.zeros((N, 16), dtype=np.uint64)
L.hb_get_stamps(b._h, st.ctypes.data_as(ctypes.c_void_p))
b.rollout_halton(1, t0=400)
L.hb_get_stamps(b._h, st.ctypes.data_as(ctypes.c_void_p))
d = np.diff(st.astype(np.int64), axis=1).astype(np.float64)
nc, ne, ni = b.counts()
pgs = d[:, 12]
pad = 4 * ((ne + 3) // 4)
A = np.stack([np.ones(N), ne, ni, ni * pad], axis=1)
x, *_ = np.linalg.lstsq(A, pgs, rcond=None)
print("PGS cycles ~ %.0f + %.1f * nefc + %.1f * sweeps + %.2f * sweeps * padded_rows   (mean %.0f, fit rms %.0f)" % (*x, pgs.mean(), np.sqrt(((A @ x - pgs) ** 2).mean())))
for k in (0, 4, 8, 12, 16, 24):
    sel = pad == k
    if sel.sum() > 20:
        print("  padded rows %2d: %4d envs, mean sweeps %5.1f, mean cycles %7.0f -> %6.1f cycles per sweep" % (k, sel.sum(), ni[sel].mean(), pgs[sel].mean(), pgs[sel].mean() / max(1, ni[sel].mean())))
