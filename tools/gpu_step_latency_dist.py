#!/usr/bin/env python3
"""How long does ONE env-step take, env by env, on the benchmark workload - and how do constraint rows / contacts move from one step to
the next?  (Diagnostic build build/libhb_stamps.so: s_memtime at the first and last stamp of every wave; 100 MHz clock.)
What the numbers are for: the two-envs-per-wave kernel holds <= 31 rows / 12 contacts per env; an env-step above that needs a whole
wave, and a launch lasts as long as its slowest wave - so the tail of this distribution, not its mean, bounds a step call."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import humanoid_mujoco_amd.engine as eng
eng.LIB_PATH = os.environ.get("HB_STAMPS_LIB", os.path.join(ROOT, "build", "libhb_stamps.so"))
import humanoid_mujoco_amd as hb
L = eng.lib()
L.hb_get_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
m = hb.Model.load(os.path.join(ROOT, "humanoid_mujoco_amd", "assets", "humanoid27.hbm"))
TICK_US = 0.01


def run(N, steps=40):
    b = hb.Batch(m, N, 0)
    b.reset(perturb=True)
    b.rollout_halton(600)
    st = np.zeros((N, 16), dtype=np.uint64)
    assert L.hb_get_stamps(b._h, st.ctypes.data_as(ctypes.c_void_p)) == 0  # arm
    dur, rows, sweeps, cons = [], [], [], []
    for t in range(steps):
        b.rollout_halton(1, t0=600 + t)
        assert L.hb_get_stamps(b._h, st.ctypes.data_as(ctypes.c_void_p)) == 0
        s = st.astype(np.int64)
        dur.append((s[:, 15] - s[:, 0]) * TICK_US)
        nc, ne, ni = b.counts()
        rows.append(ne.copy()); sweeps.append(ni.copy()); cons.append(nc.copy())
    b.close()
    return np.array(dur), np.array(rows), np.array(sweeps), np.array(cons)


for N in (256, 2048, 4096):
    dur, rows, sweeps, cons = run(N)
    d = dur.ravel()
    print("== %d envs, %d steps: wave time per env-step (us, stamp build): mean %.1f median %.1f p90 %.1f p99 %.1f p99.9 %.1f max %.1f"
          % (N, dur.shape[0], d.mean(), np.median(d), *np.percentile(d, [90, 99, 99.9]), d.max()))
    print("   slowest wave of each launch: mean %.1f min %.1f max %.1f" % (dur.max(1).mean(), dur.max(1).min(), dur.max(1).max()))
    work = (rows * sweeps).ravel()
    A = np.stack([np.ones_like(work, dtype=np.float64), work, rows.ravel()], 1)
    coef, *_ = np.linalg.lstsq(A, d, rcond=None)
    print("   fit: %.1f us + %.4f us per row update + %.3f us per row" % tuple(coef))
    if N != 4096:
        continue
    r = rows.ravel(); c = cons.ravel()
    print("   rows: mean %.1f max %d; contacts: mean %.1f max %d; sweeps mean %.1f" % (r.mean(), r.max(), c.mean(), c.max(), sweeps.mean()))
    for cap_r, cap_c in ((31, 12), (31, 24), (27, 12), (23, 12)):
        print("   env-steps above %d rows or %d contacts: %.3f %%  (per 4096-env step: %.1f)" % (cap_r, cap_c, 100 * np.mean((r > cap_r) | (c > cap_c)), 4096 * np.mean((r > cap_r) | (c > cap_c))))
    over = (rows > 31) | (cons > 12)
    print("   wave time of the env-steps above 31 rows / 12 contacts: mean %.1f max %.1f us; of the others: mean %.1f p99.9 %.1f max %.1f"
          % (dur[over].mean() if over.any() else 0, dur[over].max() if over.any() else 0, dur[~over].mean(), np.percentile(dur[~over], 99.9), dur[~over].max()))
    # prediction from the previous step: P(over now | previous step's rows <= k and contacts <= kc)
    prev_r, prev_c, now = rows[:-1], cons[:-1], over[1:]
    for k, kc in ((31, 12), (27, 10), (23, 9), (19, 8), (15, 6)):
        light = (prev_r <= k) & (prev_c <= kc)
        print("   predicted light (previous rows <= %d, contacts <= %d): %.2f %% of env-steps; of those %.4f %% overflow now (%.2f per 4096-env step); predicted heavy per step: %.1f"
              % (k, kc, 100 * light.mean(), 100 * now[light].mean(), 4096 * (now & light).mean(), 4096 * (~light).mean()))
    # how long do streaks above the capacity last
    runs = []
    for e in range(over.shape[1]):
        n = 0
        for t in range(over.shape[0]):
            if over[t, e]: n += 1
            elif n: runs.append(n); n = 0
        if n: runs.append(n)
    if runs:
        print("   streaks above capacity: %d, mean length %.1f, max %d" % (len(runs), np.mean(runs), max(runs)))
    # the heaviest env-steps
    idx = np.argsort(d)[-8:]
    print("   slowest env-steps: " + ", ".join("%.0f us (%d rows x %d sweeps, %d contacts)" % (d[i], r[i], sweeps.ravel()[i], c[i]) for i in idx))
