#!/usr/bin/env python3
"""When do the waves of a multi-step launch of the two-envs-per-wave kernel end, and where did they run?  Diagnostic build
(build/libhb_stamps.so): every wave leaves its HW_ID / XCC_ID, its block index and its end time.  4096 envs of the benchmark's steady
regime, one launch of K steps (default 250)."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import humanoid_mujoco_amd.engine as eng
eng.LIB_PATH = os.path.join(ROOT, "build", "libhb_stamps.so")
import humanoid_mujoco_amd as hb
L = eng.lib()
L.hb_get_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
m = hb.Model.load(os.path.join(ROOT, "humanoid_mujoco_amd", "assets", "humanoid27.hbm"))
N, K = 4096, int(sys.argv[1]) if len(sys.argv) > 1 else 250
b = hb.Batch(m, N, 0)
b.reset(perturb=True)
b.rollout_halton(600)
b.rollout_halton(K, 600)   # (the order of the measured launch comes from a launch like it)
b.sync()
st = np.zeros((N, 16), np.uint64)
assert L.hb_get_stamps(b._h, st.ctypes.data_as(ctypes.c_void_p)) == 0  # arm
b.rollout_halton(K, 600 + K)
b.sync()
assert L.hb_get_stamps(b._h, st.ctypes.data_as(ctypes.c_void_p)) == 0
w = st[(st[:, 3] < (N + 1) // 2) & (st[:, 4] > 0) & (st[:, 2] < 64)]  # (the record of a wave's first env; its second env's record holds the stage stamps of the last step)
blk = w[:, 3].astype(np.int64); hwid = w[:, 1].astype(np.int64); xcc = w[:, 2].astype(np.int64) & 15
end = w[:, 4].astype(np.float64) - w[:, 5].astype(np.float64)  # the wave's own duration (every XCD has its own clock: ends are not comparable, durations are)
simd = (hwid >> 4) & 3; cu = (hwid >> 8) & 15; sh = (hwid >> 12) & 1; se = (hwid >> 13) & 7
key = ((xcc * 8 + se) * 2 + sh) * 64 + cu * 4 + simd
print("%d waves recorded; wave durations (clock ticks): min %.4g, median %.4g, 90th %.4g, 99th %.4g, max %.4g; mean / max %.3f"
      % (len(w), end.min(), np.median(end), np.percentile(end, 90), np.percentile(end, 99), end.max(), end.mean() / end.max()))
u, cnt = np.unique(key, return_counts=True)
print("distinct (xcc, se, sh, cu, simd): %d; waves per SIMD: %s" % (len(u), dict(zip(*np.unique(cnt, return_counts=True)))))
# mates: the block indices that shared a SIMD
d = []
for k in u[:4000]:
    bb = np.sort(blk[key == k])
    if len(bb) == 2:
        d.append(bb[1] - bb[0])
d = np.array(d)
if len(d):
    vals, c = np.unique(d, return_counts=True)
    top = np.argsort(-c)[:6]
    print("block-index distance of the two waves of a SIMD: most common", [(int(vals[i]), int(c[i])) for i in top])
# per SIMD: the later of its two ends
late = np.array([end[key == k].max() for k in u])
print("per SIMD, the longer of its two waves: median %.4g, max %.4g; the shorter: median %.4g" % (np.median(late), late.max(), np.median([end[key == k].min() for k in u])))
order = np.argsort(blk)
print("duration by block index (thousands of ticks), every 128th block:", [int(end[order][i] / 1000) for i in range(0, len(order), 128)])
print("histogram of the ends over the spread (ten bins):", np.histogram(end, bins=10)[0].tolist())
