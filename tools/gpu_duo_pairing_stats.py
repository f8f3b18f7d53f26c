import os, sys, numpy as np
sys.path.insert(0, '.')
import humanoid_mujoco_amd as hb
m = hb.Model.load("humanoid_mujoco_amd/assets/humanoid27.hbm")
N=4096
b = hb.Batch(m, N, 0); b.reset(perturb=True); b.rollout_halton(600)
prev=None
rows=[]
for t in range(12):
    b.rollout_halton(1, 600+t)
    nc, ne, ni = b.counts()
    if prev is not None:
        pn, pe = prev
        def cost(ni_, ne_, idx):
            a, c = idx[0::2], idx[1::2]
            sw = np.maximum(ni_[a], ni_[c]); rw = np.maximum(ne_[a], ne_[c])
            return sw.mean(), (sw * (500 + 49 * ((rw + 3) // 4 * 4))).mean()
        rnd = np.random.default_rng(t).permutation(N)
        by_prev = np.lexsort((pe, pn))      # sweeps major, rows minor of the previous step
        by_prod = np.argsort(pn * pe)
        by_now = np.lexsort((ne, ni))
        solo = (ni * (500 + 32 * ((ne + 3) // 4 * 4))).mean()
        rows.append((ni.mean(), cost(ni, ne, rnd), cost(ni, ne, by_prod), cost(ni, ne, by_prev), cost(ni, ne, by_now), solo))
    prev=(ni.copy(), ne.copy())
r=np.array([[x[0], x[1][0], x[1][1], x[2][0], x[2][1], x[3][0], x[3][1], x[4][0], x[4][1], x[5]] for x in rows]).mean(0)
print("mean sweeps per env %.1f; sweeps per WAVE (max of the pair): random pairing %.1f, sorted by previous rows x sweeps %.1f, by previous sweeps then rows %.1f, by this step's own (unknowable) %.1f" % (r[0], r[1], r[3], r[5], r[7]))
print("model cycles per wave in the sweeps: random %.0f, prev product %.0f, prev sweeps-major %.0f, ideal %.0f; one-env kernel per env %.0f (x2 = %.0f)" % (r[2], r[4], r[6], r[8], r[9], 2*r[9]))
print("niter histogram:", np.bincount(np.minimum(ni,50)//5))
