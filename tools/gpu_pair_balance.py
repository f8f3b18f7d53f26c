#!/usr/bin/env python3
"""How unequal is the work of the wave pairs over a multi-step launch?  4096 envs of the benchmark's steady regime, 256 steps one launch per
call; per step the rows and sweeps of every env (hb_get_counts).  A duo wave's vector instructions per step are modelled from
profiles/r04_phase_instructions_duo.txt: 6450 outside the sweeps, a sweep of both envs rows_max x 7 + 30, a sweep of one env rows x 5 + 25.
A one-round multi-step launch lasts as long as its heaviest pair; prints the heaviest pair's total against the mean, for pairs of
neighbours in the order of the envs' mean cost (what the heavy-first order gives), random pairs, and heaviest-with-lightest."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import humanoid_mujoco_amd as hb
m = hb.Model.load(os.path.join(ROOT, "humanoid_mujoco_amd", "assets", "humanoid27.hbm"))
N, K = 4096, 256
b = hb.Batch(m, N, 0)
b.tune(fold=1)
b.reset(perturb=True)
b.rollout_halton(600)
ctrl = b.dev_alloc(K * N * m.nu * 4)
b.halton_ctrl_dev(K, 600, 0, ctrl)
rows = np.zeros((K, N), np.int32); sweeps = np.zeros((K, N), np.int32)
for t in range(K):
    b.step_dev(ctrl + t * N * m.nu * 4)
    _, ne, ni = b.counts()
    rows[t] = ne; sweeps[t] = ni


def pair_cost(ia, ib):
    ra, rb, sa, sb = rows[:, ia], rows[:, ib], sweeps[:, ia], sweeps[:, ib]
    both = np.minimum(sa, sb)
    one = np.abs(sa - sb)
    r_one = np.where(sa > sb, ra, rb)
    return (6450 + both * (np.maximum(ra, rb) * 7 + 30) + one * (r_one * 5 + 25)).sum(axis=0)


env_cost = (rows * sweeps).sum(axis=0).astype(np.float64)
print("per env, sum over %d steps of rows x sweeps: mean %.0f, std %.0f, min %.0f, max %.0f (max / mean %.2f); sweeps per step mean %.1f; env-steps at the 50-sweep cap %.1f %%"
      % (K, env_cost.mean(), env_cost.std(), env_cost.min(), env_cost.max(), env_cost.max() / env_cost.mean(), sweeps.mean(), 100.0 * (sweeps >= 50).mean()))
order = np.argsort(-env_cost)
rng = np.random.default_rng(0)
perm = rng.permutation(N)
for name, ia, ib in (("neighbours in cost order", order[0::2], order[1::2]), ("random pairs", perm[0::2], perm[1::2]), ("heaviest with lightest", order[:N // 2], order[::-1][:N // 2])):
    c = pair_cost(ia, ib).astype(np.float64)
    print("%-26s modelled VALU per pair over the launch: mean %.3e, max %.3e, max / mean %.3f, 99th percentile / mean %.3f" % (name, c.mean(), c.max(), c.max() / c.mean(), np.percentile(c, 99) / c.mean()))
solo = (6970 - 2230 + sweeps * (rows * 5 + 25)).sum(axis=0).astype(np.float64)
print("one env per wave, modelled: mean %.3e, max / mean %.3f" % (solo.mean(), solo.max() / solo.mean()))
