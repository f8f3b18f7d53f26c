#!/usr/bin/env python3
"""BASELINE config 4 throughput: 4096 envs, obs -> MLP 48-256-256-21 (tanh) -> mj_step closed loop on device."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import humanoid_mujoco_amd as hb
import torch
m = hb.Model.load(os.path.join(ROOT, "humanoid_mujoco_amd", "assets", "humanoid27.hbm"))
N, T = 4096, 300
torch.manual_seed(0)
layers = [torch.nn.Linear(m.nobs, 256), torch.nn.Linear(256, 256), torch.nn.Linear(256, m.nu)]
for segs in (0, 2, 3, 4):
    b = hb.Batch(m, N, 0)
    b.set_policy_mlp([l.weight.detach().numpy().T.copy() for l in layers], [l.bias.detach().numpy().copy() for l in layers])
    b.reset(perturb=True)
    b.pipeline(segs)
    b.rollout_policy(20); b.sync()
    b.timer_start()
    t0 = time.perf_counter()
    b.rollout_policy(T)
    ms = b.timer_stop()
    wall = time.perf_counter() - t0
    nc, ne, ni = b.counts()
    print("config 4 (%s): %d envs x %d closed-loop steps: %.1f us/step (HIP events), wall %.3f s -> %.3e env-steps/s; MLP flop/env-step 166400; mean nefc %.1f; warnings %d"
          % ("pipelined, %d segments" % segs if segs else "one launch chain", N, T, 1e3 * ms / T, wall, N * T / wall, ne.mean(), (b.status() != 0).sum()))
    b.close()
