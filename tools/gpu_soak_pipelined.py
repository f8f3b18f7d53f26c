#!/usr/bin/env python3
"""Soak of the pipelined step API (hb_batch_pipeline default: three env segments): K step calls with on-device Halton controls against
the same K steps as single rollout launches of an unpipelined batch - the final states must agree bit for bit."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import humanoid_mujoco_amd as hb
m = hb.Model.load(os.path.join(ROOT, "humanoid_mujoco_amd", "assets", "humanoid27.hbm"))
N = 4096
K = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
CH = 1000
a = hb.Batch(m, N, 0); b = hb.Batch(m, N, 0)
a.reset(perturb=True); b.reset(perturb=True)
a.pipeline(True)
ctrl = a.dev_alloc(CH * N * m.nu * 4)
stride = N * m.nu * 4
t0 = time.perf_counter()
for c in range(K // CH):
    a.sync()                                   # the tape is rewritten: the steps reading it must be done
    a.halton_ctrl_dev(CH, c * CH, 0, ctrl)     # controls t = c CH .. c CH + CH - 1, as rollout_halton draws them
    for t in range(CH): a.step_dev(ctrl + t * stride)
a.sync()
dt = time.perf_counter() - t0
t1 = time.perf_counter()
for c in range(K // CH): b.rollout_halton(CH, c * CH, 0)
b.sync()
dt2 = time.perf_counter() - t1
sa, sb = a.get_state(hb.STATE_INTEGRATION), b.get_state(hb.STATE_INTEGRATION)
print("%d pipelined step calls x %d envs (%d segments) in %.2f s (%.3e env-steps/s); the same steps as %d rollout launches in %.2f s" % (K, N, a.segments, dt, N * K / dt, K // CH, dt2))
print("final states bit-identical: %s; finite: %s; envs with warnings: %d / %d; max nefc seen at the end %d" % (np.array_equal(sa, sb), np.isfinite(sa).all(), (a.status() != 0).sum(), (b.status() != 0).sum(), a.counts()[1].max()))
