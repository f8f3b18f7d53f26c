#!/usr/bin/env python3
"""Step-API throughput (one hb_step_dev call per step, controls resident) against the number of
pipeline segments (hb_batch_pipeline)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import humanoid_mujoco_amd as hb
m = hb.Model.load(os.path.join(ROOT, "humanoid_mujoco_amd", "assets", "humanoid27.hbm"))
N, K, W = 4096, 400, 300
for npipe in [0, 2, 3, 4]:
    b = hb.Batch(m, N, 0)
    ctrl = b.dev_alloc((K + W) * N * m.nu * 4)
    b.halton_ctrl_dev(K + W, 0, 0, ctrl)
    b.reset(perturb=True)
    b.pipeline(npipe)
    b.tune(fold=1)  # (this is about one launch per call)
    stride = N * m.nu * 4
    for t in range(W): b.step_dev(ctrl + t * stride)
    b.sync()
    t0 = time.perf_counter()
    for t in range(W, W + K): b.step_dev(ctrl + t * stride)
    t1 = time.perf_counter()
    b.sync()
    dt = time.perf_counter() - t0
    print("segments %d: %.1f us/step -> %.3e env-steps/s (host enqueue %.1f us/step)" % (npipe, 1e6 * dt / K, N * K / dt, 1e6 * (t1 - t0) / K), flush=True)
    b.dev_free(ctrl); b.close()
