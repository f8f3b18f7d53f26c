#!/usr/bin/env python3
"""Soak: N step calls of 4096 envs through the library's default path (calls folded into 256-step launches of the two-envs-per-wave kernel,
paired by average cost, priorities in turns) against one launch per call of the one-env-per-wave kernel on three env segments: the final
states, counts and status words must be identical bit for bit."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import humanoid_mujoco_amd as hb
m = hb.Model.load(os.path.join(ROOT, "humanoid_mujoco_amd", "assets", "humanoid27.hbm"))
N = 4096
TOTAL = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
CH = 1000
out = []
for fold, duo in ((256, 1), (1, 0)):
    b = hb.Batch(m, N, 0)
    b.tune(fold=fold, duo=duo)
    b.reset(perturb=True)
    b.pipeline(True)
    ctrl = b.dev_alloc(CH * N * m.nu * 4)
    t0 = time.perf_counter()
    for c in range(TOTAL // CH):
        b.halton_ctrl_dev(CH, c * CH, 0, ctrl)
        for t in range(CH):
            b.step_dev(ctrl + t * N * m.nu * 4)
    b.sync()
    dt = time.perf_counter() - t0
    st = b.get_state(hb.STATE_INTEGRATION)
    out.append((st,) + tuple(b.counts()) + (b.status(),))
    print("fold %3d duo %d: %d step calls x %d envs in %.2f s (%.3e env-steps/s), last kernel %s, launches %d; finite %s, envs with warnings %d"
          % (fold, duo, TOTAL, N, dt, N * TOTAL / dt, b.last_kernel(), b.step_launches(), bool(np.isfinite(st).all()), int((out[-1][-1] != 0).sum())), flush=True)
    b.dev_free(ctrl); b.close()
print("bit-identical:", all(np.array_equal(x, y) for x, y in zip(out[0], out[1])))
