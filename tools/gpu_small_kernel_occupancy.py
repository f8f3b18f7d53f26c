"""Is hb_step_small_kernel resident at three waves per SIMD?  Unpipelined single-step launches on falling humanoids (no contacts yet:
no env leaves the fast lane), batch sizes around the residency limits (2048 = two waves per SIMD, 3072 = three): time per launch, small vs full kernel."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import humanoid_mujoco_amd as hb
from oracle_lib import HUMANOID_HBM
m = hb.Model.load(HUMANOID_HBM)
for n in (1024, 2048, 2560, 3072, 3584, 4096, 6144, 8192):
    row = []
    for tl in ("0", "1"):
        os.environ["HB_TWO_LANE"] = tl
        b = hb.Batch(m, n, 0)
        best = 1e9
        for rep in range(3):
            b.reset(perturb=True)
            b.rollout_halton(10, t0=0)   # warm
            b.sync()
            t0 = time.perf_counter()
            for t in range(40):
                b.rollout_halton(1, t0=10 + t)
            b.sync()
            best = min(best, (time.perf_counter() - t0) / 40)
        row.append(best)
        slow = int(b.lanes().sum())
        b.close()
    print("n_env %5d: full kernel %6.1f us / step, small kernel %6.1f us / step (slow-lane envs at the end: %d)" % (n, 1e6 * row[0], 1e6 * row[1], slow))
