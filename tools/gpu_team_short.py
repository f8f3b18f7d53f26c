"""Short team-robot run for counter passes: 4096 envs, 150 settle steps from the standup reset, then 30 single-step launches."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import humanoid_mujoco_amd as hb  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
m = hb.Model.load(os.path.join(ROOT, "humanoid_mujoco_amd", "assets", sys.argv[2] if len(sys.argv) > 2 else "team_robot.hbm"))
b = hb.Batch(m, n, 0)
b.reset(keyframe=0 if len(sys.argv) <= 2 else -1, perturb=True)
b.rollout_halton(150, 0, 0)
ctrl = np.zeros((n, m.nu), dtype=np.float32)
for t in range(30):
    b.step(ctrl)
b.sync()
nc, ne, _ = b.counts()
print("mean ncon %.2f nefc %.1f flagged %d" % (nc.mean(), ne.mean(), int((b.status() != 0).sum())))
nw, ns, kc = b.collision_counts(want_cycles=True)
print("work items per env %.1f (max %d), portal searches per env %.1f (max %d)" % (nw.mean(), nw.max(), ns.mean(), ns.max()))
print("narrowphase wave time by the env's contacts (thousands of cycles): " + "; ".join(
    "%d contacts: %d envs, mean %.0f, max %d" % (k, (nc == k).sum(), kc[nc == k].mean(), kc[nc == k].max()) for k in range(0, 6) if (nc == k).any()))
print("by searches: " + "; ".join("%d: mean %.0f" % (k, kc[ns == k].mean()) for k in range(0, 12) if (ns == k).any()))
