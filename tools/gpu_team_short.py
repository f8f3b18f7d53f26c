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
nw, ns = b.collision_counts()
print("work items per env %.1f (max %d), portal searches per env %.1f (max %d)" % (nw.mean(), nw.max(), ns.mean(), ns.max()))
