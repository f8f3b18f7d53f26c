#!/usr/bin/env python3
"""configs[3] closed loop with N pipeline segments, short, for a kernel trace (rocprofv3 --kernel-trace -- python3 tools/gpu_config4_trace.py 3)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import humanoid_mujoco_amd as hb
m = hb.Model.load(os.path.join(ROOT, "humanoid_mujoco_amd", "assets", "humanoid27.hbm"))
g = np.load(os.path.join(ROOT, "tests", "golden", "policy_mlp_seed0.npz"))
ws = [g["w%d" % i] for i in range(3)]; bs = [g["b%d" % i] for i in range(3)]
segs = int(sys.argv[1]) if len(sys.argv) > 1 else 3
b = hb.Batch(m, 4096, 0)
b.set_policy_mlp(ws, bs)
b.reset(perturb=True)
b.pipeline(segs)
b.rollout_policy(120); b.sync()
