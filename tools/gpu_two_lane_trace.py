#!/usr/bin/env python3
"""A short pipelined two-lane run for a kernel trace (rocprofv3 --kernel-trace -- python3 tools/gpu_two_lane_trace.py <segments>)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import humanoid_mujoco_amd as hb
m = hb.Model.load(os.path.join(ROOT, "humanoid_mujoco_amd", "assets", "humanoid27.hbm"))
N, T = 4096, 700
segs = int(sys.argv[1]) if len(sys.argv) > 1 else 2
b = hb.Batch(m, N, 0)
ctrl = b.dev_alloc(T * N * m.nu * 4)
b.halton_ctrl_dev(T, 0, 0, ctrl)
b.reset(perturb=True)
b.pipeline(segs)
stride = N * m.nu * 4
for t in range(T): b.step_dev(ctrl + t * stride)
b.sync()
print("lanes: %d envs in the slow lane at the end" % int((b.lanes() != 0).sum()))
