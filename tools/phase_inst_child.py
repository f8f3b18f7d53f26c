#!/usr/bin/env python3
"""child of tools/gpu_phase_instructions.sh: 4096 envs pre-rolled into the benchmark's steady regime, then 12 single-step launches of the
diagnostic build that leave at stamp HB_STOP_AT - 1 (0: the whole step).  Run under rocprofv3 --pmc."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import humanoid_mujoco_amd.engine as eng
eng.LIB_PATH = os.path.join(ROOT, "build", "libhb_stamps.so")
import humanoid_mujoco_amd as hb
m = hb.Model.load(os.path.join(ROOT, "humanoid_mujoco_amd", "assets", sys.argv[2] if len(sys.argv) > 2 else "humanoid27.hbm"))
N = 4096
b = hb.Batch(m, N, 0)
b.tune(duo=int(os.environ.get("PHASE_DUO", "0")))  # 2: hb_step_duo_kernel (two envs per wave), 0: hb_step_h27_kernel
b.reset(perturb=True)
b.rollout_halton(600)
b.sync()
ctrl = b.dev_alloc(N * m.nu * 4)
b.halton_ctrl_dev(1, 600, 0, ctrl)
os.environ["HB_STOP_PHASE"] = sys.argv[1]
for _ in range(12):
    b.step_dev(ctrl)   # (a wave that leaves early writes nothing: every launch sees the same state)
b.sync()
