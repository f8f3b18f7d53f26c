#!/usr/bin/env python3
"""Per-phase cycle shares of hb_step_kernel from the diagnostic build (build/libhb_stamps.so).
Shares only — never quote this build's run time (stamps serialise)."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import humanoid_mujoco_amd.engine as eng
eng.LIB_PATH = os.environ.get("HB_STAMPS_LIB", os.path.join(ROOT, "build", "libhb_stamps.so"))
import humanoid_mujoco_amd as hb
L = eng.lib()
L.hb_get_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
MODEL = sys.argv[2] if len(sys.argv) > 2 else "humanoid27.hbm"  # e.g. team_robot.hbm, humanoid27_hfield.hbm
m = hb.Model.load(os.path.join(ROOT, "humanoid_mujoco_amd", "assets", MODEL))
b = hb.Batch(m, N, 0)
b.reset(perturb=True, keyframe=0 if MODEL.startswith("team") else -1)  # the robot: the standup reset (lying)
b.rollout_halton(400)
st = np.zeros((N, 16), dtype=np.uint64)
assert L.hb_get_stamps(b._h, st.ctypes.data_as(ctypes.c_void_p)) == 0  # arm
b.rollout_halton(1, t0=400)
assert L.hb_get_stamps(b._h, st.ctypes.data_as(ctypes.c_void_p)) == 0
d = np.diff(st.astype(np.int64), axis=1).astype(np.float64)
names = ["ctrl+check", "kinematics", "geoms/com/cinert/cdof", "comVel+crb+rne tree passes", "qM", "factorM", "bias/passive/act", "collision", "makeConstraint",
         "row quantities", "half-solve", "b + AR", "PGS", "dual finish", "Euler+advance"]
if b.last_kernel() == "hb_step_duo_kernel":  # two envs per wave: no factorM slot, the half solve in two parts; cycles are per WAVE = two env-steps
    names = ["ctrl+check", "kinematics", "geoms/com/cinert/cdof", "comVel+crb+rne tree passes", "qM", "bias/passive/act", "collision", "makeConstraint",
             "row quantities", "W = elimination of M (both envs)", "C = J W", "b + AR", "PGS", "dual finish + checkAcc", "Euler+advance"]
    print("kernel: hb_step_duo_kernel (cycles per wave = per TWO env-steps)")
tot = d.sum(1)
print("model %s" % MODEL)
print("envs %d; mean cycles per env-step (one wave) %.0f, median %.0f" % (N, tot.mean(), np.median(tot)))
nc, ne, ni = b.counts()
print("mean nefc %.1f niter %.1f; mean row updates per step (nefc x sweeps) %.0f" % (ne.mean(), ni.mean(), (ne * ni).mean()))
for i, n in enumerate(names):
    print("%-24s %9.0f cycles  %5.1f %%" % (n, d[:, i].mean(), 100 * d[:, i].mean() / tot.mean()))
