#!/usr/bin/env python3
"""Compiles the reference's own robot (simulation/assets/world.xml + humanoid.xml + its STL meshes) into
humanoid_mujoco_amd/assets/team_robot.hbm, with the timestep CPUEnv sets (0.002 s, cpu_env.py:87) and two keyframes the
MJCF does not have but the env's reset uses (simulation_parameters.py:64-77):
  standup_reset   root at (0, 0, Z_INITIAL_POS_STANDUP = -0.6), INITIAL_QUAT_STANDUP = (-.5, -.5, .5, .5): lying on the floor
  standing_reset  qpos0 (Z_INITIAL_POS = -0.375, INITIAL_QUAT = (-0.707, 0, 0, 0.707))
Needs /root/reference (the compiled model is committed; the hulls in it are data derived from the reference's STL files)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = "/root/reference/simulation/assets/world.xml"
DST = os.path.join(ROOT, "humanoid_mujoco_amd", "assets", "team_robot.hbm")
# the same robot on the reference's type="plane" floor (simulation/assets/green_screen_world.xml: what simulation/__init__.py exports and
# rl/generate_policy_videos.py builds CPUEnv on): plane - hull contacts (mjc_PlaneConvex) instead of the height field's prisms
SRC_PLANE = "/root/reference/simulation/assets/green_screen_world.xml"
DST_PLANE = os.path.join(ROOT, "humanoid_mujoco_amd", "assets", "team_robot_plane.hbm")


def main():
    build(SRC, DST)
    build(SRC_PLANE, DST_PLANE)


def build(SRC, DST):
    subprocess.check_call([os.path.join(ROOT, "build", "hb_compile"), SRC, DST, "--timestep", "0.002"])
    lines = open(DST).read().splitlines()
    rec = {ln.split()[1]: ln for ln in lines if len(ln.split()) > 2}
    nq = int(rec["nq"].split()[2])
    qpos0 = [float(x) for x in rec["qpos0"].split()[3:]]
    lying = list(qpos0)
    lying[0:3] = [0.0, 0.0, -0.6]
    lying[3:7] = [-0.5, -0.5, 0.5, 0.5]
    keys = lying + qpos0
    out = []
    for ln in lines:
        t = ln.split()
        if len(t) > 1 and t[1] == "nkey":
            ln = "i nkey 2"
        elif len(t) > 1 and t[1] == "key_qpos":
            ln = "D key_qpos %d %s" % (2 * nq, " ".join("%.17g" % v for v in keys))
        elif len(t) > 1 and t[1] == "key_name":
            ln = "S key_name 2 standup_reset standing_reset"
        out.append(ln)
    open(DST, "w").write("\n".join(out) + "\n")
    print("wrote", DST)


if __name__ == "__main__":
    main()
