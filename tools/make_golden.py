#!/usr/bin/env python3
"""Generate tests/golden/humanoid27_steps.npz — one-step input/output vectors of the hot path.

PROVENANCE: produced by THIS repository's fp64 oracle (oracle/mjstep_oracle.c), not by MuJoCo:
the reference's engine is not available here ("parity unpinned").  The file pins
the oracle against regressions and gives the GPU parity tests teacher-forced states that cover
free flight, first touch-down, and contact-rich phases of the benchmark workload (SURVEY.md §8d:
Halton controls, Halton-perturbed initial states).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle_lib import GOLDEN, Oracle  # noqa: E402

ENVS = [0, 1, 2, 3, 5, 8, 13, 21]
STEPS = [0, 1, 20, 60, 90, 120, 150, 180, 210, 250, 300, 400, 500, 650, 800, 999]


def main():
    o = Oracle()
    rec = {k: [] for k in ("env", "step", "time", "qpos", "qvel", "warm", "ctrl", "qpos1", "qvel1", "qacc", "ncon", "nefc", "niter",
                           "efc_force", "con_dist", "con_pos", "con_frame", "con_geom")}
    for e in ENVS:
        o.init_env(e)
        for t in range(max(STEPS) + 1):
            c = o.ctrl_env(t, e)
            o.ctrl[:] = c
            if t in STEPS:
                rec["env"].append(e); rec["step"].append(t); rec["time"].append(o.time)
                rec["qpos"].append(o.qpos.copy()); rec["qvel"].append(o.qvel.copy()); rec["warm"].append(o.qacc_warmstart.copy())
                rec["ctrl"].append(c.copy())
            o.step()
            if t in STEPS:
                rec["qpos1"].append(o.qpos.copy()); rec["qvel1"].append(o.qvel.copy()); rec["qacc"].append(o.qacc.copy())
                rec["ncon"].append(o.ncon); rec["nefc"].append(o.nefc); rec["niter"].append(o.dint("solver_niter"))
                f = np.zeros(63); n = min(o.nefc, 63); f[:n] = o.efc_force[:n]
                rec["efc_force"].append(f)
                cd, cp, cf, cg = np.zeros(24), np.zeros((24, 3)), np.zeros((24, 9)), np.zeros((24, 2))
                for k, cc in enumerate(o.contacts()[:24]):
                    cd[k] = cc["dist"]; cp[k] = cc["pos"]; cf[k] = cc["frame"].ravel(); cg[k] = (cc["geom1"], cc["geom2"])
                rec["con_dist"].append(cd); rec["con_pos"].append(cp); rec["con_frame"].append(cf); rec["con_geom"].append(cg)
    out = {k: np.array(v) for k, v in rec.items()}
    os.makedirs(GOLDEN, exist_ok=True)
    path = os.path.join(GOLDEN, "humanoid27_steps.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: v.shape for k, v in out.items()})
    print("nefc range", out["nefc"].min(), out["nefc"].max(), "ncon range", out["ncon"].min(), out["ncon"].max())


if __name__ == "__main__":
    main()
