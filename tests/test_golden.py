"""Oracle vs the committed golden one-step vectors (tests/golden/humanoid27_steps.npz, generated
by tools/make_golden.py from the oracle itself — a regression pin, see the script's header)."""
import os

import numpy as np

from oracle_lib import GOLDEN, Oracle


def test_oracle_reproduces_golden_steps():
    g = np.load(os.path.join(GOLDEN, "humanoid27_steps.npz"))
    o = Oracle()
    n = len(g["env"])
    assert n == 128
    for k in range(n):
        o.reset()
        o.qpos[:] = g["qpos"][k]; o.qvel[:] = g["qvel"][k]; o.qacc_warmstart[:] = g["warm"][k]; o.ctrl[:] = g["ctrl"][k]
        o.L.om_data_set_time(o.d, float(g["time"][k]))
        o.step()
        assert o.ncon == g["ncon"][k] and o.nefc == g["nefc"][k]
        assert np.allclose(o.qpos, g["qpos1"][k], rtol=0, atol=1e-12)
        assert np.allclose(o.qvel, g["qvel1"][k], rtol=0, atol=1e-10)
        assert np.allclose(o.efc_force[:o.nefc], g["efc_force"][k][:o.nefc], rtol=1e-9, atol=1e-9)


def test_golden_covers_the_interesting_phases():
    g = np.load(os.path.join(GOLDEN, "humanoid27_steps.npz"))
    assert (g["nefc"] == 0).sum() >= 10      # free flight
    assert (g["nefc"] >= 12).sum() >= 10     # contact rich
    assert g["ncon"].max() >= 5
    assert np.abs(g["qvel"]).max() > 5       # fast motion
    # controls are the Halton sequence of testspeed.cc:64-80 with the index convention of SURVEY.md §8(d)
    from oracle_lib import halton
    k = 17
    e, t = int(g["env"][k]), int(g["step"][k])
    want = [2 * halton(1 + t + 1000 * e, i + 2) - 1 for i in range(21)]
    assert np.allclose(g["ctrl"][k], want, atol=1e-15)
