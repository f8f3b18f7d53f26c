"""The kernels the benchmark TIMES, on the golden states.

tests/test_gpu_parity.py::test_one_step_parity_on_golden_states and tests/test_gpu_newton.py enable the diagnostic read-outs (qacc,
efc_force, contacts), and a launch with any optional output takes the full instantiation of the step kernel.  The plain step API - what
bench.py, tools/hb_testspeed.cpp and a training loop run - takes the lean, size-specialised instantiations.  Here the
same 128 golden states (oracle/mjstep_oracle.c via tools/make_golden.py; tolerances as stated in tests/test_gpu_parity.py) go through
the plain step, and hb_last_kernel says which kernel that was:
  PGS     hb_step_h27_kernel (one env per wave; step calls of batches up to 2.5 x the chip's wave slots) and hb_step_duo_kernel (two
          envs per wave: tests/test_gpu_duo.py has that one's own tests).  bench.py's timed loop itself runs hb_step_duo_q_kernel - the
          same kernel body with the step loop inside, which the library launches for step calls enqueued back to back -: held
          bit-identical to hb_step_duo_kernel launch by launch in tests/test_gpu_fold.py, and run on the golden states below as a
          one-step launch through the rollout API
  Newton  hb_step_newton28_h27_kernel
The staged step's fast kernels are named in the tests that already run them without diagnostics: hb_step_gen_fast_h27_kernel
(tests/test_gpu_parity.py::test_heightfield_terrain_humanoid_config5) and hb_step_newton_gen20_team_kernel
(tests/test_gpu_convex.py::test_team_robot_staged_fast_pass_against_the_oracle)."""
import os

import numpy as np
import pytest

from oracle_lib import GOLDEN, HUMANOID_HBM, Oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def golden():
    return np.load(os.path.join(GOLDEN, "humanoid27_steps.npz"))


def state_of(g):
    return np.concatenate([g["time"][:, None], g["qpos"], g["qvel"], g["warm"]], axis=1)


@pytest.mark.parametrize("duo,kernel", [(0, "hb_step_h27_kernel"), (2, "hb_step_duo_kernel"), (2, "hb_step_duo_q_kernel")])
def test_pgs_golden_states_through_the_plain_step(hbmod, humanoid_model, gpu, golden, duo, kernel):
    g = golden
    n = len(g["env"])
    b = hbmod.Batch(humanoid_model, n, gpu)
    b.tune(duo=duo)
    b.set_state(hbmod.STATE_INTEGRATION, state_of(g))
    if kernel.endswith("_q_kernel"):  # the multi-step kernel: two steps' worth of controls, the launch cut off after the first
        c = np.concatenate([g["ctrl"], g["ctrl"]]).astype(np.float32)
        p = b.dev_alloc(c.nbytes)
        b.to_dev(p, c)
        b.pipeline(True)
        b.step_dev(p)
        b.step_dev(p + c.nbytes // 2)  # (two calls back to back: folded into one launch of the _q kernel ...)
        b.sync()
        assert b.last_kernel() == kernel
        # ... whose SECOND step starts from the first one's result: compare the first step through a second batch that stops there
        b2 = hbmod.Batch(humanoid_model, n, gpu)
        b2.tune(duo=duo)
        b2.set_state(hbmod.STATE_INTEGRATION, state_of(g))
        b2.step(g["ctrl"].astype(np.float32))
        b2.step(g["ctrl"].astype(np.float32))
        assert np.array_equal(b.get_state(hbmod.STATE_INTEGRATION), b2.get_state(hbmod.STATE_INTEGRATION))
        b.dev_free(p); b.close()
        b = hbmod.Batch(humanoid_model, n, gpu)
        b.tune(duo=duo)
        b.set_state(hbmod.STATE_INTEGRATION, state_of(g))
        b.step(g["ctrl"].astype(np.float32))
        kernel = "hb_step_duo_kernel"
        b2.close()
    else:
        b.step(g["ctrl"].astype(np.float32))
    assert b.last_kernel() == kernel
    q, v = b.qpos.astype(np.float64), b.qvel.astype(np.float64)
    a = b.get_state(hbmod.STATE_WARMSTART).astype(np.float64)  # qacc_warmstart of the new state = the step's qacc
    ncon, nefc, niter = b.counts()
    assert not b.status().any()
    assert np.array_equal(ncon, g["ncon"]) and np.array_equal(nefc, g["nefc"])
    # sweep counts are the oracle's except where its convergence test sits within rounding of the threshold
    off = niter != g["niter"]
    assert off.sum() <= 2 and np.abs(niter - g["niter"]).max() <= 1, (niter[off], g["niter"][off])
    assert (np.abs(q - g["qpos1"]) / np.maximum(1.0, np.abs(g["qpos1"]))).max() <= 4e-5
    assert (np.abs(v - g["qvel1"]) / np.maximum(1.0, np.abs(g["qvel1"]).max(axis=1, keepdims=True))).max() <= 4e-4
    assert (np.abs(a - g["qacc"]) / np.maximum(1.0, np.abs(g["qacc"]).max(axis=1, keepdims=True))).max() <= 4e-4
    assert np.allclose(b.time, g["time"] + 0.005, atol=1e-5)
    b.close()


def test_newton_golden_states_through_the_plain_step(hbmod, gpu, golden):
    g = golden
    n = len(g["env"])
    m = hbmod.Model.load(HUMANOID_HBM)
    m.set_opt(solver=2, iterations=100)
    b = hbmod.Batch(m, n, gpu)
    b.set_state(hbmod.STATE_INTEGRATION, state_of(g))
    b.step(g["ctrl"].astype(np.float32))
    assert b.last_kernel() == "hb_step_newton28_h27_kernel"
    q, v = b.qpos.astype(np.float64), b.qvel.astype(np.float64)
    a = b.get_state(hbmod.STATE_WARMSTART).astype(np.float64)
    ncon, nefc, niter = b.counts()
    assert not b.status().any()
    o = Oracle()
    o.set_opt(solver=2, iterations=100)
    worst = dict(qpos=0.0, qvel=0.0, qacc=0.0)
    it_o = np.zeros(n, int)
    for k in range(n):
        o.qpos[:] = g["qpos"][k]; o.qvel[:] = g["qvel"][k]; o.qacc_warmstart[:] = g["warm"][k]; o.ctrl[:] = g["ctrl"][k]
        o.step()
        assert (o.ncon, o.nefc) == (ncon[k], nefc[k])
        it_o[k] = o.dint("solver_niter")
        worst["qpos"] = max(worst["qpos"], (np.abs(q[k] - o.qpos) / np.maximum(1.0, np.abs(o.qpos))).max())
        worst["qvel"] = max(worst["qvel"], np.abs(v[k] - o.qvel).max() / max(1.0, np.abs(o.qvel).max()))
        worst["qacc"] = max(worst["qacc"], np.abs(a[k] - o.qacc_warmstart).max() / max(1.0, np.abs(o.qacc_warmstart).max()))
    print("\nnewton, plain step: worst", worst, "iterations device mean %.2f max %d, oracle mean %.2f max %d" % (niter.mean(), niter.max(), it_o.mean(), it_o.max()))
    assert worst["qpos"] <= 4e-5 and worst["qvel"] <= 4e-4 and worst["qacc"] <= 4e-4, worst
    # Newton stops on a gradient norm: fp32 reaches the floor of that test an iteration or two away from fp64 on a few states
    assert niter.max() <= 30 and np.abs(niter - it_o).max() <= 3 and (niter != it_o).mean() <= 0.25, (np.abs(niter - it_o).max(), (niter != it_o).mean())
    b.close()


@pytest.mark.parametrize("order", [0, 1])
def test_one_step_parity_in_both_contact_orders(hbmod, gpu, tmp_path, order):
    """hb_model_pair_order: body-pair-major (1: what the compiler writes, MuJoCo's collision-driver structure) and geom-major (0: rounds 1-3)
    contact order.  The device mirrors whichever the model carries: one step from oracle states of the benchmark's steady regime (fallen
    humanoids: self collisions, several bodies on the floor) against the oracle on the same ordering - contacts in the same ORDER (geom
    ids row by row), same counts, qacc / forces to the one-step tolerances.  (profiles/r04_contact_order_modes.txt: how rarely the two
    orders differ on this model, and by how much.)"""
    m = hbmod.Model.load(HUMANOID_HBM)
    assert m.pair_order() == 1 and m.pair_order(order) == order
    p = str(tmp_path / ("order%d.hbm" % order))
    m.save(p)
    o = Oracle(p)
    states, ctrls = [], []
    for e in (1, 4, 6):
        o.init_env(e)
        for t in range(700):
            c = o.ctrl_env(t, e)
            o.ctrl[:] = c
            if t >= 300 and t % 8 == 0:
                states.append(np.concatenate([[o.time], o.qpos, o.qvel, o.qacc_warmstart])); ctrls.append(c.copy())
            o.step()
    n = len(states)
    st = np.array(states).astype(np.float32).astype(np.float64)
    ct = np.array(ctrls, dtype=np.float32)
    b = hbmod.Batch(m, n, gpu)
    b.diag_enable(True)
    b.set_state(hbmod.STATE_INTEGRATION, st)
    b.step(ct)
    a, f, con = b.qacc().astype(np.float64), b.efc_force().astype(np.float64), b.contacts()
    ncon, nefc, _ = b.counts()
    multi = 0
    for k in range(n):
        o.qpos[:] = st[k, 1:1 + m.nq]; o.qvel[:] = st[k, 1 + m.nq:1 + m.nq + m.nv]; o.qacc_warmstart[:] = st[k, 1 + m.nq + m.nv:]; o.ctrl[:] = ct[k]
        o.forward()
        cs = o.contacts()
        assert (o.ncon, o.nefc) == (ncon[k], nefc[k])
        assert [(int(con[k, i, 14]), int(con[k, i, 15])) for i in range(o.ncon)] == [(c["geom1"], c["geom2"]) for c in cs]
        multi += int(o.ncon >= 4)
        assert np.abs(a[k] - o.qacc).max() / max(1.0, np.abs(o.qacc).max()) <= 4e-4
        if o.nefc:
            fo = o.efc_force[:o.nefc]
            assert np.abs(f[k, :o.nefc] - fo).max() / max(1.0, np.abs(fo).max()) <= 4e-4
    assert multi >= 20
    b.close()
