"""GPU parity of the Newton solver instantiation (hb_options.solver = 2, mjSOL_NEWTON: the reference's default solver)
against the fp64 oracle's Newton solver, through the C-ABI.

Tolerances (fp32 device vs fp64 oracle; the problem is strictly convex, so both converge to the same qacc and only the
fp32 floor of the device's termination test separates them):
  one step, teacher-forced from the golden states:  qacc 4e-4 * max(1, max|qacc|), efc_force 4e-4 * max(1, max|f|),
      qpos 4e-5 relative, qvel 4e-4 * max(1, max|qvel|) (each at most 3x the maxima of profiles/r01_parity_report_newton.txt); counts (ncon, nefc) identical
  free-running through contacts for 60 steps: qpos within 5e-4 for as long as both sides see the same constraint rows
      (Newton has no sweep-count truncation, so it tracks the oracle through contacts; a differing contact set is a
      discrete event after which the trajectories are different experiments).
"""
import os

import numpy as np
import pytest

from oracle_lib import GOLDEN, HUMANOID_HBM, Oracle

pytestmark = pytest.mark.gpu
SOL_NEWTON = 2


@pytest.fixture(scope="module")
def newton_model(hbmod):
    m = hbmod.Model.load(HUMANOID_HBM)
    m.set_opt(solver=SOL_NEWTON, iterations=100)
    return m


def newton_oracle():
    o = Oracle()
    o.set_opt(solver=SOL_NEWTON, iterations=100)
    return o


def test_options_roundtrip(hbmod, newton_model):
    o = newton_model.opt
    assert o.solver == SOL_NEWTON and o.iterations == 100 and o.ls_iterations == 50 and abs(o.ls_tolerance - 0.01) < 1e-12
    m = hbmod.Model.load(HUMANOID_HBM)
    with pytest.raises(Exception):
        m.set_opt(solver=1)  # CG is not implemented


def test_one_step_parity_on_golden_states(hbmod, newton_model, gpu):
    g = np.load(os.path.join(GOLDEN, "humanoid27_steps.npz"))
    n = len(g["env"])
    b = hbmod.Batch(newton_model, n, gpu)
    b.diag_enable(True)
    st = np.concatenate([g["time"][:, None], g["qpos"], g["qvel"], g["warm"]], axis=1)
    b.set_state(hbmod.STATE_INTEGRATION, st)
    b.step(g["ctrl"].astype(np.float32))
    q, v, a, f = b.qpos.astype(np.float64), b.qvel.astype(np.float64), b.qacc().astype(np.float64), b.efc_force().astype(np.float64)
    ncon, nefc, niter = b.counts()
    assert not b.status().any()
    o = newton_oracle()
    worst = dict(qacc=0.0, force=0.0, qpos=0.0, qvel=0.0)
    it_o = []
    for k in range(n):
        o.qpos[:] = g["qpos"][k]; o.qvel[:] = g["qvel"][k]; o.qacc_warmstart[:] = g["warm"][k]; o.ctrl[:] = g["ctrl"][k]
        o.forward()
        assert o.ncon == ncon[k] and o.nefc == nefc[k]
        it_o.append(o.dint("solver_niter"))
        worst["qacc"] = max(worst["qacc"], np.abs(a[k] - o.qacc).max() / max(1.0, np.abs(o.qacc).max()))
        ne = o.nefc
        if ne:
            worst["force"] = max(worst["force"], np.abs(f[k, :ne] - o.efc_force[:ne]).max() / max(1.0, np.abs(o.efc_force[:ne]).max()))
        o.step()
        worst["qpos"] = max(worst["qpos"], (np.abs(q[k] - o.qpos) / np.maximum(1.0, np.abs(o.qpos))).max())
        worst["qvel"] = max(worst["qvel"], np.abs(v[k] - o.qvel).max() / max(1.0, np.abs(o.qvel).max()))
    print("newton one-step worst:", worst, "iterations gpu mean/max", niter.mean(), niter.max(), "oracle", np.mean(it_o), np.max(it_o))
    assert worst["qacc"] <= 4e-4 and worst["force"] <= 4e-4 and worst["qpos"] <= 4e-5 and worst["qvel"] <= 4e-4, worst
    assert niter.max() <= 30


def test_free_running_tracks_oracle_through_contacts(hbmod, newton_model, gpu):
    """Free-running GPU vs oracle.  An env is compared for as long as both sides see the same constraint rows: once
    fp32 and fp64 disagree on whether a geom pair is within its margin (a discrete event), the trajectories are
    different experiments (parity)."""
    envs = list(range(8))
    b = hbmod.Batch(newton_model, len(envs), gpu)
    b.reset(perturb=True)
    oracles = []
    for e in envs:
        o = newton_oracle()
        o.init_env(e)
        oracles.append(o)
    alive = [True] * len(envs)
    worst, contact_steps = 0.0, 0
    for t in range(60):
        ctrl = np.stack([o.ctrl_env(t, e) for o, e in zip(oracles, envs)]).astype(np.float32)
        b.step(ctrl)
        for o, c in zip(oracles, ctrl):
            o.ctrl[:] = c
            o.step()
        q = b.qpos
        _, nefc, _ = b.counts()
        for i, o in enumerate(oracles):
            alive[i] = alive[i] and int(nefc[i]) == o.nefc
            if alive[i]:
                contact_steps += o.nefc > 0
                worst = max(worst, float(np.abs(q[i] - o.qpos).max()))
    print("newton free-running worst |dqpos| while the constraint sets agree:", worst, "env-steps with constraints:", contact_steps, "envs still in step:", sum(alive))
    assert contact_steps > 100
    assert worst <= 5e-4, worst


MODELS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "models")


@pytest.mark.parametrize("name,steps", [("ball_plane", 400), ("capsules", 300), ("chain", 300), ("pendulum_limit", 500), ("maxsize", 600)])
def test_other_models_one_step_parity_along_oracle_trajectory(hbmod, gpu, tmp_path, name, steps):
    """Small models (dense order padded to 28: elimination on the matrix cores) and the nv = 32 model (the order-32
    instantiation: Cholesky in registers), teacher-forced one-step parity along an oracle rollout."""
    m = hbmod.Model.load(os.path.join(MODELS, name + ".xml"))
    m.set_opt(solver=SOL_NEWTON, iterations=100)
    p = str(tmp_path / (name + ".hbm"))
    m.save(p)
    o = Oracle(p)
    assert o.opt("solver") == SOL_NEWTON
    o.reset(0 if name == "chain" else -1)
    rng = np.random.default_rng(11)
    states, ctrls, outs = [], [], []
    for t in range(steps):
        c = rng.uniform(-1, 1, size=max(o.nu, 1))[:o.nu]
        o.ctrl[:] = c
        if t % 10 == 0:
            states.append(np.concatenate([[o.time], o.qpos, o.qvel, o.qacc_warmstart]))
            ctrls.append(c.copy())
        o.step()
        if t % 10 == 0:
            outs.append((o.qpos.copy(), o.qvel.copy(), o.ncon, o.nefc))
    n = len(states)
    b = hbmod.Batch(m, n, gpu)
    b.set_state(hbmod.STATE_INTEGRATION, np.array(states))
    b.step(np.array(ctrls, dtype=np.float32).reshape(n, m.nu))
    q, v = b.qpos.astype(np.float64), b.qvel.astype(np.float64)
    nc, ne, ni = b.counts()
    assert not b.status().any()
    assert ni.max() <= 30
    for k, (qo, vo, nco, neo) in enumerate(outs):
        assert (nc[k], ne[k]) == (nco, neo), (k, nc[k], ne[k], nco, neo)
        assert (np.abs(q[k] - qo) / np.maximum(1, np.abs(qo))).max() <= 1e-4
        assert np.abs(v[k] - vo).max() <= 1e-3 * max(1.0, np.abs(vo).max())


def test_rollout_is_sane_and_matches_oracle_statistics(hbmod, newton_model, gpu):
    n, T = 256, 600
    b = hbmod.Batch(newton_model, n, gpu)
    b.reset(perturb=True)
    b.rollout_halton(T)
    b.sync()
    q = b.qpos
    assert np.isfinite(q).all() and np.isfinite(b.qvel).all()
    assert not b.status().any()
    assert np.allclose(np.linalg.norm(q[:, 3:7], axis=1), 1.0, atol=1e-4)
    assert q[:, 2].min() > -0.02 and q[:, 2].max() < 1.0
    o = newton_oracle()
    _, qo, st = o.rollout_threads(n, T, os.cpu_count() or 4, 0, True)
    assert abs(q[:, 2].mean() - qo[:, 2].mean()) < 0.02
    nc, ne, ni = b.counts()
    assert abs(ne.mean() - st["mean_nefc"]) < 4.0


def test_pipelined_and_rollout_are_bit_identical_to_steps(hbmod, newton_model, gpu):
    n, T = 512, 12
    ref = hbmod.Batch(newton_model, n, gpu)
    ref.reset(perturb=True)
    for t in range(T):
        ref.rollout_halton(1, t0=t)
    a = ref.get_state(hbmod.STATE_INTEGRATION)
    b = hbmod.Batch(newton_model, n, gpu)
    b.pipeline(1)
    b.reset(perturb=True)
    b.rollout_halton(T)
    b.sync()
    assert np.array_equal(a, b.get_state(hbmod.STATE_INTEGRATION))
