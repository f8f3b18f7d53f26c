"""Env adapter on the GPU (hb_env_step / hb_env_reset) against a numpy restatement of
the reference's standupReward fed with oracle quantities."""
import numpy as np
import pytest

from env_ref import obs_from_state, standup_reward
from oracle_lib import Oracle

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("solver", [0, 2])
def test_env_step_matches_reference_reward(hbmod, humanoid_model, gpu, solver):
    """solver 0: the benchmark configuration (PGS); 2: Newton, what the reference's CPUEnv runs on its model file."""
    m = humanoid_model
    if solver == 2:
        from oracle_lib import HUMANOID_HBM
        m = hbmod.Model.load(HUMANOID_HBM)
        m.set_opt(solver=2, iterations=100)
    n = 12
    env = hbmod.VecEnv(m, n, gpu, auto_reset=0, max_time=0.0)
    cfg = env.cfg
    assert cfg.n_equal == 9 and cfg.n_opposite == 0  # nine right/left actuator pairs in the 27-DoF model
    obs = env.reset()
    b = env.batch
    st0 = b.get_state(hbmod.STATE_INTEGRATION, dtype=np.float64)
    for e in range(n):
        o_ref, _ = obs_from_state(st0[e, 1:1 + m.nq], st0[e, 1 + m.nq:1 + m.nq + m.nv])
        assert np.allclose(obs[e], o_ref, atol=1e-6)
    rng = np.random.default_rng(2)
    prev = np.zeros((n, m.nu))
    o = Oracle()
    if solver == 2:
        o.set_opt(solver=2, iterations=100)
    worst = 0.0
    for t in range(60):
        act = rng.uniform(-1.2, 1.2, size=(n, m.nu)).astype(np.float32)
        st = b.get_state(hbmod.STATE_INTEGRATION, dtype=np.float64)
        obs, rew, term, trunc, info = env.step_arrays(act)
        st1 = b.get_state(hbmod.STATE_INTEGRATION, dtype=np.float64)
        assert not term.any()
        for e in range(n):
            # oracle forward at the pre-step state gives the joint torques and the self-collision flag
            o.reset()
            o.qpos[:] = st[e, 1:1 + m.nq]; o.qvel[:] = st[e, 1 + m.nq:1 + m.nq + m.nv]; o.qacc_warmstart[:] = st[e, 1 + m.nq + m.nv:]
            o.ctrl[:] = act[e]
            o.forward()
            torques = (o.qfrc_smooth + o.qfrc_constraint)[6:]
            selfcol = any(c["geom1"] != 0 for c in o.contacts())
            q1, v1 = st1[e, 1:1 + m.nq], st1[e, 1 + m.nq:1 + m.nq + m.nv]
            r_ref, te, tr = standup_reward(cfg, st1[e, 0], q1, v1, torques, prev[e], act[e].astype(np.float64), selfcol)
            worst = max(worst, abs(rew[e] - r_ref))
            assert abs(rew[e] - r_ref) <= 2e-3 * max(1.0, abs(r_ref)), (t, e, rew[e], r_ref)
            assert bool(trunc[e]) == tr
            o_ref, _ = obs_from_state(q1, v1)
            assert np.allclose(obs[e], o_ref, atol=1e-5)
        prev = act.astype(np.float64)
    assert worst < 2e-2


def test_time_limit_terminates_and_auto_resets(hbmod, humanoid_model, gpu):
    m = humanoid_model
    n = 8
    env = hbmod.VecEnv(m, n, gpu, max_time=0.0249, target_z=10.0)  # 5 steps of 5 ms; success unreachable
    env.reset()
    zeros = np.zeros((n, m.nu), np.float32)
    for t in range(4):
        obs, rew, term, trunc, info = env.step_arrays(zeros)
        assert not term.any() and not trunc.any()
    q_before = env.batch.qpos
    obs, rew, term, trunc, info = env.step_arrays(zeros)
    assert term.all() and np.allclose(rew, -100.0)
    # reset in place: time back to zero, a fresh perturbed start, and the returned obs belongs to the new episode
    assert np.allclose(env.batch.time, 0.0)
    q = env.batch.qpos
    assert not np.allclose(q, q_before)
    assert np.abs(q[:, 7:] - 0).max() <= 0.2 + 1e-6 and (q[:, 2] >= 1.282 - 1e-6).all()
    assert np.allclose(obs[:, :21], q[:, 7:])
    # a second episode starts from a different perturbation than the first
    first = env.batch.qpos.copy()
    for t in range(5):
        obs, rew, term, trunc, info = env.step_arrays(zeros)
    assert term.all()
    assert not np.allclose(env.batch.qpos, first)


def test_success_truncation_and_randomization_factor(hbmod, humanoid_model, gpu):
    m = humanoid_model
    env = hbmod.VecEnv(m, 4, gpu, randomization_factor=0.0, max_time=0.0)
    env.reset()
    assert np.allclose(env.batch.qpos, m.array("qpos0").astype(np.float32)[None])  # no perturbation at factor 0
    obs, rew, term, trunc, info = env.step_arrays(np.zeros((4, m.nu), np.float32))
    assert trunc.all() and info["is_success"].all()  # standing upright above target height == success (reward_functions.py:371-372)
    env.set_attr("randomization_factor", 0.5)
    env.reset()
    q = env.batch.qpos
    assert 0 < np.abs(q[:, 7:]).max() <= 0.1 + 1e-6
    with pytest.raises(AttributeError):
        env.set_attr("nope", 1)
    with pytest.raises(AttributeError):
        hbmod.VecEnv(m, 2, gpu, not_a_parameter=1)


def test_device_resident_step_matches_host_step(hbmod, humanoid_model, gpu):
    """hb_env_step_dev through torch tensors (policy on the same GPU) == hb_env_step through host buffers."""
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("torch sees no GPU")
    m = humanoid_model
    n = 64
    a = hbmod.VecEnv(m, n, gpu, seed=3)
    b = hbmod.VecEnv(m, n, gpu, seed=3)
    oa, ob = a.reset(), b.reset()
    assert np.array_equal(oa, ob)
    rng = np.random.default_rng(0)
    for t in range(40):
        act = rng.uniform(-1, 1, size=(n, m.nu)).astype(np.float32)
        o1, r1, te1, tr1, _ = a.step_arrays(act)
        o2, r2, te2, tr2 = b.step_torch(torch.from_numpy(act).cuda())
        assert np.array_equal(o1, o2.cpu().numpy()) and np.array_equal(r1, r2.cpu().numpy())
        assert np.array_equal(te1, te2.cpu().numpy()) and np.array_equal(tr1, tr2.cpu().numpy())


def test_step_outputs_as_views_and_through_scattered_buffers(hbmod, humanoid_model, gpu):
    """hb_env_step moves the four outputs as ONE record when the caller's buffers lie back to back (engine.py's page-locked record) and as
    four transfers otherwise; VecEnv.copy_outputs = False hands out views of that record.  All three give the same numbers."""
    import ctypes
    m = humanoid_model
    n = 96
    envs = [hbmod.VecEnv(m, n, gpu, seed=5, max_time=0.2) for _ in range(3)]  # short episodes: resets and flags inside the 60 steps
    envs[1].copy_outputs = False
    obs0 = [e.reset() for e in envs]
    assert np.array_equal(obs0[0], obs0[1]) and np.array_equal(obs0[0], obs0[2])
    L = hbmod.lib()
    o3 = np.zeros((n, m.nobs), np.float32); r3 = np.zeros(n, np.float32); te3 = np.zeros(n, np.uint8); tr3 = np.zeros(n, np.uint8)  # four separate arrays
    rng = np.random.default_rng(1)
    flags = 0
    for t in range(60):
        act = rng.uniform(-1, 1, size=(n, m.nu)).astype(np.float32)
        o1, r1, te1, tr1, _ = envs[0].step_arrays(act)
        o2, r2, te2, tr2, _ = envs[1].step_arrays(act)
        assert L.hb_env_step(envs[2].batch._h, act.ctypes.data_as(ctypes.c_void_p), 1, o3.ctypes.data_as(ctypes.c_void_p), r3.ctypes.data_as(ctypes.c_void_p),
                             te3.ctypes.data_as(ctypes.c_void_p), tr3.ctypes.data_as(ctypes.c_void_p)) == 0
        assert np.array_equal(o1, o2) and np.array_equal(r1, r2) and np.array_equal(te1, te2) and np.array_equal(tr1, tr2)
        assert te2.dtype == bool and not o2.flags.owndata  # views
        assert np.array_equal(o1, o3) and np.array_equal(r1, r3) and np.array_equal(te1, te3.astype(bool)) and np.array_equal(tr1, tr3.astype(bool))
        flags += int(te1.sum() + tr1.sum())
    assert flags > 0
    for e in envs:
        e.close()


def test_step_async_then_step_wait_equals_step(hbmod, humanoid_model, gpu):
    """stable-baselines3's VecEnv pair: step_async enqueues (hb_env_step_async: no wait), the host does something else, step_wait
    returns what step would have."""
    m = humanoid_model
    n = 80
    a = hbmod.VecEnv(m, n, gpu, seed=9, max_time=0.25)
    b = hbmod.VecEnv(m, n, gpu, seed=9, max_time=0.25)
    assert np.array_equal(a.reset(), b.reset())
    rng = np.random.default_rng(2)
    for t in range(70):
        act = rng.uniform(-1, 1, size=(n, m.nu)).astype(np.float32)
        o1, r1, te1, tr1, i1 = a.step_arrays(act)
        b.step_async(act)
        _ = float(np.linalg.norm(rng.normal(size=(64, 64)) @ rng.normal(size=(64, 64))))  # the host is free in between
        o2, r2, d2, i2 = b.step_wait()  # (stable-baselines3's tuple: obs, rewards, dones, infos - a list of dicts)
        assert np.array_equal(o1, o2) and np.array_equal(r1, r2) and np.array_equal(te1 | tr1, d2)
        assert np.array_equal(i1["done"], d2) and [("terminal_observation" in x) for x in i2] == list(d2)
    with pytest.raises(AssertionError):
        b.step_wait()
    a.close(); b.close()


def test_vecenv_reports_warning_bits(hbmod, humanoid_model, gpu):
    """The step's info carries the per-env HB_WARN_* bits (mjData.warning): an overflow or a bad-state reset of an env is
    visible to the training loop.  A NaN planted in one env's state shows up as BADQPOS for that env only."""
    n = 6
    env = hbmod.VecEnv(humanoid_model, n, gpu, auto_reset=0, max_time=0.0)
    env.warning_period = 1  # read the bits back at every step (the default refreshes them every 16th: they are sticky)
    env.reset()
    act = np.zeros((n, humanoid_model.nu), np.float32)
    _, _, _, _, info = env.step_arrays(act)
    assert info["warnings"].shape == (n,) and not info["warnings"].any() and not info["overflow"].any()
    st = env.batch.get_state(hbmod.STATE_INTEGRATION)
    st[4, 3] = np.nan
    env.batch.set_state(hbmod.STATE_INTEGRATION, st)
    obs, rew, _, _, info = env.step_arrays(act)
    assert info["warnings"][4] & hbmod.WARN_BADQPOS and not np.delete(info["warnings"], 4).any()
    assert np.isfinite(obs).all() and np.isfinite(rew).all()
    assert env.warning_counts() == {"contact_full": 0, "constraint_full": 0, "bad_qpos": 1, "bad_qvel": 0, "bad_qacc": 0}
    env.close()
    # an episode that raises a warning and ENDS (reset in place) before the next poll: hb_get_status has been cleared by the reset, the
    # poll still reports the bit - once
    env = hbmod.VecEnv(humanoid_model, n, gpu, max_time=2.5 * humanoid_model.opt.timestep, target_z=10.0)  # three-step episodes
    env.warning_period = 8
    env.reset()
    env.step_arrays(act)  # (the first step polls)
    st = env.batch.get_state(hbmod.STATE_INTEGRATION)
    st[2, 3] = np.nan
    env.batch.set_state(hbmod.STATE_INTEGRATION, st)
    for t in range(6):
        _, _, _, _, info = env.step_arrays(act)  # steps 2 .. 7: the planted env is flagged at step 2 and reset (time limit) soon after
        assert not info["warnings"].any()        # (not polled yet)
    assert not (env.batch.status() & hbmod.WARN_BADQPOS).any()  # the reset has cleared the sticky word
    _, _, _, _, info = env.step_arrays(act)      # step 8: the poll
    assert info["warnings"][2] & hbmod.WARN_BADQPOS and not np.delete(info["warnings"], 2).any()
    for t in range(8):
        _, _, _, _, info = env.step_arrays(act)
    assert not info["warnings"].any()            # reported once
    env.close()
