"""Two envs per wave (hb_step_duo_kernel, csrc/hb_step_duo.hip): the kernel the plain step API - and with it bench.py's timed loop - runs
for the 27-dof humanoid with the PGS solver.

tests/test_gpu_timed_kernels.py puts the 128 golden states of the fp64 oracle through it (and through the one-env kernel), by name.
Here the duo kernel is held bit-identical to the one-env kernels on every path it has: two envs packed at lanes 0 / 32, a heavy env packed in front of a light one, one env at a time (more than 12
contacts, rows beyond the packed capacity), an odd env count, the heavy-first order, pipelined segments, the reset paths of mj_check*.
"""
import os

import numpy as np
import pytest

from oracle_lib import GOLDEN, HUMANOID_HBM

pytestmark = pytest.mark.gpu
DUO = "hb_step_duo_kernel"


@pytest.fixture(scope="module")
def golden():
    return np.load(os.path.join(GOLDEN, "humanoid27_steps.npz"))


def pack_state(g, idx):
    return np.concatenate([g["time"][idx, None], g["qpos"][idx], g["qvel"][idx], g["warm"][idx]], axis=1)


def _reference_step(hbmod, m, gpu, state, ctrl):
    """the same step through the full one-env kernel (a launch that asks for the diagnostics takes hb_step_kernel)"""
    r = hbmod.Batch(m, len(state), gpu)
    r.diag_enable(True)
    r.set_state(hbmod.STATE_INTEGRATION, state)
    r.step(ctrl)
    assert r.last_kernel() == "hb_step_kernel"
    out = r.get_state(hbmod.STATE_INTEGRATION), r.counts(), r.status()
    r.close()
    return out


def test_duo_is_bit_identical_to_the_one_env_kernel(hbmod, humanoid_model, gpu):
    """benchmark workload, fallen humanoids: single steps through the duo kernel against the full kernel from the same states, for an
    even and an odd env count, with and without the heavy-first order, unpipelined and in three segments"""
    m = humanoid_model
    rng = np.random.default_rng(7)
    for n, pipe in ((512, False), (385, False), (1024, True)):
        b = hbmod.Batch(m, n, gpu)
        b.duo(2)
        b.reset(perturb=True)
        b.rollout_halton(300)
        assert b.last_kernel() == "hb_step_duo_q_kernel"
        if pipe:
            b.pipeline(3)
        for t in range(12):  # (the heavy-first order is re-sorted every fourth call)
            st = b.get_state(hbmod.STATE_INTEGRATION)
            ctrl = rng.uniform(-1, 1, size=(n, m.nu)).astype(np.float32)
            b.step(ctrl)
            assert b.last_kernel() == DUO
            want, wcounts, wstatus = _reference_step(hbmod, m, gpu, st, ctrl)
            got = b.get_state(hbmod.STATE_INTEGRATION)
            assert np.array_equal(got, want), (n, t, np.abs(got - want).max())
            for x, y in zip(b.counts(), wcounts):
                assert np.array_equal(x, y)
        b.close()


def partners(n, anti=16):
    """env -> the env it shares its wave with (-1: none), as hb_step_duo_kernel pairs the dispatch slots of an n-env launch without a
    heavy-first order: the first `anti` slots with the last ones, the others with their neighbour (csrc/hb_step_duo.hip: kDuoAnti)"""
    k = min(anti, n >> 2)
    p = np.full(n, -1)
    for e in range(n):
        if e < k or e >= n - k:
            p[e] = n - 1 - e
        else:
            q = e + 1 if (e - k) % 2 == 0 else e - 1
            p[e] = q if q < n - k else -1
    return p


def _collapsed_states(hbmod, m, gpu, n):
    """the collapsed regime of profiles/r01_soak_1e10.txt: humanoids dropped from a crouch with their limbs driven into the floor - many
    contacts and limit rows at once"""
    b = hbmod.Batch(m, n, gpu)
    b.reset(perturb=True)
    rng = np.random.default_rng(11)
    b.rollout_halton(150)
    ctrl = np.sign(rng.uniform(-1, 1, size=(n, m.nu))).astype(np.float32)  # saturated actuators
    for _ in range(60):
        b.step(ctrl)
    st = b.get_state(hbmod.STATE_INTEGRATION)
    b.close()
    return st, ctrl


def test_heavy_env_steps_take_the_packed_and_the_one_at_a_time_paths(hbmod, humanoid_model, gpu):
    """env-steps above 31 rows: packed in front of their partner's rows when both fit the wave's 64 row lanes, else one env at a time -
    bit-identical to the one-env kernel either way.  The states are chosen so that both paths occur (asserted on the row counts)."""
    m = humanoid_model
    n = 2048
    st, ctrl = _collapsed_states(hbmod, m, gpu, n)
    want, (ncon, nefc, niter), wstatus = _reference_step(hbmod, m, gpu, st, ctrl)
    heavy = nefc > 31
    pidx = partners(n)
    partner = np.where(pidx >= 0, nefc[pidx], 0)
    packed = heavy & (((np.maximum(nefc, partner) + 1 + 3) & ~3) + np.minimum(nefc, partner) + 1 <= 64) & (partner <= 31)
    alone = (nefc + partner + 2 > 64) | ((nefc > 31) & (partner > 31))
    print("\nrows: max %d, env-steps above 31 rows %d (packed with their partner %d), waves stepped one env at a time %d, max contacts %d"
          % (nefc.max(), heavy.sum(), packed.sum(), alone.sum() // 2, ncon.max()))
    assert heavy.sum() >= 8 and packed.sum() >= 1
    b = hbmod.Batch(m, n, gpu)
    b.duo(2)
    b.set_state(hbmod.STATE_INTEGRATION, st)
    b.step(ctrl)
    assert b.last_kernel() == DUO
    got = b.get_state(hbmod.STATE_INTEGRATION)
    bad = np.flatnonzero((got != want).any(axis=1))
    assert bad.size == 0, (bad[:8], nefc[bad[:8]], partner[bad[:8]])
    for x, y in zip(b.counts(), (ncon, nefc, niter)):
        assert np.array_equal(x, y)
    assert np.array_equal(b.status(), wstatus)
    # a heavy env beside a heavy env: force the one-env-at-a-time path by pairing the heaviest envs with each other
    idx = np.argsort(-nefc)[:128]  # (whatever the pairing: every env's partner is one of the heaviest)
    pair_state, pair_ctrl = st[idx], ctrl[idx]
    want2, counts2, _ = _reference_step(hbmod, m, gpu, pair_state, pair_ctrl)
    c = hbmod.Batch(m, len(pair_state), gpu)
    c.duo(2)
    c.set_state(hbmod.STATE_INTEGRATION, pair_state)
    c.step(pair_ctrl)
    assert c.last_kernel() == DUO
    assert np.array_equal(c.get_state(hbmod.STATE_INTEGRATION), want2)
    for x, y in zip(c.counts(), counts2):
        assert np.array_equal(x, y)
    p2 = partners(len(idx))
    assert (counts2[1] + counts2[1][p2] + 2 > 64).any()  # at least one wave could not pack its two envs
    b.close(); c.close()


def test_bad_states_reset_inside_a_shared_wave(hbmod, humanoid_model, gpu):
    """mj_checkPos / mj_checkVel / mj_checkAcc in one env of a wave: that env is reset (and, for a bad qacc, runs mj_forward a second time)
    while its partner steps normally - same bits, same warning bits as the one-env kernel"""
    m = humanoid_model
    n = 64
    b = hbmod.Batch(m, n, gpu)
    b.duo(2)
    b.reset(perturb=True)
    b.rollout_halton(250)
    st = b.get_state(hbmod.STATE_INTEGRATION)
    nq, nv = m.nq, m.nv
    st[3, 1 + 5] = np.nan              # BADQPOS
    st[10, 1 + nq + 7] = np.inf        # BADQVEL
    st[17, 1 + nq + 2] = 3e9           # a velocity that overflows the accelerations: BADQACC
    st[n - 1 - 17, 1 + nq + 4] = 3e9   # ... and its partner in the same wave as well
    st[40, 1 + nq:1 + nq + nv] = 2e9
    ctrl = np.random.default_rng(3).uniform(-1, 1, size=(n, m.nu)).astype(np.float32)
    want, wcounts, wstatus = _reference_step(hbmod, m, gpu, st, ctrl)
    b.set_state(hbmod.STATE_INTEGRATION, st)
    b.step(ctrl)
    assert b.last_kernel() == DUO
    assert np.array_equal(b.status(), wstatus) and (wstatus != 0).sum() >= 4
    got = b.get_state(hbmod.STATE_INTEGRATION)
    assert np.array_equal(got, want), np.flatnonzero((got != want).any(axis=1))
    for x, y in zip(b.counts(), wcounts):
        assert np.array_equal(x, y)
    b.close()


def test_rollouts_through_the_duo_kernel_and_the_default_choice(hbmod, humanoid_model, gpu):
    """hb_step_duo_q_kernel (the step loop inside, state on chip between steps) against the one-env rollout kernel: Halton controls
    (ctrl mode 2), recorded control tapes (mode 1), single steps with Halton controls; and which kernel a batch takes by default"""
    m = humanoid_model
    n, T = 256, 40
    a = hbmod.Batch(m, n, gpu); b = hbmod.Batch(m, n, gpu); c = hbmod.Batch(m, n, gpu)
    b.duo(2); c.duo(2)
    for x in (a, b, c):
        x.reset(perturb=True, env_offset=1000)
    a.rollout_halton(T, 0, 1000)              # a batch this small: the one-env kernel with the step loop
    assert a.last_kernel() == "hb_step_h27_q_kernel"
    b.rollout_halton(T, 0, 1000)              # two envs per wave, the step loop inside
    assert b.last_kernel() == "hb_step_duo_q_kernel"
    for t in range(T):
        c.rollout_halton(1, t, 1000)          # single steps
    assert c.last_kernel() == DUO
    sa = a.get_state(hbmod.STATE_INTEGRATION)
    assert np.array_equal(sa, b.get_state(hbmod.STATE_INTEGRATION)) and np.array_equal(sa, c.get_state(hbmod.STATE_INTEGRATION))
    for x, y in zip(a.counts(), b.counts()):
        assert np.array_equal(x, y)
    tape = np.random.default_rng(5).uniform(-1, 1, size=(T, n, m.nu)).astype(np.float32)
    a.rollout(tape); b.rollout(tape)
    assert a.last_kernel() == "hb_step_h27_q_kernel" and b.last_kernel() == "hb_step_duo_q_kernel"
    assert np.array_equal(a.get_state(hbmod.STATE_INTEGRATION), b.get_state(hbmod.STATE_INTEGRATION))
    assert np.array_equal(a.status(), b.status())
    a.close(); b.close(); c.close()
    # the default: one env per wave for the step calls of the benchmark's 4096 envs (exactly one round of duo waves: DESIGN.md 3.8), two
    # for its rollouts and for step calls from 2.5 x the chip's wave slots on
    ctrl0 = np.zeros((1, m.nu), np.float32)
    d = hbmod.Batch(m, 4096, gpu)
    d.step(np.repeat(ctrl0, 4096, axis=0)); k_unpiped = d.last_kernel()   # one launch per step call: one round of duo waves against two
    d.pipeline(True)
    d.step(np.repeat(ctrl0, 4096, axis=0)); k_step = d.last_kernel()      # env segments on their own streams: the benchmark's timed loop
    d.rollout_halton(2); k_roll = d.last_kernel()
    d.close()
    e = hbmod.Batch(m, 8192, gpu)
    e.pipeline(True)
    e.step(np.repeat(ctrl0, 8192, axis=0)); k_big = e.last_kernel()
    e.close()
    f = hbmod.Batch(m, 2048, gpu)
    f.step(np.repeat(ctrl0, 2048, axis=0)); k_small = f.last_kernel()
    f.close()
    assert (k_unpiped, k_step, k_roll, k_big, k_small) == (DUO, "hb_step_h27_kernel", "hb_step_duo_q_kernel", DUO, "hb_step_h27_kernel"), (k_unpiped, k_step, k_roll, k_big, k_small)


def test_env_adapter_steps_through_the_duo_kernel_read_the_same_torques(hbmod, humanoid_model, gpu):
    """hb_env_step's launches ask for the joint torques (qfrc_smooth + qfrc_constraint = M qacc: the reward's torque term): the
    two-envs-per-wave kernel writes them too, term for term the one-env kernel's sum - observations, rewards and episode ends identical"""
    n, T = 256, 60
    rng = np.random.default_rng(9)
    acts = rng.uniform(-1, 1, (T, n, humanoid_model.nu)).astype(np.float32)
    got, names = [], []
    for duo in (0, 2):
        env = hbmod.VecEnv(humanoid_model, n, gpu, realism=True, domain_randomization=True)
        env.batch.tune(duo=duo)
        out = [env.reset().copy()]
        for t in range(T):
            obs, rew, term, trunc, info = env.step_arrays(acts[t])
            out += [obs.copy(), rew.copy(), term.copy(), trunc.copy()]
        names.append(env.batch.last_kernel())
        got.append(out)
        env.close()
    assert names == ["hb_step_kernel", "hb_step_kernel"] or names[1].startswith("hb_step_duo"), names
    assert all(np.array_equal(a, b) for a, b in zip(got[0], got[1]))
    # (domain randomisation hands the launch per-env model parameters: that launch is not a lean one and takes the full kernel in both runs;
    # without it the second run is the duo kernel's)
    got, names = [], []
    for duo in (0, 2):
        env = hbmod.VecEnv(humanoid_model, n, gpu)
        env.batch.tune(duo=duo)
        out = [env.reset().copy()]
        for t in range(T):
            obs, rew, term, trunc, info = env.step_arrays(acts[t])
            out += [obs.copy(), rew.copy(), term.copy(), trunc.copy()]
        names.append(env.batch.last_kernel())
        got.append(out)
        env.close()
    assert names == ["hb_step_h27_q_kernel", "hb_step_duo_kernel"], names
    assert all(np.array_equal(a, b) for a, b in zip(got[0], got[1]))
    assert any(np.abs(x).max() > 0 for x in got[0][2::4])  # rewards are not all zero
