"""The staged step of the general collision path: pose kernel -> narrowphase kernel -> step kernel, with a fast
first pass of the step kernel and the full kernel behind it for the env-steps the fast one defers.  Every arrangement must give
what the single fused kernel gives (hb_batch_tune: HB_TUNE_STAGED = 0), which the convex / terrain parity tests pin against the oracle."""
import os

import numpy as np
import pytest

from oracle_lib import ROOT
from test_oracle_convex import BALL_MESH, CUBE_MESH, _hfield_xml

pytestmark = pytest.mark.gpu
TEAM_HBM = os.path.join(ROOT, "humanoid_mujoco_amd", "assets", "team_robot.hbm")
TERRAIN_HBM = os.path.join(ROOT, "humanoid_mujoco_amd", "assets", "humanoid27_hfield.hbm")


TUNE = {"HB_STAGED": "staged", "HB_FASTPASS": "fastpass"}


def _batch(hbmod, model, n, gpu, **env):
    """a batch with the given run-time choices (include/hb.h: hb_batch_tune); HB_BOX_CULL is the one switch that is baked into the model
    tables, i.e. read from the environment when the batch is created"""
    old = {k: os.environ.get(k) for k in env if k not in TUNE}
    try:
        for k, v in env.items():
            if k not in TUNE:
                os.environ[k] = str(v)
        b = hbmod.Batch(model, n, gpu)
        b.tune(**{TUNE[k]: v for k, v in env.items() if k in TUNE})
        return b
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _settled_states(hbmod, model, n, gpu, steps, keyframe=-1):
    b = hbmod.Batch(model, n, gpu)
    b.reset(keyframe=keyframe, perturb=True)
    b.rollout_halton(steps, 0, 0)
    st = b.get_state(hbmod.STATE_INTEGRATION, dtype=np.float64)
    b.close()
    return st


def _close(qa, qb, va, vb, max_fence=0.02):
    dv = np.abs(va - vb).max(axis=1) / np.maximum(1.0, np.abs(vb).max(axis=1))
    dq = np.abs(qa - qb).max(axis=1)
    fence = (dv > 2e-4) | (dq > 2e-6)
    print("   envs on a portal fence: %d of %d; the others: max |dqpos| %.2e, max rel |dqvel| %.2e" % (fence.sum(), len(fence), dq[~fence].max(), dv[~fence].max()))
    assert fence.mean() <= max_fence
    assert np.isfinite(qa).all() and np.isfinite(va).all()


def _one_step(hbmod, b, st, ctrl):
    b.set_state(hbmod.STATE_INTEGRATION, st)
    b.step(ctrl)
    nc, ne, ni = b.counts()
    return b.qpos.astype(np.float64), b.qvel.astype(np.float64), nc.copy(), ne.copy(), b.status().copy()


@pytest.mark.parametrize("path,keyframe,steps", [(TEAM_HBM, 0, 150), (TEAM_HBM, 1, 400), (TERRAIN_HBM, -1, 500)])
def test_staged_step_equals_fused_step(hbmod, gpu, path, keyframe, steps):
    m = hbmod.Model.load(path)
    n = 512
    st = _settled_states(hbmod, m, n, gpu, steps, keyframe)
    ctrl = (0.3 * np.random.default_rng(1).uniform(-1, 1, (n, m.nu))).astype(np.float32)
    staged = _batch(hbmod, m, n, gpu)
    fused = _batch(hbmod, m, n, gpu, HB_STAGED=0)
    qa, va, nca, nea, sa = _one_step(hbmod, staged, st, ctrl)
    qb, vb, ncb, neb, sb = _one_step(hbmod, fused, st, ctrl)
    same = (nca == ncb) & (nea == neb)
    print("\n%s after %d steps: contacts mean %.2f, rows mean %.1f max %d; same counts in %d of %d envs; max |dqpos| %.2e, max |dqvel| %.2e"
          % (os.path.basename(path), steps, nca.mean(), nea.mean(), nea.max(), same.sum(), n, np.abs(qa - qb)[same].max(), np.abs(va - vb)[same].max()))
    assert nca.mean() > 0.3            # the states are in contact
    assert same.mean() >= 0.995        # (the two paths evaluate the same items with the same arithmetic)
    assert np.array_equal(sa, sb)
    # The step kernels differ (one row group against four for the team robot: other summation orders).  The portal search is the same
    # source compiled into two kernels: where a search sits on the fence between two portals (tests/test_gpu_convex.py: "divergent"
    # contacts) the two builds may land on different sides, and that env's contact differs visibly.  Those are counted.
    _close(qa[same], qb[same], va[same], vb[same])
    nw, ns = staged.collision_counts()
    assert (nw >= ns).all() and ns.sum() > 0 and nw.max() <= 256
    staged.close(); fused.close()


def test_multi_step_launches_and_pipelined_segments_match_single_steps(hbmod, gpu):
    """a T-step rollout of a staged model is T x (pose, narrowphase, step) launches inside one call; pipelined segments keep their
    own stage buffers: both must reproduce repeated single steps bit for bit"""
    m = hbmod.Model.load(TEAM_HBM)
    n, T = 256, 40
    st = _settled_states(hbmod, m, n, gpu, 100, 0)
    ctrl = (0.3 * np.random.default_rng(2).uniform(-1, 1, (T, n, m.nu))).astype(np.float32)
    a = hbmod.Batch(m, n, gpu); a.set_state(hbmod.STATE_INTEGRATION, st)
    for t in range(T):
        a.step(ctrl[t])
    b = hbmod.Batch(m, n, gpu); b.set_state(hbmod.STATE_INTEGRATION, st)
    b.rollout(ctrl)
    c = hbmod.Batch(m, n, gpu); c.set_state(hbmod.STATE_INTEGRATION, st)
    c.pipeline(True)
    for t in range(T):
        c.step(ctrl[t])
    c.pipeline(False)
    assert np.array_equal(a.qpos, b.qpos) and np.array_equal(a.qvel, b.qvel)
    assert np.array_equal(a.qpos, c.qpos) and np.array_equal(a.qvel, c.qvel)
    for x in (a, b, c):
        x.close()


def test_fast_pass_defers_what_it_cannot_hold(hbmod, gpu, tmp_path):
    """Four free bodies with condim-4 / 6 contacts on a bumpy field, Newton: more than 63 rows in most steps once they have landed.
    The fast (one row group) kernel must hand exactly those env-steps to the four-group kernel: bit-identical to a batch that only
    ever runs the four-group kernel (HB_FASTPASS=0), and close to it where the fast kernel did the step itself."""
    rng = np.random.default_rng(3)
    elev = rng.uniform(0, 1, (6, 6))
    body = ('<body pos="-0.5 0.3 0.45"><freejoint/><geom type="sphere" size="0.08" condim="6"/></body>'
            '<body pos="0.4 -0.4 0.5" euler="20 40 0"><freejoint/><geom type="capsule" size="0.05 0.12" condim="3"/></body>'
            '<body pos="0.1 0.5 0.5" euler="10 20 30"><freejoint/><inertial pos="0 0 0" mass="0.5" diaginertia="0.001 0.001 0.001"/><geom type="mesh" mesh="cube" condim="4"/></body>'
            '<body pos="0.12 0.52 0.62"><freejoint/><inertial pos="0 0 0" mass="0.3" diaginertia="0.0005 0.0005 0.0005"/><geom type="mesh" mesh="ball" condim="6" friction="0.7 0.02 0.01"/></body>')
    m = hbmod.Model.from_xml_string(_hfield_xml(elev, body, nrow=6, ncol=6, size="1 1 0.3 0.2", extra=CUBE_MESH + BALL_MESH))
    m.set_opt(solver=2, iterations=100)
    n = 256
    # states along the fall and the landing: env e has run 3 e steps
    src = hbmod.Batch(m, n, gpu)
    src.reset(perturb=True)
    zero = np.zeros((n, max(1, m.nu)), np.float32)[:, :m.nu]
    states = np.zeros((n, src.state_size(hbmod.STATE_INTEGRATION)))
    for e in range(n):
        states[e] = src.get_state(hbmod.STATE_INTEGRATION, dtype=np.float64)[0]
        src.step(zero, n_substeps=3)
    src.close()
    both = _batch(hbmod, m, n, gpu)
    big = _batch(hbmod, m, n, gpu, HB_FASTPASS=0)
    qa, va, nca, nea, sa = _one_step(hbmod, both, states, zero)
    qb, vb, ncb, neb, sb = _one_step(hbmod, big, states, zero)
    over = neb > 63
    print("\nrows: mean %.1f max %d; env-steps beyond one row group: %d of %d" % (neb.mean(), neb.max(), over.sum(), n))
    assert over.sum() >= 20 and (~over).sum() >= 20
    assert np.array_equal(nca, ncb) and np.array_equal(nea, neb) and np.array_equal(sa, sb)
    assert np.array_equal(qa[over], qb[over]) and np.array_equal(va[over], vb[over])      # the same kernel stepped them
    _close(qa, qb, va, vb)                                                               # the others: one row group against four
    both.close(); big.close()


def test_bad_qacc_in_a_staged_step_is_reset_like_in_the_fused_one(hbmod, gpu):
    """mj_checkAcc inside a staged step: the fast kernels hand the env to the full kernel, which resets it and runs the second
    forward pass with a narrowphase of its own (the staged results belong to the poses before the reset)"""
    for path, key in ((TEAM_HBM, 0), (TERRAIN_HBM, -1)):
        m = hbmod.Model.load(path)
        n = 64
        st = _settled_states(hbmod, m, n, gpu, 120, key)
        spec = hbmod.STATE_INTEGRATION | hbmod.STATE_XFRC_APPLIED
        nint = st.shape[1]
        full = np.zeros((n, nint + 6 * m.nbody))
        full[:, :nint] = st
        full[5, nint + 6 * 1 + 2] = 1e14   # a wrench no solver survives: |qacc| > 1e10 on env 5
        full[9, 1 + 3] = np.nan            # and a bad qpos on env 9 (mj_checkPos, before the step)
        ctrl = np.full((n, m.nu), 0.2, np.float32)
        out = []
        for env in ({}, {"HB_STAGED": 0}):
            b = _batch(hbmod, m, n, gpu, **env)
            b.set_state(spec, full)
            b.step(ctrl)
            out.append((b.qpos.copy(), b.qvel.copy(), b.status().copy(), b.time.copy()))
            b.close()
        (qa, va, sa, ta), (qb, vb, sb, tb) = out
        assert sa[5] == hbmod.WARN_BADQACC and sa[9] == hbmod.WARN_BADQPOS and np.array_equal(sa, sb)
        assert np.isfinite(qa).all() and np.isfinite(va).all()
        assert np.array_equal(ta, tb) and abs(ta[5] - m.opt.timestep) < 1e-9 and abs(ta[9] - m.opt.timestep) < 1e-9
        _close(qa, qb, va, vb, max_fence=0.05)
        # the flagged envs were stepped by the very same kernel in both arrangements
        assert np.array_equal(qa[5], qb[5]) and np.array_equal(va[5], vb[5])


def test_oriented_box_cull_only_drops_searches_that_find_nothing(hbmod, gpu):
    """the separating-axis test of the geoms' oriented boxes (behind the bounding spheres, before the portal search) is conservative:
    the same contacts, rows and next state with and without it, fewer searches with it"""
    m = hbmod.Model.load(TEAM_HBM)
    n = 1024
    for key, steps in ((0, 120), (0, 400)):
        st = _settled_states(hbmod, m, n, gpu, steps, key)
        ctrl = (0.3 * np.random.default_rng(4).uniform(-1, 1, (n, m.nu))).astype(np.float32)
        a = _batch(hbmod, m, n, gpu)
        b = _batch(hbmod, m, n, gpu, HB_BOX_CULL=0)
        qa, va, nca, nea, sa = _one_step(hbmod, a, st, ctrl)
        qb, vb, ncb, neb, sb = _one_step(hbmod, b, st, ctrl)
        sa_n, sb_n = a.collision_counts()[1], b.collision_counts()[1]
        print("\nportal searches per env: %.2f with the box test, %.2f without; contacts %.2f" % (sa_n.mean(), sb_n.mean(), nca.mean()))
        assert np.array_equal(nca, ncb) and np.array_equal(nea, neb) and np.array_equal(sa, sb)
        assert np.array_equal(qa, qb) and np.array_equal(va, vb)
        assert (sa_n <= sb_n).all() and sa_n.sum() < sb_n.sum()
        a.close(); b.close()


def test_collision_counts_are_zero_for_a_classic_model(hbmod, humanoid_model, gpu):
    b = hbmod.Batch(humanoid_model, 32, gpu)
    b.reset(perturb=True)
    b.rollout_halton(50)
    nw, ns = b.collision_counts()
    assert not nw.any() and not ns.any()
    b.close()
