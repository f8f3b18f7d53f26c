"""the batched rollout backend of a sampling planner (MJPC's Trajectory::Rollout,
mujoco_mpc/mjpc/trajectory.cc:100-210; candidates from one state, sampling/planner.cc:342-380) — N action
sequences from one common state in one launch, with the sensor read-out the humanoid tasks' residuals use
(framepos, subtreecom, subtreelinvel: tasks/humanoid_cap/stand/task.xml:22-40), against the fp64 oracle."""
import numpy as np
import pytest

from oracle_lib import Oracle

pytestmark = pytest.mark.gpu


def oracle_sensors(o, bodies, nb):
    xpos = o.xpos.reshape(nb, 3)
    mass = o.marr("body_mass")
    com = o.subtree_com.reshape(nb, 3)[1]  # the humanoid is one tree rooted at body 1
    cvel = o.cvel.reshape(nb, 6)
    xipos = o.xipos.reshape(nb, 3)
    v = cvel[:, 3:] + np.cross(cvel[:, :3], xipos - com)  # mj_subtreeVel: a body's com moves with lin + ang x (xipos - com)
    linvel = (mass[1:, None] * v[1:]).sum(0) / mass[1:].sum()
    return np.concatenate([xpos[bodies].ravel(), com, linvel])


def test_candidates_from_one_state_with_sensor_readout(hbmod, humanoid_model, gpu):
    m = humanoid_model
    N, T, nb = 24, 30, 17
    head, foot_r, foot_l = m.name2id("body", "head"), m.name2id("body", "foot_right"), m.name2id("body", "foot_left")
    torso = m.name2id("body", "torso")
    assert min(head, foot_r, foot_l, torso) > 0
    spec = hbmod.Batch.sensor_spec([head, foot_r, foot_l], subtree_body=torso)
    # the common start: a mid-fall state from the oracle (moving, in contact)
    o = Oracle()
    o.init_env(3)
    for t in range(150):
        o.ctrl[:] = o.ctrl_env(t, 3); o.step()
    start = np.concatenate([[o.time], o.qpos, o.qvel, o.qacc_warmstart])
    rng = np.random.default_rng(6)
    actions = rng.uniform(-1, 1, size=(T, N, m.nu)).astype(np.float32)
    b = hbmod.Batch(m, N, gpu)
    b.set_state_broadcast(hbmod.STATE_INTEGRATION, start)
    assert np.allclose(b.get_state(hbmod.STATE_INTEGRATION, dtype=np.float64), start[None], atol=1e-6)
    sens, qpos = b.rollout_sensors(actions, spec, want_qpos=True)
    assert sens.shape == (T, N, 15) and qpos.shape == (T, N, m.nq)
    assert not b.status().any()  # no failure flags (CheckWarnings, utilities.cc:787-799)
    # candidate k against the oracle driven with the same action sequence, teacher-forced per step from the device
    # states (free-running fp32 vs fp64 decorrelates in contact; the sensors are a function of the state)
    for k in (0, 7, 23):
        o.reset()
        for t in range(T):
            if t == 0:
                o.qpos[:] = start[1:1 + m.nq]; o.qvel[:] = start[1 + m.nq:1 + m.nq + m.nv]
            else:
                o.qpos[:] = qpos[t - 1, k]
                o.qvel[:] = qvel_prev
            o.ctrl[:] = actions[t, k]
            o.forward()
            ref = oracle_sensors(o, [head, foot_r, foot_l], nb)
            assert np.abs(sens[t, k] - ref).max() < 2e-4 * max(1.0, np.abs(ref).max()), (k, t, np.abs(sens[t, k] - ref).max())
            # advance the oracle one step to obtain the velocity the device state has next (positions come from qpos_out)
            o.step()
            qvel_prev = o.qvel.copy()
            if t == 0:
                assert np.abs(o.qpos - qpos[0, k]).max() < 1e-4
    # the candidates differ, and the terminal read-out (mj_forward at the final state) matches a fresh evaluation
    assert np.abs(sens[-1, 0] - sens[-1, 1]).max() > 1e-4
    term = b.sensors(spec)
    q_end, v_end = b.qpos.astype(np.float64), b.qvel.astype(np.float64)
    o.reset(); o.qpos[:] = q_end[5]; o.qvel[:] = v_end[5]; o.forward()
    ref = oracle_sensors(o, [head, foot_r, foot_l], nb)
    assert np.abs(term[5] - ref).max() < 2e-4 * max(1.0, np.abs(ref).max())


def test_site_framepos(hbmod, humanoid_model, gpu):
    """framepos of a site = body position + body rotation * offset (the foot sites sp0..sp3 of the reference's humanoid task)."""
    m = humanoid_model
    foot = m.name2id("body", "foot_left")
    off = (-0.07, 0.02, 0.01)
    spec = hbmod.Batch.sensor_spec([foot, foot], offsets=[None, off])
    b = hbmod.Batch(m, 4, gpu)
    b.reset(perturb=True)
    b.rollout_halton(30)
    s = b.sensors(spec)
    o = Oracle()
    st = b.get_state(hbmod.STATE_INTEGRATION, dtype=np.float64)
    for e in range(4):
        o.reset()
        o.qpos[:] = st[e, 1:1 + m.nq]; o.qvel[:] = st[e, 1 + m.nq:1 + m.nq + m.nv]
        o.forward()
        xpos = o.xpos.reshape(-1, 3)[foot]; xmat = o.xmat.reshape(-1, 3, 3)[foot]
        assert np.allclose(s[e, :3], xpos, atol=2e-6)
        assert np.allclose(s[e, 3:6], xpos + xmat @ np.array(off), atol=2e-6)


def test_sensor_spec_checks(hbmod, humanoid_model, gpu):
    m = humanoid_model
    b = hbmod.Batch(m, 4, gpu)
    with pytest.raises(hbmod.HbError):
        b.sensors(hbmod.Batch.sensor_spec([99]))                      # no such body
    with pytest.raises(hbmod.HbError):
        b.sensors(hbmod.Batch.sensor_spec([1], subtree_body=2))       # not the root of a tree
    with pytest.raises(hbmod.HbError):
        b.sensors(hbmod.Batch.sensor_spec([]))                        # nothing to read
    out = b.sensors(hbmod.Batch.sensor_spec([1]))
    assert out.shape == (4, 3) and np.allclose(out, b.qpos[:, :3], atol=1e-6)  # torso frame position = free joint position


@pytest.mark.parametrize("pipelined", [False, True])
def test_trajectory_rollout_records_states_like_mjpc(hbmod, humanoid_model, gpu, pipelined):
    """hb_rollout_trajectory: the states after every step (qpos and qvel: what Trajectory::Rollout records,
    trajectory.cc:175-190) equal, bit for bit, the states of the same rollout taken step by step; the failure flag
    mirrors CheckWarnings."""
    m = humanoid_model
    N, T = 16, 25
    rng = np.random.default_rng(9)
    ctrl = rng.uniform(-1, 1, size=(T, N, m.nu)).astype(np.float32)
    st = np.zeros(1 + m.nq + 2 * m.nv)
    o = Oracle()
    o.init_env(2)
    st[1:1 + m.nq] = o.qpos
    a = hbmod.Batch(m, N, gpu)
    a.set_state_broadcast(hbmod.STATE_INTEGRATION, st)
    q_ref, v_ref = [], []
    for t in range(T):
        a.step(ctrl[t])
        q_ref.append(a.qpos.copy()); v_ref.append(a.qvel.copy())
    b = hbmod.Batch(m, N, gpu)
    b.pipeline(pipelined)
    b.set_state_broadcast(hbmod.STATE_INTEGRATION, st)
    q, v, failed = b.rollout_trajectory(ctrl)
    assert q.shape == (T, N, m.nq) and v.shape == (T, N, m.nv)
    assert np.array_equal(q, np.array(q_ref)) and np.array_equal(v, np.array(v_ref))
    assert not failed.any()
    # a candidate that blows up is flagged, the others are not
    bad = st.copy()
    c = hbmod.Batch(m, N, gpu)
    states = np.tile(st, (N, 1))
    states[3, 1 + m.nq + 2] = 1e12  # absurd root velocity: mj_checkVel resets the env and raises mjWARN_BADQVEL
    c.set_state(hbmod.STATE_INTEGRATION, states)
    _, _, failed = c.rollout_trajectory(ctrl[:3])
    assert failed.tolist() == [i == 3 for i in range(N)]


def test_stand_task_return_matches_mjpc_restatement(hbmod, humanoid_model, gpu):
    """hb_rollout_task_stand: residual, norms, stage costs and return of MJPC's Humanoid Stand task for N candidates,
    all evaluated on the device, against the numpy restatement (tests/mjpc_ref.py) on fp64 oracle rollouts."""
    from mjpc_ref import stand_rollout
    m = humanoid_model
    N, H, nb = 12, 24, 17
    b = hbmod.Batch(m, N, gpu)
    task = b.task_stand_default()
    assert task.n_feet == 4 and task.norm[:] == [6, 6, 0, 0, 3] and abs(task.height_goal - 1.4) < 1e-6
    o = Oracle()
    o.init_env(1)
    for t in range(40):  # a state with the feet near the ground
        o.ctrl[:] = o.ctrl_env(t, 1)
        o.step()
    q0, v0, w0 = o.qpos.copy(), o.qvel.copy(), o.qacc_warmstart.copy()
    st = np.concatenate([[0.0], q0, v0, w0])
    rng = np.random.default_rng(4)
    ctrl = rng.uniform(-0.6, 0.6, size=(H - 1, N, m.nu)).astype(np.float32)
    b.set_state_broadcast(hbmod.STATE_INTEGRATION, st)
    total, costs = b.rollout_task_stand(ctrl, task, want_costs=True)
    assert costs.shape == (H, N)
    worst = 0.0
    for e in range(0, N, 3):
        o.reset()
        o.qpos[:] = q0; o.qvel[:] = v0; o.qacc_warmstart[:] = w0
        ret, cs = stand_rollout(o, ctrl[:, e].astype(np.float64), task, nb)
        # stage costs: teacher-forcing is not possible inside a rollout, so compare the early stages tightly and the return loosely
        assert np.allclose(costs[:6, e], cs[:6], rtol=2e-3, atol=1e-3), (e, costs[:6, e], cs[:6])
        worst = max(worst, abs(total[e] - ret) / max(1.0, abs(ret)))
    assert worst < 2e-2, worst
    # the first stage is the same state for every candidate: only the control term differs
    ctrl_term = np.array([task.weight[4] * (0.3 ** 2) * (np.cosh(ctrl[0, e].astype(np.float64) / 0.3) - 1).sum() for e in range(N)])
    assert np.allclose(costs[0] - ctrl_term, (costs[0] - ctrl_term)[0], rtol=1e-4, atol=1e-3)
    # pipelined stepping (env segments on their own streams) changes nothing
    bp = hbmod.Batch(m, N, gpu)
    bp.pipeline(True)
    bp.set_state_broadcast(hbmod.STATE_INTEGRATION, st)
    tp, cp = bp.rollout_task_stand(ctrl, task, want_costs=True)
    assert np.array_equal(tp, total) and np.array_equal(cp, costs)
    # horizon 1: one mj_forward with a zero action
    t1, c1 = b.rollout_task_stand(np.zeros((0, N, m.nu), np.float32), task, want_costs=True)
    assert c1.shape == (1, N) and np.allclose(t1, c1[0])
    # a diverging candidate returns kMaxReturnValue
    states = np.tile(st, (N, 1))
    states[5, 1 + m.nq + 1] = 1e12
    b.set_state(hbmod.STATE_INTEGRATION, states)
    total, _ = b.rollout_task_stand(ctrl[:4], task)
    assert total[5] == 1e6 and (np.delete(total, 5) < 1e5).all()


def test_axis_linvel_and_subtree_sensors(hbmod, humanoid_model, gpu):
    """framexaxis / framezaxis (xbody), framelinvel (body) and subtreelinvel of a non-root body against oracle quantities."""
    from mjpc_ref import body_linvel, subtree_linvel
    m = humanoid_model
    nb = 17
    torso, pelvis, foot, waist = (m.name2id("body", n) for n in ("torso", "pelvis", "foot_right", "waist_lower"))
    spec = hbmod.Batch.sensor_spec([], axes=[(pelvis, 2), (foot, 0)], linvel_bodies=[torso, foot], subtreelinvel_bodies=[waist, foot])
    b = hbmod.Batch(m, 4, gpu)
    b.reset(perturb=True)
    b.rollout_halton(45)
    s = b.sensors(spec)
    assert s.shape == (4, 18)
    o = Oracle()
    parent = o.info["body_parentid"]
    st = b.get_state(hbmod.STATE_INTEGRATION, dtype=np.float64)
    for e in range(4):
        o.reset()
        o.qpos[:] = st[e, 1:1 + m.nq]; o.qvel[:] = st[e, 1 + m.nq:1 + m.nq + m.nv]
        o.forward()
        xmat = o.xmat.reshape(nb, 3, 3)
        ref = np.concatenate([xmat[pelvis][:, 2], xmat[foot][:, 0], body_linvel(o, torso, nb), body_linvel(o, foot, nb),
                              subtree_linvel(o, waist, nb, parent), subtree_linvel(o, foot, nb, parent)])
        assert np.allclose(s[e], ref, atol=2e-5 * max(1.0, np.abs(ref).max())), (e, s[e], ref)


def test_walk_task_return_matches_mjpc_restatement(hbmod, humanoid_model, gpu):
    """hb_rollout_task_walk against the numpy restatement of Walk::ResidualFn::Residual on fp64 oracle rollouts."""
    from mjpc_ref import walk_rollout
    m = humanoid_model
    N, H, nb = 12, 20, 17
    b = hbmod.Batch(m, N, gpu)
    task = b.task_walk_default()
    assert task.n_term == 8 and list(task.dim[:]) == [1, 1, 2, 8, 21, 2, 1, 21] and list(task.norm[:]) == [7, 8, 1, 2, 0, 7, 7, 3]
    o = Oracle()
    parent = o.info["body_parentid"]
    o.init_env(2)
    for t in range(25):
        o.ctrl[:] = o.ctrl_env(t, 2)
        o.step()
    q0, v0, w0 = o.qpos.copy(), o.qvel.copy(), o.qacc_warmstart.copy()
    st = np.concatenate([[0.0], q0, v0, w0])
    rng = np.random.default_rng(6)
    ctrl = rng.uniform(-0.5, 0.5, size=(H - 1, N, m.nu)).astype(np.float32)
    b.set_state_broadcast(hbmod.STATE_INTEGRATION, st)
    total, costs = b.rollout_task_walk(ctrl, task, want_costs=True)
    worst = 0.0
    for e in range(0, N, 3):
        o.reset()
        o.qpos[:] = q0; o.qvel[:] = v0; o.qacc_warmstart[:] = w0
        ret, cs = walk_rollout(o, ctrl[:, e].astype(np.float64), task, nb, parent)
        assert np.allclose(costs[:6, e], cs[:6], rtol=2e-3, atol=2e-3), (e, costs[:6, e], cs[:6])
        worst = max(worst, abs(total[e] - ret) / max(1.0, abs(ret)))
    assert worst < 2e-2, worst
    # a cost-term layout that does not cover the residual is refused (the reference aborts with "mismatch between total
    # user-sensor dimension and actual length of residual")
    task.dim[4] = 20
    with pytest.raises(hbmod.HbError):
        b.rollout_task_walk(ctrl, task)


def test_rollout_noise_is_an_ou_process_on_xfrc(hbmod, humanoid_model, gpu):
    """hb_rollout_noise: xfrc_applied follows x <- rate x + scale N(0,1) (Trajectory::NoisyRollout, trajectory.cc:147-156):
    stationary deviation xfrc_std, lag-one correlation exp(-timestep / xfrc_rate), independent across envs, reproducible,
    and it actually pushes the robot."""
    m = humanoid_model
    N, nb = 256, 17
    std, tau = 2.0, 0.05
    rate = np.exp(-0.005 / tau)
    b = hbmod.Batch(m, N, gpu)
    b.reset(perturb=True)
    b.rollout_noise(std, tau, seed=11)
    zeros = np.zeros((1, N, m.nu), np.float32)
    xs = []
    for t in range(120):
        b.rollout(zeros)
        xs.append(b.get_state(hbmod.STATE_XFRC_APPLIED).reshape(N, nb, 6).copy())
    xs = np.array(xs)[40:]  # past the transient from zero
    assert abs(xs.std() - std) < 0.05 * std, xs.std()
    assert abs(xs.mean()) < 0.05
    lag = (xs[1:] * xs[:-1]).mean() / (xs * xs).mean()
    assert abs(lag - rate) < 0.03, (lag, rate)
    assert abs(np.corrcoef(xs[:, 0].ravel(), xs[:, 1].ravel())[0, 1]) < 0.15  # envs draw independently (all entries of two envs)
    assert abs(np.corrcoef(xs[:, :, 3, 0].ravel(), xs[:, :, 3, 1].ravel())[0, 1]) < 0.05  # and so do entries
    q_noisy = b.qpos.copy()
    # same seed, same call sequence: the same noise
    c = hbmod.Batch(m, N, gpu)
    c.reset(perturb=True)
    c.rollout_noise(std, tau, seed=11)
    for t in range(120):
        c.rollout(zeros)
    assert np.array_equal(c.qpos, q_noisy)
    # off: xfrc stays as it is, and the trajectory differs from the noisy one
    d = hbmod.Batch(m, N, gpu)
    d.reset(perturb=True)
    for t in range(120):
        d.rollout(zeros)
    assert np.abs(d.qpos - q_noisy).max() > 1e-4


@pytest.mark.parametrize("interp", [0, 1, 2])
def test_spline_policies_on_the_device(hbmod, humanoid_model, gpu, interp):
    """hb_ctrl_tape_splines: every candidate's time spline sampled and clamped on the device equals the host evaluation of
    TimeSpline::Sample (tests/mjpc_ref.py) fed through the ordinary host tape: same returns."""
    from mjpc_ref import spline_sample
    m = humanoid_model
    N, H, P = 16, 24, 4
    o = Oracle()
    o.init_env(1)
    for t in range(40):
        o.ctrl[:] = o.ctrl_env(t, 1)
        o.step()
    st = np.concatenate([[0.0], o.qpos, o.qvel, o.qacc_warmstart])
    rng = np.random.default_rng(8 + interp)
    knots = rng.uniform(-1.4, 1.4, size=(N, P, m.nu)).astype(np.float32)  # beyond ctrlrange: the clamp matters
    time0 = 0.37
    times = (time0 + np.array([-0.01, 0.03, 0.07, 0.09])).astype(np.float32)  # first node in the past, last before the horizon's end
    h = 0.005
    tape = np.zeros((H - 1, N, m.nu), np.float32)
    lo, hi = m.array("actuator_ctrlrange").reshape(-1, 2).T
    for e in range(N):
        for t in range(H - 1):
            tape[t, e] = np.clip(spline_sample(times.astype(np.float64), knots[e].astype(np.float64), interp, float(np.float32(time0) + np.float32(t) * np.float32(h))), lo, hi)
    b = hbmod.Batch(m, N, gpu)
    task = b.task_stand_default()
    b.set_state_broadcast(hbmod.STATE_INTEGRATION, st)
    ref_total, ref_costs = b.rollout_task_stand(tape, task, want_costs=True)
    b.set_state_broadcast(hbmod.STATE_INTEGRATION, st)
    b.ctrl_tape_splines(knots, times, interp, time0, H - 1)
    total, costs = b.rollout_task_stand(("tape", H - 1), task, want_costs=True)
    assert np.allclose(costs[:4], ref_costs[:4], rtol=1e-4, atol=1e-4)
    assert np.allclose(total, ref_total, rtol=5e-3, atol=5e-3)
    # a tape shorter than the rollout is refused
    with pytest.raises(hbmod.HbError):
        b.rollout_task_stand(("tape", H + 3), task)
    # and so is a tape that another call has written over since (hb_step copies its controls into the same buffer):
    # stale controls must never be rolled out silently
    b.ctrl_tape_splines(knots, times, interp, time0, H - 1)
    b.step(np.zeros((N, m.nu), np.float32))
    with pytest.raises(hbmod.HbError):
        b.rollout_task_stand(("tape", H - 1), task)
    b.ctrl_tape_splines(knots, times, interp, time0, H - 1)
    b.rollout_task_stand(("tape", H - 1), task)  # a fresh tape is accepted again


def _oracle_transition_fd(o, x, u, warm, eps, nq, nv, nu):
    """mjd_transitionFD restated on the fp64 oracle: centered differences of one step in tangent coordinates."""
    def step(q, v, uu):
        o.reset()
        o.qpos[:] = q; o.qvel[:] = v; o.qacc_warmstart[:] = warm; o.ctrl[:] = uu
        o.step()
        return o.qpos.copy(), o.qvel.copy()

    def quat_mul(a, b):
        return np.array([a[0]*b[0]-a[1]*b[1]-a[2]*b[2]-a[3]*b[3], a[0]*b[1]+a[1]*b[0]+a[2]*b[3]-a[3]*b[2],
                         a[0]*b[2]-a[1]*b[3]+a[2]*b[0]+a[3]*b[1], a[0]*b[3]+a[1]*b[2]-a[2]*b[1]+a[3]*b[0]])

    def integrate(q, dq):  # the humanoid: one free joint (dofs 0..5), hinges after it
        q = q.copy()
        q[:3] += dq[:3]
        ang = np.linalg.norm(dq[3:6])
        if ang > 0:
            r = np.concatenate([[np.cos(ang / 2)], np.sin(ang / 2) * dq[3:6] / ang])
            q[3:7] = quat_mul(q[3:7], r); q[3:7] /= np.linalg.norm(q[3:7])
        q[7:] += dq[6:]
        return q

    def differentiate(q1, q2):
        d = np.zeros(nv)
        d[:3] = q2[:3] - q1[:3]
        c = quat_mul(q1[3:7] * np.array([1, -1, -1, -1]), q2[3:7])
        sn = np.linalg.norm(c[1:])
        ang = 2 * np.arctan2(sn, c[0])
        if ang > np.pi:
            ang -= 2 * np.pi
        d[3:6] = c[1:] * (ang / sn if sn > 1e-15 else 0.0)
        d[6:] = q2[7:] - q1[7:]
        return d

    q0, v0 = x[:nq], x[nq:]
    qn, vn = step(q0, v0, u)
    A = np.zeros((2 * nv, 2 * nv)); B = np.zeros((2 * nv, nu))
    for col in range(2 * nv + nu):
        out = []
        for sgn in (1.0, -1.0):
            q, v, uu = q0, v0.copy(), u.copy()
            if col < nv:
                dq = np.zeros(nv); dq[col] = sgn * eps
                q = integrate(q0, dq)
            elif col < 2 * nv:
                v[col - nv] += sgn * eps
            else:
                uu[col - 2 * nv] += sgn * eps
            q2, v2 = step(q, v, uu)
            out.append(np.concatenate([differentiate(qn, q2), v2 - vn]))
        d = (out[0] - out[1]) / (2 * eps)
        if col < 2 * nv:
            A[:, col] = d
        else:
            B[:, col - 2 * nv] = d
    return A, B


def test_transition_derivatives_by_batched_finite_differences(hbmod, humanoid_model, gpu):
    """hb_transition_fd (mjd_transitionFD for T points in one launch) against the same differences on the fp64 oracle."""
    m = humanoid_model
    nq, nv, nu = m.nq, m.nv, m.nu
    T, eps = 3, 2e-3
    o = Oracle()
    xs, us, ws = [], [], []
    o.init_env(4)
    for t in range(30):
        o.ctrl[:] = o.ctrl_env(t, 4)
        o.step()
        if t in (5, 17, 29):
            xs.append(np.concatenate([o.qpos, o.qvel])); us.append(0.5 * o.ctrl_env(t + 1, 4)); ws.append(o.qacc_warmstart.copy())
    per = 1 + 2 * (2 * nv + nu)
    b = hbmod.Batch(m, T * per + 5, gpu)
    A, B = b.transition_fd(np.array(xs), np.array(us), np.array(ws), eps=eps, centered=True)
    assert A.shape == (T, 2 * nv, 2 * nv) and B.shape == (T, 2 * nv, nu)
    for t in range(T):
        Ao, Bo = _oracle_transition_fd(o, xs[t], us[t], ws[t], eps, nq, nv, nu)
        sa, sb = np.abs(Ao).max(), np.abs(Bo).max()
        # fp32 rounding of x' (|qvel'| up to ~30) over 2 eps is the floor: worst entry 5e-3 of the matrix scale, typical entry 1e-4
        assert np.abs(A[t] - Ao).max() <= 5e-3 * sa, (t, np.abs(A[t] - Ao).max(), sa)
        assert np.abs(B[t] - Bo).max() <= 5e-3 * max(sb, 1e-3), (t, np.abs(B[t] - Bo).max(), sb)
        assert np.median(np.abs(A[t] - Ao)) <= 2e-4 * sa and np.median(np.abs(B[t] - Bo)) <= 2e-4 * max(sb, 1e-3)
    # sensor derivatives: the torso's frame position (objtype xbody) moves one to one with the root translation, its
    # dependence on the controls is nil (sensors are evaluated at (x, u) itself, before the step)
    torso = m.name2id("body", "torso")
    spec = hbmod.Batch.sensor_spec([torso], subtree_body=torso)
    A2, B2, C, D = b.transition_fd(np.array(xs), np.array(us), np.array(ws), eps=eps, centered=True, sensor_spec=spec)
    assert np.allclose(A2, A) and np.allclose(B2, B)
    assert C.shape == (T, 9, 2 * nv) and D.shape == (T, 9, nu)
    for t in range(T):
        assert np.allclose(C[t][:3, :3], np.eye(3), atol=2e-3)       # d xpos(torso) / d root translation
        assert np.abs(C[t][:3, nv:]).max() < 2e-3                    # positions do not depend on velocities
        assert np.allclose(C[t][6:9, nv:nv + 3], np.eye(3), atol=5e-3)  # d subtreelinvel / d root linear velocity
        assert np.abs(D[t]).max() < 2e-3
    # a batch too small for the perturbed copies is refused
    small = hbmod.Batch(m, 10, gpu)
    with pytest.raises(hbmod.HbError):
        small.transition_fd(np.array(xs), np.array(us))
