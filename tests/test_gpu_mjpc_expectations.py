"""REFERENCE-HELD VECTORS on the device: the known answers mujoco_mpc/mjpc/test holds for TimeSpline::Sample
(spline/spline_test.cc:40-355), Clamp and LinearInterpolation (agent/agent_utilities_test.cc:208-221,264-283), the cost terms and risk
transformation (tasks/task_test.cc:49-98) and state packing (state/state_test.cc:43-58, agent_utilities_test.cc:32-63,196-202), lifted
into tests/golden/mjpc_expectations.npz by tools/make_mjpc_expectations.py — literals of the reference's tests, no oracle-generated value.

The -m gpu tests run hb_ctrl_tape_splines / hb_ctrl_tape_read, hb_task_cost and hb_get/set_state (all through the C-ABI) against them;
the CPU tests hold tests/mjpc_ref.py (the numpy restatement the other planner tests use as their checker) to the same vectors."""
import json
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden", "mjpc_expectations.npz")
TWO_MOTORS = os.path.join(ROOT, "tests", "models", "two_motors.xml")
PARTICLE = os.path.join(ROOT, "tests", "golden", "particle_task.hbm")
TOL = 2e-6  # fp32 evaluation of values that are exact binary fractions or one rounding away from them; the reference compares with == (ElementsAre)


@pytest.fixture(scope="module")
def expect():
    m = json.loads(str(np.load(GOLD)["manifest"]))
    assert m["label"].startswith("REFERENCE-HELD")
    return m


def test_fixture_is_what_the_generator_lifts(expect):
    """The committed fixture equals what tools/make_mjpc_expectations.py builds (which re-checks every literal against the reference
    sources when the tree is present)."""
    import subprocess
    import sys
    before = open(GOLD, "rb").read()
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_mjpc_expectations.py")], stdout=subprocess.DEVNULL)
    assert json.loads(str(np.load(GOLD)["manifest"])) == expect
    open(GOLD, "wb").write(before)
    assert len(expect["spline"]) == 27 and sum(len(c["samples"]) for c in expect["spline"]) == 59


def test_numpy_restatement_meets_the_reference_spline_vectors(expect):
    from mjpc_ref import spline_sample
    for cs in expect["spline"]:
        if not cs["times"]:
            continue  # the empty spline: zeros by definition (spline.cc:108-111); spline_sample takes at least one node
        for t, want in cs["samples"]:
            got = spline_sample(cs["times"], cs["values"], cs["interp"], t)
            assert np.abs(got - np.array(want)).max() < 1e-14, (cs["cite"], t, got, want)


def test_numpy_restatement_meets_the_reference_cost_identities(expect):
    from mjpc_ref import terms_cost

    class T:
        pass
    tk = expect["task"]
    t = T()
    t.n_term, t.dim, t.norm, t.weight, t.norm_p, t.risk = 2, tk["dims"], tk["norms"], tk["weights"], [[0, 0], [0, 0]], 0.0
    assert abs(terms_cost(np.array(tk["residual"]), t) - tk["terms_sum"]) < 1e-18
    t.risk = tk["risk"]
    assert abs(terms_cost(np.array(tk["residual"]), t) - tk["cost_value"]) < 1e-18


@pytest.mark.gpu
def test_device_splines_meet_the_reference_vectors(hbmod, gpu, expect):
    """hb_ctrl_tape_splines + hb_ctrl_tape_read: one candidate per case, one tape step per sample time."""
    m = hbmod.Model.load(TWO_MOTORS)
    assert m.nu == 2
    b = hbmod.Batch(m, 4, 0)
    worst, n = 0.0, 0
    for cs in expect["spline"]:
        P = len(cs["times"])
        knots = np.zeros((4, P, 2), np.float32)
        if P:
            knots[:] = np.array(cs["values"], np.float32)[None]
        for t, want in cs["samples"]:
            b.ctrl_tape_splines(knots, np.array(cs["times"], np.float32), cs["interp"], t, 1)
            got = b.ctrl_tape_read(1)[0]
            assert got.shape == (4, 2)
            err = np.abs(got - np.array(want, np.float32)[None]).max()
            assert err <= TOL, (cs["cite"], cs["interp"], t, got[0], want)
            worst = max(worst, float(err)); n += 1
    assert n == 59
    print("device TimeSpline::Sample vs %d reference-held samples: worst |error| %.1e" % (n, worst))
    # the tape is gone once anything else writes the controls
    b.step(np.zeros((4, 2), np.float32))
    with pytest.raises(hbmod.HbError):
        b.ctrl_tape_read(1)
    b.close()


@pytest.mark.gpu
def test_device_clamp_meets_the_reference_vector(hbmod, gpu, expect):
    """Clamp(x, bounds) as SamplingPolicy::Action applies it to the spline's value (policy.cc:50-58), on the reference's particle model
    (ctrlrange [-1, 1]): x = {-2, 3, 0} -> {-1, 1, 0}."""
    m = hbmod.Model.load(PARTICLE)
    b = hbmod.Batch(m, 2, 0)
    x, want = expect["clamp"]["x"], expect["clamp"]["expect"]
    knots = np.array([[[x[0], x[1]]], [[x[2], x[0]]]], np.float32)
    b.ctrl_tape_splines(knots, np.array([0.0], np.float32), 0, 0.0, 3)
    got = b.ctrl_tape_read(3)
    for t in range(3):
        assert np.array_equal(got[t], np.array([[want[0], want[1]], [want[2], want[0]]], np.float32))
    b.close()


@pytest.mark.gpu
def test_device_cost_terms_meet_the_reference_identities(hbmod, gpu, expect):
    """hb_task_cost (the cost code of the hb_rollout_task_* kernels) on TasksTest.Task's residual: sum of the terms and the risk-sensitive
    value, relative 2e-6 where the reference allows 1e-5 absolute on numbers of order 1e-5; the same identities with the residual
    scaled by 1000 (cost 13.75, exp(2.75)) so that the exponential is exercised away from its linear range."""
    tk = expect["task"]
    m = hbmod.Model.load(PARTICLE)
    b = hbmod.Batch(m, 2, 0)
    r = np.array(tk["residual"])
    res = np.stack([r, 1000.0 * r])
    terms, cost = b.task_cost(res, tk["dims"], tk["norms"], tk["weights"], risk=0.0)
    c = np.array([tk["terms_sum"], 1e6 * tk["terms_sum"]])
    assert np.abs(terms.sum(1) / c - 1).max() < 2e-6 and np.abs(cost / c - 1).max() < 2e-6
    assert abs(terms[0, 0] / (5.0 * 0.5 * (r[0] ** 2 + r[1] ** 2)) - 1) < 2e-6 and abs(terms[0, 1] / (0.1 * 0.5 * (r[2] ** 2 + r[3] ** 2)) - 1) < 2e-6
    _, risky = b.task_cost(res, tk["dims"], tk["norms"], tk["weights"], risk=tk["risk"])
    want = (np.exp(tk["risk"] * c) - 1.0) / tk["risk"]
    assert abs(want[0] - tk["cost_value"]) < 1e-18
    # (exp(x) - 1) / risk in fp32 at x = 2.75e-6 loses digits to the subtraction exactly as an fp32 mju_exp would: 1e-5 absolute is the reference's bound
    assert abs(risky[0] - want[0]) < 1e-5 and abs(risky[1] / want[1] - 1) < 2e-6
    _, xml_risk = b.task_cost(res, tk["dims"], tk["norms"], tk["weights"], risk=tk["xml_risk"])
    assert abs(xml_risk[1] / ((np.exp(c[1]) - 1.0)) - 1) < 2e-6
    # a dimension mismatch is the reference's "mismatch between total user-sensor dimension and actual length of residual"
    with pytest.raises(hbmod.HbError):
        b.task_cost(res, [2, 1], tk["norms"], tk["weights"])
    b.close()


@pytest.mark.gpu
def test_device_state_packing_meets_the_reference_vectors(hbmod, gpu, expect):
    """hb_set_state / hb_get_state with mjSTATE_QPOS | mjSTATE_QVEL on the reference's particle model: State::Set's layout qpos | qvel
    (state_test.cc:43-58), the SetState / GetState round trip (agent_utilities_test.cc:43-62) and the "home" keyframe (:196-202).  The
    mocap half of State (state_test.cc:46-66) has no counterpart: the path's models have no mocap-driven dynamics (the particle's goal body
    is compiled as a static body)."""
    st = expect["state"]
    m = hbmod.Model.load(PARTICLE)
    assert (m.nq, m.nv) == (2, 2)
    b = hbmod.Batch(m, 3, 0)
    spec = hbmod.STATE_QPOS | hbmod.STATE_QVEL
    assert b.state_size(spec) == 4
    full = b.get_state(hbmod.STATE_INTEGRATION)
    full[:, 1:3] = st["qpos_fill"]
    full[:, 3:5] = st["qvel_fill"]
    b.set_state(hbmod.STATE_INTEGRATION, full)
    assert np.array_equal(b.get_state(spec), np.tile(np.array(st["expect"], np.float32), (3, 1)))
    rt = np.tile(np.array(st["roundtrip"], np.float32), (3, 1))
    b.set_state(spec, rt)
    assert np.array_equal(b.qpos, rt[:, :2]) and np.array_equal(b.qvel, rt[:, 2:])
    assert np.array_equal(b.get_state(spec), rt)
    key = m.name2id("key", st["key"])
    assert key == 0
    b.reset(keyframe=key)
    assert np.array_equal(b.qpos, np.tile(np.array(st["key_qpos"], np.float32), (3, 1)))
    b.close()
