"""The one expected result the reference holds at the mj_step boundary: RolloutTest.Particle
(mujoco_mpc/mjpc/test/agent/rollout_test.cc:67-153 with mjpc/test/testdata/particle.xml / particle_task.xml).

REFERENCE-HELD EXPECTATION, ORACLE-GENERATED VALUES: the reference holds the model, the PD policy, the horizon and the
bound on the final state (|pos - goal|_1 < 0.1, |vel|_1 < 0.1 after 99 closed-loop mj_step calls from the zero state —
a start that fails the bound by 0.2, so gain, damping sign, control clamp and the integrator are all pinned by it); the
100 x 4 state table in tests/golden/particle_rollout.npz comes from the fp64 oracle (tools/make_particle_golden.py) and
is checked here against the closed form of the same dynamics written independently.
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from make_particle_golden import GOAL, HBM, HORIZON, REF_XML, closed_form, oracle_rollout, policy  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden", "particle_rollout.npz")


def reference_bounds(states):
    """rollout_test.cc:137-145"""
    e = states[HORIZON - 1]
    assert np.abs(e[:2] - GOAL).sum() < 0.1
    assert np.abs(e[2:4]).sum() < 0.1


def test_start_state_fails_the_reference_bound():
    assert np.abs(np.zeros(2) - GOAL).sum() >= 0.1  # 0.2: the expectation is not vacuous


def test_oracle_meets_the_reference_expectation_and_the_closed_form():
    states, actions = oracle_rollout()
    reference_bounds(states)
    g = np.load(GOLD)
    assert np.array_equal(states, g["states"]) and np.array_equal(actions, g["actions"])
    cf = closed_form()
    reference_bounds(cf)
    assert np.abs(states - cf).max() < 1e-12
    # the first actions saturate the motors' ctrlrange: the clamp (mj_fwdActuation) is exercised
    assert (np.abs(actions[0]) >= 1.0).all() and np.abs(actions[-1]).max() < 1.0


@pytest.mark.skipif(not os.path.exists(REF_XML), reason="reference tree not present")
def test_committed_model_is_the_reference_xml_compiled(hbmod, tmp_path):
    m = hbmod.Model.load(REF_XML)
    out = str(tmp_path / "p.hbm")
    m.save(out)
    assert open(out).read() == open(HBM).read()
    assert (m.nq, m.nv, m.nu) == (2, 2, 2)
    assert m.opt.timestep == 0.01 and m.opt.disableflags & (1 << 4)  # <flag contact="disable"/>


@pytest.mark.gpu
def test_device_meets_the_reference_expectation(hbmod, gpu):
    """The same closed loop through hb_step (C-ABI): the policy reads the device state, the device steps."""
    g = np.load(GOLD)
    for solver in (2, 0):  # the model's own solver (none given: Newton) and PGS; with contacts off and no active limit both are idle
        m = hbmod.Model.load(HBM)
        m.set_opt(solver=solver)
        n = 64
        b = hbmod.Batch(m, n, gpu)
        b.reset()
        states = [np.concatenate([b.qpos[0], b.qvel[0]]).astype(np.float64)]
        for t in range(HORIZON - 1):
            st = np.concatenate([b.qpos, b.qvel], axis=1).astype(np.float64)
            act = np.stack([policy(s) for s in st])
            b.step(act.astype(np.float32))
            states.append(np.concatenate([b.qpos[0], b.qvel[0]]).astype(np.float64))
        states = np.array(states)
        reference_bounds(states)
        assert not b.status().any()
        # against the oracle's table: closed loop over 99 steps in fp32
        assert np.abs(states - g["states"]).max() < 2e-6, np.abs(states - g["states"]).max()
        # every env ran the same loop
        assert np.abs(b.qpos - b.qpos[0]).max() == 0.0
        b.close()
