#!/usr/bin/env python3
"""Generates tests/golden/particle_rollout.npz: the one expected result the reference itself holds at the mj_step
boundary, mujoco_mpc/mjpc/test/agent/rollout_test.cc:67-153 (RolloutTest.Particle).

What the reference holds: the model (mjpc/test/testdata/particle.xml via particle_task.xml: two slide joints, damping 1,
point mass 0.3 kg, timestep 0.01, contacts disabled, motors with gear 1 and ctrlrange [-1, 1]), the closed-loop PD policy
(P = 10, D = 2.5, goal (0.1, 0.1), zero goal velocity), the horizon (100 states = 99 mj_step calls from the zero state)
and the EXPECTATION on the last state: |pos - goal|_1 < 0.1 and |vel|_1 < 0.1.  It holds no state values.

So this file is labelled: REFERENCE-HELD EXPECTATION, ORACLE-GENERATED VALUES.  The 100 x 4 state table is produced by
oracle/mjstep_oracle.c (the fp64 restatement) running the reference's policy closed loop, and is cross-checked here against the
closed form of the same dynamics written independently in Python (semi-implicit Euler with implicit joint damping:
(m + h d) a = u - d v; v += h a; x += h v).  tests/test_particle_fixture.py asserts the reference's bounds on the oracle,
on the closed form and (-m gpu) on the device.

Run from the repository root (needs /root/reference only to recompile the model; the compiled model is committed):
    python tools/make_particle_golden.py
"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
REF_XML = "/root/reference/mujoco_mpc/mjpc/test/testdata/particle_task.xml"
HBM = os.path.join(ROOT, "tests", "golden", "particle_task.hbm")

GOAL = np.array([0.1, 0.1])
P_GAIN, D_GAIN = 10.0, 2.5
HORIZON = 100


def policy(state):
    """rollout_test.cc:83-101: action = -P (pos - goal) - D (vel - 0)."""
    return -P_GAIN * (state[:2] - GOAL) - D_GAIN * state[2:4]


def closed_form(horizon=HORIZON, mass=0.3, damping=1.0, h=0.01):
    x, v = np.zeros(2), np.zeros(2)
    states = [np.concatenate([x, v])]
    for _ in range(horizon - 1):
        u = np.clip(policy(states[-1]), -1.0, 1.0)   # ctrllimited, ctrlrange [-1, 1]; gear 1
        a = (u - damping * v) / (mass + h * damping)  # mj_Euler with implicit joint damping
        v = v + h * a
        x = x + h * v
        states.append(np.concatenate([x, v]))
    return np.array(states)


def oracle_rollout(horizon=HORIZON):
    from oracle_lib import Oracle
    o = Oracle(HBM)
    o.reset()
    states = [np.concatenate([o.qpos, o.qvel])]
    actions = []
    for _ in range(horizon - 1):
        a = policy(states[-1])
        o.ctrl[:] = a
        o.step()
        actions.append(a)
        states.append(np.concatenate([o.qpos, o.qvel]))
    return np.array(states), np.array(actions)


def main():
    if os.path.exists(REF_XML):
        subprocess.check_call([os.path.join(ROOT, "build", "hb_compile"), REF_XML, HBM])
    states, actions = oracle_rollout()
    cf = closed_form()
    assert np.abs(states - cf).max() < 1e-12, np.abs(states - cf).max()
    np.savez(os.path.join(ROOT, "tests", "golden", "particle_rollout.npz"), states=states, actions=actions, goal=GOAL,
             label=np.array("reference-held expectation (mujoco_mpc/mjpc/test/agent/rollout_test.cc:137-145), oracle-generated values"))
    e = states[-1]
    print("final |pos - goal|_1 = %.4f, |vel|_1 = %.4f (reference bound 0.1 each); start %.2f" % (np.abs(e[:2] - GOAL).sum(), np.abs(e[2:]).sum(), np.abs(states[0][:2] - GOAL).sum()))


if __name__ == "__main__":
    main()
