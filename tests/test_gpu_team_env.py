"""f1 for the reference's own robot (simulation/cpu_env.py:84-133,374-416,465-616 on simulation/assets/
world.xml): obs[30] = [q_j(12), qd_j(12), qvel[3:6], R(q_root)' (0, 0, -1)] with the joints in JOINT_NAMES order
(simulation_parameters.py:84-103), action[12] in the same order, standupReward with the reference's constants, the
standup reset (lying on the floor) with its perturbations, kp / force-range randomisation on the twelve motors."""
import os

import numpy as np
import pytest

from env_ref import obs_from_state, standup_reward
from oracle_lib import ROOT, Oracle

pytestmark = pytest.mark.gpu
TEAM_HBM = os.path.join(ROOT, "humanoid_mujoco_amd", "assets", "team_robot.hbm")
JOINT_NAMES = ["right_shoulder_pitch", "right_shoulder_roll", "right_elbow", "left_shoulder_pitch", "left_shoulder_roll", "left_elbow",
               "left_hip_roll", "left_hip_pitch", "left_knee", "right_hip_roll", "right_hip_pitch", "right_knee"]


def _team_obs(m, qpos, qvel):
    jid = [m.name2id("joint", n) for n in JOINT_NAMES]
    qadr = m.array("jnt_qposadr").astype(int)[jid]
    dadr = m.array("jnt_dofadr").astype(int)[jid]
    _, g = obs_from_state(qpos, qvel)
    return np.concatenate([qpos[qadr], qvel[dadr], qvel[3:6], g]), dadr


def test_team_config_is_the_references(hbmod, gpu):
    m = hbmod.Model.load(TEAM_HBM)
    env = hbmod.VecEnv(m, 4, gpu, team=True)
    c = env.cfg
    assert (c.target_z, c.min_z, c.max_time, c.safe_torque, c.control_frequency) == (-0.375, pytest.approx(-0.6), 10.0, 1.0, 500.0)
    assert (c.w_hvel, c.w_upright, c.w_height, c.w_torque, c.w_ctrl_change, c.w_ctrl_reg, c.w_symmetry) == (5, 10, 15, 2.5, 2, 0.5, 1)
    assert (c.self_collision_penalty, c.terminal_reward, c.upright_tol) == (-20, -100, pytest.approx(0.7))
    a = {n: m.name2id("actuator", n) for n in JOINT_NAMES}
    assert [a[n] for n in JOINT_NAMES] == list(range(12))  # the <motor> order IS JOINT_NAMES (assets/humanoid.xml:97-110)
    assert c.n_equal == 1 and tuple(c.equal_pairs[0]) == (a["left_elbow"], a["right_elbow"])
    assert c.n_opposite == 5 and tuple(c.opposite_pairs[0]) == (a["left_hip_roll"], a["right_hip_roll"]) and tuple(c.opposite_pairs[4]) == (a["left_shoulder_roll"], a["right_shoulder_roll"])
    assert c.obs_actuator_order == 1 and c.reset_quat_perturb == pytest.approx(0.1) and m.nobs == 30 and m.nu == 12
    assert m.name2id("key", "standup_reset") == c.reset_keyframe == 0
    env.close()


def test_team_reset_is_the_standup_reset_with_its_perturbations(hbmod, gpu):
    m = hbmod.Model.load(TEAM_HBM)
    n = 64
    env = hbmod.VecEnv(m, n, gpu, team=True)
    obs = env.reset()
    st = env.batch.get_state(hbmod.STATE_INTEGRATION, dtype=np.float64)
    q = st[:, 1:1 + m.nq]
    # root: xy untouched, z in [-0.6, -0.5] and then one settle step of free fall; quaternion (-.5, -.5, .5, .5) +- 0.1 per component
    assert np.abs(q[:, 0:2]).max() < 1e-3 and (q[:, 2] > -0.61).all() and (q[:, 2] < -0.49).all()
    quat = q[:, 3:7]
    assert np.abs(quat / np.linalg.norm(quat, axis=1, keepdims=True) - np.array([-.5, -.5, .5, .5])).max() < 0.16
    assert np.abs(quat - np.array([-.5, -.5, .5, .5])).max() > 0.02  # it is perturbed
    assert np.abs(q[:, 7:]).max() <= 0.2 + 0.05 and np.abs(q[:, 7:]).max() > 0.1  # joints +- 0.2 rad (JOINT_INITIAL_OFFSET_MAX)
    nc, _, _ = env.batch.counts()
    assert not nc.any()  # CPUEnv.reset starts over while anything is in contact (cpu_env.py:411-414)
    for e in range(0, n, 7):
        want, _ = _team_obs(m, q[e], st[e, 1 + m.nq:1 + m.nq + m.nv])
        assert np.allclose(obs[e], want, atol=1e-5)
    env.close()


def test_team_step_reward_and_observation_match_the_reference_formula(hbmod, gpu):
    m = hbmod.Model.load(TEAM_HBM)
    n = 16
    env = hbmod.VecEnv(m, n, gpu, team=True, auto_reset=0)
    env.reset()
    b, cfg = env.batch, env.cfg
    o = Oracle(TEAM_HBM)
    rng = np.random.default_rng(4)
    prev = np.zeros((n, m.nu))
    worst = 0.0
    selfcols = 0
    for t in range(150):
        act = (rng.uniform(-1, 1, (n, m.nu)) * (0.3 if t % 2 else 1.0)).astype(np.float32)
        st = b.get_state(hbmod.STATE_INTEGRATION, dtype=np.float64)
        obs, rew, term, trunc, info = env.step_arrays(act)
        st1 = b.get_state(hbmod.STATE_INTEGRATION, dtype=np.float64)
        assert not term.any() and not info["warnings"].any()
        for e in range(0, n, 3):
            o.reset()
            o.qpos[:] = st[e, 1:1 + m.nq]; o.qvel[:] = st[e, 1 + m.nq:1 + m.nq + m.nv]; o.qacc_warmstart[:] = st[e, 1 + m.nq + m.nv:]
            o.ctrl[:] = act[e]
            o.forward()
            q1, v1 = st1[e, 1:1 + m.nq], st1[e, 1 + m.nq:1 + m.nq + m.nv]
            want, dadr = _team_obs(m, q1, v1)
            assert np.allclose(obs[e], want, atol=1e-5)
            torques = (o.qfrc_smooth + o.qfrc_constraint)[dadr]
            selfcol = any(c["geom1"] > 3 for c in o.contacts())  # geoms 0..3 are the world's (three markers and the floor)
            selfcols += selfcol
            r_ref, te, tr = standup_reward(cfg, st1[e, 0], q1, v1, torques, prev[e], act[e].astype(np.float64), selfcol)
            worst = max(worst, abs(rew[e] - r_ref))
            # (hull against hull, the self-collision case, is where the MPR portal and with it the joint torques are least unique)
            assert abs(rew[e] - r_ref) <= (3e-2 if selfcol else 5e-3) * max(1.0, abs(r_ref)), (t, e, rew[e], r_ref)
            assert bool(trunc[e]) == tr
        prev = act.astype(np.float64)
    print("\nteam env: worst |reward - reference formula| %.2e over 150 steps; self-collision seen %d times" % (worst, selfcols))
    env.close()


def test_team_domain_randomisation_sets_the_motor_gains(hbmod, gpu):
    """cpu_env.py:214-237: every reset draws kp = JOINT_P_GAIN +- JOINT_P_GAIN_MAX_CHANGE (2 +- 0.5) and the force range
    +- JOINT_FORCE_LIMIT_MAX_CHANGE (0.05) per motor."""
    m = hbmod.Model.load(TEAM_HBM)
    n = 128
    env = hbmod.VecEnv(m, n, gpu, team=True, domain_randomization=True)
    env.reset()
    prm = env.batch.env_domain_params()
    nb, nv, nu = m.nbody, m.nv, m.nu
    nlim = 2 * 12
    o_gain = nb + 2 * nv + 2 * nlim
    gain = prm[:, o_gain:o_gain + nu]
    frc = prm[:, o_gain + 2 * nu:o_gain + 4 * nu].reshape(n, nu, 2)
    assert gain.min() >= 1.5 - 1e-6 and gain.max() <= 2.5 + 1e-6 and gain.std() > 0.2
    assert np.abs(frc[:, :, 0] + 1).max() <= 0.05 + 1e-6 and np.abs(frc[:, :, 1] - 1).max() <= 0.05 + 1e-6
    # and the physics uses them: one step with full command from rest, the hinge accelerations scale with the drawn gains
    obs, rew, term, trunc, info = env.step_arrays(np.ones((n, nu), np.float32))
    assert np.isfinite(obs).all() and np.isfinite(rew).all()
    env.close()
