"""numpy (fp64) restatement of the env adapter's reward/termination for the tests — follows
simulation/reward_functions.py:17-113,247-374 (standupReward) with the explicit parameters of
hb_env_config.  Test infrastructure only."""
import numpy as np


def scaled_exp(x):
    return np.exp(-x / 0.5)


def obs_from_state(qpos, qvel):
    q = qpos[3:7] / np.linalg.norm(qpos[3:7])
    w, x, y, z = q
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                  [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                  [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
    g = R.T @ np.array([0, 0, -1.0])
    return np.concatenate([qpos[7:], qvel[6:], qvel[3:6], g]), g


def standup_reward(cfg, time, qpos, qvel, joint_torques, prev_action, latest_action, self_collision):
    _, g = obs_from_state(qpos, qvel)
    r = cfg.w_hvel * scaled_exp(np.sum((qvel[0:2] - np.array(cfg.target_velocity[:])) ** 2))
    r += cfg.w_upright * scaled_exp(np.sum((g - np.array([0, 0, -1.0])) ** 2))
    r += np.interp(qpos[2], [cfg.min_z, cfg.target_z], [0, cfg.w_height])
    r += cfg.w_torque * np.mean(scaled_exp(np.clip(np.abs(joint_torques) - cfg.safe_torque, 0, np.inf) ** 2))
    p, l = prev_action / cfg.action_scale, latest_action / cfg.action_scale
    r += cfg.w_ctrl_change * np.mean(scaled_exp(((l - p) * cfg.control_frequency) ** 2))
    r += cfg.w_ctrl_reg * np.mean(scaled_exp(l ** 2))
    if cfg.n_equal + cfg.n_opposite:
        s = sum(scaled_exp((l[cfg.equal_pairs[k][0]] - l[cfg.equal_pairs[k][1]]) ** 2) for k in range(cfg.n_equal))
        s += sum(scaled_exp((l[cfg.opposite_pairs[k][0]] + l[cfg.opposite_pairs[k][1]]) ** 2) for k in range(cfg.n_opposite))
        r += cfg.w_symmetry * s / (cfg.n_equal + cfg.n_opposite)
    if self_collision:
        r += cfg.self_collision_penalty
    terminated = cfg.max_time > 0 and time >= cfg.max_time
    if terminated:
        r = cfg.terminal_reward
    truncated = qpos[2] >= cfg.target_z and np.max(np.abs(g[0:2])) < cfg.upright_tol
    return r, bool(terminated), bool(truncated)
