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


def control_input_reward(cfg, time, qpos, qvel, joint_torques, prev_action, latest_action, self_collision):
    """reward_functions.py:116-245 with the explicit parameters of hb_env_config (reward_kind = 1)."""
    _, g = obs_from_state(qpos, qvel)
    r = cfg.w_hvel * scaled_exp(np.sum((qvel[0:2] - np.array(cfg.target_velocity[:])) ** 2))
    r += cfg.w_upright * scaled_exp(np.sum((g - np.array([0, 0, -1.0])) ** 2))
    r += cfg.w_vvel * scaled_exp(qvel[2] ** 2)
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
    upright = np.max(np.abs(g[0:2])) < cfg.upright_tol
    terminated = (not upright) or qpos[2] < cfg.min_z_grounded
    if terminated:
        r = cfg.terminal_reward
    truncated = cfg.max_time > 0 and time >= cfg.max_time
    return r, bool(terminated), bool(truncated)


# ---- realism layer (hb_env_randomization): the device's counter-based random numbers, delay rings and push schedule

RS_ACTION, RS_JOINT_POS, RS_JOINT_VEL, RS_GYRO, RS_IMU, RS_DELAY, RS_PUSH = 1, 2, 3, 4, 5, 6, 7
_M = 0xFFFFFFFF


def _mix(h, v):
    h ^= v & _M
    h = (h * 0x9E3779B1) & _M; h ^= h >> 15
    h = (h * 0x85EBCA77) & _M; h ^= h >> 13
    h = (h * 0xC2B2AE3D) & _M; h ^= h >> 16
    return h


def rng_u32(seed, env, ep, step, stream, idx):
    h = _mix(0x6A09E667, seed)
    for v in (env, ep, step, stream, idx):
        h = _mix(h, v)
    return h


def rng_uniform(seed, env, ep, step, stream, idx):
    return np.float32((np.float32(rng_u32(seed, env, ep, step, stream, idx) >> 8) + np.float32(0.5)) * np.float32(1.0 / 16777216.0))


def rng_normal(seed, env, ep, step, stream, idx):
    u1 = float(rng_uniform(seed, env, ep, step, stream, 2 * idx))
    u2 = float(rng_uniform(seed, env, ep, step, stream, 2 * idx + 1))
    return np.sqrt(-2.0 * np.log(u1)) * np.cos(6.28318530718 * u2)


class RealismRef:
    """One env of the realism layer: what hb_action_env_kernel / envrand_observe do, in numpy."""
    SLOTS = 64

    def __init__(self, R, env_global, nu, nj, nbody, timestep, episode=0):
        self.R, self.ge, self.nu, self.nj, self.nbody = R, env_global, nu, nj, nbody
        self.dt = R.control_timestep if R.control_timestep > 0 else timestep
        self.begin_episode(episode)

    def begin_episode(self, ep):
        R = self.R
        self.ep = ep
        self.delay = []
        for c in range(4):
            u = rng_uniform(R.seed, self.ge, ep, 0, RS_DELAY, c)
            d = np.float32(np.float32(R.min_delay + u * np.float32(R.max_delay - R.min_delay)) * np.float32(R.factor))
            self.delay.append(int(min(self.SLOTS - 1, max(0, np.rint(np.float32(d / np.float32(self.dt)))))))
        self.k_act = self.k_obs = 0
        self.act_hist, self.joint_hist, self.gyro_hist, self.grav_hist = [], [], [], []
        self.push = dict(start=0.0, dur=0.0, mag=0.0, dx=0.0, dy=0.0, body=0, ev=0)

    @staticmethod
    def _delayed(hist, d, filler):
        return hist[-1 - d] if len(hist) > d else filler

    def apply_action(self, action):
        R = self.R
        kk = 0 if R.frozen_noise else self.k_act
        a = np.array(action, dtype=np.float64)
        if R.action_noise > 0:
            a = a + R.factor * R.action_noise * np.array([rng_normal(R.seed, self.ge, self.ep, kk, RS_ACTION, i) for i in range(self.nu)])
        self.act_hist.append(a)
        self.k_act += 1
        return self._delayed(self.act_hist, self.delay[0], np.zeros(self.nu))

    def push_update(self, time):
        """returns (body, fx, fy) to apply for the coming step, or None"""
        R, p = self.R, self.push
        if not R.push_enabled:
            return None
        if time >= np.float32(np.float32(p["start"]) + np.float32(p["dur"])):
            ev = p["ev"]
            u = [float(rng_uniform(R.seed, self.ge, self.ep, ev, RS_PUSH, i)) for i in range(6)]
            p["start"] = float(np.float32(time) + np.float32(R.push_min_interval) + np.float32(u[0]) * np.float32(R.push_max_interval - R.push_min_interval))
            p["dur"] = R.push_min_duration + u[1] * (R.push_max_duration - R.push_min_duration)
            p["mag"] = R.factor * (R.push_min_force + u[2] * (R.push_max_force - R.push_min_force))
            dx, dy = 2 * u[3] - 1, 2 * u[4] - 1
            n = np.hypot(dx, dy)
            p["dx"], p["dy"] = dx / n, dy / n
            p["body"] = 1 + min(self.nbody - 2, int(np.float32(u[5]) * np.float32(self.nbody - 1)))
            p["ev"] = ev + 1
        if time > p["start"] and time < p["start"] + p["dur"]:
            return p["body"], p["dx"] * p["mag"], p["dy"] * p["mag"]
        return None

    def observe(self, qpos, qvel):
        R = self.R
        kk = 0 if R.frozen_noise else self.k_obs
        nj = self.nj
        o, _ = obs_from_state(qpos, qvel)
        ja = o[:nj] + R.factor * R.joint_angle_noise * np.array([rng_normal(R.seed, self.ge, self.ep, kk, RS_JOINT_POS, i) for i in range(nj)])
        jv = o[nj:2 * nj] + R.factor * R.joint_velocity_noise * np.array([rng_normal(R.seed, self.ge, self.ep, kk, RS_JOINT_VEL, i) for i in range(nj)])
        gy = o[2 * nj:2 * nj + 3] + R.factor * R.gyro_noise * np.array([rng_normal(R.seed, self.ge, self.ep, kk, RS_GYRO, i) for i in range(3)])
        q = np.array(qpos[3:7], dtype=np.float64) + R.factor * R.imu_noise * np.array([rng_normal(R.seed, self.ge, self.ep, kk, RS_IMU, i) for i in range(4)])
        qq = np.array(qpos, dtype=np.float64); qq[3:7] = q
        _, g = obs_from_state(qq, qvel)
        self.joint_hist.append(np.concatenate([ja, jv])); self.gyro_hist.append(gy); self.grav_hist.append(g)
        self.k_obs += 1
        return np.concatenate([self._delayed(self.joint_hist, self.delay[1], np.zeros(2 * nj)),
                               self._delayed(self.gyro_hist, self.delay[2], np.zeros(3)),
                               self._delayed(self.grav_hist, self.delay[3], np.array([0, 0, -1.0]))])
