"""stable-baselines3 VecEnv contract over the env entry points of libhb.so.

Replaces the stack ``DummyVecEnv([CPUEnv, ...])`` the reference trains and benchmarks with (rl/train.py:123-136,169-232,
simulation/benchmark.py:37-53; env contract simulation/cpu_env.py:374-416,676-693).  What SB3's learners and callbacks use of a VecEnv:
  ``reset() -> obs[N, nobs]``
  ``step_async(actions)``, ``step_wait() -> (obs, rewards, dones, infos)`` and ``step(actions)`` = the two in a row, finished envs reset
      in place; ``infos`` is a list of N dicts, and for an env that finished this step
          infos[i]["terminal_observation"]  the observation of the state the episode ended in (DummyVecEnv.step_wait; SAC / PPO bootstrap
                                            from it),
          infos[i]["TimeLimit.truncated"]   = truncated and not terminated (the learner bootstraps only then),
          infos[i]["is_success"]            (cpu_env.py:688-689: the env's `truncated` - in standupReward that IS success);
  ``num_envs``, ``observation_space`` / ``action_space`` (Box-like: low, high, shape, dtype, sample(), contains()), ``get_attr``,
  ``set_attr`` (rl/randomization_adaptation_callback.py:53-54 sets "randomization_factor"), ``env_method``, ``seed``, ``env_is_wrapped``,
  ``close``.
``step_arrays(actions) -> (obs, reward, terminated, truncated, infos)`` is the array form (gymnasium's vector-env tuple, infos a dict of
arrays) for callers that do not want N dicts per step; ``step_torch`` keeps everything on the GPU.  All arithmetic is in the HIP kernels.
"""
import numpy as np

from .engine import WARN_BADQACC, WARN_BADQPOS, WARN_BADQVEL, WARN_CNSTRFULL, WARN_CONTACTFULL, Batch, Model


class Box:
    """what SB3 reads of a gymnasium.spaces.Box (gymnasium is not a dependency of this package): cpu_env.py:56-63"""

    def __init__(self, low, high, shape, dtype=np.float32, seed=0):
        self.low = np.full(shape, low, dtype=dtype)
        self.high = np.full(shape, high, dtype=dtype)
        self.shape = tuple(shape)
        self.dtype = np.dtype(dtype)
        self._rng = np.random.default_rng(seed)

    def sample(self):
        return self._rng.uniform(self.low, self.high).astype(self.dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

    def seed(self, seed=None):
        self._rng = np.random.default_rng(seed)
        return [seed]

    def __repr__(self):
        return "Box(%g, %g, %r, %s)" % (float(self.low.flat[0]), float(self.high.flat[0]), self.shape, self.dtype.name)


_NO_INFO = {}  # (shared by the envs that did not finish an episode this step: SB3 copies an info dict before it adds to one)


class VecEnv:
    def __init__(self, model, n_envs, device=0, n_substeps=1, randomization_factor=1.0, realism=False, domain_randomization=False, seed=0, team=False, **reward_overrides):
        """realism=True adds CPUEnv's sensor/action noise, delay FIFOs and pushes (hb_env_randomization), scaled by
        randomization_factor exactly like the reset perturbation; domain_randomization=True draws per-env masses, floor
        friction, joint and actuator parameters at every reset (hb_domain_randomization).  team=True: the reference's own
        constants for its own robot (hb_env_team_config: obs[30] in JOINT_NAMES order, action[12], standup reset, kp = 2)."""
        self.model = model if isinstance(model, Model) else Model.load(model)
        self.batch = Batch(self.model, n_envs, device)
        self.num_envs = int(n_envs)
        self.n_substeps = int(n_substeps)
        self.copy_outputs = True  # False: step() returns views of the page-locked transfer record, valid until the next step() (saves four host copies)
        self.cfg = self.batch.env_team_config() if team else self.batch.env_default_config()
        self.team = bool(team)
        self.cfg.reset_perturb = float(randomization_factor)
        for k, v in reward_overrides.items():
            if not hasattr(self.cfg, k):
                raise AttributeError("unknown env parameter %r" % k)
            setattr(self.cfg, k, v)
        self.batch.env_configure(self.cfg)
        self.realism = None
        if realism:
            self.realism = self.batch.env_default_randomization()
            self.realism.seed = int(seed)
            self.realism.control_timestep = float(self.model.opt.timestep * self.n_substeps)
            self._apply_realism()
        self.domain = None
        if domain_randomization:
            self.domain = self.batch.env_default_domain_randomization()
            self.domain.seed = int(seed)
            if self.team:
                self.domain.kp_nominal = 2.0  # JOINT_P_GAIN (simulation_parameters.py:35): CPUEnv overwrites the motors' gain at every reset
                self.domain.floor_bump_max = 0.1  # MAX_FLOOR_BUMP_HEIGHT
            self._apply_domain()
        # gymnasium-style space descriptions (cpu_env.py:56-63: Box(-1,1,(nu,)), Box(-10,10,(nobs,)))
        self.action_shape = (self.model.nu,)
        self.observation_shape = (self.model.nobs,)
        self.action_low, self.action_high = -1.0, 1.0
        self.action_space = Box(-1.0, 1.0, self.action_shape, seed=seed)
        self.observation_space = Box(-10.0, 10.0, self.observation_shape, seed=seed)
        self.seed_value = int(seed)
        self.render_mode = None
        self.batch.env_terminal_obs(fetch=False)  # the env kernel records the observation an episode ends in (hb_env_terminal_obs)

    @property
    def randomization_factor(self):
        return float(self.cfg.reset_perturb)

    def _apply_realism(self):
        self.realism.factor = float(self.cfg.reset_perturb)
        self.batch.env_randomize(self.realism if self.realism.factor > 0 else None)

    def _apply_domain(self):
        self.domain.factor = float(self.cfg.reset_perturb)
        self.batch.env_domain_randomize(self.domain if self.domain.factor > 0 else None)

    ATTRS = ("randomization_factor", "num_envs", "n_substeps", "reward_kind", "max_time", "seed_value", "render_mode")

    def get_attr(self, name, indices=None):
        """VecEnv.get_attr: the attribute of every (selected) env - all envs of a batch share their parameters"""
        if name not in self.ATTRS:
            raise AttributeError(name)
        v = getattr(self.cfg, name) if name in ("reward_kind", "max_time") else getattr(self, name)
        return [v] * len(self._indices(indices))

    def set_attr(self, name, value, indices=None):
        if name != "randomization_factor":
            raise AttributeError(name)
        self.cfg.reset_perturb = float(np.clip(value, 0.0, 1.0))
        self.batch.env_configure(self.cfg)
        if self.realism is not None:
            self._apply_realism()
        if self.domain is not None:
            self._apply_domain()

    def env_method(self, method_name, *args, indices=None, **kwargs):
        """VecEnv.env_method: the per-env methods SB3's tooling calls on CPUEnv: its reward / termination functions are the batch's"""
        if method_name == "get_wrapper_attr":
            return self.get_attr(args[0], indices)
        raise AttributeError("env method %r: the envs of a batch live on the GPU (hb_env_*), not in Python objects" % method_name)

    def env_is_wrapped(self, wrapper_class, indices=None):
        return [False] * len(self._indices(indices))

    def seed(self, seed=None):
        """VecEnv.seed: env i is seeded seed + i in SB3; here the counter-based generators take (seed, global env index) themselves"""
        self.seed_value = int(0 if seed is None else seed)
        for r in (self.realism, self.domain):
            if r is not None:
                r.seed = self.seed_value
        if self.realism is not None:
            self._apply_realism()
        if self.domain is not None:
            self._apply_domain()
        self.action_space.seed(self.seed_value)
        return [self.seed_value + i for i in range(self.num_envs)]

    def _indices(self, indices):
        if indices is None:
            return range(self.num_envs)
        return [indices] if isinstance(indices, int) else list(indices)

    def reset(self):
        return self.batch.env_reset()

    def step_async(self, actions):
        """stable-baselines3's VecEnv.step_async: the step is enqueued (actions up, physics, observations down) and the host returns at once"""
        self.batch.env_step(actions, self.n_substeps, wait=False)
        self._pending = True

    def step_wait(self):
        """VecEnv.step_wait: (obs, rewards, dones, infos) of the step_async before it, infos a list of dicts (module docstring)"""
        assert getattr(self, "_pending", False), "step_wait without step_async"
        self._pending = False
        self.batch.sync()
        return self._sb3(*self._finish(*self.batch.env_step_result(getattr(self, "copy_outputs", True))))

    def step(self, actions):
        """VecEnv.step = step_async + step_wait"""
        self.step_async(actions)
        return self.step_wait()

    def step_arrays(self, actions):
        """the same step as arrays: (obs, reward, terminated, truncated, infos) with infos a dict of arrays"""
        return self._finish(*self.batch.env_step(actions, self.n_substeps, copy=getattr(self, "copy_outputs", True)))

    def _sb3(self, obs, rew, term, trunc, arr):
        done = arr["done"]
        infos = [_NO_INFO] * self.num_envs
        if done.any():
            tobs = self.batch.env_terminal_obs()  # one more transfer, only on the steps an episode ends
            for i in np.flatnonzero(done):
                infos[i] = {"terminal_observation": tobs[i].copy(), "TimeLimit.truncated": bool(trunc[i] and not term[i]), "is_success": bool(trunc[i]),
                            "warnings": int(arr["warnings"][i])}
        return obs, rew, done, infos

    def _finish(self, obs, rew, term, trunc):
        done = term | trunc
        # "warnings": the per-env HB_WARN_* bits (mjData.warning, mjdata.h:54-65): a contact or constraint-row overflow (rows were dropped for
        # that env-step) or a bad-state reset is visible to the training loop instead of silently changing that env's physics.  Polled every
        # `warning_period`-th step (default 16: the poll is one more stream synchronisation and device-to-host copy on the host-action path)
        # through hb_env_warnings, which also keeps the bits of episodes that ended - and were reset in place - between two polls: nothing is
        # lost, a bit is reported at the poll after it was raised.  A fresh array at every poll.
        self._steps = getattr(self, "_steps", 0) + 1
        if getattr(self, "_warn", None) is None or self._steps % max(1, int(getattr(self, "warning_period", 16))) == 0:
            self._warn = self.batch.env_warnings()
        w = self._warn
        infos = {"is_success": trunc.copy(), "done": done, "warnings": w,  # cpu_env.py:688-689
                 "overflow": (w & (WARN_CONTACTFULL | WARN_CNSTRFULL)) != 0}
        return obs, rew, term, trunc, infos

    def warning_counts(self):
        """Number of envs currently carrying each warning bit."""
        w = self.batch.status()
        return {"contact_full": int(((w & WARN_CONTACTFULL) != 0).sum()), "constraint_full": int(((w & WARN_CNSTRFULL) != 0).sum()),
                "bad_qpos": int(((w & WARN_BADQPOS) != 0).sum()), "bad_qvel": int(((w & WARN_BADQVEL) != 0).sum()), "bad_qacc": int(((w & WARN_BADQACC) != 0).sum())}

    def step_torch(self, actions):
        """The same step with the policy on the GPU: `actions` is a float32 CUDA tensor [n_envs, nu]; returns CUDA tensors
        (obs, reward, terminated, truncated) that are rewritten by the next call.  No host transfer and no host
        synchronisation: the batch's stream waits for the caller's current torch stream, and the caller's stream for the step."""
        import torch
        a = actions.contiguous()
        if getattr(self, "_t_out", None) is None:
            dev = a.device
            self._t_out = (torch.empty((self.num_envs, self.model.nobs), dtype=torch.float32, device=dev), torch.empty(self.num_envs, dtype=torch.float32, device=dev),
                           torch.empty(self.num_envs, dtype=torch.uint8, device=dev), torch.empty(self.num_envs, dtype=torch.uint8, device=dev))
            self._t_stream = torch.cuda.ExternalStream(self.batch.stream, device=dev)  # the batch's own HIP stream, seen by torch
        assert a.dtype == torch.float32 and tuple(a.shape) == (self.num_envs, self.model.nu)
        cur = torch.cuda.current_stream(a.device)
        self._t_stream.wait_stream(cur)   # the policy's output is complete before the step reads it
        o, r, te, tr = self._t_out
        self.batch.env_step_dev(a.data_ptr(), o.data_ptr(), r.data_ptr(), te.data_ptr(), tr.data_ptr(), self.n_substeps)
        cur.wait_stream(self._t_stream)   # and the caller's stream sees the results
        a.record_stream(self._t_stream)
        return o, r, te.view(torch.bool), tr.view(torch.bool)  # (the kernel writes 0 / 1 bytes: a view, not two conversion kernels)

    def close(self):
        self.batch.close()
