"""ctypes host binding of libhb.so (include/hb.h) — the Python side of the drop-in boundary.

Mirrors how the reference's Python reaches its physics step: ``mujoco.MjModel.from_xml_path`` /
``mujoco.MjData`` / ``mujoco.mj_step`` in simulation/cpu_env.py:86,397,684 become
``Model.load`` / ``Batch`` / ``Batch.step``.  All compute happens in the HIP kernels behind the
C-ABI; this module only moves pointers.  There is no CPU fallback: creating a Batch without a
GPU raises.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# (HB_LIB: another build of the same library, e.g. build/libhb_stamps.so with the per-stage cycle stamps; there is no other backend)
LIB_PATH = os.environ.get("HB_LIB") or os.path.join(_HERE, "libhb.so")

HB_OK = 0
WARN_CONTACTFULL, WARN_CNSTRFULL, WARN_BADQPOS, WARN_BADQVEL, WARN_BADQACC = 1 << 1, 1 << 2, 1 << 4, 1 << 5, 1 << 6
STATE_TIME, STATE_QPOS, STATE_QVEL, STATE_WARMSTART, STATE_XFRC_APPLIED = 1 << 0, 1 << 1, 1 << 2, 1 << 4, 1 << 7
STATE_PHYSICS = STATE_QPOS | STATE_QVEL
STATE_INTEGRATION = STATE_TIME | STATE_QPOS | STATE_QVEL | STATE_WARMSTART

# mjtDisableBit (simulation/mujoco/include/mujoco/mjmodel.h:50-68)
DSBL_CONSTRAINT, DSBL_LIMIT, DSBL_CONTACT, DSBL_PASSIVE, DSBL_GRAVITY = 1 << 0, 1 << 3, 1 << 4, 1 << 5, 1 << 6
DSBL_CLAMPCTRL, DSBL_WARMSTART, DSBL_FILTERPARENT, DSBL_ACTUATION, DSBL_REFSAFE, DSBL_EULERDAMP = 1 << 7, 1 << 8, 1 << 9, 1 << 10, 1 << 11, 1 << 14


class HbOptions(ctypes.Structure):
    _fields_ = [("timestep", ctypes.c_double), ("gravity", ctypes.c_double * 3), ("impratio", ctypes.c_double),
                ("tolerance", ctypes.c_double), ("iterations", ctypes.c_int), ("solver", ctypes.c_int),
                ("cone", ctypes.c_int), ("integrator", ctypes.c_int), ("disableflags", ctypes.c_int),
                ("ls_iterations", ctypes.c_int), ("ls_tolerance", ctypes.c_double)]


class HbSizes(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int) for n in ("nq", "nv", "nu", "nbody", "njnt", "ngeom", "ntendon", "nM", "nkey",
                                            "npair", "nobs", "ncon_max", "nefc_max")]


class HbEnvConfig(ctypes.Structure):
    """hb_env_config (include/hb.h): reward / termination parameters of the env adapter."""
    _fields_ = [("target_velocity", ctypes.c_float * 2), ("target_z", ctypes.c_float), ("min_z", ctypes.c_float),
                ("max_time", ctypes.c_float), ("safe_torque", ctypes.c_float), ("control_frequency", ctypes.c_float),
                ("action_scale", ctypes.c_float), ("w_hvel", ctypes.c_float), ("w_upright", ctypes.c_float),
                ("w_height", ctypes.c_float), ("w_torque", ctypes.c_float), ("w_ctrl_change", ctypes.c_float),
                ("w_ctrl_reg", ctypes.c_float), ("w_symmetry", ctypes.c_float), ("self_collision_penalty", ctypes.c_float),
                ("terminal_reward", ctypes.c_float), ("upright_tol", ctypes.c_float), ("n_equal", ctypes.c_int),
                ("n_opposite", ctypes.c_int), ("equal_pairs", (ctypes.c_int * 2) * 16), ("opposite_pairs", (ctypes.c_int * 2) * 16),
                ("auto_reset", ctypes.c_int), ("reset_keyframe", ctypes.c_int), ("reset_perturb", ctypes.c_float),
                ("reward_kind", ctypes.c_int), ("w_vvel", ctypes.c_float), ("min_z_grounded", ctypes.c_float),
                ("reset_collision_mode", ctypes.c_int), ("reset_quat_perturb", ctypes.c_float), ("obs_actuator_order", ctypes.c_int)]


class HbEnvRandomization(ctypes.Structure):
    """hb_env_randomization (include/hb.h): sensor/action noise, delay FIFOs and pushes of CPUEnv, per env on the device."""
    _fields_ = [("factor", ctypes.c_float), ("seed", ctypes.c_uint), ("control_timestep", ctypes.c_float),
                ("joint_angle_noise", ctypes.c_float), ("joint_velocity_noise", ctypes.c_float), ("gyro_noise", ctypes.c_float),
                ("imu_noise", ctypes.c_float), ("action_noise", ctypes.c_float), ("min_delay", ctypes.c_float),
                ("max_delay", ctypes.c_float), ("frozen_noise", ctypes.c_int), ("push_enabled", ctypes.c_int),
                ("push_min_interval", ctypes.c_float), ("push_max_interval", ctypes.c_float), ("push_min_duration", ctypes.c_float),
                ("push_max_duration", ctypes.c_float), ("push_min_force", ctypes.c_float), ("push_max_force", ctypes.c_float)]


class HbTaskStand(ctypes.Structure):
    _fields_ = [("head_body", ctypes.c_int), ("n_feet", ctypes.c_int), ("foot_body", ctypes.c_int * 4), ("foot_offset", (ctypes.c_float * 3) * 4),
                ("subtree_body", ctypes.c_int), ("height_goal", ctypes.c_float), ("norm", ctypes.c_int * 5), ("weight", ctypes.c_float * 5),
                ("norm_p", (ctypes.c_float * 2) * 5), ("risk", ctypes.c_float)]


class HbTaskWalk(ctypes.Structure):
    _fields_ = [("torso_body", ctypes.c_int), ("pelvis_body", ctypes.c_int), ("foot_right_body", ctypes.c_int), ("foot_left_body", ctypes.c_int),
                ("waist_lower_body", ctypes.c_int), ("height_goal", ctypes.c_float), ("speed_goal", ctypes.c_float), ("n_term", ctypes.c_int),
                ("dim", ctypes.c_int * 8), ("norm", ctypes.c_int * 8), ("weight", ctypes.c_float * 8), ("norm_p", (ctypes.c_float * 2) * 8),
                ("risk", ctypes.c_float)]


class HbCostSpec(ctypes.Structure):
    """hb_cost_spec (include/hb.h): cost terms over consecutive slices of a residual vector (mjpc task.cc:71-110)."""
    _fields_ = [("n_term", ctypes.c_int), ("dim", ctypes.c_int * 8), ("norm", ctypes.c_int * 8), ("weight", ctypes.c_float * 8),
                ("norm_p", (ctypes.c_float * 2) * 8), ("risk", ctypes.c_float)]


class HbSensorSpec(ctypes.Structure):
    """hb_sensor_spec (include/hb.h): framepos bodies and the tree whose subtreecom / subtreelinvel are read out."""
    _fields_ = [("n_framepos", ctypes.c_int), ("framepos_body", ctypes.c_int * 16), ("subtree_body", ctypes.c_int),
                ("framepos_offset", (ctypes.c_float * 3) * 16),
                ("n_frameaxis", ctypes.c_int), ("frameaxis_body", ctypes.c_int * 8), ("frameaxis_which", ctypes.c_int * 8),
                ("n_framelinvel", ctypes.c_int), ("framelinvel_body", ctypes.c_int * 8),
                ("n_subtreelinvel", ctypes.c_int), ("subtreelinvel_body", ctypes.c_int * 4)]


class HbDomainRandomization(ctypes.Structure):
    """hb_domain_randomization (include/hb.h): per-env model parameters drawn at reset."""
    _fields_ = [("factor", ctypes.c_float), ("seed", ctypes.c_uint), ("friction_min_mult", ctypes.c_float), ("friction_max_mult", ctypes.c_float),
                ("max_mass_change", ctypes.c_float), ("max_external_mass", ctypes.c_float), ("armature_max_change", ctypes.c_float),
                ("stiffness_max_change", ctypes.c_float), ("margin_max_change", ctypes.c_float), ("range_max_change", ctypes.c_float),
                ("kp_nominal", ctypes.c_float), ("kp_max_change", ctypes.c_float), ("force_limit_max_change", ctypes.c_float),
                ("floor_bump_min", ctypes.c_float), ("floor_bump_max", ctypes.c_float)]


class HbError(RuntimeError):
    pass


_lib = None


def lib():
    """Load libhb.so; fails loudly when the extension has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HbError("libhb.so is missing (%s): build it with `make` or __graft_entry__.build(); "
                      "this package has no fallback path" % LIB_PATH)
    L = ctypes.CDLL(LIB_PATH)
    vp, cp, ci, cu = ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_uint
    L.hb_version.restype = cp
    L.hb_model_load.restype = vp; L.hb_model_load.argtypes = [cp, cp, ci]
    L.hb_model_load_xml_string.restype = vp; L.hb_model_load_xml_string.argtypes = [cp, cp, ci]
    L.hb_model_save.argtypes = [vp, cp, cp, ci]
    L.hb_model_free.argtypes = [vp]; L.hb_model_free.restype = None
    L.hb_model_sizes.argtypes = [vp, ctypes.POINTER(HbSizes)]
    L.hb_options_get.argtypes = [vp, ctypes.POINTER(HbOptions)]
    L.hb_options_set.argtypes = [vp, ctypes.POINTER(HbOptions)]
    L.hb_model_name2id.argtypes = [vp, cp, cp]
    L.hb_model_get_array.argtypes = [vp, cp, vp, ci]
    L.hb_batch_create.restype = vp; L.hb_batch_create.argtypes = [vp, ci, ci, cp, ci]
    L.hb_batch_free.argtypes = [vp]; L.hb_batch_free.restype = None
    L.hb_batch_n_env.argtypes = [vp]
    L.hb_batch_stream.restype = vp; L.hb_batch_stream.argtypes = [vp]
    L.hb_batch_sync.argtypes = [vp]
    L.hb_batch_pipeline.argtypes = [vp, ci]
    L.hb_batch_join.argtypes = [vp]
    L.hb_reset.argtypes = [vp, vp, ci, ci, ci]
    L.hb_step.argtypes = [vp, vp, ci]
    L.hb_step_dev.argtypes = [vp, vp, ci]
    L.hb_rollout.argtypes = [vp, vp, ci, vp]
    L.hb_rollout_dev.argtypes = [vp, vp, ci, vp]
    L.hb_rollout_halton.argtypes = [vp, ci, ci, ci, vp]
    L.hb_forward.argtypes = [vp, vp]
    L.hb_state_size.argtypes = [vp, cu]
    L.hb_get_state.argtypes = [vp, cu, vp]; L.hb_set_state.argtypes = [vp, cu, vp]
    L.hb_get_state_f64.argtypes = [vp, cu, vp]; L.hb_set_state_f64.argtypes = [vp, cu, vp]
    L.hb_get_obs.argtypes = [vp, vp, vp, vp, vp]
    L.hb_get_status.argtypes = [vp, vp]
    L.hb_get_counts.argtypes = [vp, vp, vp, vp]
    L.hb_last_kernel.argtypes = [vp]; L.hb_last_kernel.restype = ctypes.c_char_p
    L.hb_batch_step_launches.argtypes = [vp]; L.hb_batch_step_launches.restype = ctypes.c_longlong
    L.hb_batch_tune.argtypes = [vp, ci, ci]
    L.hb_model_pair_order.argtypes = [vp, ci]
    L.hb_env_terminal_obs.argtypes = [vp, vp]
    L.hb_env_warnings.argtypes = [vp, vp]
    L.hb_batch_device_name.argtypes = [vp, cp, ci]
    L.hb_get_collision_counts.argtypes = [vp, vp, vp, vp]
    L.hb_batch_segments.argtypes = [vp]
    L.hb_diag_enable.argtypes = [vp, ci]
    L.hb_get_qacc.argtypes = [vp, vp]; L.hb_get_efc_force.argtypes = [vp, vp]; L.hb_get_contacts.argtypes = [vp, vp]
    L.hb_env_default_config.argtypes = [vp, ctypes.POINTER(HbEnvConfig)]
    L.hb_env_team_config.argtypes = [vp, ctypes.POINTER(HbEnvConfig)]
    L.hb_env_configure.argtypes = [vp, ctypes.POINTER(HbEnvConfig)]
    L.hb_env_default_randomization.argtypes = [vp, ctypes.POINTER(HbEnvRandomization)]
    L.hb_env_randomize.argtypes = [vp, ctypes.POINTER(HbEnvRandomization)]
    L.hb_env_default_domain_randomization.argtypes = [vp, ctypes.POINTER(HbDomainRandomization)]
    L.hb_env_domain_randomize.argtypes = [vp, ctypes.POINTER(HbDomainRandomization)]
    L.hb_env_get_domain_params.argtypes = [vp, vp]
    L.hb_state_to_proto.argtypes = [vp, ci, vp, ci]
    L.hb_state_from_proto.argtypes = [vp, ci, cp, ci]
    L.hb_sensor_size.argtypes = [ctypes.POINTER(HbSensorSpec)]
    L.hb_set_state_broadcast.argtypes = [vp, cu, vp]
    L.hb_set_state_broadcast_f64.argtypes = [vp, cu, vp]
    L.hb_rollout_sensors.argtypes = [vp, vp, ci, ctypes.POINTER(HbSensorSpec), vp, vp]
    L.hb_rollout_trajectory.argtypes = [vp, vp, ci, vp, vp, vp]
    L.hb_rollout_noise.argtypes = [vp, ctypes.c_float, ctypes.c_float, ctypes.c_uint]
    L.hb_ctrl_tape_splines.argtypes = [vp, vp, vp, ci, ci, ctypes.c_double, ci]
    L.hb_ctrl_tape_read.argtypes = [vp, ci, vp]
    L.hb_task_cost.argtypes = [vp, vp, ci, ci, ctypes.POINTER(HbCostSpec), vp, vp]
    L.hb_transition_fd.argtypes = [vp, vp, vp, vp, ci, ctypes.c_double, ci, vp, vp]
    L.hb_transition_fd_sensors.argtypes = [vp, vp, vp, vp, ci, ctypes.c_double, ci, ctypes.POINTER(HbSensorSpec), vp, vp, vp, vp]
    L.hb_task_stand_default.argtypes = [vp, ctypes.POINTER(HbTaskStand)]
    L.hb_rollout_task_stand.argtypes = [vp, vp, ci, ctypes.POINTER(HbTaskStand), vp, vp]
    L.hb_task_walk_default.argtypes = [vp, ctypes.POINTER(HbTaskWalk)]
    L.hb_rollout_task_walk.argtypes = [vp, vp, ci, ctypes.POINTER(HbTaskWalk), vp, vp]
    L.hb_sensors.argtypes = [vp, vp, ctypes.POINTER(HbSensorSpec), vp]
    L.hb_env_reset.argtypes = [vp, vp]
    L.hb_env_step.argtypes = [vp, vp, ci, vp, vp, vp, vp]
    L.hb_env_step_async.argtypes = [vp, vp, ci, vp, vp, vp, vp]
    L.hb_env_step_dev.argtypes = [vp, vp, ci, vp, vp, vp, vp]
    L.hb_policy_set_mlp.argtypes = [vp, ci, vp, vp, vp]
    L.hb_policy_eval.argtypes = [vp, vp]
    L.hb_rollout_policy.argtypes = [vp, ci, vp]
    L.hb_dev_alloc.restype = vp; L.hb_dev_alloc.argtypes = [vp, ctypes.c_uint64]
    L.hb_dev_free.restype = None; L.hb_dev_free.argtypes = [vp, vp]
    L.hb_host_alloc.restype = vp; L.hb_host_alloc.argtypes = [ctypes.c_uint64]
    L.hb_host_free.restype = None; L.hb_host_free.argtypes = [vp]
    L.hb_memcpy_h2d.argtypes = [vp, vp, vp, ctypes.c_uint64]; L.hb_memcpy_d2h.argtypes = [vp, vp, vp, ctypes.c_uint64]
    L.hb_halton_ctrl_dev.argtypes = [vp, ci, ci, ci, vp]
    L.hb_timer_start.argtypes = [vp]; L.hb_timer_stop.argtypes = [vp, ctypes.POINTER(ctypes.c_float)]
    L.hb_step_timing.argtypes = [vp, ci]; L.hb_step_timing_read.argtypes = [vp, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ci)]
    _lib = L
    return L


def _check(rc, what):
    if rc != HB_OK:
        raise HbError("%s failed with code %d" % (what, rc))


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


class _Pinned:
    """numpy array over page-locked host memory (hb_host_alloc): copies to and from it are DMA transfers."""

    def __init__(self, shape, dtype):
        self.nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        self.ptr = lib().hb_host_alloc(max(1, self.nbytes))
        if not self.ptr:
            raise HbError("hb_host_alloc(%d) failed" % self.nbytes)
        buf = (ctypes.c_char * max(1, self.nbytes)).from_address(self.ptr)
        self.array = np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)

    def free(self):
        if self.ptr:
            self.array = None
            lib().hb_host_free(ctypes.c_void_p(self.ptr))
            self.ptr = None


class Model:
    """Compiled model (replaces mujoco.MjModel for this path)."""

    def __init__(self, handle):
        self._h = handle
        self._read_sizes()

    def _read_sizes(self):
        s = HbSizes()
        _check(lib().hb_model_sizes(self._h, ctypes.byref(s)), "hb_model_sizes")
        for name, _ in HbSizes._fields_:
            setattr(self, name, getattr(s, name))

    @classmethod
    def load(cls, path):
        err = ctypes.create_string_buffer(1024)
        h = lib().hb_model_load(os.fsencode(path), err, len(err))
        if not h:
            raise HbError("hb_model_load(%s): %s" % (path, err.value.decode()))
        return cls(h)

    @classmethod
    def from_xml_string(cls, xml):
        err = ctypes.create_string_buffer(1024)
        h = lib().hb_model_load_xml_string(xml.encode(), err, len(err))
        if not h:
            raise HbError("hb_model_load_xml_string: %s" % err.value.decode())
        return cls(h)

    def save(self, path):
        err = ctypes.create_string_buffer(1024)
        if lib().hb_model_save(self._h, os.fsencode(path), err, len(err)) != HB_OK:
            raise HbError("hb_model_save: %s" % err.value.decode())

    def __del__(self):
        try:
            if self._h:
                lib().hb_model_free(self._h)
                self._h = None
        except Exception:
            pass

    @property
    def opt(self):
        o = HbOptions()
        _check(lib().hb_options_get(self._h, ctypes.byref(o)), "hb_options_get")
        return o

    def set_opt(self, **kw):
        o = self.opt
        for k, v in kw.items():
            if k == "gravity":
                for i in range(3):
                    o.gravity[i] = v[i]
            else:
                setattr(o, k, v)
        _check(lib().hb_options_set(self._h, ctypes.byref(o)), "hb_options_set")
        self._read_sizes()  # (the solver selects the instantiation: contact and row capacities follow it)

    def pair_order(self, order=-1):
        """contact order of mj_collision (include/hb.h: hb_model_pair_order): 1 body pairs first (the compiler's default), 0 geom pairs
        ascending; -1: query.  Returns the order in force."""
        rc = lib().hb_model_pair_order(self._h, int(order))
        if rc < 0:
            _check(rc, "hb_model_pair_order")
        return rc

    def name2id(self, kind, name):
        return lib().hb_model_name2id(self._h, kind.encode(), name.encode())

    def array(self, field):
        n = lib().hb_model_get_array(self._h, field.encode(), None, 0)
        if n < 0:
            raise KeyError(field)
        out = np.zeros(n, dtype=np.float64)
        lib().hb_model_get_array(self._h, field.encode(), _ptr(out), n)
        return out


class Batch:
    """n_env environments resident on one GPU (replaces n_env mujoco.MjData blocks)."""

    def __init__(self, model, n_env, device=0):
        self.model = model
        self.n_env = int(n_env)
        err = ctypes.create_string_buffer(1024)
        self._h = lib().hb_batch_create(model._h, self.n_env, int(device), err, len(err))
        if not self._h:
            raise HbError("hb_batch_create: %s" % err.value.decode())

    def close(self):
        if getattr(self, "_h", None):
            lib().hb_batch_free(self._h)
            self._h = None
        for pb in (getattr(self, "_env_pin", None) or {}).values():
            if isinstance(pb, _Pinned):
                pb.free()
        self._env_pin = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def stream(self):
        return lib().hb_batch_stream(self._h)

    def sync(self):
        _check(lib().hb_batch_sync(self._h), "hb_batch_sync")

    def pipeline(self, on=True):
        """Pipelined stepping (hb_batch_pipeline): env segments (three by default, two when the process' streams cannot have a hardware queue each) on their own streams, so the slow tail of
        one step overlaps the next step.  Results are identical; see include/hb.h for the stream contract."""
        _check(lib().hb_batch_pipeline(self._h, int(on)), "hb_batch_pipeline")  # True: the default; 2..8: that many

    @property
    def segments(self):
        """env segments a step call is cut into at the moment (1: unpipelined)"""
        return lib().hb_batch_segments(self._h)

    def join(self):
        _check(lib().hb_batch_join(self._h), "hb_batch_join")

    def reset(self, mask=None, keyframe=-1, perturb=False, env_offset=0):
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        _check(lib().hb_reset(self._h, _ptr(m), int(keyframe), int(bool(perturb)), int(env_offset)), "hb_reset")

    def step(self, ctrl, n_substeps=1):
        c = np.ascontiguousarray(ctrl, dtype=np.float32)
        assert c.shape == (self.n_env, self.model.nu), c.shape
        _check(lib().hb_step(self._h, _ptr(c), int(n_substeps)), "hb_step")

    def step_dev(self, ctrl_ptr, n_substeps=1):
        """ctrl_ptr: device address (int) of a float32 [n_env, nu] array; asynchronous."""
        _check(lib().hb_step_dev(self._h, ctypes.c_void_p(ctrl_ptr), int(n_substeps)), "hb_step_dev")

    def forward(self, ctrl=None):
        c = None if ctrl is None else np.ascontiguousarray(ctrl, dtype=np.float32)
        _check(lib().hb_forward(self._h, _ptr(c)), "hb_forward")

    def rollout(self, ctrl, want_qpos=False):
        c = np.ascontiguousarray(ctrl, dtype=np.float32)
        T = c.shape[0]
        assert c.shape == (T, self.n_env, self.model.nu), c.shape
        q = np.zeros((T, self.n_env, self.model.nq), dtype=np.float32) if want_qpos else None
        _check(lib().hb_rollout(self._h, _ptr(c), T, _ptr(q)), "hb_rollout")
        return q

    def rollout_dev(self, ctrl_ptr, T, qpos_out_ptr=None):
        _check(lib().hb_rollout_dev(self._h, ctypes.c_void_p(ctrl_ptr), int(T), ctypes.c_void_p(qpos_out_ptr or 0)), "hb_rollout_dev")

    def rollout_halton(self, T, t0=0, env_offset=0, qpos_out_ptr=None):
        _check(lib().hb_rollout_halton(self._h, int(T), int(t0), int(env_offset), ctypes.c_void_p(qpos_out_ptr or 0)), "hb_rollout_halton")

    def state_size(self, spec):
        return lib().hb_state_size(self._h, spec)

    def get_state(self, spec=STATE_INTEGRATION, dtype=np.float32):
        w = self.state_size(spec)
        out = np.zeros((self.n_env, w), dtype=dtype)
        fn = lib().hb_get_state if dtype == np.float32 else lib().hb_get_state_f64
        _check(fn(self._h, spec, _ptr(out)), "hb_get_state")
        return out

    def set_state(self, spec, values):
        dt = np.float64 if np.asarray(values).dtype == np.float64 else np.float32
        v = np.ascontiguousarray(values, dtype=dt)
        assert v.shape == (self.n_env, self.state_size(spec)), v.shape
        fn = lib().hb_set_state if dt == np.float32 else lib().hb_set_state_f64
        _check(fn(self._h, spec, _ptr(v)), "hb_set_state")

    @property
    def qpos(self):
        return self.get_state(STATE_QPOS)

    @property
    def qvel(self):
        return self.get_state(STATE_QVEL)

    @property
    def time(self):
        return self.get_state(STATE_TIME)[:, 0]

    def obs(self, want_reward=True):
        o = np.zeros((self.n_env, self.model.nobs), dtype=np.float32)
        r = np.zeros(self.n_env, dtype=np.float32) if want_reward else None
        te = np.zeros(self.n_env, dtype=np.uint8)
        tr = np.zeros(self.n_env, dtype=np.uint8)
        _check(lib().hb_get_obs(self._h, _ptr(o), _ptr(r), _ptr(te), _ptr(tr)), "hb_get_obs")
        return o, r, te.astype(bool), tr.astype(bool)

    def status(self):
        s = np.zeros(self.n_env, dtype=np.int32)
        _check(lib().hb_get_status(self._h, _ptr(s)), "hb_get_status")
        return s

    def counts(self):
        a, b, c = (np.zeros(self.n_env, dtype=np.int32) for _ in range(3))
        _check(lib().hb_get_counts(self._h, _ptr(a), _ptr(b), _ptr(c)), "hb_get_counts")
        return a, b, c

    TUNE = {"duo": 0, "lean": 1, "sized": 2, "staged": 3, "fastpass": 4, "narrow_prim": 5, "schedule": 6, "reorder_period": 7, "policy_lean": 8, "fold": 9}

    def tune(self, **knobs):
        """run-time choices between kernels / schedules that give the same results (include/hb.h: hb_batch_tune, HB_TUNE_*), e.g.
        tune(duo=2, staged=0)"""
        for k, v in knobs.items():
            _check(lib().hb_batch_tune(self._h, self.TUNE[k], int(v)), "hb_batch_tune")

    def duo(self, mode=1):
        """two envs per wave for the lean launches of the 27-dof humanoid's PGS kernel: 0 never, 1 where it pays (default), 2 always"""
        self.tune(duo=mode)

    def device_name(self):
        buf = ctypes.create_string_buffer(256)
        _check(lib().hb_batch_device_name(self._h, buf, 256), "hb_batch_device_name")
        return buf.value.decode()

    def step_launches(self):
        """number of step launches so far (include/hb.h: hb_batch_step_launches)"""
        return int(lib().hb_batch_step_launches(self._h))

    def last_kernel(self):
        """name of the step kernel the batch's last step / rollout / forward launch ran (include/hb.h: hb_last_kernel)"""
        return lib().hb_last_kernel(self._h).decode()

    def collision_counts(self, want_cycles=False):
        """(work items, portal searches[, narrowphase wave time in 1024-cycle units]) of every env's last step (general collision
        models; zeros otherwise)."""
        a, b, c = (np.zeros(self.n_env, dtype=np.int32) for _ in range(3))
        _check(lib().hb_get_collision_counts(self._h, _ptr(a), _ptr(b), _ptr(c)), "hb_get_collision_counts")
        return (a, b, c) if want_cycles else (a, b)

    def diag_enable(self, on=True):
        _check(lib().hb_diag_enable(self._h, int(on)), "hb_diag_enable")

    def qacc(self):
        out = np.zeros((self.n_env, self.model.nv), dtype=np.float32)
        _check(lib().hb_get_qacc(self._h, _ptr(out)), "hb_get_qacc")
        return out

    def efc_force(self):
        out = np.zeros((self.n_env, self.model.nefc_max), dtype=np.float32)
        _check(lib().hb_get_efc_force(self._h, _ptr(out)), "hb_get_efc_force")
        return out

    def contacts(self):
        out = np.zeros((self.n_env, self.model.ncon_max, 16), dtype=np.float32)
        _check(lib().hb_get_contacts(self._h, _ptr(out)), "hb_get_contacts")
        return out

    # ---- wire format: agent.proto State of one env
    def state_to_proto(self, env):
        n = lib().hb_state_to_proto(self._h, int(env), None, 0)
        if n < 0:
            _check(n, "hb_state_to_proto")
        buf = ctypes.create_string_buffer(n)
        _check(min(0, lib().hb_state_to_proto(self._h, int(env), buf, n)), "hb_state_to_proto")
        return buf.raw

    def state_from_proto(self, env, data):
        _check(lib().hb_state_from_proto(self._h, int(env), bytes(data), len(data)), "hb_state_from_proto")

    # ---- planner rollouts (MJPC Trajectory::Rollout analogue)
    @staticmethod
    def sensor_spec(framepos_bodies=(), subtree_body=-1, offsets=None, axes=(), linvel_bodies=(), subtreelinvel_bodies=()):
        """offsets: per frame, the site's position in its body frame (None / missing: the body frame itself);
        axes: (body, which) pairs, which = 0 (framexaxis) or 2 (framezaxis); linvel_bodies: framelinvel (objtype body);
        subtreelinvel_bodies: subtreelinvel of further bodies."""
        sp = HbSensorSpec()
        sp.n_framepos = len(framepos_bodies)
        for k, bd in enumerate(framepos_bodies):
            sp.framepos_body[k] = int(bd)
            if offsets is not None and offsets[k] is not None:
                for i in range(3):
                    sp.framepos_offset[k][i] = float(offsets[k][i])
        sp.subtree_body = int(subtree_body)
        sp.n_frameaxis = len(axes)
        for k, (bd, which) in enumerate(axes):
            sp.frameaxis_body[k] = int(bd); sp.frameaxis_which[k] = int(which)
        sp.n_framelinvel = len(linvel_bodies)
        for k, bd in enumerate(linvel_bodies):
            sp.framelinvel_body[k] = int(bd)
        sp.n_subtreelinvel = len(subtreelinvel_bodies)
        for k, bd in enumerate(subtreelinvel_bodies):
            sp.subtreelinvel_body[k] = int(bd)
        return sp

    def set_state_broadcast(self, spec, state):
        """One state record on every env (the common start of a sampling planner's candidates)."""
        s = np.ascontiguousarray(state)
        if s.dtype == np.float64:
            _check(lib().hb_set_state_broadcast_f64(self._h, spec, _ptr(s)), "hb_set_state_broadcast_f64")
        else:
            s = s.astype(np.float32)
            _check(lib().hb_set_state_broadcast(self._h, spec, _ptr(s)), "hb_set_state_broadcast")

    def rollout_sensors(self, ctrl, spec, want_qpos=False):
        """ctrl [T, n_env, nu] -> sensors [T, n_env, ns] (evaluated before each step's integration), qpos [T, n_env, nq] or None."""
        c = np.ascontiguousarray(ctrl, dtype=np.float32)
        T = c.shape[0]
        assert c.shape == (T, self.n_env, self.model.nu), c.shape
        ns = lib().hb_sensor_size(ctypes.byref(spec))
        if ns <= 0:
            raise HbError("hb_sensor_size: invalid sensor spec")
        out = np.zeros((T, self.n_env, ns), dtype=np.float32)
        q = np.zeros((T, self.n_env, self.model.nq), dtype=np.float32) if want_qpos else None
        _check(lib().hb_rollout_sensors(self._h, _ptr(c), T, ctypes.byref(spec), _ptr(out), _ptr(q)), "hb_rollout_sensors")
        return out, q

    def transition_fd(self, x, u, warmstart=None, eps=1e-3, centered=True, sensor_spec=None):
        """mjd_transitionFD for T points at once: x [T, nq + nv], u [T, nu] -> A [T, 2nv, 2nv], B [T, 2nv, nu] (float64)."""
        xs = np.ascontiguousarray(x, dtype=np.float64)
        us = np.ascontiguousarray(u, dtype=np.float64)
        T, nv, nu = xs.shape[0], self.model.nv, self.model.nu
        assert xs.shape == (T, self.model.nq + nv) and us.shape == (T, nu)
        w = None if warmstart is None else np.ascontiguousarray(warmstart, dtype=np.float64)
        A = np.zeros((T, 2 * nv, 2 * nv)); B = np.zeros((T, 2 * nv, nu))
        if sensor_spec is None:
            _check(lib().hb_transition_fd(self._h, _ptr(xs), _ptr(us), _ptr(w), T, float(eps), int(bool(centered)), _ptr(A), _ptr(B)), "hb_transition_fd")
            return A, B
        ns = lib().hb_sensor_size(ctypes.byref(sensor_spec))
        C = np.zeros((T, ns, 2 * nv)); D = np.zeros((T, ns, nu))
        _check(lib().hb_transition_fd_sensors(self._h, _ptr(xs), _ptr(us), _ptr(w), T, float(eps), int(bool(centered)), ctypes.byref(sensor_spec),
                                              _ptr(A), _ptr(B), _ptr(C), _ptr(D)), "hb_transition_fd_sensors")
        return A, B, C, D

    def ctrl_tape_splines(self, knots, times, interpolation, time0, T):
        """knots [n_env, P, nu], times [P] -> the action tape of T steps on the device (SamplingPolicy::Action per candidate);
        pass ("tape", T) as `ctrl` to rollout_task_stand / rollout_task_walk."""
        k = np.ascontiguousarray(knots, dtype=np.float32)
        tm = np.ascontiguousarray(times, dtype=np.float32)
        assert k.shape == (self.n_env, len(tm), self.model.nu), k.shape
        _check(lib().hb_ctrl_tape_splines(self._h, _ptr(k) if len(tm) else None, _ptr(tm) if len(tm) else None, len(tm), int(interpolation), float(time0), int(T)),
               "hb_ctrl_tape_splines")

    def ctrl_tape_read(self, T):
        """The tape ctrl_tape_splines left on the device: [T, n_env, nu] (TimeSpline::Sample per candidate and step time)."""
        out = np.zeros((int(T), self.n_env, self.model.nu), dtype=np.float32)
        _check(lib().hb_ctrl_tape_read(self._h, int(T), _ptr(out)), "hb_ctrl_tape_read")
        return out

    def task_cost(self, residual, dims, norms, weights, norm_p=None, risk=0.0):
        """BaseResidualFn::CostTerms / CostValue on the device for residual [n, n_residual] -> (terms [n, n_term], cost [n])."""
        r = np.ascontiguousarray(residual, dtype=np.float32)
        n, nres = r.shape
        sp = HbCostSpec()
        sp.n_term = len(dims)
        for k in range(len(dims)):
            sp.dim[k] = int(dims[k]); sp.norm[k] = int(norms[k]); sp.weight[k] = float(weights[k])
            if norm_p is not None:
                sp.norm_p[k][0] = float(norm_p[k][0]); sp.norm_p[k][1] = float(norm_p[k][1])
        sp.risk = float(risk)
        terms = np.zeros((n, len(dims)), dtype=np.float32)
        cost = np.zeros(n, dtype=np.float32)
        _check(lib().hb_task_cost(self._h, _ptr(r), n, nres, ctypes.byref(sp), _ptr(terms), _ptr(cost)), "hb_task_cost")
        return terms, cost

    def _tape_or_ctrl(self, ctrl):
        if isinstance(ctrl, tuple) and ctrl[0] == "tape":
            return int(ctrl[1]) + 1, ctypes.c_void_p(1)  # HB_CTRL_TAPE
        c = np.ascontiguousarray(ctrl, dtype=np.float32)
        H = c.shape[0] + 1
        assert c.shape == (H - 1, self.n_env, self.model.nu), c.shape
        self._keep = c
        return H, (_ptr(c) if H > 1 else None)

    def task_stand_default(self):
        t = HbTaskStand()
        _check(lib().hb_task_stand_default(self.model._h, ctypes.byref(t)), "hb_task_stand_default")
        return t

    def rollout_task_stand(self, ctrl, task, want_costs=False):
        """ctrl [horizon - 1, n_env, nu] -> (total_return [n_env], stage costs [horizon, n_env] or None) of MJPC's Humanoid Stand task."""
        H, cp = self._tape_or_ctrl(ctrl)
        total = np.zeros(self.n_env, dtype=np.float32)
        costs = np.zeros((H, self.n_env), dtype=np.float32) if want_costs else None
        _check(lib().hb_rollout_task_stand(self._h, cp, H, ctypes.byref(task), _ptr(total), _ptr(costs)), "hb_rollout_task_stand")
        return total, costs

    def task_walk_default(self):
        t = HbTaskWalk()
        _check(lib().hb_task_walk_default(self.model._h, ctypes.byref(t)), "hb_task_walk_default")
        return t

    def rollout_task_walk(self, ctrl, task, want_costs=False):
        """ctrl [horizon - 1, n_env, nu] -> (total_return [n_env], stage costs [horizon, n_env] or None) of MJPC's Humanoid Walk task."""
        H, cp = self._tape_or_ctrl(ctrl)
        total = np.zeros(self.n_env, dtype=np.float32)
        costs = np.zeros((H, self.n_env), dtype=np.float32) if want_costs else None
        _check(lib().hb_rollout_task_walk(self._h, cp, H, ctypes.byref(task), _ptr(total), _ptr(costs)), "hb_rollout_task_walk")
        return total, costs

    def rollout_noise(self, xfrc_std, xfrc_rate, seed=0):
        """Ornstein-Uhlenbeck noise on xfrc_applied for the rollouts that follow (Trajectory::NoisyRollout); std 0: off."""
        _check(lib().hb_rollout_noise(self._h, float(xfrc_std), float(xfrc_rate), int(seed)), "hb_rollout_noise")

    def rollout_trajectory(self, ctrl):
        """ctrl [T, n_env, nu] -> (qpos [T, n_env, nq], qvel [T, n_env, nv], failed [n_env]): the states after every step."""
        c = np.ascontiguousarray(ctrl, dtype=np.float32)
        T = c.shape[0]
        assert c.shape == (T, self.n_env, self.model.nu), c.shape
        q = np.zeros((T, self.n_env, self.model.nq), dtype=np.float32)
        v = np.zeros((T, self.n_env, self.model.nv), dtype=np.float32)
        f = np.zeros(self.n_env, dtype=np.int32)
        _check(lib().hb_rollout_trajectory(self._h, _ptr(c), T, _ptr(q), _ptr(v), _ptr(f)), "hb_rollout_trajectory")
        return q, v, f.astype(bool)

    def sensors(self, spec, ctrl=None):
        ns = lib().hb_sensor_size(ctypes.byref(spec))
        if ns <= 0:
            raise HbError("hb_sensor_size: invalid sensor spec")
        out = np.zeros((self.n_env, ns), dtype=np.float32)
        c = None if ctrl is None else np.ascontiguousarray(ctrl, dtype=np.float32)
        _check(lib().hb_sensors(self._h, _ptr(c), ctypes.byref(spec), _ptr(out)), "hb_sensors")
        return out

    # ---- env adapter (CPUEnv.step/reset analogue)
    def env_default_config(self):
        c = HbEnvConfig()
        _check(lib().hb_env_default_config(self.model._h, ctypes.byref(c)), "hb_env_default_config")
        return c

    def env_team_config(self):
        """The reference's own values for its own robot (hb_env_team_config)."""
        c = HbEnvConfig()
        _check(lib().hb_env_team_config(self.model._h, ctypes.byref(c)), "hb_env_team_config")
        return c

    def env_configure(self, cfg):
        _check(lib().hb_env_configure(self._h, ctypes.byref(cfg)), "hb_env_configure")

    def env_default_randomization(self):
        r = HbEnvRandomization()
        _check(lib().hb_env_default_randomization(self.model._h, ctypes.byref(r)), "hb_env_default_randomization")
        return r

    def env_randomize(self, cfg):
        """Install (cfg.factor > 0) or remove (None) the realism layer; call before env_reset."""
        _check(lib().hb_env_randomize(self._h, ctypes.byref(cfg) if cfg is not None else None), "hb_env_randomize")

    def env_default_domain_randomization(self):
        d = HbDomainRandomization()
        _check(lib().hb_env_default_domain_randomization(self.model._h, ctypes.byref(d)), "hb_env_default_domain_randomization")
        return d

    def env_domain_randomize(self, cfg):
        """Install (cfg.factor > 0) or remove (None) per-env model parameters; call before env_reset."""
        _check(lib().hb_env_domain_randomize(self._h, ctypes.byref(cfg) if cfg is not None else None), "hb_env_domain_randomize")

    def env_domain_params(self):
        """[n_env, stride] float32 (layout in include/hb.h), or None when domain randomisation is off."""
        stride = lib().hb_env_get_domain_params(self._h, None)
        if stride < 0:
            _check(stride, "hb_env_get_domain_params")
        if stride == 0:
            return None
        out = np.zeros((self.n_env, stride), dtype=np.float32)
        _check(min(0, lib().hb_env_get_domain_params(self._h, _ptr(out))), "hb_env_get_domain_params")
        return out

    def env_reset(self):
        o = np.zeros((self.n_env, self.model.nobs), dtype=np.float32)
        _check(lib().hb_env_reset(self._h, _ptr(o)), "hb_env_reset")
        return o

    def env_step(self, action, n_substeps=1, copy=True, wait=True):
        """action [n_env, nu] -> (obs, reward, terminated, truncated).  The transfer buffers are page-locked and reused; the four
        outputs are one record in device and host memory, so they come back in one transfer.  copy=True (default): the returned
        arrays are fresh copies, the caller may keep them; copy=False: views of the transfer buffer, valid until the next call.
        wait=False: enqueue only (hb_env_step_async) and return None; sync(), then env_step_result()."""
        pin = getattr(self, "_env_pin", None)
        if pin is None:
            n, nobs = self.n_env, self.model.nobs
            ob, rb = n * nobs * 4, n * 4
            out = _Pinned((ob + rb + 2 * n,), np.uint8)
            pin = self._env_pin = dict(a=_Pinned((n, self.model.nu), np.float32), out=out,
                                       o=out.array[:ob].view(np.float32).reshape(n, nobs), r=out.array[ob:ob + rb].view(np.float32),
                                       te=out.array[ob + rb:ob + rb + n], tr=out.array[ob + rb + n:], off=(0, ob, ob + rb, ob + rb + n))
        a = np.asarray(action, dtype=np.float32)
        assert a.shape == (self.n_env, self.model.nu), a.shape
        if getattr(self, "_env_pending", False):  # a step enqueued with wait=False may still be reading the action buffer / writing the outputs
            self.sync()
            self._env_pending = False
        pin["a"].array[...] = a
        base, off = pin["out"].ptr, pin["off"]
        fn = lib().hb_env_step_async if wait is False else lib().hb_env_step
        _check(fn(self._h, ctypes.c_void_p(pin["a"].ptr), int(n_substeps), ctypes.c_void_p(base + off[0]), ctypes.c_void_p(base + off[1]),
                  ctypes.c_void_p(base + off[2]), ctypes.c_void_p(base + off[3])), "hb_env_step")
        if wait is False:
            self._env_pending = True
            return None
        return self.env_step_result(copy)

    def env_step_result(self, copy=True):
        """the outputs of the last env_step (after env_step(..., wait=False): call sync() first)"""
        pin = getattr(self, "_env_pin", None)
        if pin is None:
            raise HbError("env_step_result before any env_step")
        if not copy:
            return pin["o"], pin["r"], pin["te"].view(bool), pin["tr"].view(bool)
        return pin["o"].copy(), pin["r"].copy(), pin["te"].astype(bool), pin["tr"].astype(bool)

    def env_warnings(self):
        """HB_WARN_* bits every env has raised since the previous call, across in-place resets (include/hb.h: hb_env_warnings)"""
        w = np.zeros(self.n_env, dtype=np.int32)
        _check(lib().hb_env_warnings(self._h, _ptr(w)), "hb_env_warnings")
        return w

    def env_terminal_obs(self, fetch=True):
        """[n_env, nobs] observations of the states episodes ended in (include/hb.h: hb_env_terminal_obs); fetch=False only switches the
        recording on"""
        if not fetch:
            _check(lib().hb_env_terminal_obs(self._h, None), "hb_env_terminal_obs")
            return None
        out = np.zeros((self.n_env, self.model.nobs), dtype=np.float32)
        _check(lib().hb_env_terminal_obs(self._h, _ptr(out)), "hb_env_terminal_obs")
        return out

    def env_step_dev(self, action_ptr, obs_ptr, reward_ptr, terminated_ptr, truncated_ptr, n_substeps=1):
        """hb_env_step_dev: device pointers (e.g. torch tensors' data_ptr()), asynchronous on the batch's stream: a policy on
        the same GPU never sees the host.  Work the caller enqueued on other streams must be finished (or ordered before
        this call through hb_batch_stream); call sync() before reading the outputs from another stream."""
        _check(lib().hb_env_step_dev(self._h, ctypes.c_void_p(action_ptr), int(n_substeps), ctypes.c_void_p(obs_ptr), ctypes.c_void_p(reward_ptr),
                                     ctypes.c_void_p(terminated_ptr), ctypes.c_void_p(truncated_ptr)), "hb_env_step_dev")

    # ---- policy in the loop (BASELINE config 4)
    def set_policy_mlp(self, weights, biases):
        """weights[l]: [in, out] float32 (transpose of torch.nn.Linear.weight); tanh after every layer."""
        ws = [np.ascontiguousarray(w, dtype=np.float32) for w in weights]
        bs = [np.ascontiguousarray(x, dtype=np.float32) for x in biases]
        sizes = (ctypes.c_int * (len(ws) + 1))(*([ws[0].shape[0]] + [w.shape[1] for w in ws]))
        wp = (ctypes.c_void_p * len(ws))(*[w.ctypes.data for w in ws])
        bp = (ctypes.c_void_p * len(bs))(*[x.ctypes.data for x in bs])
        _check(lib().hb_policy_set_mlp(self._h, len(ws), sizes, wp, bp), "hb_policy_set_mlp")

    def policy_eval(self):
        out = np.zeros((self.n_env, self.model.nu), dtype=np.float32)
        _check(lib().hb_policy_eval(self._h, _ptr(out)), "hb_policy_eval")
        return out

    def rollout_policy(self, T, qpos_out_ptr=None):
        _check(lib().hb_rollout_policy(self._h, int(T), ctypes.c_void_p(qpos_out_ptr or 0)), "hb_rollout_policy")

    # ---- device buffers / timing helpers (no HIP headers or torch needed on the caller's side)
    def dev_alloc(self, nbytes):
        p = lib().hb_dev_alloc(self._h, int(nbytes))
        if not p:
            raise HbError("hb_dev_alloc(%d) failed" % nbytes)
        return p

    def dev_free(self, p):
        lib().hb_dev_free(self._h, ctypes.c_void_p(p))

    def to_dev(self, p, arr):
        a = np.ascontiguousarray(arr)
        _check(lib().hb_memcpy_h2d(self._h, ctypes.c_void_p(p), _ptr(a), a.nbytes), "hb_memcpy_h2d")

    def from_dev(self, p, shape, dtype=np.float32):
        out = np.zeros(shape, dtype=dtype)
        _check(lib().hb_memcpy_d2h(self._h, _ptr(out), ctypes.c_void_p(p), out.nbytes), "hb_memcpy_d2h")
        return out

    def halton_ctrl_dev(self, T, t0, env_offset, out_ptr):
        _check(lib().hb_halton_ctrl_dev(self._h, int(T), int(t0), int(env_offset), ctypes.c_void_p(out_ptr)), "hb_halton_ctrl_dev")

    def timer_start(self):
        _check(lib().hb_timer_start(self._h), "hb_timer_start")

    def timer_stop(self):
        ms = ctypes.c_float()
        _check(lib().hb_timer_stop(self._h, ctypes.byref(ms)), "hb_timer_stop")
        return ms.value

    def step_timing(self, enable=True):
        _check(lib().hb_step_timing(self._h, int(enable)), "hb_step_timing")

    def step_timing_read(self):
        us, n = ctypes.c_float(), ctypes.c_int()
        _check(lib().hb_step_timing_read(self._h, ctypes.byref(us), ctypes.byref(n)), "hb_step_timing_read")
        return us.value, n.value
