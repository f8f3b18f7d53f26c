"""ctypes binding of oracle/liboracle.so for the tests, smoke() and bench.py's cpu_baseline leg.

TEST INFRASTRUCTURE: the product package (humanoid_mujoco_amd/) never imports this module.
"""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.environ.get("HB_ORACLE_SO") or os.path.join(ORACLE_DIR, "liboracle.so")  # HB_ORACLE_SO: e.g. a sanitizer build (tools/asan_host.sh)
GOLDEN = os.path.join(ROOT, "tests", "golden")
HUMANOID_HBM = os.path.join(ROOT, "humanoid_mujoco_amd", "assets", "humanoid27.hbm")

_lib = None


def build_oracle(force=False):
    src = os.path.join(ORACLE_DIR, "mjstep_oracle.c")
    if os.environ.get("HB_ORACLE_SO"):
        return ORACLE_SO
    if force or not os.path.exists(ORACLE_SO) or os.path.getmtime(ORACLE_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])
    return ORACLE_SO


def lib():
    global _lib
    if _lib is None:
        build_oracle()
        L = ctypes.CDLL(ORACLE_SO)
        vp, cp, ci, cd = ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_double
        pd, pi = ctypes.POINTER(cd), ctypes.POINTER(ci)
        L.om_load.restype = vp; L.om_load.argtypes = [cp, cp, ci]
        L.om_make_data.restype = vp; L.om_make_data.argtypes = [vp]
        L.om_free_data.argtypes = [vp]
        L.om_reset.argtypes = [vp, vp, ci]
        L.om_forward.argtypes = [vp, vp]
        L.om_step.argtypes = [vp, vp]
        L.om_data_ptr.restype = pd; L.om_data_ptr.argtypes = [vp, vp, cp, pi]
        L.om_model_ptr.restype = pd; L.om_model_ptr.argtypes = [vp, cp, pi]
        L.om_model_int.restype = ci; L.om_model_int.argtypes = [vp, cp]
        L.om_model_set_int.argtypes = [vp, cp, ci]
        L.om_model_set_dbl.argtypes = [vp, cp, cd]
        L.om_model_dbl.restype = cd; L.om_model_dbl.argtypes = [vp, cp]
        L.om_data_int.restype = ci; L.om_data_int.argtypes = [vp, cp]
        L.om_data_time.restype = cd; L.om_data_time.argtypes = [vp]
        L.om_data_set_time.argtypes = [vp, cd]
        L.om_contact_get.argtypes = [vp, ci, pd]
        L.om_efc_types.argtypes = [vp, pi, pi]
        L.om_stats.argtypes = [vp, pd]
        L.om_halton.restype = cd; L.om_halton.argtypes = [ci, ci]
        L.om_init_env.argtypes = [vp, vp, ci]
        L.om_ctrl_env.argtypes = [vp, pd, ci, ci]
        L.om_pgs_reverts.restype = ctypes.c_long; L.om_pgs_reverts.argtypes = []
        L.om_rollout_threads.restype = ctypes.c_longlong
        L.om_rollout_threads.argtypes = [vp, ci, ci, ci, ci, vp, vp]
        L.om_mpr_test.restype = ci
        L.om_mpr_test.argtypes = [ci, pd, pd, pd, pd, ci, ci, pd, pd, pd, pd, ci, cd, pd, pd, pd]
        L.om_set_contact_order.argtypes = [ci, ctypes.c_uint]
        L.om_rollout_window.restype = ctypes.c_longlong
        L.om_rollout_window.argtypes = [vp, ci, ci, ci, ci, ci, vp, vp, pd]
        _lib = L
    return _lib


def parse_hbm(path):
    """Independent (Python) reader of the .hbm text model: dict name -> int/float/np.ndarray/list."""
    out = {}
    with open(path) as f:
        first = f.readline()
        assert first.startswith("HBM1"), "not an HBM1 file"
        for line in f:
            line = line.strip()
            if line == "END":
                break
            if not line or line[0] == "#":
                continue
            tok = line.split()
            kind, name = tok[0], tok[1]
            if kind == "i":
                out[name] = int(tok[2])
            elif kind == "d":
                out[name] = float(tok[2])
            elif kind == "I":
                out[name] = np.array([int(x) for x in tok[3:]], dtype=np.int64)
            elif kind == "D":
                out[name] = np.array([float(x) for x in tok[3:]], dtype=np.float64)
            elif kind == "S":
                out[name] = ["" if x == "-" else x for x in tok[3:]]
    return out


class Oracle:
    """One model + one data block of the fp64 oracle."""

    def __init__(self, hbm_path=HUMANOID_HBM):
        self.L = lib()
        err = ctypes.create_string_buffer(512)
        self.m = self.L.om_load(hbm_path.encode(), err, 512)
        if not self.m:
            raise RuntimeError("oracle: " + err.value.decode())
        self.d = self.L.om_make_data(self.m)
        self.info = parse_hbm(hbm_path)
        for k in ("nq", "nv", "nu", "nbody", "njnt", "ngeom", "ntendon", "nM"):
            setattr(self, k, self.L.om_model_int(self.m, k.encode()))

    def __del__(self):
        try:
            if self.d:
                self.L.om_free_data(self.d)
        except Exception:
            pass

    # ---- arrays are live views into the oracle's memory
    def arr(self, name):
        n = ctypes.c_int()
        p = self.L.om_data_ptr(self.m, self.d, name.encode(), ctypes.byref(n))
        if not p:
            raise KeyError(name)
        return np.ctypeslib.as_array(p, (max(n.value, 0),))

    def marr(self, name):
        n = ctypes.c_int()
        p = self.L.om_model_ptr(self.m, name.encode(), ctypes.byref(n))
        if not p:
            raise KeyError(name)
        return np.ctypeslib.as_array(p, (n.value,))

    def __getattr__(self, name):
        if name.startswith("_") or name in ("L", "m", "d", "info"):
            raise AttributeError(name)
        try:
            return self.arr(name)
        except KeyError:
            raise AttributeError(name)

    def dint(self, name):
        return self.L.om_data_int(self.d, name.encode())

    @property
    def ncon(self):
        return self.dint("ncon")

    @property
    def nefc(self):
        return self.dint("nefc")

    @property
    def time(self):
        return self.L.om_data_time(self.d)

    def set_opt(self, **kw):
        for k, v in kw.items():
            if k in ("iterations", "disableflags", "solver", "ls_iterations", "mpr_iterations"):
                self.L.om_model_set_int(self.m, k.encode(), int(v))
            else:
                self.L.om_model_set_dbl(self.m, k.encode(), float(v))

    def opt(self, name):
        if name in ("iterations", "disableflags", "solver"):
            return self.L.om_model_int(self.m, name.encode())
        return self.L.om_model_dbl(self.m, name.encode())

    def reset(self, key=-1):
        self.L.om_reset(self.m, self.d, key)

    def forward(self):
        self.L.om_forward(self.m, self.d)

    def step(self, n=1):
        for _ in range(n):
            self.L.om_step(self.m, self.d)

    def init_env(self, e):
        self.L.om_init_env(self.m, self.d, e)

    def ctrl_env(self, t, e):
        c = np.zeros(self.nu)
        self.L.om_ctrl_env(self.m, c.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), t, e)
        return c

    def contacts(self):
        out = []
        buf = (ctypes.c_double * 23)()
        for k in range(self.ncon):
            self.L.om_contact_get(self.d, k, buf)
            a = np.array(buf[:])
            out.append(dict(dist=a[0], pos=a[1:4].copy(), frame=a[4:13].reshape(3, 3).copy(), dim=int(a[13]),
                            geom1=int(a[14]), geom2=int(a[15]), efc_address=int(a[16]), friction=a[17], friction5=a[18:23].copy()))
        return out

    def efc_types(self):
        n = self.nefc
        t = (ctypes.c_int * max(n, 1))()
        i = (ctypes.c_int * max(n, 1))()
        self.L.om_efc_types(self.d, t, i)
        return np.array(t[:n]), np.array(i[:n])

    def dense_M(self):
        """Dense mass matrix from the ancestor-chain sparse qM."""
        nv = self.nv
        M = np.zeros((nv, nv))
        qM = self.arr("qM")
        par, madr = self.info["dof_parentid"], self.info["dof_Madr"]
        for i in range(nv):
            a = madr[i]
            j = i
            while j >= 0:
                M[i, j] = M[j, i] = qM[a]
                a += 1
                j = par[j]
        return M

    def pgs_reverts(self):
        """PGS row updates undone by the reference's cost-change test since the library was loaded."""
        return int(self.L.om_pgs_reverts())

    def rollout_threads(self, n_env, nstep, nthread, env_offset=0, want_qpos=False):
        q = np.zeros((n_env, self.nq)) if want_qpos else None
        st = (ctypes.c_double * 5)()
        n = self.L.om_rollout_threads(self.m, n_env, nstep, nthread, env_offset,
                                      q.ctypes.data if q is not None else None, ctypes.addressof(st))
        return n, q, dict(mean_ncon=st[0], mean_nefc=st[1], mean_iter=st[2], max_ncon=st[3], max_nefc=st[4])

    def rollout_window(self, n_env, t_pre, nstep, nthread, env_offset=0, want_qpos=False):
        """t_pre untimed steps then nstep timed ones per env (the benchmark's pre-rolled window); returns
        (timed env-steps, qpos, stats over the timed steps, env-steps/s of the timed window with all threads busy)."""
        q = np.zeros((n_env, self.nq)) if want_qpos else None
        st = (ctypes.c_double * 5)()
        rate = ctypes.c_double(0.0)
        n = self.L.om_rollout_window(self.m, n_env, t_pre, nstep, nthread, env_offset,
                                     q.ctypes.data if q is not None else None, ctypes.addressof(st), ctypes.byref(rate))
        return n, q, dict(mean_ncon=st[0], mean_nefc=st[1], mean_iter=st[2], max_ncon=st[3], max_nefc=st[4]), rate.value


def mpr(obj1, obj2, margin=0.0):
    """libccd-MPR restatement on two free objects: obj = dict(type=2 sphere | 3 capsule | 7 mesh, pos, mat (3x3), size, vert (n x 3)).
    Returns (result, depth, dir, pos); result 0 = intersecting."""
    L = lib()
    P = ctypes.POINTER(ctypes.c_double)

    def pack(o):
        pos = np.ascontiguousarray(o.get("pos", np.zeros(3)), dtype=np.float64)
        mat = np.ascontiguousarray(o.get("mat", np.eye(3)), dtype=np.float64).reshape(9)
        size = np.zeros(3); sz = np.atleast_1d(np.asarray(o.get("size", []), dtype=np.float64)); size[:len(sz)] = sz
        vert = np.ascontiguousarray(o.get("vert", np.zeros((1, 3))), dtype=np.float64)
        return int(o["type"]), pos, mat, size, vert

    t1, p1, m1, s1, v1 = pack(obj1)
    t2, p2, m2, s2, v2 = pack(obj2)
    depth = ctypes.c_double(0.0)
    d = np.zeros(3); x = np.zeros(3)
    r = L.om_mpr_test(t1, p1.ctypes.data_as(P), m1.ctypes.data_as(P), s1.ctypes.data_as(P), v1.ctypes.data_as(P), len(v1),
                      t2, p2.ctypes.data_as(P), m2.ctypes.data_as(P), s2.ctypes.data_as(P), v2.ctypes.data_as(P), len(v2),
                      float(margin), ctypes.byref(depth), d.ctypes.data_as(P), x.ctypes.data_as(P))
    return r, depth.value, d, x


def halton(index, base):
    """Python restatement of the radical inverse (mju_Halton, mujoco.h:1231), for cross-checks."""
    f, r = 1.0 / base, 0.0
    while index > 0:
        r += f * (index % base)
        index //= base
        f /= base
    return r


# ---- rounding fences: a discrete decision (which MPR portal a search ends on, whether a geom pair is inside its margin) may fall the
# other way between the fp32 device and the fp64 oracle when the state sits within a rounding of the decision's boundary.  The parity
# tests do not skip such a case on suspicion: they require PROOF that the oracle itself gives the device's answer from a state no
# further away than fp32 rounding.  Anything that cannot be proved this way is a failure.
FENCE_MAGNITUDES = (6e-8, 2e-7, 6e-7)  # half an fp32 ulp at 1, and the few ulps the fp32 kinematics of a 7-link chain accumulates
FENCE_TRIES = 48


def load_state(o, state, ctrl):
    """mjSTATE_INTEGRATION record [time, qpos, qvel, qacc_warmstart] + ctrl into the oracle's data (after a reset)."""
    nq, nv = o.nq, o.nv
    o.reset()
    o.L.om_data_set_time(o.d, float(state[0]))
    o.qpos[:] = state[1:1 + nq]
    o.qvel[:] = state[1 + nq:1 + nq + nv]
    o.qacc_warmstart[:] = state[1 + nq + nv:1 + nq + 2 * nv]
    o.ctrl[:] = ctrl


def prove_rounding_fence(o, state, ctrl, accept, seed=0):
    """Search the fp32-rounding neighbourhood of `state` (qpos perturbed by FENCE_MAGNITUDES x N(0,1) x max(1, |qpos|)) for a state from which
    the oracle reproduces the device: accept(o) is called with the perturbed state loaded (nothing evaluated yet) and returns True on
    a match.  Returns the perturbation magnitude that matched (the oracle is left in accept's final state), or None: no proof."""
    rng = np.random.default_rng(seed)
    state = np.asarray(state, dtype=np.float64)
    nq = o.nq
    for mag in FENCE_MAGNITUDES:
        for _ in range(FENCE_TRIES):
            st = state.copy()
            st[1:1 + nq] += mag * rng.standard_normal(nq) * np.maximum(1.0, np.abs(st[1:1 + nq]))
            load_state(o, st, ctrl)
            if accept(o):
                return mag
    return None
