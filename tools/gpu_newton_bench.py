#!/usr/bin/env python3
"""PGS (benchmark configuration) against the Newton instantiation (the reference's default solver) on configs[1]:
step API (pipelined and single-launch) and rollout throughput, solver iteration statistics, and - with the stamps
build - the per-phase cycle shares of the Newton kernel."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
if "--phases" in sys.argv or "--probe" in sys.argv:
    import humanoid_mujoco_amd.engine as eng
    eng.LIB_PATH = os.environ.get("HB_STAMPS_LIB", os.path.join(ROOT, "build", "libhb_probe.so" if "--probe" in sys.argv else "libhb_stamps.so"))
import humanoid_mujoco_amd as hb
import humanoid_mujoco_amd.engine as eng
HBM = os.path.join(ROOT, "humanoid_mujoco_amd", "assets", "humanoid27.hbm")
N, K, W = 4096, 600, 300


def model(solver):
    m = hb.Model.load(HBM)
    if solver == 2:
        m.set_opt(solver=2, iterations=100)
    return m


if "--probe" in sys.argv:
    L = eng.lib()
    L.hb_get_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    b = hb.Batch(model(2), N, 0)
    b.reset(perturb=True)
    b.rollout_halton(400)
    st = np.zeros((N, 16), dtype=np.uint64)
    assert L.hb_get_stamps(b._h, st.ctypes.data_as(ctypes.c_void_p)) == 0
    b.rollout_halton(1, t0=400)
    assert L.hb_get_stamps(b._h, st.ctypes.data_as(ctypes.c_void_p)) == 0
    nc, ne, ni = b.counts()
    sel = ne > 0
    acc = st[sel, :8].astype(np.float64)
    names = ["constraint update, J'f, cost", "Hessian (MFMA + rows)", "Cholesky", "solve", "M v, J v", "line search", "-", "-"]
    print("Newton sections, cycles per env-step over %d envs with constraints (mean iterations %.2f):" % (sel.sum(), ni[sel].mean()))
    for i in range(6):
        print("%-32s %9.0f  (%.0f per iteration)" % (names[i], acc[:, i].mean(), acc[:, i].sum() / max(1, ni[sel].sum())))
    sys.exit(0)

if "--phases" in sys.argv:
    L = eng.lib()
    L.hb_get_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    b = hb.Batch(model(2), N, 0)
    b.reset(perturb=True)
    b.rollout_halton(400)
    st = np.zeros((N, 16), dtype=np.uint64)
    assert L.hb_get_stamps(b._h, st.ctypes.data_as(ctypes.c_void_p)) == 0
    b.rollout_halton(1, t0=400)
    assert L.hb_get_stamps(b._h, st.ctypes.data_as(ctypes.c_void_p)) == 0
    d = np.diff(st.astype(np.int64), axis=1).astype(np.float64)
    names = ["ctrl+check", "kinematics", "geoms/com/cinert/cdof", "comVel+crb+rne tree passes", "qM", "(no factorM)", "bias/passive/act", "collision", "makeConstraint",
             "row quantities", "dense M, chol(M), qacc_smooth", "warm start choice", "Newton iterations", "write-back", "Euler (dense) + advance"]
    tot = d.sum(1)
    nc, ne, ni = b.counts()
    print("Newton kernel, %d envs: mean cycles per env-step %.0f, median %.0f; mean nefc %.1f, mean iterations %.2f (max %d)" % (N, tot.mean(), np.median(tot), ne.mean(), ni.mean(), ni.max()))
    for i, n in enumerate(names):
        print("%-32s %9.0f cycles  %5.1f %%" % (n, d[:, i].mean(), 100 * d[:, i].mean() / tot.mean()))
    # least-squares split of the iteration phase: fixed + per-iteration
    A = np.stack([np.ones(N), ni.astype(np.float64)], axis=1)
    c = np.linalg.lstsq(A, d[:, 12], rcond=None)[0]
    print("Newton iteration phase ~ %.0f + %.0f cycles per iteration" % (c[0], c[1]))
    sys.exit(0)

for solver, name in ((0, "PGS/50 (benchmark configuration)"), (2, "Newton/100 (reference default)")):
    m = model(solver)
    for npipe in (0, 2):
        b = hb.Batch(m, N, 0)
        ctrl = b.dev_alloc((K + W) * N * m.nu * 4)
        b.halton_ctrl_dev(K + W, 0, 0, ctrl)
        b.reset(perturb=True)
        b.pipeline(npipe)
        stride = N * m.nu * 4
        for t in range(W): b.step_dev(ctrl + t * stride)
        b.sync()
        t0 = time.perf_counter()
        for t in range(W, W + K): b.step_dev(ctrl + t * stride)
        b.sync()
        dt = time.perf_counter() - t0
        nc, ne, ni = b.counts()
        print("%-34s step API, %d segments: %6.1f us/step -> %.3e env-steps/s (mean nefc %.1f, mean solver iterations %.2f, max %d)" % (name, npipe, 1e6 * dt / K, N * K / dt, ne.mean(), ni.mean(), ni.max()), flush=True)
        b.dev_free(ctrl); b.close()
    b = hb.Batch(m, N, 0)
    b.reset(perturb=True)
    b.rollout_halton(W); b.sync()
    t0 = time.perf_counter()
    b.rollout_halton(K, t0=W); b.sync()
    dt = time.perf_counter() - t0
    print("%-34s rollout (one launch):     %6.1f us/step -> %.3e env-steps/s; status flags %d" % (name, 1e6 * dt / K, N * K / dt, int(np.count_nonzero(b.status()))), flush=True)
    b.close()

# contact-rich regime: every env lying on the floor with its servos off (zero controls), ~45 constraint rows per env
import ctypes
hb.lib().hb_memcpy_h2d.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64]
for solver, name in ((0, "PGS/50 (benchmark configuration)"), (2, "Newton/100 (reference default)")):
    m = model(solver)
    b = hb.Batch(m, N, 0)
    zeros = b.dev_alloc(N * m.nu * 4)
    z = np.zeros((N, m.nu), np.float32)
    assert hb.lib().hb_memcpy_h2d(b._h, ctypes.c_void_p(zeros), z.ctypes.data_as(ctypes.c_void_p), z.nbytes) == 0
    b.reset(perturb=True)
    b.pipeline(2)
    for t in range(1500): b.step_dev(zeros)
    b.sync()
    t0 = time.perf_counter()
    for t in range(300): b.step_dev(zeros)
    b.sync()
    dt = time.perf_counter() - t0
    nc, ne, ni = b.counts()
    print("%-34s collapsed on the floor, zero controls, step API 2 segments: %6.1f us/step -> %.3e env-steps/s (mean nefc %.1f, mean solver iterations %.2f)" % (name, 1e6 * dt / 300, N * 300 / dt, ne.mean(), ni.mean()), flush=True)
    b.dev_free(zeros); b.close()
