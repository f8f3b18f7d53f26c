"""Oracle Newton solver (mjSOL_NEWTON, the reference's default: its humanoid XML sets no solver) checked without MuJoCo:
  * it minimises the same convex problem as the dual PGS solver: a PGS run to convergence lands on the same qacc;
  * optimality: the gradient M qacc - qfrc_smooth - J' f vanishes with f = -D min(0, J qacc - aref);
  * it needs a handful of iterations on the benchmark workload.
"""
import numpy as np

from oracle_lib import Oracle

SOL_PGS, SOL_NEWTON = 0, 2


def run(solver, iterations, T, env, tolerance=1e-8):
    o = Oracle()
    o.set_opt(solver=solver, iterations=iterations, tolerance=tolerance)
    o.init_env(env)
    qpos, qacc, niter = [], [], []
    for t in range(T):
        o.ctrl[:] = o.ctrl_env(t, env)
        o.step()
        qpos.append(o.qpos.copy()); qacc.append(o.qacc.copy()); niter.append(o.dint("solver_niter"))
    return o, np.array(qpos), np.array(qacc), np.array(niter)


def test_newton_agrees_with_converged_pgs():
    # teacher-forced: both solvers start every step from the Newton trajectory's state
    env, T = 3, 120
    on = Oracle(); on.set_opt(solver=SOL_NEWTON, iterations=100)
    op = Oracle(); op.set_opt(solver=SOL_PGS, iterations=20000, tolerance=1e-14)
    on.init_env(env)
    worst = 0.0
    active = 0
    for t in range(T):
        c = on.ctrl_env(t, env)
        op.qpos[:] = on.qpos; op.qvel[:] = on.qvel; op.qacc_warmstart[:] = on.qacc_warmstart
        on.ctrl[:] = c; op.ctrl[:] = c
        on.forward(); op.forward()
        assert on.nefc == op.nefc
        if on.nefc:
            active += 1
            scale = max(1.0, np.abs(op.qacc).max())
            worst = max(worst, np.abs(on.qacc - op.qacc).max() / scale)
            fs = max(1.0, np.abs(op.efc_force).max())
            assert np.abs(on.efc_force[:on.nefc] - op.efc_force[:on.nefc]).max() / fs < 1e-5
        on.step()
    assert active > 40
    assert worst < 1e-6, worst


def test_newton_solution_is_stationary():
    env = 5
    o = Oracle(); o.set_opt(solver=SOL_NEWTON, iterations=100)
    o.init_env(env)
    nv = o.nv
    checked = 0
    for t in range(150):
        o.ctrl[:] = o.ctrl_env(t, env)
        o.forward()
        n = o.nefc
        if n:
            J = o.efc_J[: n * nv].reshape(n, nv)
            jar = J @ o.qacc - o.efc_aref[:n]
            f = np.where(jar < 0, -o.efc_D[:n] * jar, 0.0)
            assert np.allclose(f, o.efc_force[:n], rtol=1e-9, atol=1e-9)
            grad = o.dense_M() @ o.qacc - o.qfrc_smooth - J.T @ f
            assert np.abs(grad).max() < 1e-5 * max(1.0, np.abs(o.qfrc_smooth).max()), (t, np.abs(grad).max())
            assert np.allclose(o.qfrc_constraint, J.T @ f, atol=1e-9)
            checked += 1
        o.step()
    assert checked > 50


def test_newton_iteration_counts_on_benchmark_workload():
    _, qpos, _, niter = run(SOL_NEWTON, 100, 300, 1)
    assert np.isfinite(qpos).all()
    assert niter.max() <= 20 and niter.mean() < 5.0, (niter.max(), niter.mean())


def test_pgs_50_is_not_the_converged_solution_but_newton_is_iteration_independent():
    # the deviation the benchmark configuration carries (PGS, 50 sweeps) is visible; Newton does not depend on its cap
    _, qa, _, _ = run(SOL_NEWTON, 100, 150, 3)
    _, qb, _, _ = run(SOL_NEWTON, 30, 150, 3)
    assert np.abs(qa - qb).max() < 1e-9
    _, qp, _, _ = run(SOL_PGS, 50, 150, 3)
    assert np.abs(qa[:40] - qp[:40]).max() < 5e-3  # close early on ...
    assert np.abs(qa - qp).max() > 1e-4            # ... but not the same trajectory


def test_newton_agrees_with_converged_pgs_on_other_models(tmp_path):
    """Same cross-check on the small test models: condim-1 pairs, multi-tree contact, slide joints, joint limits."""
    import os
    import humanoid_mujoco_amd as hb
    models = os.path.join(os.path.dirname(os.path.abspath(__file__)), "models")
    for name, steps in (("ball_plane", 300), ("capsules", 300), ("chain", 200)):
        p = str(tmp_path / (name + ".hbm"))
        hb.Model.load(os.path.join(models, name + ".xml")).save(p)
        on, op = Oracle(p), Oracle(p)
        on.set_opt(solver=SOL_NEWTON, iterations=100)
        op.set_opt(solver=SOL_PGS, iterations=50000, tolerance=1e-15)
        on.reset(0 if name == "chain" else -1)
        rng = np.random.default_rng(5)
        active, worst = 0, 0.0
        for t in range(steps):
            c = rng.uniform(-1, 1, size=max(on.nu, 1))[:on.nu]
            op.qpos[:] = on.qpos; op.qvel[:] = on.qvel; op.qacc_warmstart[:] = on.qacc_warmstart
            on.ctrl[:] = c; op.ctrl[:] = c
            on.forward(); op.forward()
            assert on.nefc == op.nefc
            if on.nefc:
                active += 1
                worst = max(worst, np.abs(on.qacc - op.qacc).max() / max(1.0, np.abs(op.qacc).max()))
            on.step()
        assert active > 20, (name, active)
        assert worst < 1e-5, (name, worst)
