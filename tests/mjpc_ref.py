"""numpy restatement of the MJPC pieces the device evaluates for a planner iteration (test infrastructure):
mjpc::Norm (mujoco_mpc/mjpc/norm.cc:50-208, value only), Stand::ResidualFn::Residual
(mujoco_mpc/mjpc/tasks/humanoid/stand/stand.cc:41-104), BaseResidualFn::CostValue (mjpc/task.cc:71-110) and
Trajectory::UpdateReturn (mjpc/trajectory.cc:312-326), fed with fp64 oracle quantities."""
import numpy as np


def norm(kind, x, p=0.0, q=0.0):
    x = np.atleast_1d(np.asarray(x, dtype=np.float64))
    if kind == 0:
        return 0.5 * float(x @ x)
    if kind == 1:
        return float(((x @ x) ** (q / 2) + p ** q) ** (1 / q) - p)
    if kind == 2:
        return float(np.sqrt(x @ x + p * p) - p)
    if kind == 3:
        return float((p * p * (np.cosh(x / p) - 1.0)).sum())
    if kind == 5:
        return float((np.abs(x) ** p).sum())
    if kind == 6:
        return float((np.sqrt(x * x + p * p) - p).sum())
    if kind == 7:
        return float(((np.abs(x) ** q + p ** q) ** (1 / q) - p).sum())
    if kind == 8:
        return float((p * np.log1p(np.exp(x / p))).sum()) if p > 0 else float(np.maximum(x, 0).sum())
    return float(x[0])


def stand_residual(o, task, nb):
    """Residual of the oracle's current (forwarded) state and ctrl."""
    xpos = o.xpos.reshape(nb, 3)
    xmat = o.xmat.reshape(nb, 3, 3)
    feet = np.array([xpos[task.foot_body[k]] + xmat[task.foot_body[k]] @ np.array(task.foot_offset[k][:]) for k in range(task.n_feet)])
    head = o.xipos.reshape(nb, 3)[task.head_body]  # framepos objtype="body": the inertial frame
    mass = o.marr("body_mass")
    com = o.subtree_com.reshape(nb, 3)[task.subtree_body]
    cvel = o.cvel.reshape(nb, 6)
    xipos = o.xipos.reshape(nb, 3)
    v = cvel[:, 3:] + np.cross(cvel[:, :3], xipos - com)
    linvel = (mass[1:, None] * v[1:]).sum(0) / mass[1:].sum()  # the humanoid is one tree: subtree of the torso = all bodies
    height = head[2] - feet[:, 2].mean() - task.height_goal
    capture = com + 0.2 * linvel
    balance = np.linalg.norm(feet[:, :2].mean(0) - capture[:2])
    return [np.array([height]), np.array([balance]), linvel[:2].copy(), o.qvel[6:].copy(), o.ctrl.copy()]


def stand_cost(res, task):
    c = sum(task.weight[k] * norm(task.norm[k], res[k], task.norm_p[k][0], task.norm_p[k][1]) for k in range(5))
    if abs(task.risk) >= 1e-6:
        c = (np.exp(task.risk * c) - 1.0) / task.risk
    return c


def stand_rollout(o, ctrl, task, nb):
    """Trajectory::Rollout for one candidate: horizon = len(ctrl) + 1 states, last action repeated for the final forward."""
    costs = []
    for t in range(len(ctrl)):
        o.ctrl[:] = ctrl[t]
        o.forward()  # what mj_step's forward pass (and the user-sensor callback in it) sees
        costs.append(stand_cost(stand_residual(o, task, nb), task))
        o.step()
    if len(ctrl):
        o.ctrl[:] = ctrl[-1]
    else:
        o.ctrl[:] = 0
    o.forward()
    costs.append(stand_cost(stand_residual(o, task, nb), task))
    return float(np.mean(costs)), np.array(costs)


# ---- sensors as MuJoCo evaluates them, from oracle quantities -------------------------------------------------------
def body_linvel(o, body, nb):
    """framelinvel, objtype body (mj_objectVelocity at the inertial frame origin, world axes): cvel is the spatial
    velocity at the tree's com, so the point velocity is lin + ang x (xipos - com)."""
    cvel = o.cvel.reshape(nb, 6)[body]
    com = o.subtree_com.reshape(nb, 3)[1]  # one tree, rooted at body 1
    return cvel[3:] + np.cross(cvel[:3], o.xipos.reshape(nb, 3)[body] - com)


def subtree_linvel(o, root, nb, parent):
    """subtreelinvel of any body: linear momentum of its subtree over the subtree's mass (mj_subtreeVel)."""
    mass = o.marr("body_mass")
    members = [b for b in range(1, nb) if root in _ancestors(b, parent)]
    mom = sum(mass[b] * body_linvel(o, b, nb) for b in members)
    return mom / sum(mass[b] for b in members)


def _ancestors(b, parent):
    out = []
    while b > 0:
        out.append(b)
        b = parent[b]
    return out


def walk_residual(o, task, nb, parent):
    """Walk::ResidualFn::Residual (mujoco_mpc/mjpc/tasks/humanoid/walk/walk.cc:44-163)."""
    xipos = o.xipos.reshape(nb, 3)
    xmat = o.xmat.reshape(nb, 3, 3)
    up = lambda b: xmat[b][:, 2]
    fwd = lambda b: xmat[b][:, 0]
    res = []
    torso_height = xipos[task.torso_body][2]
    res.append(torso_height - task.height_goal)
    fr, fl = xipos[task.foot_right_body], xipos[task.foot_left_body]
    res.append(0.5 * (fl[2] + fr[2]) - xipos[task.pelvis_body][2] - 0.2)
    com = o.subtree_com.reshape(nb, 3)[task.torso_body]
    comvel = subtree_linvel(o, task.torso_body, nb, parent)
    cp = com + 0.3 * comvel
    cp[2] = 1e-3
    axis = fr - fl
    axis[2] = 1e-3
    n = np.linalg.norm(axis)
    axis = axis / n
    length = 0.5 * n - 0.05
    center = 0.5 * (fr + fl)
    t = float(np.clip((cp - center) @ axis, -length, length))
    pcp = axis * t + center
    standing = torso_height / np.sqrt(torso_height ** 2 + 0.45 ** 2) - 0.4
    res += list(standing * (cp[:2] - pcp[:2]))
    z = np.array([0.0, 0.0, 1.0])
    res.append(up(task.torso_body)[2] - 1.0)
    res.append(0.3 * (up(task.pelvis_body)[2] - 1.0))
    res += list(0.1 * standing * (up(task.foot_right_body) - z))
    res += list(0.1 * standing * (up(task.foot_left_body) - z))
    res += list(o.qpos[7:])
    f = fwd(task.torso_body)[:2] + fwd(task.pelvis_body)[:2] + fwd(task.foot_right_body)[:2] + fwd(task.foot_left_body)[:2]
    f = f / np.linalg.norm(f)
    com_vel = 0.5 * (subtree_linvel(o, task.waist_lower_body, nb, parent)[:2] + body_linvel(o, task.torso_body, nb)[:2])
    res.append(standing * (com_vel @ f - task.speed_goal))
    move = com_vel - 0.5 * body_linvel(o, task.foot_right_body, nb)[:2] - 0.5 * body_linvel(o, task.foot_left_body, nb)[:2]
    res += list(standing * move)
    res += list(o.ctrl)
    return np.array(res)


def terms_cost(res, task):
    """BaseResidualFn::CostTerms / CostValue (mjpc/task.cc:71-110): the terms' dims slice the residual in order."""
    c, sh = 0.0, 0
    for k in range(task.n_term):
        c += task.weight[k] * norm(task.norm[k], res[sh:sh + task.dim[k]], task.norm_p[k][0], task.norm_p[k][1])
        sh += task.dim[k]
    assert sh == len(res)
    if abs(task.risk) >= 1e-6:
        c = (np.exp(task.risk * c) - 1.0) / task.risk
    return c


def walk_rollout(o, ctrl, task, nb, parent):
    costs = []
    for t in range(len(ctrl)):
        o.ctrl[:] = ctrl[t]
        o.forward()
        costs.append(terms_cost(walk_residual(o, task, nb, parent), task))
        o.step()
    o.ctrl[:] = ctrl[-1] if len(ctrl) else 0
    o.forward()
    costs.append(terms_cost(walk_residual(o, task, nb, parent), task))
    return float(np.mean(costs)), np.array(costs)


def spline_sample(times, values, interp, t):
    """TimeSpline::Sample (mujoco_mpc/mjpc/spline/spline.cc:103-156) with Slope / CubicCoefficients (:240-277);
    values [P, dim]; interp 0 zero-order, 1 linear, 2 cubic."""
    times = np.asarray(times, dtype=np.float64)
    values = np.asarray(values, dtype=np.float64)
    P = len(times)
    up = int(np.searchsorted(times, t, side="right"))  # std::upper_bound
    if up == P:
        return values[P - 1].copy()
    if up == 0:
        return values[0].copy()
    lo = up - 1
    x = (t - times[lo]) / (times[up] - times[lo])
    if interp == 0:
        return values[lo].copy()
    if interp == 1:
        return values[lo] * (1 - x) + values[up] * x

    def slope(k):
        if k == 0:
            return (values[1] - values[0]) / (times[1] - times[0])
        back = (values[k] - values[k - 1]) / (times[k] - times[k - 1])
        if k == P - 1:
            return back
        return 0.5 * (values[k + 1] - values[k]) / (times[k + 1] - times[k]) + 0.5 * back

    h = times[up] - times[lo]
    c = (2 * x ** 3 - 3 * x ** 2 + 1, (x ** 3 - 2 * x ** 2 + x) * h, -2 * x ** 3 + 3 * x ** 2, (x ** 3 - x ** 2) * h)
    return c[0] * values[lo] + c[1] * slope(lo) + c[2] * values[up] + c[3] * slope(up)
