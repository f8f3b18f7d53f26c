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
