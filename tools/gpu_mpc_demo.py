#!/usr/bin/env python3
"""Closed-loop predictive sampling on the device back end (a demonstration of SURVEY §8 f3, not part of the product): MJPC's
sampling planner loop (planners/sampling/planner.cc:151-187, 342-380) written against the C-ABI - spline policies with P nodes,
N noisy candidates per iteration evaluated by hb_ctrl_tape_splines + hb_rollout_task_stand, the best one becomes the nominal,
its action drives a one-env "plant" for agent_timestep (3 physics steps).  Task: Humanoid Stand from the squat keyframe.
Reports the task cost along the way and the time per planner iteration."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import humanoid_mujoco_amd as hb
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
TASK = sys.argv[2] if len(sys.argv) > 2 else "stand"   # "stand" (from the squat keyframe) or "walk" (from the upright pose)
ITERS = 120 if TASK == "stand" else 300
HBM = os.path.join(ROOT, "humanoid_mujoco_amd", "assets", "humanoid27.hbm")
m = hb.Model.load(HBM)
m.set_opt(solver=2, iterations=100)                      # Newton, as MJPC runs the model
h = m.opt.timestep
P, horizon_s, agent_dt, explore = 3, 0.35, 0.015, 0.05   # sampling_spline_points, agent_horizon, agent_timestep, sampling_exploration
H = int(round(horizon_s / h)) + 1
sub = int(round(agent_dt / h))
plant = hb.Batch(m, 1, 0)
plant.reset(keyframe=m.name2id("key", "squat") if TASK == "stand" else -1)
cand = hb.Batch(m, N, 0)
task = cand.task_stand_default() if TASK == "stand" else cand.task_walk_default()
evaluate = cand.rollout_task_stand if TASK == "stand" else cand.rollout_task_walk
lo, hi = m.array("actuator_ctrlrange").reshape(-1, 2).T
rng = np.random.default_rng(0)
times = np.linspace(0.0, horizon_s, P).astype(np.float32)
nominal = np.zeros((P, m.nu), np.float32)
t_now, iters, t_plan = 0.0, 0, 0.0
print("task: Humanoid %s" % TASK.capitalize())
print("predictive sampling: %d candidates x %d states, %d spline nodes, plant advances %d steps per iteration" % (N, H, P, sub))
x_start = float(plant.qpos[0][0])
for it in range(ITERS):
    st = plant.get_state(hb.STATE_INTEGRATION, dtype=np.float64)[0]
    st[0] = 0.0
    t0 = time.perf_counter()
    # candidates: the nominal (candidate 0) and noisy copies of its nodes (AddNoiseToPolicy: std = exploration * range)
    knots = np.repeat(nominal[None], N, axis=0)
    knots[1:] += (rng.standard_normal((N - 1, P, m.nu)) * explore * (hi - lo)).astype(np.float32)
    np.clip(knots, lo, hi, out=knots)
    cand.set_state_broadcast(hb.STATE_INTEGRATION, st)
    cand.ctrl_tape_splines(knots, times, 2, 0.0, H - 1)
    ret, _ = evaluate(("tape", H - 1), task)
    best = int(np.argmin(ret))
    t_plan += time.perf_counter() - t0
    iters += 1
    nominal = knots[best].copy()
    # act: the winner's spline at the plant's time, for agent_timestep
    from mjpc_ref import spline_sample
    for k in range(sub):
        u = np.clip(spline_sample(times, nominal, 2, k * h), lo, hi).astype(np.float32)
        plant.step(u[None])
    # shift the plan by the time that passed (the next iteration's clock starts at zero again)
    shifted = np.array([spline_sample(times, nominal, 2, min(float(tk) + agent_dt, horizon_s)) for tk in times], np.float32)
    nominal = shifted
    if it % (ITERS // 12) == 0 or it == ITERS - 1:
        q = plant.qpos[0]
        print("iteration %3d (t = %.2f s): best return %.3f (nominal %.3f), torso height %.3f, forward travel %.2f m, planner %.2f ms/iteration"
              % (it, (it + 1) * agent_dt, ret[best], ret[0], q[2], q[0] - x_start, 1e3 * t_plan / iters), flush=True)
print("plant status flags:", int(plant.status()[0]), "| %.0f planner iterations per second with %d candidates" % (iters / t_plan, N))
