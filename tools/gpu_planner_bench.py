#!/usr/bin/env python3
"""One sampling-planner iteration on the device: N candidates from one state, MJPC's Humanoid Stand task
(agent_horizon 0.35 s at agent_timestep 0.015 s -> 24 states), rollouts + residuals + costs + returns in one call,
N floats back.  Host action tape upload included (that is what a planner hands over)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import humanoid_mujoco_amd as hb
from oracle_lib import Oracle
m = hb.Model.load(os.path.join(ROOT, "humanoid_mujoco_amd", "assets", "humanoid27.hbm"))
o = Oracle(); o.init_env(1)
for t in range(40):
    o.ctrl[:] = o.ctrl_env(t, 1); o.step()
st = np.concatenate([[0.0], o.qpos, o.qvel, o.qacc_warmstart])
H = 24
for solver, name in ((0, "PGS/50"), (2, "Newton/100")):
    if solver == 2:
        m.set_opt(solver=2, iterations=100)
    for N in (1024, 4096, 16384):
        b = hb.Batch(m, N, 0)
        task = b.task_stand_default()
        rng = np.random.default_rng(0)
        ctrl = rng.uniform(-0.5, 0.5, size=(H - 1, N, m.nu)).astype(np.float32)
        b.set_state_broadcast(hb.STATE_INTEGRATION, st)
        b.rollout_task_stand(ctrl, task)
        reps = 10
        t0 = time.perf_counter()
        for _ in range(reps):
            b.set_state_broadcast(hb.STATE_INTEGRATION, st)
            total, _ = b.rollout_task_stand(ctrl, task)
        dt = (time.perf_counter() - t0) / reps
        # the same iteration with the candidates' spline policies evaluated on the device (3 nodes, cubic: task.xml)
        knots = rng.uniform(-0.5, 0.5, size=(N, 3, m.nu)).astype(np.float32)
        times = np.array([0.0, 0.175, 0.35], np.float32)
        b.set_state_broadcast(hb.STATE_INTEGRATION, st)
        b.ctrl_tape_splines(knots, times, 2, 0.0, H - 1)
        b.rollout_task_stand(("tape", H - 1), task)
        t2 = time.perf_counter()
        for _ in range(reps):
            b.set_state_broadcast(hb.STATE_INTEGRATION, st)
            b.ctrl_tape_splines(knots, times, 2, 0.0, H - 1)
            stotal, _ = b.rollout_task_stand(("tape", H - 1), task)
        dts = (time.perf_counter() - t2) / reps
        print("%-10s %6d candidates x %d states, spline policies sampled on the device: %.2f ms per planner iteration" % (name, N, H, 1e3 * dts), flush=True)
        wt = b.task_walk_default()
        b.set_state_broadcast(hb.STATE_INTEGRATION, st)
        b.rollout_task_walk(ctrl, wt)
        t1 = time.perf_counter()
        for _ in range(reps):
            b.set_state_broadcast(hb.STATE_INTEGRATION, st)
            wtotal, _ = b.rollout_task_walk(ctrl, wt)
        dtw = (time.perf_counter() - t1) / reps
        print("%-10s %6d candidates x %d states, Walk task: %.2f ms per planner iteration; best return %.3f" % (name, N, H, 1e3 * dtw, float(wtotal.min())), flush=True)
        print("%-10s %6d candidates x %d states: %.2f ms per planner iteration (%.3e env-steps/s incl. tape upload, cost evaluation and return download); best return %.3f, failures %d"
              % (name, N, H, 1e3 * dt, N * (H - 1) / dt, float(total.min()), int((total >= 1e6).sum())), flush=True)
        b.close()
