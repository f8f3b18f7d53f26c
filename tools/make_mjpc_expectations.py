#!/usr/bin/env python3
"""Lifts the known-answer expectations the reference's own tests hold for pieces of this path into
tests/golden/mjpc_expectations.npz (a JSON manifest of cases: inputs -> expected outputs, each with its file:line).

REFERENCE-HELD VECTORS: every number below is a literal (or the closed-form expression) written in a reference test; nothing here is
produced by this repository's oracle or by tests/mjpc_ref.py.  Sources (relative to /root/reference/mujoco_mpc/mjpc/test):
  spline/spline_test.cc:40-355        TimeSpline::Sample — zero-order, linear, cubic Hermite, boundary behaviour
  agent/agent_utilities_test.cc:208-221,264-283   Clamp to bounds; LinearInterpolation inside / below / above
  tasks/task_test.cc:49-98            cost terms (two quadratic norms, weights 5.0 and 0.1) and the risk transformation
  state/state_test.cc:43-58           State packing: qpos | qvel
  agent/agent_utilities_test.cc:32-63,196-202     SetState / GetState round trip; keyframe "home" qpos
When /root/reference is present the script checks that each cited line range still contains the literals a case was lifted from;
the committed .npz is what the tests read (the reference tree does not travel to the GPU box).

What is adapted, and only this: the reference's splines have dim 1, 2 or 10; the device evaluates splines of dimension nu of a model,
so cases run on a two-actuator model (dim 1 cases use column 0; the dim-10 "Empty" case checks nu zeros).  Container behaviour
(Reserve, DiscardBefore, iterators, move / copy) has no device counterpart: such cases enter with the node set the container holds at
the time of the Sample() call.

    python tools/make_mjpc_expectations.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/mujoco_mpc/mjpc/test"
OUT = os.path.join(ROOT, "tests", "golden", "mjpc_expectations.npz")
ZERO, LINEAR, CUBIC = 0, 1, 2  # spline.h:30-34
ALL = (ZERO, LINEAR, CUBIC)

spline = []  # {cite, must, interp, times, values, samples: [[t, [expected...]], ...]}


def sp(cite, must, interps, nodes, samples):
    for it in interps:
        spline.append({"cite": cite, "must": must, "interp": it, "times": [float(n[0]) for n in nodes], "values": [list(map(float, n[1])) for n in nodes],
                       "samples": [[float(t), list(map(float, v))] for t, v in samples]})


F = "spline/spline_test.cc"
sp(F + ":40-49", ["TimeSpline spline(/*dim=*/10)", "spline.Sample(2.0", "EXPECT_EQ(v, 0.0)"], (ZERO,), [], [(2.0, [0.0, 0.0])])
sp(F + ":51-62", ["spline.AddNode(1.0, {1.0, 2.0})", "{0.0, 2.0, 4.0}", "ElementsAre(1.0, 2.0)"], ALL, [(1.0, [1, 2])],
   [(0.0, [1, 2]), (2.0, [1, 2]), (4.0, [1, 2])])
sp(F + ":64-80", ["spline.AddNode(1.0, {1.0, 2.0})", "n.values()[0] = 3.0", "n.values()[1] = 4.0", "spline.Sample(3), ElementsAre(3.0, 4.0)"], ALL,
   [(1.0, [1, 2]), (2.0, [3, 4])], [(0, [1, 2]), (1, [1, 2]), (2, [3, 4]), (3, [3, 4])])
sp(F + ":82-96", ["spline.AddNode(0.0, {0.0, 1.0})", "spline.Sample(3), ElementsAre(3.0, 4.0)"], (ZERO,),
   [(0.0, [0, 1]), (1.0, [1, 2]), (2.0, [2, 3]), (3.0, [3, 4])], [(0, [0, 1]), (1, [1, 2]), (2, [2, 3]), (3, [3, 4])])
sp(F + ":115-122", ["kZeroSpline", "spline.Sample(1.5), ElementsAre(1.0, 2.0)"], (ZERO,), [(1.0, [1, 2]), (2.0, [3, 4])], [(1.5, [1, 2])])
sp(F + ":124-131", ["kLinearSpline", "spline.Sample(1.5), ElementsAre(2.0, 3.0)"], (LINEAR,), [(1.0, [1, 2]), (2.0, [3, 4])], [(1.5, [2, 3])])
sp(F + ":133-140", ["kCubicSpline", "spline.Sample(1.5), ElementsAre(2.0, 3.0)"], (CUBIC,), [(1.0, [1, 2]), (2.0, [3, 4])], [(1.5, [2, 3])])
sp(F + ":142-149", ["spline.AddNode(0.0, {1.0, 2.0})", "spline.AddNode(3.0, {3.0, 4.0})", "spline.Sample(1.5), ElementsAre(2.0, 3.0)"], (CUBIC,),
   [(0.0, [1, 2]), (1.0, [1, 2]), (2.0, [3, 4]), (3.0, [3, 4])], [(1.5, [2, 3])])
# "Known solution for this spline": y = -x^3 + 2 x^2 for x = 0, 0.125, ..., 1 (spline_test.cc:151-159); dim 1 -> column 0, column 1 is zero
sp(F + ":151-159", ["spline.AddNode(-1.0, {1.0})", "spline.AddNode(0.0, {0.0})", "spline.AddNode(1.0, {1.0})", "x += 0.125", "-std::pow(x, 3) + 2 * std::pow(x, 2)"],
   (CUBIC,), [(-1.0, [1, 0]), (0.0, [0, 0]), (1.0, [1, 0])], [(x, [-x ** 3 + 2 * x ** 2, 0.0]) for x in np.arange(0.0, 1.0 + 1e-9, 0.125)])
# DiscardBefore: the node sets the container holds before and after DiscardBefore(3.0) (spline_test.cc:161-206)
nodes4 = [(1.0, [1, 2]), (2.0, [2, 3]), (3.0, [3, 4]), (4.0, [4, 5])]
sp(F + ":161-181", ["spline.AddNode(4.0, {4.0, 5.0})", "spline.Sample(1.0), ElementsAre(1.0, 2.0)", "spline.Sample(0.0), ElementsAre(1.0, 2.0)"], ALL, nodes4,
   [(1.0, [1, 2]), (0.0, [1, 2])])
sp(F + ":180-206", ["spline.DiscardBefore(3.0)", "EXPECT_EQ(discarded, 1)", "spline.Sample(1.0), ElementsAre(2.0, 3.0)"], (CUBIC,), nodes4[1:], [(1.0, [2, 3])])
sp(F + ":180-206", ["spline.DiscardBefore(3.0)", "EXPECT_EQ(discarded, 2)", "spline.Sample(1.0), ElementsAre(3.0, 4.0)"], (ZERO, LINEAR), nodes4[2:], [(1.0, [3, 4])])
sp(F + ":208-231", ["spline.AddNode(6.0, {6.0})", "spline.DiscardBefore(6.0), 3", "EXPECT_EQ(spline.Sample(1.0)[0], 6.0)"], (ZERO,), [(6.0, [6, 0])], [(1.0, [6, 0])])
sp(F + ":233-247", ["kLinearSpline", "spline.AddNode(3.0, {4.0, 5.0})", "spline.Sample(2.5), ElementsAre(3.0, 4.0)"], (LINEAR,),
   [(1.0, [1, 2]), (2.0, [2, 3]), (3.0, [4, 5])], [(2.5, [3, 4])])
sp(F + ":249-262", ["spline.AddNode(4.0, {4.0, 5.0})", "spline2.Sample(2.5), ElementsAre(2.0, 3.0)"], (ZERO,), nodes4, [(2.5, [2, 3])])
sp(F + ":285-310", ["kLinearSpline", "spline.DiscardBefore(2.0)", "spline.AddNode(5.0, {5.0, 6.0})", "spline2.Sample(1.5), ElementsAre(2.0, 3.0)",
                    "spline2.Sample(2.5), ElementsAre(2.5, 3.5)"], (LINEAR,), [(2.0, [2, 3]), (3.0, [3, 4]), (4.0, [4, 5]), (5.0, [5, 6])],
   [(1.5, [2, 3]), (2.5, [2.5, 3.5])])
sp(F + ":341-355", ["spline.Sample(0), ElementsAre(1.0, 2.0)", "spline.Clear()", "spline.Sample(0), ElementsAre(0.0, 0.0)"], (ZERO,), [(1.0, [1, 2])], [(0, [1, 2])])
sp(F + ":341-355", ["spline.Clear()", "spline.Sample(0), ElementsAre(0.0, 0.0)"], (ZERO,), [], [(0, [0, 0])])
sp(F + ":341-355", ["spline.AddNode(1.0, {3.0, 4.0})", "spline.Sample(1), ElementsAre(3.0, 4.0)"], (ZERO,), [(1.0, [3, 4])], [(1, [3, 4])])
# LinearInterpolation (agent_utilities_test.cc:264-283): x {1, 2}, y {1, 2}; inside, below, above
sp("agent/agent_utilities_test.cc:264-283", ["std::vector<double> x{1.0, 2.0}", "double y[2] = {1.0, 2.0}", "EXPECT_NEAR(y1, 1.5", "EXPECT_NEAR(y2, 1.0", "EXPECT_NEAR(y3, 2.0"],
   (LINEAR,), [(1.0, [1, 0]), (2.0, [2, 0])], [(1.5, [1.5, 0]), (0.5, [1.0, 0]), (2.5, [2.0, 0])])

# Clamp(x, bounds, 3) with bounds +-1 (agent_utilities_test.cc:208-221): on the particle model (ctrlrange [-1, 1], nu = 2) as one-node
# splines, the way SamplingPolicy::Action clamps what the spline returns (policy.cc:50-58); three values -> two candidates
clamp = {"cite": "agent/agent_utilities_test.cc:208-221", "must": ["bounds[6] = {-1.0, 1.0, -1.0, 1.0, -1.0, 1.0}", "x[3] = {-2.0, 3.0, 0.0}", "EXPECT_NEAR(x[0], -1.0",
                                                                     "EXPECT_NEAR(x[1], 1.0", "EXPECT_NEAR(x[2], 0.0"],
         "x": [-2.0, 3.0, 0.0], "expect": [-1.0, 1.0, 0.0]}

# TasksTest.Task (task_test.cc:49-98)
r = [1.0e-3, 2.0e-3, 3.0e-3, 4.0e-3]
c = 5.0 * 0.5 * (r[0] * r[0] + r[1] * r[1]) + 0.1 * 0.5 * (r[2] * r[2] + r[3] * r[3])  # task_test.cc:82-85
task = {"cite": "tasks/task_test.cc:49-98",
        "must": ["task.weight[0], 5.0", "task.weight[1], 0.1", "NormType::kQuadratic", "dim_norm_residual[0], 2", "dim_norm_residual[1], 2",
                 "residual[] = {1.0e-3, 2.0e-3, 3.0e-3, 4.0e-3}", "c += 5.0 * 0.5 * mju_dot(residual, residual, 2)",
                 "c += 0.1 * 0.5 * mju_dot(residual + 2, residual + 2, 2)", "task.risk = 0.2", "(mju_exp(task.risk * c) - 1.0) / task.risk"],
        "dims": [2, 2], "norms": [0, 0], "weights": [5.0, 0.1], "residual": r, "terms_sum": c, "risk": 0.2, "cost_value": (np.exp(0.2 * c) - 1.0) / 0.2,
        "xml_risk": 1.0}

# StateTest (state_test.cc:43-58) and AgentUtilitiesTest.State / ByName (agent_utilities_test.cc:32-63, 196-202)
state = {"cite": "state/state_test.cc:43-58", "must": ["mju_fill(data->qpos, 1.0, model->nq)", "mju_fill(data->qvel, 2.0, model->nv)", "state.state_[0], 1.0",
                                                         "state.state_[2], 2.0"], "qpos_fill": 1.0, "qvel_fill": 2.0, "expect": [1.0, 1.0, 2.0, 2.0],
         "cite2": "agent/agent_utilities_test.cc:32-63", "must2": ["state[4] = {0.1, 0.2, 0.3, 0.4}", "data->qpos[1], state[1]", "data->qvel[0], state[2]"],
         "roundtrip": [0.1, 0.2, 0.3, 0.4],
         "cite3": "agent/agent_utilities_test.cc:196-202", "must3": ["KeyQPosByName(model, data, \"home\")", "qpos_home[0], 1.0", "qpos_home[1], 2.0"],
         "key": "home", "key_qpos": [1.0, 2.0]}


def check_source(cite, must):
    path, span = cite.rsplit(":", 1)
    a, b = map(int, span.split("-"))
    full = os.path.join(REF, path)
    text = " ".join(" ".join(open(full).read().split("\n")[a - 1:b]).split())
    for m in must:
        assert " ".join(m.split()) in text, (cite, m)


def main():
    if os.path.isdir(REF):
        for cs in spline:
            check_source(cs["cite"], cs["must"])
        check_source(clamp["cite"], clamp["must"])
        check_source(task["cite"], task["must"])
        check_source(state["cite"], state["must"])
        check_source(state["cite2"], state["must2"])
        check_source(state["cite3"], state["must3"])
        print("every case found in the reference sources at its cited lines")
    else:
        print("reference tree not present: literals not re-checked")
    manifest = {"label": "REFERENCE-HELD EXPECTATIONS lifted from mujoco_mpc/mjpc/test (see tools/make_mjpc_expectations.py); no oracle-generated value",
                "spline": spline, "clamp": clamp, "task": task, "state": state}
    np.savez(OUT, manifest=np.array(json.dumps(manifest, sort_keys=True)))
    print("wrote", OUT, "-", len(spline), "spline cases,", sum(len(c["samples"]) for c in spline), "samples")


if __name__ == "__main__":
    sys.exit(main())
