#!/usr/bin/env python3
"""How much could MuJoCo's (unknown here) contact ORDER move this path's results?  One of the restatement's unverifiable choices
(DESIGN.md §2): the oracle and the device emit contacts in static (geom1, geom2) pair order; MuJoCo 3.1 visits sorted body pairs and
traverses a BVH inside each.  The order changes nothing physical — the constraint problem is the same convex problem — but
Gauss-Seidel cut at 50 sweeps is order-dependent.  This tool re-orders the oracle's contact list (om_set_contact_order: reversed
inside every body pair, fully reversed, eight seeded shuffles) on the 128 golden states of the benchmark workload and reports the
largest change of qacc, qfrc_constraint, the next qvel and the matched per-contact normal forces, for PGS/50 (the benchmark
configuration), Newton/100 (the reference's default) and PGS run to convergence.

    python tools/contact_order_sensitivity.py [> profiles/r03_contact_order.txt]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle_lib import GOLDEN, Oracle  # noqa: E402

ORDERS = [(1, 0, "reversed inside each body pair"), (2, 0, "fully reversed")] + [(3, s, "shuffle %d" % s) for s in range(8)]


def one_step(o, g, k):
    o.reset()
    o.L.om_data_set_time(o.d, float(g["time"][k]))
    o.qpos[:] = g["qpos"][k]; o.qvel[:] = g["qvel"][k]; o.qacc_warmstart[:] = g["warm"][k]; o.ctrl[:] = g["ctrl"][k]
    o.forward()
    qacc, qfc = o.qacc.copy(), o.qfrc_constraint.copy()
    # normal force per contact = the sum of its pyramid rows' forces (condim 3: four rows; condim 1: one), keyed by what identifies a contact
    fn = {}
    f = o.efc_force[:o.nefc]
    for c in o.contacts():
        rows = 1 if c["dim"] == 1 else 2 * (c["dim"] - 1)
        fn[(c["geom1"], c["geom2"], tuple(np.round(c["pos"], 9)))] = float(f[c["efc_address"]:c["efc_address"] + rows].sum())
    niter = o.dint("solver_niter")
    o.step()
    return qacc, qfc, o.qvel.copy(), fn, niter


def sensitivity(solver_name, **opt):
    g = np.load(os.path.join(GOLDEN, "humanoid27_steps.npz"))
    o = Oracle()
    o.set_opt(**opt)
    n = len(g["time"])
    worst = dict(qacc=0.0, qfrc=0.0, qvel=0.0, fn=0.0)
    per_order = {}
    multi = sweeps = capped = 0
    per_state = []
    for k in range(n):
        o.L.om_set_contact_order(0, 0)
        qa0, qf0, qv0, fn0, it0 = one_step(o, g, k)
        if o.ncon >= 2:
            multi += 1
        sweeps += it0
        capped += int(it0 >= opt.get("iterations", 1 << 30))
        per_state.append(0.0)
        for mode, seed, label in ORDERS:
            o.L.om_set_contact_order(mode, seed)
            qa, qf, qv, fn, _ = one_step(o, g, k)
            assert set(fn) == set(fn0)
            d = dict(qacc=np.abs(qa - qa0).max() / max(1.0, np.abs(qa0).max()), qfrc=np.abs(qf - qf0).max() / max(1.0, np.abs(qf0).max()),
                     qvel=np.abs(qv - qv0).max() / max(1.0, np.abs(qv0).max()),
                     fn=max([abs(fn[c] - fn0[c]) for c in fn0], default=0.0) / max([1.0] + [abs(x) for x in fn0.values()]))
            for key, x in d.items():
                worst[key] = max(worst[key], x)
            per_order[label] = max(per_order.get(label, 0.0), d["qacc"])
            per_state[-1] = max(per_state[-1], d["qacc"])
    o.L.om_set_contact_order(0, 0)
    print("%-28s states with >= 2 contacts: %d of %d; mean iterations %.1f, at the cap in %d states" % (solver_name, multi, n, sweeps / n, capped))
    print("    max over states and orders:  d qacc %.2e   d qfrc_constraint %.2e   d qvel(next) %.2e   d normal force %.2e   (relative to max(1, |.|_inf))"
          % (worst["qacc"], worst["qfrc"], worst["qvel"], worst["fn"]))
    ps = np.sort(per_state)
    print("    d qacc per state (max over orders): median %.1e, 90th percentile %.1e, states above 1e-6: %d" % (ps[len(ps) // 2], ps[int(0.9 * len(ps))], int((ps > 1e-6).sum())))
    worst["per_state"] = ps
    print("    d qacc by order: " + ", ".join("%s %.1e" % (k, v) for k, v in per_order.items()))
    return worst


def main():
    print("contact-order sensitivity on tests/golden/humanoid27_steps.npz (128 teacher-forced states of the benchmark workload), fp64 oracle")
    print("baseline order: static (geom1, geom2) pair order, what oracle and device emit\n")
    r = {}
    r["pgs50"] = sensitivity("PGS, <= 50 sweeps, tol 1e-8", solver=0, iterations=50, tolerance=1e-8)
    r["newton"] = sensitivity("Newton, <= 100 iterations", solver=2, iterations=100, tolerance=1e-8)
    r["pgs_conv"] = sensitivity("PGS to convergence", solver=0, iterations=20000, tolerance=1e-14)
    return r


if __name__ == "__main__":
    main()
