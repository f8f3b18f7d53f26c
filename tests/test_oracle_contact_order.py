"""Contact ORDER is one of the restatement's choices that cannot be checked against MuJoCo here.  What can be checked is
how much it matters: tools/contact_order_sensitivity.py re-orders the oracle's contact list on the 128 golden states.  Asserted here:
the converged solvers do not care (Newton to 1e-10, PGS run to convergence to 1e-6: one convex problem, one minimiser), so the order can
only move results through the 50-sweep cut of the benchmark configuration — and there it does, measurably (the numbers are in
profiles/r03_contact_order.txt and DESIGN.md §2)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_contact_order_moves_only_the_truncated_pgs(capsys):
    from contact_order_sensitivity import main
    r = main()
    assert r["newton"]["qacc"] < 1e-10 and r["newton"]["fn"] < 1e-10
    assert r["pgs_conv"]["qacc"] < 1e-6 and r["pgs_conv"]["fn"] < 1e-5
    # PGS cut at 50 sweeps: order-dependent wherever the cap binds; bounded by what was measured, a regression guard
    assert 1e-4 < r["pgs50"]["qacc"] < 0.1 and r["pgs50"]["fn"] < 0.5
    ps = r["pgs50"]["per_state"]
    assert ps[len(ps) // 2] < 1e-4  # the median state (PGS converged inside 50 sweeps) is order-free to solver tolerance
