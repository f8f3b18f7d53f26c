"""Two-lane stepping of the 27-dof PGS kernel: single-step calls run hb_step_small_kernel (31 rows, 12 contacts, three
waves per SIMD); an env whose step overflows that instantiation is stepped by the full kernel on the slow lane until the next rebalance
point.  The results must be those of the full kernel alone (HB_TWO_LANE=0), whatever the lane an env is in:

  * teacher-forced, step by step, on the benchmark workload and on the collapsed regime (zero controls, every humanoid on the floor:
    up to 62 rows, a large part of the batch in the slow lane);
  * free-running over many steps with a short rebalance window, pipelined and unpipelined, across resets and host reads;
  * the oracle agrees with the small kernel's one-step results exactly as it does with the full kernel's (same tolerances as
    tests/test_gpu_parity.py: this kernel IS the benchmark's hot path now)."""
import os

import numpy as np
import pytest

from oracle_lib import GOLDEN

pytestmark = pytest.mark.gpu


def _batch(hbmod, model, n, gpu, two_lane, window=None):
    old = {k: os.environ.get(k) for k in ("HB_TWO_LANE", "HB_LANE_WINDOW")}
    os.environ["HB_TWO_LANE"] = "1" if two_lane else "0"
    if window is not None:
        os.environ["HB_LANE_WINDOW"] = str(window)
    try:
        return hbmod.Batch(model, n, gpu)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _equal_or_report(a, b, what):
    if np.array_equal(a, b):
        return 0.0
    d = np.abs(a.astype(np.float64) - b.astype(np.float64)).max()
    raise AssertionError("%s: two-lane and full-kernel results differ (max |d| %.3e, %d of %d entries)" % (what, d, int((a != b).sum()), a.size))


@pytest.mark.parametrize("regime", ["benchmark", "collapsed"])
def test_two_lane_steps_equal_the_full_kernel_teacher_forced(hbmod, humanoid_model, gpu, regime):
    """Every step from the SAME state on both batches: bit-identical states, counts and status, with envs in both lanes."""
    n = 768
    full = _batch(hbmod, humanoid_model, n, gpu, False)
    two = _batch(hbmod, humanoid_model, n, gpu, True, window=8)
    assert two.lanes().sum() == 0
    full.reset(perturb=True)
    pre = 700 if regime == "collapsed" else 300
    if regime == "collapsed":
        full.rollout(np.zeros((pre, n, humanoid_model.nu), np.float32))
    else:
        full.rollout_halton(pre)
    seen_slow = 0
    max_nefc = 0
    for t in range(48):
        st = full.get_state(hbmod.STATE_INTEGRATION)
        two.set_state(hbmod.STATE_INTEGRATION, st)
        if regime == "collapsed":
            ctrl = np.zeros((n, humanoid_model.nu), np.float32)
            full.step(ctrl); two.step(ctrl)
        else:
            full.rollout_halton(1, t0=pre + t); two.rollout_halton(1, t0=pre + t)
        _equal_or_report(full.get_state(hbmod.STATE_INTEGRATION), two.get_state(hbmod.STATE_INTEGRATION), "step %d" % t)
        for a, c in zip(full.counts(), two.counts()):
            assert np.array_equal(a, c)
        assert np.array_equal(full.status(), two.status())
        seen_slow = max(seen_slow, int(two.lanes().sum()))
        max_nefc = max(max_nefc, int(full.counts()[1].max()))
    print("\n%s: max nefc %d, up to %d of %d envs in the slow lane" % (regime, max_nefc, seen_slow, n))
    assert not full.lanes().any()
    if regime == "collapsed":
        assert max_nefc > 31 and seen_slow > 0  # the slow lane really ran
    full.close(); two.close()


@pytest.mark.parametrize("pipelined", [False, True])
def test_two_lane_free_running_equals_the_full_kernel(hbmod, humanoid_model, gpu, pipelined):
    """300 free-running steps (lanes filling and rebalancing every 8 steps), a masked reset and a host read in the middle, then a
    multi-step launch (full kernel: every env, slow lane joined first) and more single steps: bit-identical to the full kernel alone."""
    n, T = 1000, 150
    res = []
    slow_total = 0
    for two_lane in (False, True):
        b = _batch(hbmod, humanoid_model, n, gpu, two_lane, window=8)
        if pipelined:
            b.pipeline(True)
        b.reset(perturb=True)
        b.rollout(np.zeros((500, n, humanoid_model.nu), np.float32))  # down on the floor: rows around and above the small capacity
        for t in range(T):
            b.rollout_halton(1, t0=t)
            if two_lane and t % 10 == 9:
                slow_total += int(b.lanes().sum())
        mid = b.get_state(hbmod.STATE_INTEGRATION)
        b.reset(mask=(np.arange(n) % 5 == 0).astype(np.uint8), perturb=True)
        for t in range(T, 2 * T):
            b.rollout_halton(1, t0=t)
        b.rollout_halton(7, t0=2 * T)
        for t in range(2 * T + 7, 2 * T + 20):
            b.rollout_halton(1, t0=t)
        res.append((mid, b.get_state(hbmod.STATE_INTEGRATION), b.status(), b.counts()))
        b.close()
    _equal_or_report(res[0][0], res[1][0], "state after %d steps" % T)
    _equal_or_report(res[0][1], res[1][1], "final state")
    assert np.array_equal(res[0][2], res[1][2])
    for a, c in zip(res[0][3], res[1][3]):
        assert np.array_equal(a, c)
    print("\npipelined %s: slow-lane envs sampled over the run: %d" % (pipelined, slow_total))
    assert slow_total > 0


def test_small_kernel_one_step_parity_against_the_oracle(hbmod, humanoid_model, gpu):
    """The 128 golden states through the two-lane path WITHOUT diagnostics (diagnostics select the full kernel): next state within the
    one-step tolerances of tests/test_gpu_parity.py, counts identical - for the states the small kernel holds and for the ones it hands over."""
    g = np.load(os.path.join(GOLDEN, "humanoid27_steps.npz"))
    n = len(g["env"])
    b = _batch(hbmod, humanoid_model, n, gpu, True)
    st = np.concatenate([g["time"][:, None], g["qpos"], g["qvel"], g["warm"]], axis=1)
    b.set_state(hbmod.STATE_INTEGRATION, st)
    b.step(g["ctrl"].astype(np.float32))
    q, v = b.qpos.astype(np.float64), b.qvel.astype(np.float64)
    ncon, nefc, _ = b.counts()
    assert not b.status().any()
    assert np.array_equal(ncon, g["ncon"]) and np.array_equal(nefc, g["nefc"])
    dq = np.abs(q - g["qpos1"]) / np.maximum(1.0, np.abs(g["qpos1"]))
    vs = np.maximum(1.0, np.abs(g["qvel1"]).max(axis=1, keepdims=True))
    dv = np.abs(v - g["qvel1"]) / vs
    slow = b.lanes().astype(bool)
    print("\nsmall kernel: %d states, %d handed to the slow lane (nefc max %d); worst qpos %.2e qvel %.2e" % (n, int(slow.sum()), int(nefc.max()), dq.max(), dv.max()))
    assert dq.max() <= 4e-5 and dv.max() <= 4e-4
    assert np.array_equal(slow, (g["nefc"] > 31) | (g["ncon"] > 12))
    b.close()
