"""Step calls enqueued back to back run as ONE launch (include/hb.h: hb_step_dev, HB_TUNE_FOLD; csrc/hb_api.cpp: fold_steps).

The reference steps a thread's envs in a loop with nothing between the steps (simulation/mujoco/sample/testspeed.cc:93-96: mj_step
after mj_step); hb_step_dev is asynchronous, so K calls in a row are the same thing to the caller as one launch of K steps - which
has no batch-wide barrier between the steps.  Held here: the folded launches give bit-identical states, counts and status to one
launch per call, for every pattern of calls that folds or has to stop folding."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
N = 512  # (three segments of a pipelined batch: 170 / 171 envs, an odd number for the two-envs-per-wave kernel)
T = 150


def _ctrl(hbmod, b, nu, t0=7):
    p = b.dev_alloc(T * N * nu * 4)
    b.halton_ctrl_dev(T, t0, 0, p)
    return p, N * nu * 4


def _policy(m):
    rng = np.random.default_rng(11)
    sizes = [m.nobs, 32, m.nu]
    return ([(0.2 * rng.standard_normal((sizes[i], sizes[i + 1]))).astype(np.float32) for i in range(2)],
            [(0.1 * rng.standard_normal(sizes[i + 1])).astype(np.float32) for i in range(2)])


def _everything(hbmod, b):
    return (b.get_state(hbmod.STATE_INTEGRATION),) + tuple(b.counts()) + (b.status(),)


def _same(x, y):
    return all(np.array_equal(a, c) for a, c in zip(x, y))


@pytest.mark.parametrize("duo,pipelined,kernel", [(2, 1, "hb_step_duo_q_kernel"), (2, 2, "hb_step_duo_q_kernel"), (0, 1, "hb_step_h27_q_kernel"), (1, 3, "hb_step_h27_q_kernel"), (2, 0, "hb_step_duo_kernel")])
def test_folded_step_calls_are_bit_identical(hbmod, humanoid_model, gpu, duo, pipelined, kernel):
    """(an unpipelined batch does not fold: its caller was never asked to join before using the batch's stream - include/hb.h)"""
    m = humanoid_model
    got = []
    for fold in (1, 64):
        folds = fold > 1 and pipelined
        b = hbmod.Batch(m, N, gpu)
        b.tune(duo=duo, fold=fold)
        b.reset(perturb=True)
        b.pipeline(pipelined)
        ctrl, stride = _ctrl(hbmod, b, m.nu)
        for t in range(T):  # 150 calls: two full launches of 64 steps and one of 22
            b.step_dev(ctrl + t * stride)
        b.sync()
        assert b.last_kernel() == (kernel if folds else kernel.replace("_q_kernel", "_kernel"))
        got.append(_everything(hbmod, b))
        b.dev_free(ctrl)
        b.close()
    assert _same(got[0], got[1])
    assert got[0][1].max() >= 3  # contacts were there: fallen humanoids


def test_folded_calls_of_other_models_and_solvers(hbmod, gpu):
    """every primitive-geometry model folds when its waves fit on the chip at once: the humanoid under Newton (hb_step_newton28_lean_q_kernel)
    and a model without the humanoid's size signature, the particle of tests/golden/particle_task.hbm (the generic kernels)"""
    import os
    from oracle_lib import GOLDEN, HUMANOID_HBM
    newton = hbmod.Model.load(HUMANOID_HBM)
    newton.set_opt(solver=2, iterations=100)
    particle = hbmod.Model.load(os.path.join(GOLDEN, "particle_task.hbm"))
    for m, kernel in ((newton, "hb_step_newton28_lean_q_kernel"), (particle, None)):
        got, names, launches = [], [], []
        for fold in (1, 256):
            b = hbmod.Batch(m, N, gpu)
            b.tune(fold=fold)
            b.reset(perturb=True)
            b.pipeline(True)
            p = b.dev_alloc(T * N * max(m.nu, 1) * 4)
            b.halton_ctrl_dev(T, 3, 0, p)
            n0 = b.step_launches()
            for t in range(T):
                b.step_dev(p + t * N * m.nu * 4)
            b.sync()
            launches.append(b.step_launches() - n0)
            names.append(b.last_kernel())
            got.append(_everything(hbmod, b))
            b.dev_free(p)
            b.close()
        assert _same(got[0], got[1])
        assert launches == [T, 1] and (kernel is None or names[1] == kernel), (launches, names)


def test_fold_stops_where_it_has_to(hbmod, humanoid_model, gpu):
    """reads, writes into the control buffer through the batch's API, substeps, the host-control step and a rollout between the calls"""
    m = humanoid_model
    rng = np.random.default_rng(3)
    host_ctrl = rng.uniform(-1, 1, (N, m.nu)).astype(np.float32)
    got = []
    for fold in (1, 256, 5):
        b = hbmod.Batch(m, N, gpu)
        b.tune(fold=fold, duo=2)
        b.reset(perturb=True)
        b.pipeline(True)
        b.set_policy_mlp(*_policy(m))
        ctrl, stride = _ctrl(hbmod, b, m.nu)
        one = b.dev_alloc(stride)
        mid = []
        for t in range(40):
            b.step_dev(ctrl + t * stride)
        mid.append(b.get_state(hbmod.STATE_INTEGRATION))       # a read in the middle
        for t in range(40, 60):                                  # the closed-loop pattern: ONE buffer, rewritten between the calls
            b.halton_ctrl_dev(1, 7 + t, 0, one)
            b.step_dev(one)
        for t in range(60, 70):
            b.step_dev(ctrl + t * stride, 3)                     # substeps: the same controls three steps long
        b.step(host_ctrl)                                        # host controls (synchronous)
        for t in range(70, 80):
            b.step_dev(ctrl + t * stride)
        b.rollout_halton(9, 500)                                 # another kind of launch behind held step calls
        for t in range(76, 80):
            b.step_dev(ctrl + t * stride)
        b.rollout_policy(3)                                      # the closed loop launches its segments itself (fork_pipes): held calls first
        for t in range(80, 90):
            b.step_dev(ctrl + t * stride)
        mid.append(b.counts()[0])
        for t in range(90, 100):
            b.step_dev(ctrl + t * stride)
        b.tune(reorder_period=2)                                 # a knob turned between step calls
        for t in range(100, 110):
            b.step_dev(ctrl + t * stride)
        b.sync()
        got.append(tuple(mid) + _everything(hbmod, b))
        b.dev_free(ctrl); b.dev_free(one)
        b.close()
    assert _same(got[0], got[1]) and _same(got[0], got[2])


def test_callers_own_work_on_the_batch_stream_between_folded_calls(hbmod, humanoid_model, gpu):
    """the rule of a pipelined batch (include/hb.h: hb_batch_pipeline): hb_batch_join, then the caller's own work on the batch's stream.  Here a
    torch copy rewrites ONE control buffer between the step calls, on a stream handle fetched once"""
    torch = pytest.importorskip("torch")
    m = humanoid_model
    dev = torch.device("cuda", gpu)
    rng = np.random.default_rng(5)
    seq = torch.from_numpy(rng.uniform(-1, 1, (30, N, m.nu)).astype(np.float32)).to(dev)
    torch.cuda.synchronize(dev)
    b = hbmod.Batch(m, N, gpu)
    b.tune(duo=2)
    b.reset(perturb=True)
    b.pipeline(True)
    ref = hbmod.Batch(m, N, gpu)
    ref.tune(fold=1, duo=2)
    ref.reset(perturb=True)
    ref.pipeline(True)
    buf = torch.zeros((N, m.nu), dtype=torch.float32, device=dev)
    s = torch.cuda.ExternalStream(b.stream, device=dev)
    for t in range(30):
        b.join()
        with torch.cuda.stream(s):
            buf.copy_(seq[t])
        b.step_dev(buf.data_ptr())
        if t % 3 == 0:
            b.step_dev(buf.data_ptr())  # and two calls in a row on the same controls
            ref.step_dev(seq[t].data_ptr())
        ref.step_dev(seq[t].data_ptr())
    b.sync(); ref.sync()
    assert np.array_equal(b.get_state(hbmod.STATE_INTEGRATION), ref.get_state(hbmod.STATE_INTEGRATION))
    b.close(); ref.close()


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_random_sequences_of_calls_give_the_same_states_folded_or_not(hbmod, humanoid_model, gpu, seed):
    """a few hundred calls in random order - step calls with one to three substeps, reads, state writes, masked resets, rollouts, forward passes,
    knobs, pipelining switched on and off, joins, the timers - on one batch that folds and one that does not: the same states at every read"""
    m = humanoid_model
    n = 256
    batches = []
    for fold in (1, 256):
        b = hbmod.Batch(m, n, gpu)
        b.tune(fold=fold, duo=2 if seed else 1)
        b.reset(perturb=True)
        b.pipeline(True)
        batches.append(b)
    ctrls = [_ctrl(hbmod, b, m.nu, t0=11 + seed) for b in batches]
    rng = np.random.default_rng(100 + seed)
    reads = [[], []]
    t = 0
    for op in range(260):
        k = int(rng.integers(0, 20))
        arg = rng.integers(0, 1 << 30)
        for i, b in enumerate(batches):
            r = np.random.default_rng(int(arg))  # the same draw for both batches
            p, stride = ctrls[i]
            if k < 11:
                b.step_dev(p + (t % T) * stride, int(r.integers(1, 4)) if k == 10 else 1)
            elif k == 11:
                reads[i].append(b.get_state(hbmod.STATE_INTEGRATION))
            elif k == 12:
                st = b.get_state(hbmod.STATE_INTEGRATION)
                st[:, 1 + m.nq:] *= 0.5
                b.set_state(hbmod.STATE_INTEGRATION, st)
            elif k == 13:
                b.reset(mask=(r.uniform(size=n) < 0.1).astype(np.uint8), perturb=True, env_offset=int(r.integers(0, 50)))
            elif k == 14:
                b.rollout_halton(int(r.integers(1, 12)), int(r.integers(0, 100)))
            elif k == 15:
                b.forward()
            elif k == 16:
                b.tune(reorder_period=int(r.integers(1, 6)))
            elif k == 17:
                b.pipeline(int(r.integers(0, 4)))
            elif k == 18:
                b.join()
            else:
                b.timer_start(); b.step_dev(p + (t % T) * stride); b.timer_stop()
        t += 1
    for i, b in enumerate(batches):
        b.sync()
        reads[i].append(b.get_state(hbmod.STATE_INTEGRATION))
        reads[i] += list(b.counts()) + [b.status()]
    assert len(reads[0]) == len(reads[1]) and all(np.array_equal(x, y) for x, y in zip(reads[0], reads[1]))
    for (b, (p, _)) in zip(batches, ctrls):
        b.dev_free(p)
        b.close()
