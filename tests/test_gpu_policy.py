"""BASELINE config 4: 2-hidden-layer tanh MLP policy (48 -> 256 -> 256 -> 21, sizes from
simulation/hyperparam_config.py:21-27) evaluated on the GPU inside the rollout loop, against numpy."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def make_policy(nobs, nu, hidden=256, seed=0):
    """nn.Linear-style initialisation U(-1/sqrt(fan_in), 1/sqrt(fan_in)) from a fixed numpy seed, as [in, out]
    (numpy rather than torch so the GPU suite does not pay a torch import on a fresh box)."""
    rng = np.random.default_rng(seed)
    sizes = [nobs, hidden, hidden, nu]
    ws, bs = [], []
    for a, b in zip(sizes[:-1], sizes[1:]):
        k = 1.0 / np.sqrt(a)
        ws.append(rng.uniform(-k, k, size=(a, b)).astype(np.float32))
        bs.append(rng.uniform(-k, k, size=b).astype(np.float32))
    return ws, bs


def mlp_ref(obs, ws, bs):
    x = obs.astype(np.float64)
    for w, b in zip(ws, bs):
        x = np.tanh(x @ w.astype(np.float64) + b.astype(np.float64))
    return x


def test_policy_eval_matches_numpy(hbmod, humanoid_model, gpu):
    m = humanoid_model
    n = 1000  # not a multiple of 32: exercises the tile edge
    b = hbmod.Batch(m, n, gpu)
    b.reset(perturb=True)
    b.rollout_halton(120)
    ws, bs = make_policy(m.nobs, m.nu)
    b.set_policy_mlp(ws, bs)
    ctrl = b.policy_eval()
    obs, _, _, _ = b.obs(want_reward=False)
    ref = mlp_ref(obs, ws, bs)
    assert ctrl.shape == (n, m.nu)
    assert np.abs(ctrl - ref).max() < 2e-5
    assert np.abs(ctrl).max() < 1.0 and np.abs(ctrl).std() > 0.01


@pytest.mark.parametrize("hidden", [(33,), (64, 17, 130), (300, 64), (512,)])
def test_policy_shapes(hbmod, humanoid_model, gpu, hidden):
    """Odd widths (K swept in pairs: pad column), fewer than eight output tiles (K split over waves), up to
    four layers — the one-launch kernel; a layer wider than 256 takes the layer-by-layer path.  Both against numpy."""
    m = humanoid_model
    n = 77
    rng = np.random.default_rng(5)
    sizes = [m.nobs, *hidden, m.nu]
    ws = [rng.uniform(-0.3, 0.3, size=(a, c)).astype(np.float32) for a, c in zip(sizes[:-1], sizes[1:])]
    bs = [rng.uniform(-0.3, 0.3, size=c).astype(np.float32) for c in sizes[1:]]
    b = hbmod.Batch(m, n, gpu)
    b.reset(perturb=True)
    b.rollout_halton(50)
    b.set_policy_mlp(ws, bs)
    ctrl = b.policy_eval()
    obs, _, _, _ = b.obs(want_reward=False)
    assert np.abs(ctrl - mlp_ref(obs, ws, bs)).max() < 5e-5


def test_closed_loop_rollout_matches_host_loop(hbmod, humanoid_model, gpu):
    m = humanoid_model
    n, T = 64, 25
    ws, bs = make_policy(m.nobs, m.nu)
    a = hbmod.Batch(m, n, gpu)
    c = hbmod.Batch(m, n, gpu)
    for x in (a, c):
        x.reset(perturb=True)
        x.set_policy_mlp(ws, bs)
    a.rollout_policy(T)
    a.sync()
    for t in range(T):  # same loop driven from the host: policy_eval then step
        c.step(c.policy_eval())
    assert np.array_equal(a.get_state(hbmod.STATE_INTEGRATION), c.get_state(hbmod.STATE_INTEGRATION))
    # and against the numpy policy in the loop (fp32 tanh vs fp64: tiny control differences only)
    d = hbmod.Batch(m, n, gpu)
    d.reset(perturb=True)
    for t in range(5):
        obs, _, _, _ = d.obs(want_reward=False)
        d.step(mlp_ref(obs, ws, bs).astype(np.float32))
    e = hbmod.Batch(m, n, gpu)
    e.reset(perturb=True)
    e.set_policy_mlp(ws, bs)
    e.rollout_policy(5)
    assert np.abs(d.qpos - e.qpos).max() < 1e-4


def test_pipelined_policy_rollout(hbmod, humanoid_model, gpu):
    """With hb_batch_pipeline on, every env segment runs its own obs -> MLP -> mj_step chain on its own stream.  The segments'
    policy kernel is the LDS-free one (it runs beside the other segment's step kernel instead of waiting for its LDS): the closed
    loop is bit-identical for any number of segments, and equal to the unpipelined loop (the LDS kernel: another summation order
    in the last layer) to rounding - a few steps on, before the contact dynamics amplifies the last bits."""
    m = humanoid_model
    n, T = 600, 30
    ws, bs = make_policy(m.nobs, m.nu)
    out, early = [], []
    for segs in (0, 2, 3):
        b = hbmod.Batch(m, n, gpu)
        b.reset(perturb=True)
        b.set_policy_mlp(ws, bs)
        b.pipeline(segs)
        b.rollout_policy(4)
        early.append(b.get_state(hbmod.STATE_INTEGRATION, dtype=np.float64))
        b.rollout_policy(T)
        b.rollout_policy(3)
        out.append(b.get_state(hbmod.STATE_INTEGRATION))
        assert not b.status().any()
    assert np.array_equal(out[1], out[2])
    nq = m.nq
    assert np.abs(early[0][:, 1:1 + nq] - early[1][:, 1:1 + nq]).max() < 1e-4  # (measured 1.3e-5: the bound of the numpy-policy comparison above)
    assert np.isfinite(out[0]).all() and np.isfinite(out[1]).all()


def test_policy_argument_checks(hbmod, humanoid_model, gpu):
    m = humanoid_model
    b = hbmod.Batch(m, 8, gpu)
    with pytest.raises(hbmod.HbError):
        b.rollout_policy(3)  # no policy installed
    with pytest.raises(hbmod.HbError):
        b.set_policy_mlp([np.zeros((m.nobs + 1, m.nu), np.float32)], [np.zeros(m.nu, np.float32)])  # wrong input width


def _fixture_policy():
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "policy_mlp_seed0.npz"))
    return [g["w0"], g["w1"], g["w2"]], [g["b0"], g["b1"], g["b2"]], g


def test_config4_fixture_policy_on_4096_envs(hbmod, humanoid_model, gpu):
    """BASELINE configs[3] as SURVEY.md 8(d) config 4 specifies it: the `torch.manual_seed(0)` default-`nn.Linear` policy
    (tests/golden/policy_mlp_seed0.npz, tools/make_policy_fixture.py) on 4096 envs.  (1) The fixture's own probe: the numpy evaluation of
    the exported weights reproduces the stored torch-checked outputs; (2) hb_policy_eval on 4096 device observations against numpy;
    (3) 40 closed-loop steps of hb_rollout_policy equal the host-driven loop (policy_eval, then step) bit for bit, the pipelined form (other
    policy kernel) to rounding, and both stay within fp32 reach of a loop whose policy runs in fp64 numpy."""
    m = humanoid_model
    ws, bs, g = _fixture_policy()
    assert [w.shape for w in ws] == [(48, 256), (256, 256), (256, 21)] and m.nobs == 48 and m.nu == 21
    assert np.abs(mlp_ref(g["probe"], ws, bs) - g["probe_out"]).max() < 1e-12
    n, T = 4096, 40
    a = hbmod.Batch(m, n, gpu)
    a.reset(perturb=True)
    a.rollout_halton(150)          # falling, first contacts
    a.set_policy_mlp(ws, bs)
    ctrl = a.policy_eval()
    obs, _, _, _ = a.obs(want_reward=False)
    assert ctrl.shape == (n, 21) and np.abs(ctrl - mlp_ref(obs, ws, bs)).max() < 2e-5
    st0 = a.get_state(hbmod.STATE_INTEGRATION)
    finals = []
    for mode in ("launch chain", "pipelined", "host loop"):
        b = hbmod.Batch(m, n, gpu)
        b.set_policy_mlp(ws, bs)
        b.set_state(hbmod.STATE_INTEGRATION, st0)
        if mode == "pipelined":
            b.pipeline(True)
        if mode == "host loop":
            for _ in range(T):
                b.step(b.policy_eval())
        else:
            b.rollout_policy(T)
        finals.append(b.get_state(hbmod.STATE_INTEGRATION))
        assert not (b.status() & (hbmod.WARN_BADQPOS | hbmod.WARN_BADQVEL | hbmod.WARN_BADQACC)).any()
        b.close()
    assert np.array_equal(finals[0], finals[2])
    # the pipelined loop evaluates the policy with the LDS-free kernel (hb_policy_lean_kernel: another summation order), so it equals the
    # launch chain to rounding, amplified over 40 closed-loop steps of a contact-rich humanoid
    nq = m.nq
    dq = np.abs(finals[1][:, 1:1 + nq] - finals[2][:, 1:1 + nq]).max(1)
    # measured: median 2.5e-6, 99 % of the envs within 3e-5, six of 4096 above 1e-3 (an env whose contact set changes a step earlier), worst 0.07
    assert np.median(dq) < 2e-5 and np.quantile(dq, 0.99) < 3e-4 and (dq > 1e-3).sum() < n // 100 and dq.max() < 0.5
    # fp64 policy in the loop for 5 steps: the controls differ by fp32 tanh rounding only
    c = hbmod.Batch(m, n, gpu)
    d = hbmod.Batch(m, n, gpu)
    for x in (c, d):
        x.set_state(hbmod.STATE_INTEGRATION, st0)
    c.set_policy_mlp(ws, bs)
    for _ in range(5):
        c.step(c.policy_eval())
        o, _, _, _ = d.obs(want_reward=False)
        d.step(mlp_ref(o, ws, bs).astype(np.float32))
    dq = np.abs(c.qpos - d.qpos).max(1)
    print("fp64 policy, 5 closed-loop steps: qpos median %.3g, 99 %% %.3g, max %.3g, above 1e-4: %d" % (np.median(dq), np.quantile(dq, 0.99), dq.max(), (dq > 1e-4).sum()))
    # measured: median 1.2e-7, 99 % within 3.6e-7, one env of 4096 at 2.2e-3 (its contact set changes inside the five steps)
    assert np.median(dq) < 1e-6 and np.quantile(dq, 0.99) < 5e-6 and (dq > 1e-4).sum() <= 4 and dq.max() < 0.05
