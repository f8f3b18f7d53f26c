#!/usr/bin/env python3
"""tests/golden/policy_mlp_seed0.npz: the policy of BASELINE configs[3] exactly as SURVEY.md §8(d) config 4 specifies it -
`torch.manual_seed(0)`, default `nn.Linear` initialisation, 48 -> 256 -> 256 -> 21 (sizes from the reference's
simulation/hyperparam_config.py:21-27 / rl/train.py:163-167), tanh on every layer - exported as [in, out] float32 weights and biases,
plus 64 probe observations with the fp64 numpy output of the network on them (a fixture the tests can use without importing torch).

    python tools/make_policy_fixture.py
"""
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden", "policy_mlp_seed0.npz")
SIZES = (48, 256, 256, 21)


def main():
    torch.manual_seed(0)
    layers = [torch.nn.Linear(a, b) for a, b in zip(SIZES[:-1], SIZES[1:])]
    ws = [l.weight.detach().numpy().T.copy().astype(np.float32) for l in layers]
    bs = [l.bias.detach().numpy().copy().astype(np.float32) for l in layers]
    rng = np.random.default_rng(0)
    probe = rng.uniform(-2, 2, size=(64, SIZES[0])).astype(np.float32)
    x = probe.astype(np.float64)
    for w, b in zip(ws, bs):
        x = np.tanh(x @ w.astype(np.float64) + b.astype(np.float64))
    with torch.no_grad():
        y = torch.from_numpy(probe)
        for l in layers:
            y = torch.tanh(l(y))
    assert np.abs(y.numpy() - x).max() < 1e-5  # the export is the torch network
    np.savez_compressed(OUT, w0=ws[0], w1=ws[1], w2=ws[2], b0=bs[0], b1=bs[1], b2=bs[2], probe=probe, probe_out=x,
                        label=np.array("torch.manual_seed(0), default nn.Linear init, 48-256-256-21, tanh (SURVEY.md 8d config 4); torch " + torch.__version__))
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
