#!/usr/bin/env python3
"""Throughput of the env adapter (CPUEnv.step analogue): VecEnv.step with host actions and host observations, plain
and with the full sim-to-real layer (noise, delays, pushes, domain randomisation)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
try:  # torch first: its device probe must run before this process opens the GPU through libhb
    import torch
    HAVE_TORCH = torch.cuda.is_available() and torch.zeros(1, device="cuda").is_cuda
except ImportError:
    HAVE_TORCH = False
import humanoid_mujoco_amd as hb
model = os.path.join(ROOT, "humanoid_mujoco_amd", "assets", "humanoid27.hbm")
N, T = 4096, 600
rng = np.random.default_rng(0)
acts = rng.uniform(-1, 1, size=(8, N, 21)).astype(np.float32)
for label, kw in (("plain", {}), ("plain, outputs as views of the transfer record", dict(_views=True)), ("realism + domain randomisation", dict(realism=True, domain_randomization=True, seed=1))):
    views = kw.pop("_views", False)
    env = hb.VecEnv(model, N, 0, randomization_factor=1.0, target_z=10.0, max_time=2.0, **kw)  # success unreachable: episodes run 400 steps
    env.copy_outputs = not views
    env.reset()
    for t in range(20):
        env.step_arrays(acts[t % 8])
    t0 = time.perf_counter()
    done = 0
    for t in range(T):
        obs, rew, term, trunc, info = env.step_arrays(acts[t % 8])
        done += int(info["done"].sum())
    dt = time.perf_counter() - t0
    print("VecEnv.step (%s): %d envs x %d steps in %.3f s -> %.3e env-steps/s (%.0f us/step, host round trip included); %d episodes ended"
          % (label, N, T, dt, N * T / dt, 1e6 * dt / T, done))
    env.close()

# the same loop with the policy on the GPU (torch): actions, observations, rewards and flags never leave the device
if True:
    if not HAVE_TORCH:
        print("torch sees no GPU: device-resident loop skipped")
    else:
        for label, kw in (("plain", {}), ("realism + domain randomisation", dict(realism=True, domain_randomization=True, seed=1))):
            env = hb.VecEnv(model, N, 0, randomization_factor=1.0, target_z=10.0, max_time=2.0, **kw)
            env.reset()
            w = torch.randn(48, 21, device="cuda") * 0.1
            obs = torch.zeros(N, 48, device="cuda")
            for t in range(20):
                obs, rew, term, trunc = env.step_torch(torch.tanh(obs @ w))
            t0 = time.perf_counter()
            for t in range(T):
                obs, rew, term, trunc = env.step_torch(torch.tanh(obs @ w))  # a linear policy stands in for the network
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            print("VecEnv.step_torch (%s): %d envs x %d steps in %.3f s -> %.3e env-steps/s (%.0f us/step; policy, env and physics on the GPU, no host transfer)"
                  % (label, N, T, dt, N * T / dt, 1e6 * dt / T))
            env.close()

        # Two half-size envs, each with its policy on its own torch stream: the two obs -> policy -> step chains are
        # independent, so one half's policy and env kernels run under the other half's physics (a closed loop over ONE
        # batch cannot overlap anything with its own step).
        H = N // 2
        envs = [hb.VecEnv(model, H, 0, randomization_factor=1.0, target_z=10.0, max_time=2.0, seed=s) for s in (0, 1)]
        streams = [torch.cuda.Stream(), torch.cuda.Stream()]
        w = torch.randn(48, 21, device="cuda") * 0.1
        obs = []
        for e in envs:
            e.reset()
            obs.append(torch.zeros(H, 48, device="cuda"))
        torch.cuda.synchronize()
        def sweep(n):
            for t in range(n):
                for k in (0, 1):
                    with torch.cuda.stream(streams[k]):
                        obs[k] = envs[k].step_torch(torch.tanh(obs[k] @ w))[0]
        sweep(20)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sweep(T)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("VecEnv.step_torch, two half-size envs on two torch streams: 2 x %d envs x %d steps in %.3f s -> %.3e env-steps/s (%.0f us per step of both)"
              % (H, T, dt, N * T / dt, 1e6 * dt / T))
        for e in envs:
            e.close()

