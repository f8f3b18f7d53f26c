#!/usr/bin/env python3
"""bench.py — headline benchmark: env-steps/s of the batched mj_step path, 27-DoF humanoid,
4096 envs per GPU, Halton random actions (BASELINE.json configs[1]).

One "step" = one hb_step_dev call = one physics step of every env of this rank's batch, with
state and controls already resident in HBM.  The timed loop runs with hb_batch_pipeline on: each call
enqueues the batch as two env segments on two streams, so the slow tail of one step overlaps the
next (same results, tests/test_gpu_parity.py::test_pipelined_stepping_is_bit_identical).  The roofline
object is measured on a second, unpipelined leg (one 4096-block launch per step, HIP events on the
launch stream) so that it is a per-launch figure comparable with the rocprofv3 kernel trace.  N>1: one process per GPU (torch.distributed over
RCCL for the barrier and the max-over-ranks reduction only — the path has no data collective;
envs shard by env_offset = rank * 4096), weak scaling.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ENVS_PER_GPU = 4096
ALGO_BYTES_PER_ENV_STEP = 748  # fp32 x [read qpos 28 + qvel 27 + warmstart 27 + ctrl 21 + time 1; write 28+27+27+1]
HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s


def shard_range(n_total, world, rank):
    """Contiguous block partition of env indices: env e lives on rank floor(e*G/N)."""
    lo = n_total * rank // world
    hi = n_total * (rank + 1) // world
    return lo, hi


def cpu_baseline(target_seconds=12.0):
    """The oracle (kind "port") on this box's host cores, on a bounded sample of the same workload.
    Shape of simulation/mujoco/sample/testspeed.cc:203-210: shared model, a chunk of envs per thread."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_lib import Oracle
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    o = Oracle()
    # calibration on a sample large enough to load every core, including the contact-rich later steps
    t0 = time.perf_counter()
    n, _, _ = o.rollout_threads(max(cores * 8, 256), 300, cores)
    rate = n / (time.perf_counter() - t0)
    # ~target_seconds of CPU work: the benchmark's batch (or a small multiple of it on many-core hosts), up to 1000 steps
    steps = int(max(50, min(1000, rate * target_seconds / ENVS_PER_GPU)))
    mult = int(min(4, max(1, rate * target_seconds // (ENVS_PER_GPU * steps))))
    n_env = max(cores, (ENVS_PER_GPU * mult // cores) * cores)
    t0 = time.perf_counter()
    n, _, st = o.rollout_threads(n_env, steps, cores)
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": "%d envs x %d steps of the same Halton workload, fp64 oracle/mjstep_oracle.c (CPU restatement, not libmujoco), %d threads, %.1f s"
                      % (n_env, steps, cores, dt),
            "mean_nefc": st["mean_nefc"]}


def load_traffic():
    """HBM bytes per launch from the rocprofv3 PMC passes committed under profiles/ (or None)."""
    p = os.path.join(ROOT, "profiles", "traffic_latest.json")
    if os.path.exists(p):
        try:
            return json.load(open(p))
        except Exception:
            return None
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--envs-per-gpu", type=int, default=ENVS_PER_GPU)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-rollout", action="store_true", help="skip the second (single-launch rollout) measurement, e.g. under rocprofv3")
    ap.add_argument("--no-newton", action="store_true", help="skip the third measurement (same workload with the reference's default solver, Newton)")
    ap.add_argument("--no-pipeline", action="store_true", help="time the unpipelined step API (one launch per step) as `value`")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    backend = os.environ.get("HB_BENCH_BACKEND", "nccl")  # "gloo": rehearsal of the N>1 path on a box with fewer GPUs than ranks
    red_dev = "cpu"
    if world > 1:
        import torch
        import torch.distributed as dist_
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist_.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
            red_dev = "cuda"
        else:
            dist_.init_process_group(backend=backend)
        dist = dist_

    import humanoid_mujoco_amd as hb
    model = hb.Model.load(os.path.join(ROOT, "humanoid_mujoco_amd", "assets", "humanoid27.hbm"))
    n_env = args.envs_per_gpu
    lo, hi = shard_range(n_env * world, world, rank)
    assert hi - lo == n_env
    device = local_rank
    if backend != "nccl":
        import ctypes
        ndev = ctypes.c_int(0)
        ctypes.CDLL("libamdhip64.so").hipGetDeviceCount(ctypes.byref(ndev))
        device = local_rank % max(1, ndev.value)
    batch = hb.Batch(model, n_env, device)  # raises without a GPU: no CPU fallback
    K, W = args.steps, args.warmup
    nu = model.nu
    # controls for every timed step live in HBM, generated there (testspeed.cc:64-80)
    ctrl = batch.dev_alloc((K + W) * n_env * nu * 4)
    batch.halton_ctrl_dev(K + W, 0, lo, ctrl)
    batch.reset(perturb=True, env_offset=lo)
    stride = n_env * nu * 4

    def barrier():
        batch.sync()  # the batch's own HIP stream(s)
        if dist is not None:
            dist.barrier()
            if backend == "nccl":
                import torch
                torch.cuda.synchronize()  # an RCCL barrier is enqueued on a stream: wait for it on the host as well

    pipelined = not args.no_pipeline
    batch.pipeline(pipelined)
    for t in range(W):
        batch.step_dev(ctrl + t * stride)
    barrier()
    batch.timer_start()
    t0 = time.perf_counter()
    for t in range(W, W + K):
        batch.step_dev(ctrl + t * stride)
    region_ms = batch.timer_stop()  # HIP events around the whole timed region on the batch's stream (joins the segments); also drains it
    batch.sync()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        import torch
        tt = torch.tensor([elapsed], device=red_dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        dist.barrier()
    status = batch.status()
    nc, ne, ni = batch.counts()

    # Roofline leg: the dominant kernel as ONE launch per step (all 4096 envs), unpipelined, continuing from
    # the state the timed loop left; HIP events around KR back-to-back launches on the launch stream.
    batch.pipeline(False)
    KR = min(K, 200)
    for t in range(5):
        batch.step_dev(ctrl + (W + t) * stride)
    batch.sync()
    batch.timer_start()
    for t in range(KR):
        batch.step_dev(ctrl + (W + t) * stride)
    launch_us = 1e3 * batch.timer_stop() / KR

    # Second measurement, reported beside `value`: the same K steps as ONE hb_rollout_dev launch (state
    # resident on chip, each env advancing through its own K steps without a per-step batch barrier) —
    # the shape of the reference's C++ harness simulation/mujoco/sample/testspeed.cc:84-103,203-210.
    elapsed_rollout = None
    if not args.no_rollout:
        batch.reset(perturb=True, env_offset=lo)
        batch.rollout_dev(ctrl, W)
        barrier()
        t1 = time.perf_counter()
        batch.rollout_dev(ctrl + W * stride, K)
        batch.sync()
        elapsed_rollout = time.perf_counter() - t1
    if dist is not None and elapsed_rollout is not None:
        import torch
        tt = torch.tensor([elapsed_rollout], device=red_dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed_rollout = float(tt.item())
        dist.barrier()

    # Third measurement (N = 1 only), reported beside `value`: the same workload and step API with the solver the
    # reference's own model file selects (none given = mjOption's default, Newton / 100 iterations) instead of the
    # benchmark configuration's PGS / 50.
    newton = None
    if world == 1 and not args.no_newton:
        nm = hb.Model.load(os.path.join(ROOT, "humanoid_mujoco_amd", "assets", "humanoid27.hbm"))
        nm.set_opt(solver=2, iterations=100)
        nb = hb.Batch(nm, n_env, device)
        nb.reset(perturb=True, env_offset=lo)
        nb.pipeline(pipelined)
        for t in range(W):
            nb.step_dev(ctrl + t * stride)
        nb.sync()
        t2 = time.perf_counter()
        for t in range(W, W + K):
            nb.step_dev(ctrl + t * stride)
        nb.sync()
        el = time.perf_counter() - t2
        _, _, nit = nb.counts()
        newton = {"value": n_env * K / el, "unit": "env-steps/s", "ms_per_step": 1e3 * el / K, "mean_iterations": float(nit.mean()),
                  "envs_with_warnings": int((nb.status() != 0).sum()),
                  "what": "same workload and step API, solver = Newton (mjOption default: what the reference's humanoid.xml runs), 100 iterations max, tolerance 1e-8"}
        nb.close()

    if rank == 0:
        value = n_env * world * K / elapsed
        achieved = ALGO_BYTES_PER_ENV_STEP * n_env / (launch_us * 1e-6) / 1e9
        traffic = load_traffic()
        out = {
            "metric": "env-steps/sec (whole node), 27-DoF humanoid, 4096 envs/GPU",
            "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": 1e3 * elapsed / K, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "configs[1]: %d parallel humanoid envs per GPU, Halton random actions, fp32, PGS<=50 iters tol 1e-8, dt 0.005, one mj_step of every env per hb_step_dev call%s"
                                   % (n_env, " (pipelined: 2 env segments on 2 streams)" if pipelined else ""),
                       "model": "27-DoF humanoid (assets/humanoid27.hbm)", "envs_per_gpu": n_env, "global_envs": n_env * world,
                       "sharding": "env blocks by rank, no collective"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": (traffic or {}).get("hbm_bytes_per_launch"),
                         "kernel": "hb_step_kernel", "avg_launch_us": launch_us, "launches": KR,
                         "launch_shape": "%d blocks x 64 lanes, one env per block, unpipelined leg" % n_env,
                         "timed_region_ms_per_step": region_ms / K,
                         "algorithmic_bytes_per_launch": ALGO_BYTES_PER_ENV_STEP * n_env,
                         "note": "path is latency/VALU bound, not HBM bound (SURVEY.md §8d); see DESIGN.md"},
            "state_check": {"envs_with_warnings": int((status != 0).sum()), "mean_ncon": float(nc.mean()), "mean_nefc": float(ne.mean()),
                            "mean_pgs_iters": float(ni.mean())},
        }
        if traffic:
            out["roofline"]["traffic_source"] = traffic.get("source")
            # what actually bounds the kernel (from the same PMC passes; informational): share of the chip's fp32 VALU
            # lane-op rate the kernel issues, and the share of SIMD cycles with the MFMA pipe busy
            for k in ("valu_issue_frac_of_peak", "mfma_busy_frac"):
                if k in traffic:
                    out["roofline"][k] = traffic[k]
        if elapsed_rollout is not None:
            out["rollout"] = {"value": n_env * world * K / elapsed_rollout, "unit": "env-steps/s", "ms_per_step": 1e3 * elapsed_rollout / K,
                              "what": "same K steps as one hb_rollout_dev launch per GPU (no per-step batch barrier; testspeed.cc shape)"}
        if newton is not None:
            out["newton"] = newton
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    batch.dev_free(ctrl)
    batch.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
