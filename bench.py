#!/usr/bin/env python3
"""bench.py — headline benchmark: env-steps/s of the batched mj_step path, 27-DoF humanoid,
4096 envs per GPU, Halton random actions (BASELINE.json configs[1]).

One "step" = one hb_step_dev call = one physics step of every env of this rank's batch, with
state and controls already resident in HBM.  Before anything is timed every env is rolled through
PREROLL (600) untimed steps of the same workload, so that a short timed window (the driver runs
--steps 20) samples the steady regime — the humanoids on the floor, about ten constraint rows per
env — and not the contact-free fall that follows the reset.  The CPU leg times the same window.
The timed loop runs with hb_batch_pipeline on.  Its K hb_step_dev calls are enqueued back to back, nothing else in between - the
reference's own loop, mj_step after mj_step (simulation/mujoco/sample/testspeed.cc:93-96) - and for this workload the library runs such
calls as launches of up to 256 steps of the two-envs-per-wave kernel, step t on the controls of call t (include/hb.h: hb_step_dev;
same states bit for bit, tests/test_gpu_fold.py): no env waits for the batch's slowest one between steps.  The roofline object is then
those launches themselves: they run one after the other on the batch's stream, so the HIP events around the timed region divided by
their number (hb_batch_step_launches) is the launch duration the rocprofv3 kernel trace shows.  Where the library does not fold (fewer
envs, other models, --duo 0, --no-pipeline) each call enqueues the batch as three env segments on three streams, so that the slow tail
of one step overlaps the next (tests/test_gpu_parity.py::test_pipelined_stepping_is_bit_identical), and the roofline object is measured
on a second, unpipelined leg (one launch per step, HIP events on the launch stream) - which is always run and reported as
roofline.single_step.

Launching.  `python3 bench.py --gpus N`: with WORLD_SIZE unset the parent process — before it makes
any HIP or torch call — starts N children of itself, one per GPU (RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_ADDR / MASTER_PORT set, the thread-per-device shape of simulation/mujoco/sample/testspeed.cc:
165-183,203-210 as processes), relays rank 0's JSON line and exits non-zero if any child failed.
Under torchrun (`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`) the
environment already names the rank and the process just runs it.  Ranks use torch.distributed (RCCL)
for the barrier and the max-over-ranks reduction only — the path has no data collective; envs shard
by env_offset = rank * 4096, weak scaling.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ENVS_PER_GPU = 4096
PREROLL = 600                  # untimed steps from the reset before any measurement (the fall takes ~300)
ALGO_BYTES_PER_ENV_STEP = 748  # fp32 x [read qpos 28 + qvel 27 + warmstart 27 + ctrl 21 + time 1; write 28+27+27+1]
HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s


def shard_range(n_total, world, rank):
    """Contiguous block partition of env indices: env e lives on rank floor(e*G/N)."""
    lo = n_total * rank // world
    hi = n_total * (rank + 1) // world
    return lo, hi


def free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn_ranks(n, argv, script=None, timeout=None):
    """Start n child processes of this script, one rank each, and relay rank 0's stdout.

    Called before the parent has touched HIP or torch (a process that has initialised the GPU must not
    exec or fork GPU work).  Returns the exit code: 0 only if every child exited 0."""
    script = script or os.path.abspath(__file__)
    port = os.environ.get("MASTER_PORT") or str(free_port())
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": port, "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")})
        out = subprocess.PIPE if r == 0 else sys.stderr
        procs.append(subprocess.Popen([sys.executable, script] + list(argv), env=env, stdout=out))
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()  # rank 0's pipe is drained while everybody runs (it prints one line at the very end)
    rc = 0
    t_end = None if timeout is None else time.time() + timeout
    while True:
        codes = [p.poll() for p in procs]
        if all(c is not None for c in codes):
            break
        if any(c not in (None, 0) for c in codes) or (t_end is not None and time.time() > t_end):
            # one rank failed (or the run timed out): the others would wait for it at the rendezvous; stop exactly those processes
            if not any(c not in (None, 0) for c in codes):
                rc = 124
            for p in procs:
                if p.poll() is None:
                    p.kill()
            for p in procs:
                p.wait()
            break
        time.sleep(0.05)
    own = [p.returncode for p in procs if p.returncode > 0]  # exit codes of ranks that failed by themselves (killed ones are negative)
    for r, p in enumerate(procs):
        if p.returncode != 0:
            sys.stderr.write("bench.py: rank %d exited with code %d\n" % (r, p.returncode))
            rc = rc or (own[0] if own else 1)
    reader.join(timeout=5)
    sys.stdout.write(b"".join(c for c in chunks if c).decode(errors="replace"))
    sys.stdout.flush()
    return rc


def probe_libmujoco():
    """Opportunistic probe promised by BASELINE.md §3 / SURVEY.md §8(d): is a MuJoCo already installed on this box?
    Never installs or fetches anything.  Absent -> {"libmujoco": "absent"}; present -> version, and, when a model file
    is reachable (HB_MUJOCO_XML, or the reference tree), real mj_step timing and one-step deltas on the golden states."""
    import importlib.util
    info = {"libmujoco": "absent"}
    spec = None
    try:
        spec = importlib.util.find_spec("mujoco")
    except Exception:
        spec = None
    try:
        out = subprocess.run(["ldconfig", "-p"], capture_output=True, text=True, timeout=10).stdout
        libs = [ln.split("=>")[-1].strip() for ln in out.splitlines() if "mujoco" in ln.lower()]
    except Exception:
        libs = []
    if libs:
        info = {"libmujoco": "shared library present", "libs": libs[:4]}
    if spec is None:
        return info
    try:
        import mujoco  # noqa: only reached when the package is already installed
        import numpy as np
        info = {"libmujoco": "python package present", "version": getattr(mujoco, "__version__", "?")}
        xml = os.environ.get("HB_MUJOCO_XML") or "/root/reference/simulation/mujoco/model/humanoid/humanoid.xml"
        if not os.path.exists(xml):
            info["note"] = "no MJCF file reachable (set HB_MUJOCO_XML): timing and deltas skipped"
            return info
        m = mujoco.MjModel.from_xml_path(xml)
        m.opt.solver, m.opt.iterations = 0, 50  # the benchmark configuration: PGS / 50 (mjSOL_PGS = 0)
        d = mujoco.MjData(m)
        g = np.load(os.path.join(ROOT, "tests", "golden", "humanoid27_steps.npz"))
        dq = dv = 0.0
        for k in range(len(g["env"])):
            mujoco.mj_resetData(m, d)
            d.time = g["time"][k]; d.qpos[:] = g["qpos"][k]; d.qvel[:] = g["qvel"][k]; d.qacc_warmstart[:] = g["warm"][k]; d.ctrl[:] = g["ctrl"][k]
            mujoco.mj_step(m, d)
            dq = max(dq, float(np.abs(d.qpos - g["qpos1"][k]).max()))
            dv = max(dv, float(np.abs(d.qvel - g["qvel1"][k]).max()))
        info["one_step_delta_vs_golden"] = {"max_abs_dqpos": dq, "max_abs_dqvel": dv, "states": int(len(g["env"]))}
        mujoco.mj_resetData(m, d)
        t0 = time.perf_counter()
        n = 0
        while time.perf_counter() - t0 < 3.0:
            for _ in range(200):
                mujoco.mj_step(m, d)
            n += 200
        info["mj_step_per_s_one_core"] = n / (time.perf_counter() - t0)
    except Exception as ex:  # a probe must never fail the benchmark
        info["error"] = repr(ex)[:200]
    return info


def cpu_baseline(target_seconds=12.0):
    """The oracle (kind "port") on this box's host cores, on a bounded sample of the same workload AND the same
    window as the GPU leg: every env is rolled through PREROLL untimed steps, then timed.
    Shape of simulation/mujoco/sample/testspeed.cc:203-210: shared model, a chunk of envs per thread."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_lib import Oracle
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    o = Oracle()
    # calibration: a few envs per core over the start of the window
    _, _, _, rate = o.rollout_window(cores * 2, PREROLL, 100, cores)
    # ~target_seconds of CPU work in all (pre-roll included): envs x (PREROLL + steps)
    steps = 200
    n_env = int(rate * target_seconds / (PREROLL + steps))
    n_env = max(cores, min(ENVS_PER_GPU, (n_env // cores) * cores))
    t0 = time.perf_counter()
    n, _, st, rate = o.rollout_window(n_env, PREROLL, steps, cores)
    dt = time.perf_counter() - t0
    return {"value": rate, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": "%d envs x %d timed steps after %d untimed pre-roll steps each (the GPU leg's window) of the same Halton workload, fp64 oracle/mjstep_oracle.c "
                      "(CPU restatement, not libmujoco), %d threads, %.1f s in all; value = sum over threads of timed env-steps / that thread's timed seconds"
                      % (n_env, steps, PREROLL, cores, dt),
            "mean_nefc": st["mean_nefc"]}


def load_traffic(kernel=None):
    """HBM bytes per launch from the rocprofv3 PMC passes committed under profiles/ (or None).  The file's top level is the one-env-per-wave
    single-step kernel's (hb_step_h27_kernel, as in rounds 1-3); "by_kernel" holds the passes of the other kernels the timed loop can run."""
    p = os.path.join(ROOT, "profiles", "traffic_latest.json")
    if os.path.exists(p):
        try:
            d = json.load(open(p))
            if kernel and kernel in d.get("by_kernel", {}):
                return d["by_kernel"][kernel]
            return None if (kernel and d.get("kernel", "hb_step_h27_kernel") != kernel) else d
        except Exception:
            return None
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--envs-per-gpu", type=int, default=ENVS_PER_GPU)
    ap.add_argument("--preroll", type=int, default=PREROLL, help="untimed steps from the reset before any measurement")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-rollout", action="store_true", help="skip the second (single-launch rollout) measurement, e.g. under rocprofv3")
    ap.add_argument("--no-newton", action="store_true", help="skip the third measurement (same workload with the reference's default solver, Newton)")
    ap.add_argument("--no-team", action="store_true", help="skip the fourth measurement (the reference's own robot, assets/team_robot.hbm)")
    ap.add_argument("--no-pipeline", action="store_true", help="time the unpipelined step API (one launch per step) as `value`")
    ap.add_argument("--duo", type=int, default=None, choices=(0, 1, 2), help="two envs per wave (hb_batch_tune HB_TUNE_DUO): 1 where it pays (default), 0 never, 2 always - the "
                    "counter passes of tools/gpu_round.sh hold the unpipelined launches to one kernel with it")
    ap.add_argument("--fold", type=int, default=None, help="most hb_step_dev calls the library may run as one launch (hb_batch_tune HB_TUNE_FOLD; default: the library's, 256); "
                    "1 = one launch per call on three env segments, the timed loop of rounds 1-3")
    ap.add_argument("--dry-run", action="store_true", help="launch / rendezvous / reduction only, no GPU work (CPU rehearsal of the N>1 path with HB_BENCH_BACKEND=gloo)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # one process per GPU, started before this process touches HIP or torch
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and rank == 0:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d: the launcher's world size is what runs\n" % (args.gpus, world))
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

    def reduce_max(x):
        if dist is None:
            return x
        import torch
        tt = torch.tensor([x], device=red_dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item())

    def gather(obj):
        """every rank's value on rank 0 (a list in rank order); None on the other ranks"""
        if dist is None:
            return [obj]
        out = [None] * world if rank == 0 else None
        dist.gather_object(obj, out, dst=0)
        return out

    K, W, PRE = args.steps, args.warmup, max(0, args.preroll)
    n_env = args.envs_per_gpu
    lo, hi = shard_range(n_env * world, world, rank)
    assert hi - lo == n_env

    if args.dry_run:
        if dist is not None:
            dist.barrier()
        own = 1e-3 * (1 + rank)
        elapsed = reduce_max(own)
        ranks = gather({"rank": rank, "elapsed_s": own, "device": "dry-run", "segments": 0, "envs": [lo, hi]})
        if dist is not None:
            dist.barrier()
        if rank == 0:
            print(json.dumps({"metric": "env-steps/sec (whole node), 27-DoF humanoid, 4096 envs/GPU", "dry_run": True, "value": None, "n_gpus": world,
                              "ranks_seen": dist.get_world_size() if dist is not None else 1, "ranks": ranks,
                              "steps": K, "warmup": W, "max_over_ranks_s": elapsed, "shard_of_last_rank": list(shard_range(n_env * world, world, world - 1))}), flush=True)
        if dist is not None:
            dist.destroy_process_group()
        return

    import humanoid_mujoco_amd as hb
    model = hb.Model.load(os.path.join(ROOT, "humanoid_mujoco_amd", "assets", "humanoid27.hbm"))
    device = local_rank
    if backend != "nccl":
        import ctypes
        ndev = ctypes.c_int(0)
        ctypes.CDLL("libamdhip64.so").hipGetDeviceCount(ctypes.byref(ndev))
        device = local_rank % max(1, ndev.value)
    batch = hb.Batch(model, n_env, device)  # raises without a GPU: no CPU fallback
    duo_default = args.duo if args.duo is not None else int(os.environ.get("HB_DUO", "1"))
    batch.tune(duo=duo_default)
    if args.fold is not None:
        batch.tune(fold=args.fold)
    nu = model.nu
    # controls for every timed step live in HBM, generated there (testspeed.cc:64-80); step indices continue behind the pre-roll
    ctrl = batch.dev_alloc((K + W) * n_env * nu * 4)
    batch.halton_ctrl_dev(K + W, PRE, lo, ctrl)
    stride = n_env * nu * 4

    def preroll(b):
        b.reset(perturb=True, env_offset=lo)
        if PRE > 0:
            b.rollout_halton(PRE, 0, lo)  # untimed: every env through its fall, controls t = 0 .. PRE-1
        b.sync()

    def barrier():
        batch.sync()  # the batch's own HIP stream(s)
        if dist is not None:
            dist.barrier()
            if backend == "nccl":
                import torch
                torch.cuda.synchronize()  # an RCCL barrier is enqueued on a stream: wait for it on the host as well

    preroll(batch)
    pipelined = not args.no_pipeline
    batch.pipeline(pipelined)
    nseg = batch.segments
    for t in range(W):
        batch.step_dev(ctrl + t * stride)
    barrier()
    launches0 = batch.step_launches()
    batch.timer_start()
    t0 = time.perf_counter()
    for t in range(W, W + K):
        batch.step_dev(ctrl + t * stride)
    region_ms = batch.timer_stop()  # HIP events around the whole timed region on the batch's stream (joins the segments); also drains it
    batch.sync()
    elapsed_own = elapsed = time.perf_counter() - t0
    timed_kernel = batch.last_kernel()  # what the library says the timed loop's launches ran (include/hb.h: hb_last_kernel)
    # the library runs step calls enqueued back to back as one launch of up to 256 steps where that pays (include/hb.h: hb_step_dev)
    timed_launches = batch.step_launches() - launches0
    folded = 0 < timed_launches < K
    if dist is not None:
        elapsed = reduce_max(elapsed)
        dist.barrier()
    ranks = gather({"rank": rank, "elapsed_s": elapsed_own, "value": n_env * K / elapsed_own, "device": batch.device_name(), "segments": 1 if folded else nseg, "envs": [lo, hi],
                    "kernel": timed_kernel, "launches": timed_launches})
    status = batch.status()
    nc, ne, ni = batch.counts()

    # Roofline leg: the dominant kernel as ONE launch per step (all 4096 envs), unpipelined, continuing from
    # the state the timed loop left; HIP events around KR back-to-back launches on the launch stream.
    # (When the timed loop's calls were folded into multi-step launches, those launches ARE the roofline object's - they run back to back
    # on the batch's stream, inside the HIP events of the timed region - and this leg is reported beside them as `single_step`.)
    batch.pipeline(False)
    # (the library picks the kernel of a step call by the launch's shape: hold it to the one the timed loop ran)
    batch.tune(duo=2 if "duo" in timed_kernel else 0)
    KR = min(K, 200)
    for t in range(5):
        batch.step_dev(ctrl + (W + min(t, K - 1)) * stride)
    batch.sync()
    batch.timer_start()
    for t in range(KR):
        batch.step_dev(ctrl + (W + t) * stride)
    launch_us = 1e3 * batch.timer_stop() / KR
    roofline_kernel = batch.last_kernel()
    batch.tune(duo=duo_default)

    # Second measurement, reported beside `value`: the same K steps as ONE hb_rollout_dev launch (state
    # resident on chip, each env advancing through its own K steps without a per-step batch barrier) —
    # the shape of the reference's C++ harness simulation/mujoco/sample/testspeed.cc:84-103,203-210.
    elapsed_rollout = None
    if not args.no_rollout:
        preroll(batch)
        batch.rollout_dev(ctrl, W)
        barrier()
        t1 = time.perf_counter()
        batch.rollout_dev(ctrl + W * stride, K)
        batch.sync()
        elapsed_rollout = time.perf_counter() - t1
        rollout_kernel = batch.last_kernel()
    if dist is not None and elapsed_rollout is not None:
        elapsed_rollout = reduce_max(elapsed_rollout)
        dist.barrier()

    # Third measurement (N = 1 only), reported beside `value`: the same workload and step API with the solver the
    # reference's own model file selects (none given = mjOption's default, Newton / 100 iterations) instead of the
    # benchmark configuration's PGS / 50.
    newton = None
    if world == 1 and not args.no_newton:
        nm = hb.Model.load(os.path.join(ROOT, "humanoid_mujoco_amd", "assets", "humanoid27.hbm"))
        nm.set_opt(solver=2, iterations=100)
        nb = hb.Batch(nm, n_env, device)
        preroll(nb)
        nb.pipeline(pipelined)
        for t in range(W):
            nb.step_dev(ctrl + t * stride)
        nb.sync()
        t2 = time.perf_counter()
        for t in range(W, W + K):
            nb.step_dev(ctrl + t * stride)
        nb.sync()
        el = time.perf_counter() - t2
        _, nne, nit = nb.counts()
        newton = {"value": n_env * K / el, "unit": "env-steps/s", "ms_per_step": 1e3 * el / K, "mean_iterations": float(nit.mean()), "mean_nefc": float(nne.mean()),
                  "envs_with_warnings": int((nb.status() != 0).sum()),
                  "what": "same workload, window and step API, solver = Newton (mjOption default: what the reference's humanoid.xml runs), 100 iterations max, tolerance 1e-8"}
        nb.close()

    # Fourth measurement (N = 1 only): the reference's OWN robot (simulation/assets/world.xml: mesh hulls, condim 6, height-field floor,
    # Newton, dt 0.002), same step API, from its standing reset after 200 settling steps, small random motor commands.
    team = None
    if world == 1 and not args.no_team:
        import numpy as np
        tm = hb.Model.load(os.path.join(ROOT, "humanoid_mujoco_amd", "assets", "team_robot.hbm"))
        tb = hb.Batch(tm, n_env, device)
        tb.reset(keyframe=1, perturb=True)
        KT = min(K, 300)
        tctrl_h = (0.3 * np.random.default_rng(0).uniform(-1, 1, (W + KT, n_env, tm.nu))).astype(np.float32)
        tctrl = tb.dev_alloc(tctrl_h.nbytes)
        tb.to_dev(tctrl, tctrl_h)
        tstride = n_env * tm.nu * 4
        tb.rollout_halton(200, 0, 0)
        tb.pipeline(pipelined)
        for t in range(W):
            tb.step_dev(tctrl + t * tstride)
        tb.sync()
        t2 = time.perf_counter()
        for t in range(W, W + KT):
            tb.step_dev(tctrl + t * tstride)
        tb.sync()
        el = time.perf_counter() - t2
        tnc, tne, tni = tb.counts()
        tnw, tns = tb.collision_counts()
        team = {"value": n_env * KT / el, "unit": "env-steps/s", "ms_per_step": 1e3 * el / KT, "steps": KT, "mean_ncon": float(tnc.mean()), "mean_nefc": float(tne.mean()),
                "mean_newton_iterations": float(tni.mean()), "portal_searches_per_env": float(tns.mean()), "envs_with_warnings": int((tb.status() != 0).sum()),
                "what": "the reference's own robot (assets/team_robot.hbm: 18 dofs, 9 mesh hulls, condim 6, 8 x 8 height field, Newton), %d envs, staged step "
                        "(pose, narrowphase, step kernels), same step API and pipelining as `value`" % n_env}
        team["kernel"] = tb.last_kernel()
        # roofline of the robot's dominant kernel, the narrowphase launch (48 % of its step): per env-step it reads the geoms' world poses
        # (40 B x 13 geoms) and its work items (16 B each) and writes one result record per item (64 B); times and lane utilisation from
        # the committed rocprofv3 passes (tools/gpu_team_counters.sh), not from this run
        try:
            tc = json.load(open(os.path.join(ROOT, "profiles", "team_counters_latest.json")))
            # (the committed passes run unpipelined launches, which take the two-envs-per-wave form of the same kernel body: hb_narrow2_kernel)
            nkname = "hb_narrow_kernel" if tc.get("hb_narrow_kernel") else "hb_narrow2_kernel"
            nk = tc.get(nkname) or {}
            algo = n_env * (40 * tm.ngeom + (16 + 64) * float(tnw.mean()))
            if nk.get("avg_us"):
                team["roofline"] = {"bound": "hbm", "kernel": nkname, "achieved": algo / (nk["avg_us"] * 1e-6) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                    "frac": algo / (nk["avg_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS, "avg_launch_us": nk["avg_us"], "algorithmic_bytes_per_launch": algo,
                                    "traffic": (nk.get("hbm_bytes_per_launch") or {}).get("fetch_x2_gfx950", 0) + (nk.get("hbm_bytes_per_launch") or {}).get("write", 0) or None,
                                    "active_lanes_per_valu_instruction": nk.get("active_lanes_per_valu_instruction"),
                                    "wait_inst_frac_of_wave_cycles": nk.get("sq_wait_inst_any_frac_of_wave_cycles"),
                                    "note": "four lanes per portal search, sixteen searches per wave; the launch lasts as long as its longest search - a chain of dependent fp64 operations and hull-table loads (DESIGN.md 3.6)",
                                    "source": "committed profile, not this run: profiles/team_counters_latest.json (tools/gpu_team_counters.sh)"}
        except Exception:
            pass
        tb.dev_free(tctrl)
        tb.close()
        # the same at the batch size that fills the chip under the narrowphase's latency chain (the robot against the batch size)
        NB = 32768
        tb = hb.Batch(tm, NB, device)
        tb.reset(keyframe=1, perturb=True)
        tb.rollout_halton(200, 0, 0)
        tb.pipeline(pipelined)
        KB = 40
        tctrl = tb.dev_alloc(KB * NB * tm.nu * 4)
        tb.halton_ctrl_dev(KB, 200, 0, tctrl)
        for t in range(5):
            tb.step_dev(tctrl + t * NB * tm.nu * 4)
        tb.sync()
        t2 = time.perf_counter()
        for t in range(KB):
            tb.step_dev(tctrl + t * NB * tm.nu * 4)
        tb.sync()
        el = time.perf_counter() - t2
        team["at_32768_envs"] = {"value": NB * KB / el, "unit": "env-steps/s", "ms_per_step": 1e3 * el / KB, "steps": KB, "envs_with_warnings": int((tb.status() != 0).sum()),
                                 "what": "same model and step API, 32768 envs on the GPU, on-device Halton controls"}
        tb.dev_free(tctrl)
        tb.close()

    if rank == 0:
        value = n_env * world * K / elapsed
        single_step = {"kernel": roofline_kernel, "avg_launch_us": launch_us, "launches": KR,
                       "achieved": ALGO_BYTES_PER_ENV_STEP * n_env / (launch_us * 1e-6) / 1e9,
                       "launch_shape": ("%d blocks x 64 lanes, two envs per block, one step per launch, unpipelined" % ((n_env + 1) // 2)) if "duo" in roofline_kernel
                                       else "%d blocks x 64 lanes, one env per block, one step per launch, unpipelined" % n_env}
        if folded:
            steps_per_launch = K / timed_launches
            roofline_kernel = timed_kernel
            launch_us = 1e3 * region_ms / timed_launches
            algo_per_launch = ALGO_BYTES_PER_ENV_STEP * n_env * steps_per_launch
            KR = timed_launches
            launch_shape = "%d blocks x 64 lanes, two envs per block, %s steps per launch (the timed loop's hb_step_dev calls, folded), one launch after the other on the batch's stream" % (
                (n_env + 1) // 2, ("%d" % steps_per_launch) if steps_per_launch == int(steps_per_launch) else ("%.1f" % steps_per_launch))
        else:
            steps_per_launch = 1
            algo_per_launch = ALGO_BYTES_PER_ENV_STEP * n_env
            launch_shape = single_step["launch_shape"] + " leg"
        achieved = algo_per_launch / (launch_us * 1e-6) / 1e9
        traffic = load_traffic(roofline_kernel)
        out = {
            "metric": "env-steps/sec (whole node), 27-DoF humanoid, 4096 envs/GPU",
            "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": 1e3 * elapsed / K, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "ranks_seen": dist.get_world_size() if dist is not None else 1, "ranks": ranks,
            "config": {"workload": "configs[1]: %d parallel humanoid envs per GPU, Halton random actions, fp32, PGS<=50 iters tol 1e-8, dt 0.005, one mj_step of every env per "
                                   "hb_step_dev call%s; every env pre-rolled %d untimed steps from the perturbed reset (steady regime: fallen humanoids, ~10 constraint rows), "
                                   "then %d warm-up and %d timed steps"
                                   % (n_env, (" (the %d calls enqueued back to back, which the library ran as %d launches of the two-envs-per-wave kernel: include/hb.h hb_step_dev)" % (K, timed_launches)) if folded
                                      else (" (pipelined: %d env segments on %d streams)" % (nseg, nseg)) if pipelined else "", PRE, W, K),
                       "model": "27-DoF humanoid (assets/humanoid27.hbm)", "envs_per_gpu": n_env, "global_envs": n_env * world, "preroll_steps": PRE,
                       "sharding": "env blocks by rank, no collective"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": ((traffic or {}).get("hbm_bytes_per_step") * steps_per_launch) if (traffic or {}).get("hbm_bytes_per_step") else (traffic or {}).get("hbm_bytes_per_launch"),
                         "kernel": roofline_kernel, "avg_launch_us": launch_us, "launches": KR, "steps_per_launch": steps_per_launch,
                         "launch_shape": launch_shape,
                         # the launches the TIMED loop makes
                         "timed_shape": ({"kernel": timed_kernel, "launches": timed_launches, "steps_per_launch": steps_per_launch, "segments": 1, "ms_per_step": 1e3 * elapsed / K,
                                          "what": "the roofline object's own launches: HIP events around the timed region / launches"} if folded else
                                         {"kernel": timed_kernel, "launches": timed_launches, "segments": nseg, "envs_per_segment": [n_env * (c + 1) // nseg - n_env * c // nseg for c in range(nseg)],
                                          "ms_per_step": 1e3 * elapsed / K,
                                          "segment_launch_us_from_profile": (traffic or {}).get("segment_launch_us")}),
                         "single_step": single_step,
                         "timed_region_ms_per_step": region_ms / K,
                         "algorithmic_bytes_per_launch": algo_per_launch,
                         "note": "path is latency/VALU bound, not HBM bound (SURVEY.md §8d); see DESIGN.md"},
            "state_check": {"envs_with_warnings": int((status != 0).sum()), "mean_ncon": float(nc.mean()), "mean_nefc": float(ne.mean()),
                            "mean_pgs_iters": float(ni.mean())},
        }
        if traffic:
            out["roofline"]["traffic_source"] = "committed profile, not this run: " + str(traffic.get("source"))
            # what actually bounds the kernel (from the same PMC passes; informational): share of the chip's fp32 VALU
            # lane-op rate the kernel issues, and the share of SIMD cycles with the MFMA pipe busy
            for k in ("valu_issue_frac_of_peak", "mfma_busy_frac", "valu"):
                if k in traffic:
                    out["roofline"][k] = traffic[k]
            # (the profile's own launch time: `achieved` of that run, for comparison with this one.  Multi-step launches: the committed traces
            # are of the default command (250 steps per launch) and of the driver's (`--steps 20`: one launch of 20 steps, which ends on
            # its slowest wave): the one whose launches are shaped like this run's)
            prof = traffic.get("driver_cmd") if (traffic.get("driver_cmd") and steps_per_launch <= 2 * traffic["driver_cmd"].get("steps_per_launch", 20)) else traffic
            if "avg_launch_ns_kernel_trace" in prof:
                out["roofline"]["profile_avg_launch_us"] = 1e-3 * prof["avg_launch_ns_kernel_trace"]
            if "us_per_step_kernel_trace" in prof:  # (the kernel trace's duration of the timed launches / their steps)
                out["roofline"]["profile_us_per_step"] = prof["us_per_step_kernel_trace"]
                out["roofline"]["profile_steps_per_launch"] = prof.get("steps_per_launch", prof.get("steps_per_launch_kernel_trace"))
        if elapsed_rollout is not None:
            out["rollout"] = {"value": n_env * world * K / elapsed_rollout, "unit": "env-steps/s", "ms_per_step": 1e3 * elapsed_rollout / K,
                              "kernel": rollout_kernel,
                              "what": "same K steps as one hb_rollout_dev launch per GPU (no per-step batch barrier; testspeed.cc shape)"}
        if newton is not None:
            out["newton"] = newton
        if team is not None:
            out["team_robot"] = team
        if not args.no_cpu_baseline:  # (one number per box: rank 0, whatever N)
            out["cpu_baseline"] = cpu_baseline()
            out["cpu_baseline"].update(probe_libmujoco())
        print(json.dumps(out), flush=True)
    batch.dev_free(ctrl)
    batch.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
