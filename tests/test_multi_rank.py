"""N>1 path on CPU: world_size-2 gloo run of bench.py's sharding logic.

The hot path has no data collective (independent envs): ranks own contiguous env
blocks selected by env_offset, and only the timing (MAX) and a state checksum are reduced.  The
oracle plays the stepper here (CPU, test-only) to prove that sharded results equal the unsharded
ones bit for bit and that the reductions bench.py relies on behave.
"""
import multiprocessing as mp
import os
import socket
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_total, steps, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    from bench import shard_range
    from oracle_lib import Oracle
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    lo, hi = shard_range(n_total, world, rank)
    o = Oracle()
    _, qpos, _ = o.rollout_threads(hi - lo, steps, 2, env_offset=lo, want_qpos=True)
    # what bench.py reduces: max elapsed over ranks, plus here a gather of the shard results
    t = torch.tensor([0.5 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    parts = [torch.zeros((n_total // world, qpos.shape[1]), dtype=torch.float64) for _ in range(world)]
    dist.all_gather(parts, torch.from_numpy(qpos))
    dist.barrier()
    if rank == 0:
        q.put((float(t.item()), torch.cat(parts).numpy(), (lo, hi)))
    dist.destroy_process_group()


def test_shard_range_partitions_exactly():
    sys.path.insert(0, ROOT)
    from bench import shard_range
    for n, w in ((32768, 8), (8192, 2), (4096, 1), (10, 3)):
        spans = [shard_range(n, w, r) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
    assert shard_range(32768, 8, 3) == (3 * 4096, 4 * 4096)  # config 3: env e on GPU floor(e/4096)


def test_two_rank_gloo_sharded_equals_unsharded():
    world, n_total, steps = 2, 8, 120
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_total, steps, q)) for r in range(world)]
    for p in procs:
        p.start()
    tmax, gathered, span0 = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert tmax == 1.5 and span0 == (0, 4)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_lib import Oracle
    _, whole, _ = Oracle().rollout_threads(n_total, steps, 2, env_offset=0, want_qpos=True)
    assert np.array_equal(gathered, whole)


def _run_bench(argv, extra_env=None, timeout=180):
    import subprocess
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    env["HB_BENCH_BACKEND"] = "gloo"
    env.update(extra_env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, env=env, capture_output=True, text=True, timeout=timeout)


def test_bench_gpus_flag_spawns_one_rank_per_gpu():
    """`python3 bench.py --gpus 2` with WORLD_SIZE unset (the way the driver calls it when it does not use torchrun): the
    parent must start two ranks itself and relay rank 0's line.  --dry-run keeps it to launch, rendezvous and the
    max-over-ranks reduction (no GPU here); the GPU legs behind it are the single-rank code path."""
    import json
    r = _run_bench(["--gpus", "2", "--steps", "7", "--warmup", "3", "--dry-run"])
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(line) == 1, r.stdout
    out = json.loads(line[0])
    assert out["n_gpus"] == 2 and out["steps"] == 7 and out["warmup"] == 3 and out["dry_run"] is True
    assert out["max_over_ranks_s"] == 2e-3                  # MAX over ranks of (1 + rank) ms
    assert out["shard_of_last_rank"] == [4096, 8192]        # env blocks by rank
    # what the 8-GPU record needs of every rank: how many ranks the process group really had, and per rank its own time, device and env block
    assert out["ranks_seen"] == 2 and [r["rank"] for r in out["ranks"]] == [0, 1]
    assert [r["elapsed_s"] for r in out["ranks"]] == [1e-3, 2e-3] and [r["envs"] for r in out["ranks"]] == [[0, 4096], [4096, 8192]]
    assert all("device" in r and "segments" in r for r in out["ranks"])


def test_bench_under_a_launcher_keeps_the_launchers_world():
    """The torchrun shape: RANK / WORLD_SIZE already in the environment -> no second spawn, the process runs its rank."""
    import json
    import subprocess
    port = str(_free_port())
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=port, HB_BENCH_BACKEND="gloo")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"], env=env, stdout=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=180)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs)
    assert json.loads(outs[0].strip().splitlines()[-1])["n_gpus"] == 2 and not any(ln.startswith("{") for ln in outs[1].splitlines())


def test_bench_spawn_reports_a_failed_rank():
    """A rank that dies must fail the whole run (non-zero exit) instead of leaving the others at the rendezvous."""
    sys.path.insert(0, ROOT)
    import tempfile
    from bench import spawn_ranks
    with tempfile.NamedTemporaryFile("w", suffix=".py", delete=False) as f:
        f.write("import os, sys, time\nif os.environ['RANK'] == '1': sys.exit(3)\ntime.sleep(60)\n")
        script = f.name
    try:
        assert spawn_ranks(2, [], script=script, timeout=30) == 3
    finally:
        os.unlink(script)
