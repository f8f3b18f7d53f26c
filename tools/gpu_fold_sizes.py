#!/usr/bin/env python3
"""How long K hb_step_dev calls take: one launch per call on three env segments ("calls"), folded into launches of up to 64 / 256 steps
(include/hb.h: hb_step_dev, HB_TUNE_FOLD), and hb_rollout_dev of the same K steps.  4096 envs of the 27-dof humanoid in the
benchmark's steady regime; every figure the median of 15 repetitions (host clock around enqueue + hb_batch_sync), us per step.
FORCE_SEGMENTS=1 (an experiment of round 4, with a library that still cut multi-step launches into segments) is what
profiles/r04_fold_sizes.txt was measured with: it shows why multi-step launches are not segmented."""
import os, statistics, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import humanoid_mujoco_amd as hb
m = hb.Model.load(os.path.join(ROOT, "humanoid_mujoco_amd", "assets", "humanoid27.hbm"))
if os.environ.get("SOLVER") == "newton":
    m.set_opt(solver=2, iterations=100)
N = int(os.environ.get("N_ENV", "4096"))
KMAX = 256
b = hb.Batch(m, N, 0)
b.reset(perturb=True)
b.rollout_halton(600)
b.sync()
ctrl = b.dev_alloc(KMAX * N * m.nu * 4)
b.halton_ctrl_dev(KMAX, 600, 0, ctrl)
stride = N * m.nu * 4
st = b.get_state(hb.STATE_INTEGRATION)


def run(kind, K, seg, duo):
    b.pipeline(seg)
    b.tune(duo=duo, fold={"calls": 1, "folded64": 64, "folded256": 256}.get(kind, 1))
    ts = []
    for rep in range(16):
        b.set_state(hb.STATE_INTEGRATION, st)
        # (warm: the same shape once, untimed, so that the heavy-first order of this segmentation exists)
        if rep == 0:
            b.rollout_dev(ctrl, 8)
            b.sync()
            continue
        t0 = time.perf_counter()
        if kind == "rollout":
            b.rollout_dev(ctrl, K)
        else:
            for t in range(K):
                b.step_dev(ctrl + t * stride)
        b.sync()
        ts.append(time.perf_counter() - t0)
    return 1e6 * statistics.median(ts) / K, b.last_kernel()


print("%d envs; us per step (median of 15)" % N)
print("%-8s %4s %9s %4s %10s  %s" % ("kind", "K", "segments", "duo", "us/step", "kernel"))
DUOS = tuple(int(x) for x in os.environ["DUO"].split(",")) if os.environ.get("DUO") else ((1,) if os.environ.get("SOLVER") == "newton" else (1, 0))
KS = tuple(int(x) for x in os.environ["KS"].split(",")) if os.environ.get("KS") else (5, 20, 64, 256)
for duo in DUOS:
    for K in KS:
        for kind, seg in (("calls", 3), ("folded64", 3), ("folded256", 3), ("rollout", 0), ("rollout", 3)):
            us, k = run(kind, K, seg, duo)
            print("%-8s %4d %9d %4d %10.1f  %s" % (kind, K, max(seg, 1), duo, us, k), flush=True)
