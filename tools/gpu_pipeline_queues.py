#!/usr/bin/env python3
"""Step-API throughput against the number of pipeline segments, every case in a process of its own (the
stream -> hardware-queue assignment depends on what the process created before), each case repeated.
Output: us per step of 4096 envs (segments in use); "segments 1" is hb_batch_pipeline's default (probe).
Cases: GPU_MAX_HW_QUEUES, the scheduling knobs of hb_batch_tune (TUNE_<knob>=<value> in the child's environment).
(Rounds 1-2 ran the segments on extra streams BESIDE the batch's own: gpurun_out/r03q/queues.txt, queues2.txt.)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, time
sys.path.insert(0, %r)
import humanoid_mujoco_amd as hb
npipe = int(sys.argv[1])
m = hb.Model.load(os.path.join(%r, "humanoid_mujoco_amd", "assets", "humanoid27.hbm"))
N, K, W = 4096, 400, 300
b = hb.Batch(m, N, 0)
b.tune(**{k[5:].lower(): int(v) for k, v in os.environ.items() if k.startswith("TUNE_")})
ctrl = b.dev_alloc((K + W) * N * m.nu * 4)
b.halton_ctrl_dev(K + W, 0, 0, ctrl)
b.reset(perturb=True)
b.pipeline(npipe)
b.tune(fold=1)  # (this is about one launch per call)
stride = N * m.nu * 4
for t in range(W): b.step_dev(ctrl + t * stride)
b.sync()
t0 = time.perf_counter()
for t in range(W, W + K): b.step_dev(ctrl + t * stride)
b.sync()
dt = time.perf_counter() - t0
print("%%.1f(%%d)" %% (1e6 * dt / K, b.segments))
''' % (ROOT, ROOT)
cases = [("default", {}, (1, 2, 3, 4)), ("GPU_MAX_HW_QUEUES=8", {"GPU_MAX_HW_QUEUES": "8"}, (2, 3, 4, 5, 6, 8)),
         ("GPU_MAX_HW_QUEUES=16", {"GPU_MAX_HW_QUEUES": "16"}, (2, 3, 4, 5, 6, 8))]
cases += [("TUNE_SCHEDULE=0", {"TUNE_SCHEDULE": "0"}, (1, 2)), ("TUNE_REORDER_PERIOD=1", {"TUNE_REORDER_PERIOD": "1"}, (1,)),
          ("TUNE_REORDER_PERIOD=8", {"TUNE_REORDER_PERIOD": "8"}, (1,)), ("TUNE_DUO=2", {"TUNE_DUO": "2"}, (0, 1, 2, 3)), ("TUNE_DUO=0", {"TUNE_DUO": "0"}, (0, 1, 3))]
if len(sys.argv) > 1: cases = [c for c in cases if c[0] in sys.argv[1:]]
for name, extra, pipes in cases:
    for npipe in pipes:
        res = []
        for rep in range(2):
            env = dict(os.environ); env.update(extra)
            out = subprocess.run([sys.executable, "-c", CHILD, str(npipe)], env=env, capture_output=True, text=True, timeout=300)
            res.append(out.stdout.strip().splitlines()[-1] if out.returncode == 0 and out.stdout.strip() else "rc%d" % out.returncode)
        print("%-22s segments %d: %s us/step" % (name, npipe, " ".join(res)), flush=True)
