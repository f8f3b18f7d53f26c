#!/usr/bin/env python3
"""Per-stage dynamic instruction counts of hb_step_kernel from the passes of tools/gpu_phase_instructions.sh (per wave = per env-step)."""
import csv, glob, os, sys
src = sys.argv[1]
DUO = os.environ.get("PHASE_DUO", "0") == "2"
names = ["prologue (tables, state, SGPR spill stores)", "ctrl+check", "kinematics", "geoms/com/cinert/cdof", "comVel+crb+rne tree passes", "qM", "factorM", "bias/passive/act", "collision",
         "makeConstraint", "row quantities", "half-solve", "b + AR", "PGS", "dual finish", "Euler+advance", "(the 12 probed launches of the whole-step pass advance the state: not a stage)"]
if DUO:  # hb_step_duo_kernel's stamps: no factorM slot, the half solve in two parts; a wave = TWO env-steps
    names = ["prologue (state and controls of both envs in)", "ctrl+check", "kinematics", "geoms/com/cinert/cdof", "comVel+crb+rne tree passes", "qM", "bias/passive/act", "collision",
             "makeConstraint (count, place, write)", "row quantities", "W: elimination of M, both envs side by side", "C = J W", "b + AR (block diagonal)", "PGS (side by side)",
             "dual finish + checkAcc", "Euler (H, both envs side by side) + advance", "-"]
WAVES = 2048.0 if DUO else 4096.0


def read(prefix, k):
    fs = glob.glob(os.path.join(src, "%s%d" % (prefix, k), "*counter_collection.csv")) + glob.glob(os.path.join(src, "%s%d" % (prefix, k), "*", "*counter_collection.csv"))
    acc = {}
    rows = [r for f in fs for r in csv.DictReader(open(f)) if any(k in r["Kernel_Name"] for k in (("hb_step_duo_kernel(",) if DUO else ("hb_step_kernel(", "hb_step_lean_kernel(", "hb_step_h27_kernel("))) and int(r["Grid_Size"]) == int(WAVES) * 64]
    # the last 12 dispatches per counter are the probed ones (the pre-roll is one multi-step dispatch of the same kernel)
    by = {}
    for r in rows: by.setdefault(r["Counter_Name"], []).append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    for c, v in by.items():
        v.sort(); v = [x for _, x in v[-10:]]
        acc[c] = sum(v) / len(v) / WAVES
    return acc
cum = {}
for k in list(range(1, 17)) + [0]:
    a = read("k", k); a.update(read("m", k)); cum[k] = a
order = list(range(1, 17))
cols = ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_SMEM", "SQ_INSTS_VALU_MFMA_MOPS_F32", "SQ_THREAD_CYCLES_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES"]
print("hb_step_duo_kernel: per wave = per TWO env-steps" if DUO else "hb_step_h27_kernel")
print("per wave (= per env-step; two with the duo kernel), 4096 envs in the benchmark's steady regime; each row = counters up to this stamp minus counters up to the one before")
print("%-46s %8s %8s %7s %7s %7s %9s %11s %12s" % ("stage", "VALU", "SALU", "LDS", "VMEM", "SMEM", "MFMA ops", "lanes/VALU", "wave cycles*"))
prev = {c: 0.0 for c in cols}
tot = cum[16]
for i, k in enumerate(order):
    d = {c: cum[k].get(c, 0.0) - prev[c] for c in cols}
    lanes = d["SQ_THREAD_CYCLES_VALU"] / d["SQ_ACTIVE_INST_VALU"] if d["SQ_ACTIVE_INST_VALU"] > 0 else 0.0
    print("%-46s %8.0f %8.0f %7.0f %7.0f %7.0f %9.0f %11.1f %12.0f" % (names[i], d["SQ_INSTS_VALU"], d["SQ_INSTS_SALU"], d["SQ_INSTS_LDS"], d["SQ_INSTS_VMEM_RD"], d["SQ_INSTS_SMEM"],
                                                                   d["SQ_INSTS_VALU_MFMA_MOPS_F32"] / 8.0, lanes, d["SQ_WAVE_CYCLES"] * 4))
    prev = {c: cum[k].get(c, 0.0) for c in cols}
lanes = tot["SQ_THREAD_CYCLES_VALU"] / tot["SQ_ACTIVE_INST_VALU"]
print("%-46s %8.0f %8.0f %7.0f %7.0f %7.0f %9.0f %11.1f %12.0f" % ("whole step (up to the state write)", tot["SQ_INSTS_VALU"], tot["SQ_INSTS_SALU"], tot["SQ_INSTS_LDS"], tot["SQ_INSTS_VMEM_RD"], tot["SQ_INSTS_SMEM"],
                                                               tot["SQ_INSTS_VALU_MFMA_MOPS_F32"] / 8.0, lanes, tot["SQ_WAVE_CYCLES"] * 4))
print("(* SQ_WAVE_CYCLES x 4 of a chip full of waves that all leave at the same stamp: not the stage's share of a real step's time)")
