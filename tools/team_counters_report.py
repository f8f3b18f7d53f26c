#!/usr/bin/env python3
"""Per-kernel summary of tools/gpu_team_counters.sh's rocprofv3 passes (means over the steady single-step launches)."""
import collections, csv, glob, json, os, sys
out = sys.argv[1]
KERNELS = tuple(sys.argv[2:]) or ("hb_narrow_kernel", "hb_narrow2_kernel", "hb_narrow_prim_kernel", "hb_pose_kernel", "hb_step_newton_gen20_team_kernel", "hb_step_newton_big20_kernel")


def short(name):
    for k in KERNELS:
        if k + "(" in name:
            return k
    return None


res = collections.defaultdict(dict)
GRID = int(os.environ.get("REPORT_GRID", "0"))  # only dispatches of this grid size (threads), e.g. the unpipelined single-step launches
tr = glob.glob(os.path.join(out, "trace", "*kernel_trace.csv")) or glob.glob(os.path.join(out, "trace", "*", "*kernel_trace.csv"))
if tr:
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(tr[0])):
        k = short(r["Kernel_Name"])
        if k:
            if GRID and int(r.get("Grid_Size") or r["Grid_Size_X"]) != GRID:
                continue
            d[k].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    for k, v in d.items():
        v = sorted(v)[: max(1, len(v) - 1)]  # (the 150-step settle launch of the rollout is not a per-step launch)
        res[k]["launches"] = len(v); res[k]["avg_us"] = 1e-3 * sum(v) / len(v)
for sub in ("sq", "valu", "mem", "hbm", "hbmw", "mfma"):
    fs = glob.glob(os.path.join(out, sub, "*counter_collection.csv")) or glob.glob(os.path.join(out, sub, "*", "*counter_collection.csv"))
    if not fs:
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    meta = {}
    for r in csv.DictReader(open(fs[0])):
        k = short(r["Kernel_Name"])
        if k:
            if GRID and int(r["Grid_Size"]) != GRID:
                continue
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta[k] = {x: r[x] for x in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Scratch_Size", "LDS_Block_Size", "Grid_Size", "Workgroup_Size") if x in r}
    for k, cs in agg.items():
        for c, v in cs.items():
            v = sorted(v)
            v = v[: max(1, len(v) - 1)] if len(v) > 3 else v
            res[k][c] = sum(v) / len(v)
        res[k]["dispatch"] = meta[k]
for k, r in res.items():
    w = r.get("SQ_WAVES")
    if w:
        r["per_wave"] = {c: r[c] / w for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM", "SQ_WAVE_CYCLES") if c in r}
    if r.get("SQ_ACTIVE_INST_VALU") and r.get("SQ_THREAD_CYCLES_VALU"):
        r["active_lanes_per_valu_instruction"] = r["SQ_THREAD_CYCLES_VALU"] / r["SQ_ACTIVE_INST_VALU"]
    if r.get("SQ_WAVE_CYCLES"):
        for c in ("SQ_WAIT_INST_ANY", "SQ_WAIT_ANY"):
            if c in r:
                r[c.lower() + "_frac_of_wave_cycles"] = r[c] / r["SQ_WAVE_CYCLES"]
    if r.get("SQ_INSTS_VALU_MFMA_MOPS_F32") is not None and r.get("SQ_BUSY_CU_CYCLES"):
        r["mfma_busy_frac_of_busy_cu_cycles"] = r.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / r["SQ_BUSY_CU_CYCLES"]
    if r.get("SQ_ACTIVE_INST_VALU") and r.get("SQ_BUSY_CU_CYCLES"):
        r["valu_pipe_busy_frac"] = r["SQ_ACTIVE_INST_VALU"] / r["SQ_BUSY_CU_CYCLES"]
    if "FETCH_SIZE" in r:
        r["hbm_bytes_per_launch"] = {"fetch_raw": 1024 * r["FETCH_SIZE"], "fetch_x2_gfx950": 2048 * r["FETCH_SIZE"], "write": 1024 * r.get("WRITE_SIZE", 0)}
print(json.dumps(res, indent=1, sort_keys=True))
