#!/usr/bin/env python3
"""Copy the judged summaries of a tools/gpu_round.sh run from gpurun_out/<tag>/ into profiles/.

usage: tools/collect_profiles.py <tag> <round-label>   e.g. r01f r01
Writes profiles/<round>_kernel_stats.csv, <round>_counters.json, <round>_bench.json, <round>_testspeed.txt and
profiles/traffic_latest.json (read by bench.py for roofline.traffic).
HBM traffic follows /opt/skills/guides/MI355X_MICROARCH.md §HBM: separate --pmc passes for FETCH_SIZE and
WRITE_SIZE, bytes = counter * 1024, and FETCH_SIZE doubled (gfx950 reports half of a streaming read).
"""
import collections, csv, glob, json, os, shutil, sys

def is_step(name):
    """the benchmark's single-step launches: the size-specialised lean instantiation of the PGS step kernel (hb_step_lean_kernel for other models; hb_step_kernel before they existed, and in
    launches that carry an optional input or output)"""
    return "hb_step_h27_kernel(" in name or "hb_step_lean_kernel(" in name or "hb_step_kernel(" in name
    # (hb_step_duo_kernel, two envs per wave, has its own passes: tools/gpu_duo_counters.sh -> profiles/<round>_counters_duo.json)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, rnd = sys.argv[1], sys.argv[2]
src = os.path.join(ROOT, "gpurun_out", tag)
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)

stats = glob.glob(os.path.join(src, "prof_trace", "*", "*_kernel_stats.csv"))[0]
shutil.copy(stats, os.path.join(dst, rnd + "_kernel_stats.csv"))
# The bench command runs the step kernel in two shapes: 2048-block launches (pipelined `value` leg, two per step,
# overlapping) and 4096-block launches (the unpipelined roofline leg + its warm-up).  The per-dispatch trace
# separates them by grid size; the 4096-block average is the one bench.py's roofline.avg_launch_us must agree with.
avg_ns = None
calls = 0
by_grid = collections.defaultdict(list)
trace = glob.glob(os.path.join(src, "prof_trace", "*", "*_kernel_trace.csv"))
if trace:
    for row in csv.DictReader(open(trace[0])):
        if is_step(row["Kernel_Name"]):
            by_grid[int(row.get("Grid_Size") or row["Grid_Size_X"])].append(float(row["End_Timestamp"]) - float(row["Start_Timestamp"]))
    shutil.copy(trace[0], os.path.join(src, "kernel_trace_full.csv"))
full = 4096 * 64


def steady(v):
    """bench.py pre-rolls every env through 600 steps in ONE launch of the same kernel and grid: that dispatch (hundreds of times the
    one-step launches in every duration and instruction counter) is not a sample of the per-step launch; drop what exceeds 5x the median"""
    if not v:
        return v
    med = sorted(v)[len(v) // 2]
    return [x for x in v if x <= 5 * med] if med > 0 else v


# (since the timed loop's calls are folded, the bench command's own trace holds no one-env single-step launches: tools/gpu_round.sh traces
# `bench.py --no-pipeline --duo 0 --fold 1` for them)
trace1 = glob.glob(os.path.join(src, "prof_trace_h27", "*", "*_kernel_trace.csv"))
if trace1 and not by_grid.get(full):
    for row in csv.DictReader(open(trace1[0])):
        if is_step(row["Kernel_Name"]):
            by_grid[int(row.get("Grid_Size") or row["Grid_Size_X"])].append(float(row["End_Timestamp"]) - float(row["Start_Timestamp"]))
    st1 = glob.glob(os.path.join(src, "prof_trace_h27", "*", "*_kernel_stats.csv"))
    if st1:
        shutil.copy(st1[0], os.path.join(dst, rnd + "_kernel_stats_h27.csv"))
by_grid = collections.defaultdict(list, {g: steady(v) for g, v in by_grid.items()})
if by_grid.get(full):
    avg_ns = sum(by_grid[full]) / len(by_grid[full]); calls = len(by_grid[full])
else:
    for row in csv.DictReader(open(stats)):
        if is_step(row["Name"]):
            avg_ns = float(row["AverageNs"]); calls = int(row["Calls"])
counters = {}
meta = {}
for d in ("prof_fetch", "prof_write", "prof_sq", "prof_lds", "prof_mfma", "prof_valu"):
    fs = glob.glob(os.path.join(src, d, "*", "*_counter_collection.csv"))
    if not fs:
        continue
    agg = collections.defaultdict(list)
    for row in csv.DictReader(open(fs[0])):
        if is_step(row["Kernel_Name"]) and int(row["Grid_Size"]) == 4096 * 64:
            agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
            meta = {k: row[k] for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Scratch_Size", "Workgroup_Size", "Grid_Size")}
    for k, v in agg.items():
        v = steady(v)
        counters[k] = {"mean": sum(v) / len(v), "min": min(v), "max": max(v), "dispatches": len(v)}
n_env = 4096
out = {"kernel": "hb_step_h27_kernel", "launch": "4096 envs (one wave each), 1 step per launch, bench.py workload",
       "avg_launch_ns_kernel_trace": avg_ns, "kernel_trace_calls": calls,
       "kernel_trace_by_grid": {str(g): {"calls": len(v), "avg_ns": sum(v) / len(v)} for g, v in by_grid.items()},
       "dispatch_meta": meta, "counters_per_launch": counters}
if "FETCH_SIZE" in counters and "WRITE_SIZE" in counters:
    fetch, write = counters["FETCH_SIZE"]["mean"] * 1024, counters["WRITE_SIZE"]["mean"] * 1024
    hbm = 2 * fetch + write
    out["hbm"] = {"fetch_bytes_raw": fetch, "write_bytes": write, "hbm_bytes_per_launch": hbm, "algorithmic_bytes_per_launch": 748 * n_env,
                  "note": "FETCH_SIZE doubled per the gfx950 correction (calibrated for 16 B/lane streams; this kernel reads dwords, so this is an upper estimate)"}
    latest = {"hbm_bytes_per_launch": hbm, "source": "profiles/%s_counters.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, tools/gpu_round.sh)" % rnd}
    json.dump(latest, open(os.path.join(dst, "traffic_latest.json"), "w"), indent=1)
if "SQ_INSTS_VALU" in counters:
    w = counters["SQ_WAVES"]["mean"]
    out["per_wave"] = {k: counters[k]["mean"] / w for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_SMEM", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY",
                                                               "SQ_ACTIVE_INST_VALU", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE") if k in counters}
    if avg_ns:
        # fp32 VALU utilisation: wave64 VALU instruction = 64 lanes; peak 157.3 TFLOP/s = 78.6e12 lane-FMA/s
        out["valu_lane_ops_per_s"] = counters["SQ_INSTS_VALU"]["mean"] * 64 / (avg_ns * 1e-9)
        out["valu_issue_frac_of_peak"] = out["valu_lane_ops_per_s"] / 78.6e12
# Real lane utilisation: SQ_INSTS_VALU counts a wave instruction as one whatever its EXEC mask, and this kernel's
# stages run lane = body (17), dof (27), constraint row (~11).  SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU is the mean number of active
# lanes per VALU instruction: it reads 63.6 - 63.7 on kernels whose lanes are all active (the runtime's copyBuffer, hb_reset_kernel of
# the same pass: printed beside the step kernel's figure as the check of that reading).
fv = glob.glob(os.path.join(src, "prof_valu", "*", "*_counter_collection.csv"))
if fv:
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(fv[0])):
        k = "step" if (is_step(row["Kernel_Name"]) and int(row["Grid_Size"]) == full) else ("full" if ("hb_reset_kernel" in row["Kernel_Name"] or "copyBuffer" in row["Kernel_Name"]) else None)
        if k:
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    def ratio(k):
        t, a = steady(acc[k].get("SQ_THREAD_CYCLES_VALU", [])), steady(acc[k].get("SQ_ACTIVE_INST_VALU", []))
        return sum(t) / sum(a) if t and a and sum(a) > 0 else None
    rs, rh = ratio("step"), ratio("full")
    if rs:
        util = min(1.0, rs / 64.0)
        out["valu_lanes"] = {"mean_active_lanes_per_valu_instruction": rs, "active_lane_fraction": util, "same_ratio_on_full_wave_kernels": rh,
                             "note": "SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU; full-wave kernels of the same pass (hb_reset_kernel, copyBuffer) read ~63.7"}
        if "valu_lane_ops_per_s" in out:
            out["valu_useful_lane_ops_per_s"] = out["valu_lane_ops_per_s"] * util
            out["valu_useful_frac_of_peak"] = out["valu_useful_lane_ops_per_s"] / 78.6e12
json.dump(out, open(os.path.join(dst, rnd + "_counters.json"), "w"), indent=1)


def timed_kernel_profile():
    """The kernel bench.py's timed loop runs since its step calls are folded: hb_step_duo_q_kernel, launches of up to 256 steps of the two-envs-per-wave
    kernel.  PMC passes: `bench.py --steps 256 --warmup 5 --no-rollout --no-newton --no-team` - the process' LAST dispatch of that kernel is the
    timed loop's one launch of 256 steps (before it: the pre-roll's 600 steps and the warm-up's 5; the single-step leg behind it runs
    hb_step_duo_kernel).  Kernel trace: the bench command itself (--steps 1000: the last ceil(1000 / 256) dispatches)."""
    name, steps, waves = "hb_step_duo_q_kernel(", 256, 2048
    q = {}
    for d in ("prof_q_fetch", "prof_q_write", "prof_q_sq", "prof_q_lds", "prof_q_mfma", "prof_q_valu", "prof_q_icache"):
        fs = glob.glob(os.path.join(src, d, "*", "*_counter_collection.csv"))
        if not fs:
            continue
        rows = [r for r in csv.DictReader(open(fs[0])) if name in r["Kernel_Name"]]
        if not rows:
            continue
        last = max(int(r["Dispatch_Id"]) for r in rows)
        one = {}
        for r in rows:
            if int(r["Dispatch_Id"]) == last:
                one[r["Counter_Name"]] = one.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
                q["_meta"] = {k: r[k] for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Scratch_Size", "Workgroup_Size", "Grid_Size")}
        for k, v in one.items():
            q.setdefault(k, v)  # (a counter collected in two passes: the first pass' reading)
    res = {"kernel": "hb_step_duo_q_kernel", "pmc_launch": "%d blocks x 64 lanes (two envs each), %d steps in the launch" % (waves, steps), "counters_of_that_launch": q}
    if trace:
        rows = [(int(r["Dispatch_Id"]) if "Dispatch_Id" in r else i, float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
                for i, r in enumerate(csv.DictReader(open(trace[0]))) if name in r["Kernel_Name"]]
        rows.sort()
        K, n = 1000, 4  # (tools/gpu_round.sh: --steps 1000 = launches of 256, 256, 256, 232 steps)
        if len(rows) >= n + 2:
            t = [x for _, x in rows[-n:]]
            res["kernel_trace"] = {"timed_launches": n, "steps": K, "launch_ns": t, "avg_launch_ns": sum(t) / n, "us_per_step": 1e-3 * sum(t) / K}
    # the driver's command: `bench.py --gpus 1 --steps 20 --warmup 5` - its timed loop is the process' last dispatch of the kernel, 20 steps
    td = glob.glob(os.path.join(src, "prof_trace_driver", "*", "*_kernel_trace.csv"))
    if td:
        rows = sorted((int(r["Dispatch_Id"]), float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) for r in csv.DictReader(open(td[0])) if name in r["Kernel_Name"])
        if len(rows) >= 3:
            res["kernel_trace_driver_cmd"] = {"command": "bench.py --gpus 1 --steps 20 --warmup 5", "timed_launches": 1, "steps": 20, "launch_ns": rows[-1][1], "us_per_step": 1e-3 * rows[-1][1] / 20}
        sd = glob.glob(os.path.join(src, "prof_trace_driver", "*", "*_kernel_stats.csv"))
        if sd:
            shutil.copy(sd[0], os.path.join(dst, rnd + "_kernel_stats_driver_cmd.csv"))
    if "FETCH_SIZE" in q and "WRITE_SIZE" in q:
        fetch, write = q["FETCH_SIZE"] * 1024, q["WRITE_SIZE"] * 1024
        res["hbm"] = {"fetch_bytes_raw_per_step": fetch / steps, "write_bytes_per_step": write / steps, "hbm_bytes_per_step": (2 * fetch + write) / steps,
                      "algorithmic_bytes_per_step": 748 * n_env, "note": "FETCH_SIZE doubled per the gfx950 correction; the launch's counters / its 256 steps"}
    if "SQ_INSTS_VALU" in q and "SQ_WAVES" in q:
        res["per_env_step"] = {k: q[k] / (2.0 * q["SQ_WAVES"] * steps) for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_SMEM", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY",
                                                                                 "SQ_WAIT_INST_ANY", "SQ_LDS_BANK_CONFLICT", "SQ_INSTS_MFMA") if k in q}
    if "SQ_THREAD_CYCLES_VALU" in q and q.get("SQ_ACTIVE_INST_VALU"):
        res["active_lanes_per_valu_instruction"] = q["SQ_THREAD_CYCLES_VALU"] / q["SQ_ACTIVE_INST_VALU"]
    if "SQ_VALU_MFMA_BUSY_CYCLES" in q and q.get("SQ_BUSY_CU_CYCLES"):
        res["mfma_busy_frac"] = q["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * q["SQ_BUSY_CU_CYCLES"])
    if "SQ_ACTIVE_INST_VALU" in q and q.get("SQ_BUSY_CU_CYCLES"):
        res["valu_pipe_busy_frac"] = q["SQ_ACTIVE_INST_VALU"] / q["SQ_BUSY_CU_CYCLES"]
    if "SQC_ICACHE_MISSES" in q and q.get("SQC_ICACHE_REQ"):
        res["icache_miss_frac_of_requests"] = q["SQC_ICACHE_MISSES"] / q["SQC_ICACHE_REQ"]
    if "SQ_WAIT_INST_ANY" in q and q.get("SQ_WAVE_CYCLES"):
        res["sq_wait_inst_any_frac_of_wave_cycles"] = q["SQ_WAIT_INST_ANY"] / q["SQ_WAVE_CYCLES"]
    return res if (q or "kernel_trace" in res) else None


tk = timed_kernel_profile()
if tk:
    json.dump(tk, open(os.path.join(dst, rnd + "_counters_timed_kernel.json"), "w"), indent=1)
if "valu_issue_frac_of_peak" in out and os.path.exists(os.path.join(dst, "traffic_latest.json")):
    latest = json.load(open(os.path.join(dst, "traffic_latest.json")))
    # the launches of the TIMED loop: env segments (grids smaller than the whole batch) of the same kernel, from the kernel trace of the bench command
    seg = {g: v for g, v in by_grid.items() if g != full and len(v) >= 10}
    if seg:
        latest["segment_launch_us"] = {str(g // 64) + " envs": 1e-3 * sum(v) / len(v) for g, v in sorted(seg.items())}
    latest["valu_issue_frac_of_peak"] = out["valu_issue_frac_of_peak"]
    latest["avg_launch_ns_kernel_trace"] = avg_ns
    if "hbm" in out:
        latest["fetch_bytes_raw"] = out["hbm"]["fetch_bytes_raw"]; latest["write_bytes"] = out["hbm"]["write_bytes"]
    if "valu_lanes" in out:
        latest["valu"] = {"bound": "valu", "issued_lane_ops_per_s": out["valu_lane_ops_per_s"], "active_lane_fraction": out["valu_lanes"]["active_lane_fraction"],
                          "achieved": out["valu_useful_lane_ops_per_s"], "peak": 78.6e12, "unit": "fp32 lane-op/s", "frac": out["valu_useful_frac_of_peak"],
                          "note": "SQ_INSTS_VALU x 64 x (active lanes / 64) / launch time; peak = 157.3 TFLOP/s / 2 flop per fma"}
    if "SQ_VALU_MFMA_BUSY_CYCLES" in counters and "SQ_BUSY_CU_CYCLES" in counters:
        latest["mfma_busy_frac"] = counters["SQ_VALU_MFMA_BUSY_CYCLES"]["mean"] / (4.0 * counters["SQ_BUSY_CU_CYCLES"]["mean"])
    if "SQ_ACTIVE_INST_VALU" in counters and "SQ_BUSY_CU_CYCLES" in counters and "valu" in latest:
        # SQ_ACTIVE_INST_VALU counts one per wave64 VALU instruction issued (4 clocks on the SIMD's 16 lanes); a busy CU has 4 SIMDs:
        # 4 x ACTIVE / (4 x BUSY_CU cycles).  Whole unpipelined launch, its ragged tail included; 85 MFMAs per wave come on top (mfma_busy_frac)
        latest["valu"]["pipe_busy_frac"] = counters["SQ_ACTIVE_INST_VALU"]["mean"] / counters["SQ_BUSY_CU_CYCLES"]["mean"]
        latest["valu"]["instructions_per_wave"] = counters["SQ_INSTS_VALU"]["mean"] / counters["SQ_WAVES"]["mean"]
    latest["kernel"] = "hb_step_h27_kernel"
    if tk:
        e = {"source": "profiles/%s_counters_timed_kernel.json (rocprofv3 --pmc passes of `bench.py --steps 256`, the timed loop's one launch of 256 steps; tools/gpu_round.sh)" % rnd}
        if "hbm" in tk:
            e["hbm_bytes_per_step"] = tk["hbm"]["hbm_bytes_per_step"]; e["fetch_bytes_raw_per_step"] = tk["hbm"]["fetch_bytes_raw_per_step"]; e["write_bytes_per_step"] = tk["hbm"]["write_bytes_per_step"]
        if "kernel_trace" in tk:
            e["us_per_step_kernel_trace"] = tk["kernel_trace"]["us_per_step"]; e["avg_launch_ns_kernel_trace"] = tk["kernel_trace"]["avg_launch_ns"]
            e["steps_per_launch_kernel_trace"] = tk["kernel_trace"]["steps"] / tk["kernel_trace"]["timed_launches"]
        if "kernel_trace_driver_cmd" in tk:  # (a single launch of 20 steps: what `bench.py --steps 20` times)
            e["driver_cmd"] = {"us_per_step_kernel_trace": tk["kernel_trace_driver_cmd"]["us_per_step"], "avg_launch_ns_kernel_trace": tk["kernel_trace_driver_cmd"]["launch_ns"], "steps_per_launch": 20}
        if "per_env_step" in tk and "active_lanes_per_valu_instruction" in tk and "kernel_trace" in tk:
            issued = tk["per_env_step"]["SQ_INSTS_VALU"] * n_env * 64 / (tk["kernel_trace"]["us_per_step"] * 1e-6)
            util = min(1.0, tk["active_lanes_per_valu_instruction"] / 64.0)
            e["valu_issue_frac_of_peak"] = issued / 78.6e12
            e["valu"] = {"bound": "valu", "issued_lane_ops_per_s": issued, "active_lane_fraction": util, "achieved": issued * util, "peak": 78.6e12, "unit": "fp32 lane-op/s",
                         "frac": issued * util / 78.6e12, "instructions_per_env_step": tk["per_env_step"]["SQ_INSTS_VALU"],
                         "note": "SQ_INSTS_VALU x 64 x (active lanes / 64) of the 256-step launch / its duration in the kernel trace; peak = 157.3 TFLOP/s / 2 flop per fma"}
            if "valu_pipe_busy_frac" in tk:
                e["valu"]["pipe_busy_frac"] = tk["valu_pipe_busy_frac"]
        if "mfma_busy_frac" in tk:
            e["mfma_busy_frac"] = tk["mfma_busy_frac"]
        latest.setdefault("by_kernel", {})["hb_step_duo_q_kernel"] = e
    # the single-step form of the same kernel body (unpipelined step calls from 3072 envs, the closed loops): tools/gpu_duo_counters.sh's report, if it has been copied
    pd = os.path.join(dst, rnd + "_counters_duo.json")
    if os.path.exists(pd):
        dk = json.load(open(pd)).get("hb_step_duo_kernel", {})
        hb_ = dk.get("hbm_bytes_per_launch")
        if hb_:
            latest.setdefault("by_kernel", {})["hb_step_duo_kernel"] = {
                "source": "profiles/%s_counters_duo.json (tools/gpu_duo_counters.sh: rocprofv3 --pmc passes of `bench.py --no-pipeline --duo 2`, FETCH_SIZE and WRITE_SIZE in passes of their own)" % rnd,
                "hbm_bytes_per_launch": hb_["fetch_x2_gfx950"] + hb_["write"], "fetch_bytes_raw": hb_["fetch_raw"], "write_bytes": hb_["write"],
                "avg_launch_ns_kernel_trace": 1e3 * dk.get("avg_us", 0.0),
                "valu": {"bound": "valu", "instructions_per_wave": dk.get("per_wave", {}).get("SQ_INSTS_VALU"),
                         "active_lane_fraction": (dk.get("active_lanes_per_valu_instruction") or 0.0) / 64.0, "pipe_busy_frac": dk.get("valu_pipe_busy_frac")}}
    json.dump(latest, open(os.path.join(dst, "traffic_latest.json"), "w"), indent=1)
# second kernel trace (bench.py --no-pipeline with its Newton leg): every hb_* kernel's 4096-block launches
tn = glob.glob(os.path.join(src, "prof_trace_newton", "*", "*_kernel_trace.csv"))
if tn:
    per = collections.defaultdict(list)
    for row in csv.DictReader(open(tn[0])):
        if "hb_step" in row["Kernel_Name"] and int(row.get("Grid_Size") or row["Grid_Size_X"]) == full:
            per[row["Kernel_Name"].split("(")[0]].append(float(row["End_Timestamp"]) - float(row["Start_Timestamp"]))
    json.dump({"command": "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-rollout --no-pipeline",
               "launch": "4096 blocks x 64 lanes, one step per launch",
               "kernels": {k: {"calls": len(v), "avg_ns": sum(v) / len(v), "min_ns": min(v), "max_ns": max(v)} for k, v in per.items()}},
              open(os.path.join(dst, rnd + "_kernel_trace_solvers.json"), "w"), indent=1)
    st2 = glob.glob(os.path.join(src, "prof_trace_newton", "*", "*_kernel_stats.csv"))
    if st2:
        shutil.copy(st2[0], os.path.join(dst, rnd + "_kernel_stats_solvers.csv"))
for f, name in (("phase_profile.txt", rnd + "_phase_profile.txt"), ("parity_report.txt", rnd + "_parity_report.txt"), ("config4.txt", rnd + "_config4.txt"), ("config5.txt", rnd + "_config5.txt"), ("pipeline_sweep.txt", rnd + "_pipeline_sweep.txt"), ("vecenv.txt", rnd + "_vecenv.txt"), ("soak.txt", rnd + "_soak.txt"), ("soak_newton.txt", rnd + "_soak_newton.txt"), ("pgs_fit.txt", rnd + "_pgs_fit.txt"), ("planner.txt", rnd + "_planner.txt"), ("mpc_demo.txt", rnd + "_mpc_demo.txt"), ("latency.txt", rnd + "_latency.txt"), ("drift_nocontact.txt", rnd + "_drift_nocontact.txt"), ("parity_report_newton.txt", rnd + "_parity_report_newton.txt"), ("newton_bench.txt", rnd + "_newton_bench.txt"), ("newton_phases.txt", rnd + "_newton_phases.txt"), ("newton_sections.txt", rnd + "_newton_sections.txt"), ("bench.json", rnd + "_bench.json"), ("testspeed.log", rnd + "_testspeed.txt"), ("testspeed_newton.log", rnd + "_testspeed_newton.txt"), ("pytest_gpu.log", rnd + "_pytest_gpu.txt"), ("smoke.log", rnd + "_smoke.txt"), ("team_bench.txt", rnd + "_team_bench.txt"), ("phase_config5.txt", rnd + "_phase_config5.txt"), ("phase_team.txt", rnd + "_phase_team.txt"), ("phase_team_fused.txt", rnd + "_phase_team_fused.txt"), ("pipeline_queues.txt", rnd + "_pipeline_queues.txt"), ("phase_instructions.txt", rnd + "_phase_instructions.txt"), ("soak_pipelined.txt", rnd + "_soak_pipelined.txt"), ("config4_physics_only.txt", rnd + "_config4_physics_only.txt"), ("testspeed_stages.log", rnd + "_testspeed_stages.txt"), ("testspeed_team.log", rnd + "_testspeed_team.txt"), ("team_short.txt", rnd + "_team_short.txt"), ("duo_sizes.txt", rnd + "_duo_sizes.txt"), ("vecenv_sb3.txt", rnd + "_vecenv_sb3.txt"), ("step_latency_dist.txt", rnd + "_step_latency_dist.txt"), ("fold_sizes.txt", rnd + "_fold_sizes_now.txt"),
                ("prof_team/t_kernel_stats.csv", rnd + "_kernel_stats_team.csv"), ("prof_config5/t_kernel_stats.csv", rnd + "_kernel_stats_config5.csv")):
    p = os.path.join(src, f)
    if os.path.exists(p):
        shutil.copy(p, os.path.join(dst, name))
print(json.dumps({k: out[k] for k in out if k in ("avg_launch_ns_kernel_trace", "hbm", "per_wave", "valu_issue_frac_of_peak")}, indent=1))
