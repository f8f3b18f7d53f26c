#!/bin/bash
# One GPU-box session: parity tests, bench, rocprofv3 kernel trace + PMC passes.
# usage: tools/gpu_round.sh <tag>   (outputs under gpurun_out/<tag>/)
set -o pipefail
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
# the diagnostic libraries (stage stamps, Newton sections) must be of this source: built here if the box has not got them from the snapshot
[ build/libhb_stamps.so -nt humanoid_mujoco_amd/libhb.so ] || echo "note: build/libhb_stamps.so is older than libhb.so: run make build/libhb_stamps.so build/libhb_probe.so before the round" | tee $OUT/progress.log
echo "== pytest -m gpu" | tee -a $OUT/progress.log
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/progress.log
tail -5 $OUT/pytest_gpu.log
echo "== smoke" | tee -a $OUT/progress.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/progress.log
echo "== bench" | tee -a $OUT/progress.log
timeout -k 10 600 python bench.py --steps 1000 --warmup 20 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" | tee -a $OUT/progress.log
cat $OUT/bench.json
echo "== testspeed (C++ host)" | tee -a $OUT/progress.log
timeout -k 10 300 ./build/hb_testspeed humanoid_mujoco_amd/assets/humanoid27.hbm 1000 4096 > $OUT/testspeed.log 2>&1; echo "testspeed rc=$?" | tee -a $OUT/progress.log
cat $OUT/testspeed.log
timeout -k 10 300 ./build/hb_testspeed humanoid_mujoco_amd/assets/humanoid27.hbm 1000 4096 0 Newton > $OUT/testspeed_newton.log 2>&1; echo "testspeed newton rc=$?" | tee -a $OUT/progress.log
timeout -k 10 300 ./build/hb_testspeed_stamps humanoid_mujoco_amd/assets/humanoid27.hbm 300 4096 > $OUT/testspeed_stages.log 2>&1; timeout -k 10 300 ./build/hb_testspeed humanoid_mujoco_amd/assets/team_robot.hbm 500 4096 > $OUT/testspeed_team.log 2>&1
echo "== rocprofv3 kernel trace" | tee -a $OUT/progress.log
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_trace -- python3 bench.py --steps 1000 --warmup 20 --no-cpu-baseline --no-rollout --no-newton --no-team > $OUT/prof_trace.log 2>&1; echo "trace rc=$?" | tee -a $OUT/progress.log
# the driver's own command (--steps 20 --warmup 5: its timed loop is ONE launch of 20 steps, which ends on its slowest wave - a different shape from the 250-step launches above)
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_trace_driver -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-rollout --no-newton --no-team > $OUT/prof_trace_driver.log 2>&1; echo "trace driver cmd rc=$?" | tee -a $OUT/progress.log
# the one-env-per-wave single-step kernel (hb_step_h27_kernel: batches below 4096 envs, closed loops), one launch per step: its launch durations for the PMC passes below
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_trace_h27 -- python3 bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-rollout --no-newton --no-team --no-pipeline --duo 0 --fold 1 > $OUT/prof_trace_h27.log 2>&1; echo "trace h27 rc=$?" | tee -a $OUT/progress.log
echo "== rocprofv3 pmc FETCH" | tee -a $OUT/progress.log
timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/prof_fetch -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-rollout --no-newton --no-team --no-pipeline --duo 0 > $OUT/prof_fetch.log 2>&1; echo "fetch rc=$?" | tee -a $OUT/progress.log
timeout -k 10 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/prof_write -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-rollout --no-newton --no-team --no-pipeline --duo 0 > $OUT/prof_write.log 2>&1; echo "write rc=$?" | tee -a $OUT/progress.log
timeout -k 10 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/prof_sq -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-rollout --no-newton --no-team --no-pipeline --duo 0 > $OUT/prof_sq.log 2>&1; echo "sq rc=$?" | tee -a $OUT/progress.log
timeout -k 10 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/prof_lds -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-rollout --no-newton --no-team --no-pipeline --duo 0 > $OUT/prof_lds.log 2>&1; echo "lds rc=$?" | tee -a $OUT/progress.log
# real lane utilisation of the VALU instructions (SQ_INSTS_VALU counts a wave instruction whatever its EXEC mask): thread-cycles over
# instruction-cycles = active lanes per VALU instruction (63.7 on the full-wave kernels of the same pass) - collect_profiles.py
timeout -k 10 600 rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES --output-format csv -d $OUT/prof_valu -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-rollout --no-newton --no-team --no-pipeline --duo 0 > $OUT/prof_valu.log 2>&1; echo "valu rc=$?" | tee -a $OUT/progress.log
timeout -k 10 600 rocprofv3 --pmc SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --output-format csv -d $OUT/prof_mfma -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-rollout --no-newton --no-team --no-pipeline --duo 0 > $OUT/prof_mfma.log 2>&1; echo "mfma rc=$?" | tee -a $OUT/progress.log
# the kernel the timed loop runs (its step calls folded: hb_step_duo_q_kernel): the same passes on `bench.py --steps 256`, whose timed loop is ONE launch of 256 steps
QARGS="--steps 256 --warmup 5 --no-cpu-baseline --no-rollout --no-newton --no-team"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/prof_q_fetch -- python3 bench.py $QARGS > $OUT/prof_q_fetch.log 2>&1; echo "q fetch rc=$?" | tee -a $OUT/progress.log
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/prof_q_write -- python3 bench.py $QARGS > $OUT/prof_q_write.log 2>&1; echo "q write rc=$?" | tee -a $OUT/progress.log
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/prof_q_sq -- python3 bench.py $QARGS > $OUT/prof_q_sq.log 2>&1; echo "q sq rc=$?" | tee -a $OUT/progress.log
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/prof_q_lds -- python3 bench.py $QARGS > $OUT/prof_q_lds.log 2>&1; echo "q lds rc=$?" | tee -a $OUT/progress.log
timeout -k 10 300 rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/prof_q_valu -- python3 bench.py $QARGS > $OUT/prof_q_valu.log 2>&1; echo "q valu rc=$?" | tee -a $OUT/progress.log
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --output-format csv -d $OUT/prof_q_mfma -- python3 bench.py $QARGS > $OUT/prof_q_mfma.log 2>&1; echo "q mfma rc=$?" | tee -a $OUT/progress.log
# (waves of a 256-step launch drift apart: does the instruction cache still hold what 8 waves per CU are running?)
timeout -k 10 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/prof_q_icache -- python3 bench.py $QARGS > $OUT/prof_q_icache.log 2>&1; echo "q icache rc=$?" | tee -a $OUT/progress.log
timeout -k 10 200 python tools/gpu_pipeline_sweep.py > $OUT/pipeline_sweep.txt 2>&1; echo "sweep rc=$?" | tee -a $OUT/progress.log
timeout -k 10 200 python tools/gpu_phase_profile.py 4096 > $OUT/phase_profile.txt 2>&1; echo "phase rc=$?" | tee -a $OUT/progress.log
timeout -k 10 200 python tools/gpu_phase_profile.py 4096 humanoid27_hfield.hbm > $OUT/phase_config5.txt 2>&1; timeout -k 10 200 python tools/gpu_phase_profile.py 4096 team_robot.hbm > $OUT/phase_team.txt 2>&1; timeout -k 10 400 python tools/gpu_pipeline_queues.py default GPU_MAX_HW_QUEUES=8 GPU_MAX_HW_QUEUES=16 > $OUT/pipeline_queues.txt 2>&1; timeout -k 10 900 bash tools/gpu_phase_instructions.sh $OUT/phase_inst > $OUT/phase_instructions.txt 2>&1
timeout -k 10 200 python tools/gpu_parity_report.py > $OUT/parity_report.txt 2>&1; timeout -k 10 200 python tools/gpu_parity_report.py newton > $OUT/parity_report_newton.txt 2>&1; echo "parity rc=$?" | tee -a $OUT/progress.log
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_trace_newton -- python3 bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-rollout --no-pipeline > $OUT/prof_trace_newton.log 2>&1; echo "newton trace rc=$?" | tee -a $OUT/progress.log
timeout -k 10 300 python tools/gpu_soak.py > $OUT/soak.txt 2>&1; timeout -k 10 300 python tools/gpu_soak_pipelined.py 100000 > $OUT/soak_pipelined.txt 2>&1; timeout -k 10 200 python tools/gpu_config4_physics_only.py > $OUT/config4_physics_only.txt 2>&1; timeout -k 10 300 python tools/gpu_soak.py newton > $OUT/soak_newton.txt 2>&1; timeout -k 10 200 python tools/gpu_vecenv_bench.py > $OUT/vecenv.txt 2>&1; timeout -k 10 200 python tools/gpu_pgs_fit.py > $OUT/pgs_fit.txt 2>&1; timeout -k 10 200 python tools/gpu_config4.py > $OUT/config4.txt 2>&1; timeout -k 10 200 python tools/gpu_config5.py > $OUT/config5.txt 2>&1; timeout -k 10 300 python tools/gpu_team_bench.py > $OUT/team_bench.txt 2>&1; timeout -k 10 300 python tools/gpu_planner_bench.py > $OUT/planner.txt 2>&1; timeout -k 10 300 python tools/gpu_drift_nocontact.py > $OUT/drift_nocontact.txt 2>&1; timeout -k 10 300 python tools/gpu_mpc_demo.py 4096 > $OUT/mpc_demo.txt 2>&1; timeout -k 10 300 python tools/gpu_latency.py > $OUT/latency.txt 2>&1; timeout -k 10 300 python tools/gpu_fold_sizes.py > $OUT/fold_sizes.txt 2>&1; timeout -k 10 300 python tools/gpu_newton_bench.py > $OUT/newton_bench.txt 2>&1; timeout -k 10 200 python tools/gpu_newton_bench.py --phases > $OUT/newton_phases.txt 2>&1; timeout -k 10 200 python tools/gpu_newton_bench.py --probe > $OUT/newton_sections.txt 2>&1; echo "configs rc=$?" | tee -a $OUT/progress.log
# kernel split of a staged step: the reference's own robot and the terrain humanoid (configs[4])
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_team -o t -- python3 tools/gpu_team_short.py > $OUT/team_short.txt 2>&1; timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_config5 -o t -- python3 tools/gpu_team_short.py 8192 humanoid27_hfield.hbm > $OUT/config5_short.txt 2>&1; echo "staged traces rc=$?" | tee -a $OUT/progress.log
find $OUT -name "*.csv" | head -40
echo done | tee -a $OUT/progress.log
