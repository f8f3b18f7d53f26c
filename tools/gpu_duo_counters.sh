#!/bin/bash
# rocprofv3 passes over the unpipelined single-step launches of the benchmark workload (bench.py --no-pipeline, 4096 envs) for ONE kernel choice:
#   tools/gpu_duo_counters.sh <out dir> <HB_DUO value> <kernel name> <grid threads>
# e.g. gpurun_out/r04_duo 2 hb_step_duo_kernel 131072   /   gpurun_out/r04_solo 0 hb_step_h27_kernel 262144
OUT=$1; export HB_DUO=$2; K=$3; export REPORT_GRID=$4
mkdir -p $OUT
export TMPDIR=/tmp
B="python3 bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-rollout --no-newton --no-team --no-pipeline --duo $HB_DUO"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $B > $OUT/trace.txt 2>&1; echo "trace rc=$?"
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/sq -o t -- $B > $OUT/sq.txt 2>&1; echo "sq rc=$?"
timeout -k 10 300 rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES --output-format csv -d $OUT/valu -o t -- $B > $OUT/valu.txt 2>&1; echo "valu rc=$?"
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES --output-format csv -d $OUT/mem -o t -- $B > $OUT/mem.txt 2>&1; echo "mem rc=$?"
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/mfma -o t -- $B > $OUT/mfma.txt 2>&1; echo "mfma rc=$?"
# (FETCH_SIZE and WRITE_SIZE in separate passes, as /opt/skills/guides/MI355X_MICROARCH.md prescribes: together the profiler aborts)
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/hbm -o t -- $B > $OUT/hbm.txt 2>&1; echo "hbm fetch rc=$?"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/hbmw -o t -- $B > $OUT/hbmw.txt 2>&1; echo "hbm write rc=$?"
python3 tools/team_counters_report.py $OUT $K | tee $OUT/report.txt
