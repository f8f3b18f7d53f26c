#!/bin/bash
# Dynamic instruction counts per stage of hb_step_kernel: the diagnostic build run sixteen times under rocprofv3 --pmc, every run leaving
# the kernel at one stamp further; tools/phase_instructions_report.py differences the counters.   usage: gpu_phase_instructions.sh <outdir> [model.hbm]
OUT=${1:-gpurun_out/phase_inst}; MODEL=${2:-humanoid27.hbm}
mkdir -p $OUT; OUT=$(cd $OUT && pwd); ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp; export TMPDIR=/tmp
for k in 1 2 3 4 5 6 7 8 9 10 11 12 13 14 15 16 0; do
  timeout -k 10 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAVES --output-format csv -d $OUT/k$k -o c -- python3 $ROOT/tools/phase_inst_child.py $k $MODEL > $OUT/k$k.log 2>&1 || { echo "pass $k failed"; exit 1; }
  timeout -k 10 120 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAIT_INST_ANY SQ_WAVES --output-format csv -d $OUT/m$k -o c -- python3 $ROOT/tools/phase_inst_child.py $k $MODEL > $OUT/m$k.log 2>&1 || { echo "pass m$k failed"; exit 1; }
done
cd $ROOT; python3 tools/phase_instructions_report.py $OUT
