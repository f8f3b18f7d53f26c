#!/bin/bash
# rocprofv3 passes over the staged step of the reference's own robot (tools/gpu_team_short.py: 4096 envs, 30 single-step launches): kernel
# trace, then PMC passes (own runs, no trace domains beside them) for the narrowphase, pose and step kernels.
# usage: tools/gpu_team_counters.sh <out dir under gpurun_out/>
OUT=${1:-gpurun_out/team_counters}
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 tools/gpu_team_short.py > $OUT/short.txt 2>&1; echo "trace rc=$?"
timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/sq -o t -- python3 tools/gpu_team_short.py > $OUT/sq.txt 2>&1; echo "sq rc=$?"
timeout -k 10 200 rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES --output-format csv -d $OUT/valu -o t -- python3 tools/gpu_team_short.py > $OUT/valu.txt 2>&1; echo "valu rc=$?"
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_ANY SQ_INSTS_FLAT SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES --output-format csv -d $OUT/mem -o t -- python3 tools/gpu_team_short.py > $OUT/mem.txt 2>&1; echo "mem rc=$?"
# (FETCH_SIZE and WRITE_SIZE in separate passes, as /opt/skills/guides/MI355X_MICROARCH.md prescribes: together the profiler aborts)
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/hbm -o t -- python3 tools/gpu_team_short.py > $OUT/hbm.txt 2>&1; echo "hbm fetch rc=$?"
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/hbmw -o t -- python3 tools/gpu_team_short.py > $OUT/hbmw.txt 2>&1; echo "hbm write rc=$?"
python3 tools/team_counters_report.py $OUT | tee $OUT/report.txt
