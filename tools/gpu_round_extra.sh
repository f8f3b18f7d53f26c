#!/bin/bash
# Second GPU-box session of a round (tools/gpu_round.sh is the first): the two-envs-per-wave kernel (batch-size sweep, PMC counters, per-stage
# instruction table, stage cycles), the robot's kernels (PMC counters), the per-env step latency distribution.
# usage: tools/gpu_round_extra.sh <tag>   (outputs under gpurun_out/<tag>/)
set -o pipefail
TAG=${1:-r04x}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
OUT=$OUT/duo_sizes timeout -k 10 400 bash tools/gpu_duo_sizes.sh > $OUT/duo_sizes.txt 2>&1; echo "duo sizes rc=$?" | tee -a $OUT/progress.log
timeout -k 10 600 bash tools/gpu_duo_counters.sh $OUT/duo 2 hb_step_duo_kernel 131072 > $OUT/duo_counters.log 2>&1; echo "duo counters rc=$?" | tee -a $OUT/progress.log
timeout -k 10 600 bash tools/gpu_team_counters.sh $OUT/team > $OUT/team_counters.log 2>&1; echo "team counters rc=$?" | tee -a $OUT/progress.log
PHASE_DUO=2 timeout -k 10 900 bash tools/gpu_phase_instructions.sh $OUT/phase_inst_duo > $OUT/phase_instructions_duo.txt 2>&1
HB_DUO=2 timeout -k 10 200 python tools/gpu_phase_profile.py 4096 > $OUT/phase_profile_duo.txt 2>&1
timeout -k 10 300 python tools/gpu_step_latency_dist.py > $OUT/step_latency_dist.txt 2>&1
echo done | tee -a $OUT/progress.log
