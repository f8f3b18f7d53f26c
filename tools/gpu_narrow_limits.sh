#!/bin/bash
# Where the robot's narrowphase launch spends its time: the diagnostic build (build/libhb_stamps.so) with every portal search cut off after
# a support calls and every hull climb after b rounds (HB_MPR_LIMIT=a,b: wrong contacts, the point is the kernel's duration), under
# rocprofv3 --kernel-trace --stats.   usage: tools/gpu_narrow_limits.sh <outdir>
OUT=${1:-gpurun_out/narrow_limits}
mkdir -p $OUT; export TMPDIR=/tmp
for lim in 0,0 1,0 1,99 2,99 3,99 4,99 6,99 99,0 99,1 99,99; do
  HB_LIB=$PWD/build/libhb_stamps.so HB_MPR_LIMIT=$lim timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p_$lim -o t -- python3 tools/gpu_team_short.py > $OUT/log_$lim.txt 2>&1
  python3 - <<PY
import csv, glob
for f in glob.glob("$OUT/p_$lim/**/t_kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "hb_narrow" in r["Name"]: print("support calls, climb rounds <= %-6s %-22s avg %7.1f us" % ("$lim", r["Name"].split("(")[0], float(r["AverageNs"]) / 1e3))
PY
  grep 'by searches' $OUT/log_$lim.txt
done
