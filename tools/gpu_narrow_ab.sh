#!/bin/bash
# The robot's narrowphase after a change: parity tests of the convex path, the kernels' launch durations (rocprofv3 kernel trace of
# tools/gpu_team_short.py: unpipelined launches, the pair form hb_narrow2_kernel) and the step throughput (tools/gpu_team_bench.py).
# usage: tools/gpu_narrow_ab.sh <outdir>
OUT=${1:-gpurun_out/narrow_ab}
mkdir -p $OUT; export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_convex.py tests/test_gpu_staged.py tests/test_gpu_team_env.py -x -q > $OUT/pytest.txt 2>&1; tail -3 $OUT/pytest.txt
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o t -- python3 tools/gpu_team_short.py > $OUT/team_short.txt 2>&1
cat $OUT/team_short.txt
python3 - <<PY
import csv, glob
for f in glob.glob("$OUT/prof/**/t_kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "hb_" in r["Name"]: print("%-60s calls %6s avg %9.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
timeout -k 10 300 python tools/gpu_team_bench.py > $OUT/team_bench.txt 2>&1; cat $OUT/team_bench.txt
