#!/bin/bash
# AddressSanitizer + UBSan pass over the host-side model compiler (CPU only; GPU sanitizers are not available on this pool):
# every test MJCF and the compiled benchmark model through compile -> save -> load -> save, plus malformed inputs.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${TMPDIR:-/tmp}/hb_asan
mkdir -p $OUT
g++ -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -std=c++17 -o $OUT/hb_compile $ROOT/tools/hb_compile.cpp \
    $ROOT/humanoid_mujoco_amd/csrc/mjcf.cpp $ROOT/humanoid_mujoco_amd/csrc/setconst.cpp $ROOT/humanoid_mujoco_amd/csrc/model_io.cpp $ROOT/humanoid_mujoco_amd/csrc/mesh.cpp
fail=0
for f in $ROOT/tests/models/*.xml $ROOT/humanoid_mujoco_amd/assets/*.hbm; do
  $OUT/hb_compile $f $OUT/a.hbm > $OUT/log.txt 2>&1 || { echo "FAIL $f"; cat $OUT/log.txt; fail=1; }
  $OUT/hb_compile $OUT/a.hbm $OUT/b.hbm >> $OUT/log.txt 2>&1 || { echo "FAIL reload $f"; fail=1; }
  cmp -s $OUT/a.hbm $OUT/b.hbm || { echo "round trip differs: $f"; fail=1; }
  grep -q "AddressSanitizer\|runtime error" $OUT/log.txt && { echo "sanitizer report: $f"; cat $OUT/log.txt; fail=1; }
done
printf '<mujoco><worldbody><body><joint type="ball"/><geom size="1"/></body></worldbody></mujoco>' > $OUT/bad1.xml
printf '<mujoco><worldbody><body><geom type="box" size="1 1 1"/></body></worldbody>' > $OUT/bad2.xml
printf '<mujoco><worldbody><body childclass="zzz"><joint/><geom size="0.1"/></body></worldbody></mujoco>' > $OUT/bad3.xml
printf '<mujoco><worldbody><body><joint/><geom size="0.1"' > $OUT/bad4.xml
head -c 300 $ROOT/humanoid_mujoco_amd/assets/humanoid27.hbm > $OUT/bad5.hbm
for f in $OUT/bad1.xml $OUT/bad2.xml $OUT/bad3.xml $OUT/bad4.xml $OUT/bad5.hbm; do
  if $OUT/hb_compile $f $OUT/c.hbm > $OUT/log.txt 2>&1; then echo "accepted a malformed input: $f"; fail=1; fi
  grep -q "AddressSanitizer\|runtime error" $OUT/log.txt && { echo "sanitizer report: $f"; cat $OUT/log.txt; fail=1; }
done
# the fp64 oracle under the same sanitizers, driven by its own tests
gcc -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -shared -fPIC -o $OUT/liboracle_asan.so $ROOT/oracle/mjstep_oracle.c -lm -lpthread
( cd $ROOT && ASAN_OPTIONS=detect_leaks=0 LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) HB_ORACLE_SO=$OUT/liboracle_asan.so \
    python -m pytest tests/test_oracle_kat.py tests/test_oracle_newton.py tests/test_oracle_convex.py tests/test_oracle_contact_order.py -x -q > $OUT/oracle.txt 2>&1 ) || { echo "oracle under sanitizers: FAIL"; tail -20 $OUT/oracle.txt; fail=1; }
grep -q "AddressSanitizer\|runtime error" $OUT/oracle.txt && { echo "sanitizer report in the oracle run"; fail=1; }
[ $fail = 0 ] && echo "asan/ubsan host pass: clean (model compiler, oracle)"
exit $fail
