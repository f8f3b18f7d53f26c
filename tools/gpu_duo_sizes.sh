mkdir -p ${OUT:-gpurun_out/duo_sizes}
# FOLD=1 (default here): one launch per step call, the comparison this table is about; FOLD=256: the library's default, calls folded
for n in 2048 4096 8192 16384 32768; do
  for duo in 1 0; do
    HB_DUO=$duo timeout -k 10 200 python bench.py --steps 300 --warmup 20 --envs-per-gpu $n --fold ${FOLD:-1} --no-cpu-baseline --no-team --no-newton --no-rollout > ${OUT:-gpurun_out/duo_sizes}/b_${n}_$duo.json 2> ${OUT:-gpurun_out/duo_sizes}/b_${n}_$duo.err
    python -c "
import json; d=json.load(open('${OUT:-gpurun_out/duo_sizes}/b_${n}_$duo.json')); print('envs $n duo $duo fold ${FOLD:-1}: %.3e env-steps/s, %.1f us/step pipelined, %.1f us unpipelined launch' % (d['value'], 1e3*d['ms_per_step'], d['roofline']['single_step']['avg_launch_us']))"
  done
done
