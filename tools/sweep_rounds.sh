#!/bin/bash
export KIFS_TUNING=1  # the overrides below are honoured only with this set
# Round length of the ray re-queuing (KIFS_ROUND_STEPS; 0 = one wave per block): tools/sweep_rounds.sh
run() { python bench.py --cpu-seconds 0 "$@" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   ', d['config']['workload'], 'B', d['config']['frames_per_launch'], 'ms/step', d['ms_per_step'], 'Mpix/s', d['value'])"; }
for k in 0 8 16 32 64; do
  echo "== KIFS_ROUND_STEPS=$k"
  export KIFS_ROUND_STEPS=$k
  run --steps 300
  run --steps 500 --frames-per-launch 1
  run --workload cfg4_julia_4096 --steps 60 --frames-per-launch 1
  run --workload cfg3_sierpinski_1080p --steps 200
  run --workload cfg5_sierpinski_8k_orbit --orbit --steps 40 --warmup 4 --frames-per-launch 1
  run --workload ref_julia_1080p --steps 200
done
