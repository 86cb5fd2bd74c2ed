#!/bin/bash
# Round length of the multi-chunk rounds (a tile's rays that fit one chunk are marched to the end since round 3):
# headline and friends, KIFS_ROUND_STEPS forced in turn.   tools/sweep_rounds_wave.sh
export KIFS_TUNING=1
run() { python bench.py --cpu-seconds 0 --no-secondary --steps 80 --warmup 10 "$@" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   ', d['config']['workload'], 'B', d['config']['frames_per_launch'], 'Mpix/s', d['value'], d['roofline']['kernel'])"; }
for k in 4 6 8 10 12 16 24; do
  echo "== KIFS_ROUND_STEPS=$k"
  export KIFS_ROUND_STEPS=$k
  run
  run --workload cfg3_sierpinski_1080p
  run --workload ref_julia_1080p
  run --workload cfg4_julia_4096 --steps 20 --frames-per-launch 16
done
