#!/bin/bash
# Launch shapes of the bunny's mid-size batches (rules::BUNNY_PAIR_FROM / BUNNY_COOP_FROM in kifs_schedule.cpp) after the
# chunk tickets: four lanes per ray with one or two tiles per workgroup, or four waves per 64 rays.  GPU box, repo root.
export KIFS_TUNING=1
O=${1:-gpurun_out/r04/sweep_bunny_shapes.txt}; mkdir -p $(dirname $O); : > $O
run() { # batch env...
  b=$1; shift
  r=$(env "$@" python bench.py --workload n2_bunny_1080p --steps 40 --warmup 8 --cpu-seconds 0 --no-secondary --frames-per-launch $b 2>/dev/null | grep "^{" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"]["kernel"])')
  echo "batch=$b $* : $r" | tee -a $O
}
for b in 4 8 12 16 24 32; do
  run $b KIFS_BUNNY_COOP=0 KIFS_GROUP_TILES=1
  run $b KIFS_BUNNY_COOP=0 KIFS_GROUP_TILES=2
  run $b KIFS_BUNNY_COOP=1
done
