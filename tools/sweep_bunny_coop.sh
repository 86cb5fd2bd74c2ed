#!/bin/bash
# The bunny's throughput path: four waves per 64 rays (render_bunny_coop_kernel) against four lanes per ray
# (render_group_kernel<0, 5, T>), per batch size, round length and tiles per workgroup.  GPU box, repo root.
O=gpurun_out/r03; mkdir -p $O; : > $O/sweep_bunny_coop.txt
run() { # coop rounds tiles batch
  r=$(KIFS_TUNING=1 KIFS_BUNNY_COOP=$1 KIFS_ROUND_STEPS=$2 KIFS_GROUP_TILES=$3 python bench.py --workload n2_bunny_1080p --steps 40 --warmup 8 --cpu-seconds 0 --no-secondary --frames-per-launch $4 | grep "^{" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')
  echo "coop=$1 rounds=$2 tiles=$3 batch=$4 : $r" | tee -a $O/sweep_bunny_coop.txt
}
for b in 2 4 8 16 24 32 48; do
  for coop in 1 0; do
    for rounds in 8 16 32; do
      for tiles in 1 2; do
        run $coop $rounds $tiles $b
      done
    done
  done
done
