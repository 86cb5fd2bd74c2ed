#!/bin/bash
# Launch-shape thresholds of the re-queuing path (rules::PAIR_FROM_* / WAVE_FROM_* in kifs_schedule.cpp) re-swept with
# round 4's render_group_kernel (chunks drawn by ticket): every shape forced in turn per workload and batch size.
# KIFS_GROUP_TILES: 0 = one wave per tile, 1 / 2 = tiles per 256-thread workgroup.  GPU box, repo root.
export KIFS_TUNING=1
O=${1:-gpurun_out/r04/sweep_group_shapes.txt}; mkdir -p $(dirname $O); : > $O
run() { # workload batch tiles
  r=$(KIFS_GROUP_TILES=$3 python bench.py --workload $1 --steps 40 --warmup 8 --cpu-seconds 0 --no-secondary --frames-per-launch $2 2>/dev/null | grep "^{" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"]["kernel"])')
  echo "$1 batch=$2 tiles=$3 : $r" | tee -a $O
}
for b in 4 8 12 16 24 32; do for t in 1 2 0; do run n1_genjulia_1080p $b $t; done; done
for b in 2 4 8 12 16 20 24 32; do for t in 1 2 0; do run cfg2_julia_1080p $b $t; done; done
for b in 4 8 16 24 32 48; do for t in 1 2 0; do run cfg3_sierpinski_1080p $b $t; done; done
for b in 1 2 4; do for t in 1 2 0; do run cfg4_julia_4096 $b $t; done; done
