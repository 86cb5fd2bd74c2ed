#!/bin/bash
# Throw-away A/B of bunny_sdf_coop builds (KIFS_COOP_VARIANT: bit 0 branch-free sine, bit 1 weights loaded ahead of
# the barriers), libraries prebuilt under build_variants/ and loaded from there (KIFS_LIB_VARIANT: the tree's own library
# is never replaced).  GPU box, repo root.
O=gpurun_out/r03; mkdir -p $O; : > $O/sweep_bunny_variants.txt
run() { # label env... batch
  b=$1; shift
  r=$(env KIFS_TUNING=1 KIFS_LIB_VARIANT=$PWD/build_variants/libkifs_v$V.so "$@" python bench.py --workload n2_bunny_1080p --steps 40 --warmup 8 --cpu-seconds 0 --no-secondary --frames-per-launch $b | grep "^{" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')
  echo "v=$V batch=$b $* : $r" | tee -a $O/sweep_bunny_variants.txt
}
for V in 0 1 2 3; do
  run 48 KIFS_GROUP_TILES=2
  run 48 KIFS_GROUP_TILES=1
  run 16 KIFS_GROUP_TILES=1
  run 8 KIFS_GROUP_TILES=1
  run 1 KIFS_BUNNY_BLOCK=1
  run 1 KIFS_BUNNY_BLOCK=0
done
V=3
for b in 2 4 8; do
  run $b KIFS_ROUND_STEPS=0 KIFS_BUNNY_BLOCK=1
  run $b KIFS_ROUND_STEPS=0 KIFS_BUNNY_BLOCK=0
done
