#!/bin/bash
export KIFS_TUNING=1  # the overrides below are honoured only with this set
# Residency / feedback sweep of one workload: tools/sweep_pad.sh <workload> [bench flags]
w=$1; shift
run() { python bench.py --workload $w --steps 60 --warmup 12 --cpu-seconds 0 "$@" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel_ms'])"; }
echo "== $w default"; run "$@"
echo "== $w feedback off"; KIFS_TILE_FEEDBACK=0 run "$@"
echo "== $w feedback all"; KIFS_TILE_FEEDBACK=2 run "$@"
for pad in 0 18000 36000 50000 72000 100000; do echo "== $w pad $pad"; KIFS_LDS_PAD=$pad run "$@"; done
