#!/bin/bash
# Same-box A/B of two builds of the library (KIFS_LIB_VARIANT: loaded from where they lie, the tree's own library is
# never replaced): alternating runs of bench.py per workload@B, value and kernel per run.  GPU box, repo root.
#   tools/ab_variants.sh OUT.txt A.so B.so workload@B [workload@B ...]
O=$1; A=$2; B=$3; shift 3
mkdir -p $(dirname $O); : > $O
for item in "$@"; do
  w=${item%@*}; b=${item#*@}
  for rep in 1 2 3; do
    for v in $A $B; do
      r=$(KIFS_TUNING=1 KIFS_LIB_VARIANT=$PWD/$v python bench.py --workload $w --steps 60 --warmup 10 --cpu-seconds 0 --no-secondary --frames-per-launch $b 2>/dev/null | grep "^{" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"]["kernel"])')
      echo "$item $(basename $v) : $r" | tee -a $O
    done
  done
done
