#!/bin/bash
# Round length of the bunny's throughput kernels (KIFS_ROUND_STEPS), per batch size.  GPU box, repo root.
O=gpurun_out/r03; mkdir -p $O; : > $O/sweep_bunny_rounds.txt
for b in ${BATCHES:-48 24 16}; do for coop in ${COOPS:-1 0}; do for r in ${ROUNDS:-1 2 3 4 6 8}; do
  v=$(KIFS_TUNING=1 KIFS_BUNNY_COOP=$coop KIFS_ROUND_STEPS=$r python bench.py --workload n2_bunny_1080p --steps 40 --warmup 8 --cpu-seconds 0 --no-secondary --frames-per-launch $b | grep "^{" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"]["kernel"])')
  echo "batch=$b coop=$coop rounds=$r : $v" | tee -a $O/sweep_bunny_rounds.txt
done; done; done
