#!/bin/bash
# The round's whole profile in one go, on the GPU box, from the repo root: the headline's bench lines, rocprofv3 kernel
# stats and counter passes (profile_round.sh), every workload's kernel stats and counters (profile_workloads.sh), the stall
# attribution passes (profile_stalls.sh).  Afterwards, in the container: python tools/profile_report.py rNN.
# About 12 minutes of box time.  `quick` as the first argument skips the long bench sweep of profile_round.sh's other
# workloads (KIFS_PROFILE_QUICK=1).
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
[ "$1" = quick ] && export KIFS_PROFILE_QUICK=1
bash $R/tools/profile_round.sh > $R/gpurun_out/profile_round.log 2>&1; echo "profile_round done"
bash $R/tools/profile_workloads.sh > $R/gpurun_out/profile_workloads.log 2>&1; echo "profile_workloads done"
bash $R/tools/profile_stalls.sh cfg2_julia_1080p@48 cfg2_julia_1080p@8 cfg2_julia_1080p@1 cfg3_sierpinski_1080p@48 cfg3_sierpinski_1080p@8 \
     ref_julia_1080p@48 n1_genjulia_1080p@48 n2_bunny_1080p@8 n2_bunny_1080p@48 > $R/gpurun_out/profile_stalls.log 2>&1; echo "profile_stalls done"
