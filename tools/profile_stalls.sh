#!/bin/bash
# Stall attribution for the kernels whose vector pipes are busy less than 0.70 of the time at the plain rate
# (VERDICT r03 item 3): per workload@B three rocprofv3 --pmc passes of their own (never combined with a trace,
# MI355X_MICROARCH.md) that say where a wave's non-VALU cycles go -- waiting on any instruction, waiting on LDS,
# issuing scalar / LDS / branch / memory instructions.  Run on the GPU box from the repo root; results land in
# gpurun_out/prof_stalls/<workload>@<B>/ and tools/profile_stalls_summary.py condenses them into
# profiles/rNN/stalls.json + a Markdown table.
#   tools/profile_stalls.sh [workload@B ...]
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_stalls
mkdir -p $O
LIST="$@"
[ -z "$LIST" ] && LIST="cfg2_julia_1080p@48 cfg3_sierpinski_1080p@48 ref_julia_1080p@48 n1_genjulia_1080p@48 n2_bunny_1080p@8 n2_bunny_1080p@48 cfg2_julia_1080p@1"
cp $R/kifs_raymarching_amd/libkifs_hip.so.srchash $O/srchash.txt  # which library the counters belong to
cd /tmp && export TMPDIR=/tmp
PA="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU"
PB="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_VMEM SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS"
PC="SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_IFETCH SQ_INST_LEVEL_LDS SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE"
for item in $LIST; do
  w=${item%@*}; b=${item#*@}
  D=$O/$item; rm -rf $D; mkdir -p $D
  cp $O/srchash.txt $D/srchash.txt  # (the summaries skip directories left over from another library's run)
  B="$R/bench.py --workload $w --frames-per-launch $b --cpu-seconds 0 --no-secondary --settle-ms 20 --steps 16 --warmup 4"
  rocprofv3 --pmc $PA -d $D/pmc_a -o t --output-format csv -- python3 $B > $D/pmc_a.log 2>&1
  rocprofv3 --pmc $PB -d $D/pmc_b -o t --output-format csv -- python3 $B > $D/pmc_b.log 2>&1
  rocprofv3 --pmc $PC -d $D/pmc_c -o t --output-format csv -- python3 $B > $D/pmc_c.log 2>&1
  for f in $(find $D -name "*counter_collection.csv"); do
    head -1 $f > $f.render; grep "render_" $f | tail -600 >> $f.render || true; mv $f.render $f
  done
  echo "stalls profiled $item"
done
