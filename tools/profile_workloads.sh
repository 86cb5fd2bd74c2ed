#!/bin/bash
# Counter evidence for EVERY workload (tools/profile_round.sh covers the headline in depth): per workload, at the
# batch size given, one rocprofv3 kernel-trace pass and three --pmc passes of their own (SQ + GRBM; WRITE_SIZE;
# FETCH_SIZE -- counters never share a run with a trace, MI355X_MICROARCH.md).  Run on the GPU box from the repo
# root; results land in gpurun_out/prof_wl/<workload>@<B>/ and tools/profile_workloads_summary.py condenses them
# into profiles/rNN/workloads_pmc.json + the table of profiles/rNN/README.md.
#   tools/profile_workloads.sh [workload@B ...]      (default: the list below)
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_wl
mkdir -p $O
LIST="$@"
[ -z "$LIST" ] && LIST="cfg2_julia_1080p@48 cfg3_sierpinski_1080p@48 cfg4_julia_4096@48 ref_julia_1080p@48 \
n1_genjulia_1080p@48 n2_bunny_1080p@48 n2_bunny_1080p@8 cfg2_julia_1080p@8 cfg5_sierpinski_8k_orbit@8 cfg5_sierpinski_8k_orbit_shadows@8 cfg1_julia_256@48"
cp $R/kifs_raymarching_amd/libkifs_hip.so.srchash $O/srchash.txt  # which library the counters belong to
cd /tmp && export TMPDIR=/tmp
SQ="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"
for item in $LIST; do
  w=${item%@*}; b=${item#*@}
  case $w in cfg5*|cfg4*) steps=8; warm=3;; *) steps=24; warm=6;; esac
  D=$O/$item; rm -rf $D; mkdir -p $D
  cp $O/srchash.txt $D/srchash.txt  # (the summaries skip directories left over from another library's run)
  B="$R/bench.py --workload $w --frames-per-launch $b --cpu-seconds 0 --no-secondary --settle-ms 20"
  rocprofv3 --kernel-trace --stats --output-format csv -d $D/trace -o t -- python3 $B --steps $((steps * 4)) --warmup $warm > $D/trace.log 2>&1
  rocprofv3 --pmc $SQ -d $D/pmc_sq -o t --output-format csv -- python3 $B --steps $steps --warmup $warm > $D/pmc_sq.log 2>&1
  rocprofv3 --pmc WRITE_SIZE -d $D/pmc_write -o t --output-format csv -- python3 $B --steps $steps --warmup $warm > $D/pmc_write.log 2>&1
  rocprofv3 --pmc FETCH_SIZE -d $D/pmc_fetch -o t --output-format csv -- python3 $B --steps $steps --warmup $warm > $D/pmc_fetch.log 2>&1
  # keep what travels back small: the per-dispatch rows of the render kernels only
  for f in $(find $D -name "*counter_collection.csv"); do
    head -1 $f > $f.render; grep "render_" $f | tail -400 >> $f.render || true; mv $f.render $f
  done
  find $D -name "*kernel_trace.csv" -delete
  echo "profiled $item"
done
