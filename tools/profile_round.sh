#!/bin/bash
# Round profile: bench lines, rocprofv3 kernel stats and the PMC passes for the headline workload.
# Run on the GPU box from the repo root; results land in gpurun_out/prof/ (tools/profile_summary.py
# turns them into the files kept under profiles/rNN/).  Counters are collected in passes of their own
# (--pmc only; never combined with a trace), as MI355X_MICROARCH.md prescribes.
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof
rm -rf $O; mkdir -p $O
cd $R
python bench.py > $O/cfg2_bench.json
echo "bench default done"
: > $O/other_workloads_bench.jsonl
[ -n "$KIFS_PROFILE_QUICK" ] && SWEEP="" || SWEEP="cfg1_julia_256 cfg3_sierpinski_1080p cfg4_julia_4096 ref_julia_1080p n1_genjulia_1080p n2_bunny_1080p"
for w in $SWEEP; do
  for b in 48 8 1; do
    python bench.py --workload $w --steps 100 --warmup 12 --cpu-seconds $([ $b = 48 ] && echo 4 || echo 0) --no-secondary --frames-per-launch $b >> $O/other_workloads_bench.jsonl
  done
  echo "bench $w done"
done
[ -n "$KIFS_PROFILE_QUICK" ] && SWEEP5="" || SWEEP5="cfg5_sierpinski_8k_orbit cfg5_sierpinski_8k_orbit_shadows"
for w in $SWEEP5; do
  python bench.py --workload $w --steps 40 --warmup 6 --cpu-seconds 0 --no-secondary --frames-per-launch 1 >> $O/other_workloads_bench.jsonl
  python bench.py --workload $w --steps 12 --warmup 3 --cpu-seconds 0 --no-secondary --frames-per-launch 4 >> $O/other_workloads_bench.jsonl
  # the workload as BASELINE.json names it: the WHOLE 120-frame orbit, every frame resident (15.9 GB), per launch size
  for b in 4 8 16 24; do
    python bench.py --workload $w --whole-orbit --steps 3 --warmup 1 --cpu-seconds 0 --no-secondary --frames-per-launch $b >> $O/other_workloads_bench.jsonl
  done
  echo "bench $w done"
done
cp $R/kifs_raymarching_amd/libkifs_hip.so.srchash $O/srchash.txt  # which library the counters belong to
cd /tmp && export TMPDIR=/tmp
B="$R/bench.py --cpu-seconds 0 --no-secondary"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o cfg2 -- python3 $B --steps 200 --warmup 20 > $O/trace.log 2>&1
echo "trace done"
rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o cfg2 --output-format csv -- python3 $B --steps 12 --warmup 4 > $O/pmc_write.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o cfg2 --output-format csv -- python3 $B --steps 12 --warmup 4 > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -d $O/pmc_sq -o cfg2 --output-format csv -- python3 $B --steps 12 --warmup 4 > $O/pmc_sq.log 2>&1
echo "pmc default done"
# the same three passes for 8 frames per launch (render_group_kernel) and one frame per launch (render_kernel)
for b in 8 1; do
  rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write_b$b -o cfg2 --output-format csv -- python3 $B --steps 12 --warmup 4 --frames-per-launch $b > $O/pmc_write_b$b.log 2>&1
  rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch_b$b -o cfg2 --output-format csv -- python3 $B --steps 12 --warmup 4 --frames-per-launch $b > $O/pmc_fetch_b$b.log 2>&1
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -d $O/pmc_sq_b$b -o cfg2 --output-format csv -- python3 $B --steps 12 --warmup 4 --frames-per-launch $b > $O/pmc_sq_b$b.log 2>&1
done
echo "pmc b8 b1 done"
find $O -name "*.csv" | head -40
