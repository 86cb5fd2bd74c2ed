#!/bin/bash
# Round profile: bench lines, rocprofv3 kernel stats and the PMC passes for the headline workload.
# Run on the GPU box from the repo root; results land in gpurun_out/prof/ (copy into profiles/rNN/).
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof
mkdir -p $O
cd $R
python bench.py > $O/cfg2_bench.json
python bench.py --frames-per-launch 1 --cpu-seconds 0 > $O/cfg2_single_frame_bench.json
: > $O/other_workloads_bench.jsonl
for w in cfg1_julia_256 cfg3_sierpinski_1080p cfg4_julia_4096 ref_julia_1080p n1_genjulia_1080p n2_bunny_1080p; do
  python bench.py --workload $w --steps 200 --warmup 20 --cpu-seconds 4 >> $O/other_workloads_bench.jsonl
  python bench.py --workload $w --steps 200 --warmup 20 --cpu-seconds 0 --frames-per-launch 1 >> $O/other_workloads_bench.jsonl
done
for w in cfg5_sierpinski_8k_orbit cfg5_sierpinski_8k_orbit_shadows; do
  python bench.py --workload $w --orbit --steps 60 --warmup 6 --cpu-seconds 0 --frames-per-launch 1 >> $O/other_workloads_bench.jsonl
  python bench.py --workload $w --orbit --steps 15 --warmup 3 --cpu-seconds 0 --frames-per-launch 4 >> $O/other_workloads_bench.jsonl
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o cfg2 -- python3 $R/bench.py --steps 200 --warmup 20 --cpu-seconds 0 > $O/trace.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o cfg2 --output-format csv -- python3 $R/bench.py --steps 20 --warmup 4 --cpu-seconds 0 > $O/pmc_write.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o cfg2 --output-format csv -- python3 $R/bench.py --steps 20 --warmup 4 --cpu-seconds 0 > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $O/pmc_sq -o cfg2 --output-format csv -- python3 $R/bench.py --steps 20 --warmup 4 --cpu-seconds 0 > $O/pmc_sq.log 2>&1
ls -R $O | head -40
