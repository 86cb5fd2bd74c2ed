# Round check on the GPU box: GPU tests, then a few bench lines.
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
if [ -z "$SKIP_TESTS" ]; then
python -m pytest tests -m gpu -x -q > gpurun_out/r02/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"
tail -5 gpurun_out/r02/pytest_gpu.log
[ $rc -ne 0 ] && exit $rc
fi
b() { echo "== ${ENVV[*]} $*"; env "${ENVV[@]}" python bench.py --cpu-seconds 0 --no-secondary --steps 100 --warmup 20 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   ', d['value'], 'Mpix/s', d['ms_per_step'], 'ms/step kernel', d['roofline']['kernel'], d['roofline']['kernel_ms'])"; }
ENVV=(A=1); b; b --frames-per-launch 16; b --frames-per-launch 32; b --camera fixed; b --frames-per-launch 1
for w in cfg3_sierpinski_1080p cfg4_julia_4096 ref_julia_1080p n1_genjulia_1080p n2_bunny_1080p; do
  b --workload $w --steps 40 --warmup 8;  b --workload $w --steps 40 --warmup 8 --frames-per-launch 32
done
b --workload cfg5_sierpinski_8k_orbit --steps 10 --warmup 3 --frames-per-launch 4
b --workload cfg5_sierpinski_8k_orbit --steps 10 --warmup 3 --frames-per-launch 1
