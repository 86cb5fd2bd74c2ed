# Round check on the GPU box: GPU tests, then a few bench lines.
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
if [ -z "$SKIP_TESTS" ]; then
python -m pytest tests -m gpu -x -q --durations=5 > gpurun_out/r02/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"
tail -12 gpurun_out/r02/pytest_gpu.log
[ $rc -ne 0 ] && exit $rc
fi
b() { echo "== $*"; python bench.py --cpu-seconds 0 --no-secondary --steps 100 --warmup 20 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   ', d['value'], 'Mpix/s', d['ms_per_step'], 'ms/step', round(d['ms_per_step']/d['config']['frames_per_launch'],4), 'ms/frame kernel', d['roofline']['kernel'], d['roofline']['kernel_ms'])"; }
[ -n "$SKIP_BENCH" ] && exit 0
b "$@"
