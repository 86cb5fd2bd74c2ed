set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
python -m pytest tests -m gpu -x -q > gpurun_out/r02/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r02/pytest_gpu.log
tail -5 gpurun_out/r02/pytest_gpu.log
python bench.py --steps 200 --warmup 20 > gpurun_out/r02/bench_default.json 2> gpurun_out/r02/bench_default.err; echo "bench rc=$?"
cat gpurun_out/r02/bench_default.json
bash tools/rehearse_multi.sh > gpurun_out/r02/rehearse.log 2>&1; echo "rehearse rc=$?"
cat gpurun_out/r02/rehearse.log | tail -20
