cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
rm -f gpurun_out/r02/sweep_raw.jsonl
for gt in 0 1 2; do
  KIFS_TUNING=1 KIFS_GROUP_TILES=$gt python tools/cliff_sweep.py --quick --tag group_tiles=$gt --out gpurun_out/r02/sweep_raw.jsonl > /dev/null 2>gpurun_out/r02/sweep_$gt.err || exit 1
  echo "done $gt"
done
python tools/cliff_sweep.py --quick --tag default --out gpurun_out/r02/sweep_raw.jsonl > /dev/null 2>>gpurun_out/r02/sweep_d.err
wc -l gpurun_out/r02/sweep_raw.jsonl
