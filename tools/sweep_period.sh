for p in 3 4 8 16; do
export KIFS_TUNING=1  # the overrides below are honoured only with this set
  for w in "cfg2_julia_1080p --orbit" "cfg4_julia_4096" "ref_julia_1080p --orbit"; do
    echo "period $p $w"
    KIFS_FEEDBACK_PERIOD=$p python bench.py --workload $w --steps 720 --warmup 48 --cpu-seconds 0 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel_ms'])"
  done
done
