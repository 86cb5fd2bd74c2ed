export KIFS_TUNING=1
for b in 16 20 24 28 32 40 48; do
  for v in w2off w2lds; do
    r=$(KIFS_LIB_VARIANT=$PWD/build_variants/libkifs_$v.so KIFS_BUNNY_COOP=0 KIFS_GROUP_TILES=2 python bench.py --workload n2_bunny_1080p --steps 40 --warmup 8 --cpu-seconds 0 --no-secondary --frames-per-launch $b 2>/dev/null | grep "^{" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"]["kernel"])')
    echo "batch=$b $v pairs : $r"
  done
  r=$(KIFS_LIB_VARIANT=$PWD/build_variants/libkifs_w2off.so KIFS_BUNNY_COOP=1 python bench.py --workload n2_bunny_1080p --steps 40 --warmup 8 --cpu-seconds 0 --no-secondary --frames-per-launch $b 2>/dev/null | grep "^{" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"]["kernel"])')
  echo "batch=$b coop : $r"
done
