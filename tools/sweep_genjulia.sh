export KIFS_TUNING=1
run() { python bench.py --workload n1_genjulia_1080p --steps 30 --warmup 6 --cpu-seconds 0 --no-secondary --frames-per-launch $1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   B', d['config']['frames_per_launch'], 'Mpix/s', d['value'], 'ms/frame', round(d['ms_per_step']/d['config']['frames_per_launch'],4), d['roofline']['kernel'])"; }
for shape in 0 1 2; do for r in 4 8 16 32; do echo "== GROUP_TILES=$shape ROUND_STEPS=$r"; KIFS_GROUP_TILES=$shape KIFS_ROUND_STEPS=$r run 48; done; done
echo "== default"; unset KIFS_TUNING; run 48; run 96
