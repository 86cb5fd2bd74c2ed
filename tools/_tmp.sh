cd $GRAFT_REPO_ROOT
b() { echo -n "${ENVV[*]} : "; env "${ENVV[@]}" python bench.py --cpu-seconds 0 --no-secondary --steps 150 --warmup 20 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['workload'], d['config']['frames_per_launch'], d['value'], 'Mpix/s', round(d['ms_per_step']/d['config']['frames_per_launch'],4), 'ms/frame', d['roofline']['kernel'], d['roofline']['kernel_ms'])"; }
for r in 16 4 6 8 10 12 24 32; do ENVV=(KIFS_TUNING=1 KIFS_ROUND_STEPS=$r); b; done
for r in 16 8 4; do ENVV=(KIFS_TUNING=1 KIFS_ROUND_STEPS=$r); b --workload cfg3_sierpinski_1080p; done
