#!/bin/bash
# usage: tools/sweep_pad.sh "<pads>" "<workloads>"
for pad in $1; do for wl in $2; do
  KIFS_LDS_PAD=$pad timeout -k 10 200 python bench.py --workload $wl --steps 200 --warmup 20 --cpu-seconds 0 2>/dev/null > /tmp/sweep.json
  python -c "import json; d=json.load(open('/tmp/sweep.json')); print('pad', $pad, d['config']['workload'], d['value'], 'Mpix/s', d['roofline']['kernel_ms'], 'ms')"
done; done
