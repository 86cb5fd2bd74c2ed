#!/bin/bash
# Rehearsal of bench.py's N > 1 control flow on ONE GPU: ranks share cuda:0 and talk over gloo.
# (RCCL itself needs one GPU per rank; the driver runs that at round end.)
set -e
run() { python -m torch.distributed.run --nnodes=1 --nproc-per-node $1 --master-addr 127.0.0.1 --master-port $2 \
        bench.py --gpus $1 --backend gloo --share-device --check --steps 6 --warmup 3 --cpu-seconds 0 "${@:3}" \
        | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['config']['parallelism'], '| check', d.get('gathered_frame_equals_single_gpu_frame'), '| Mpix/s', d['value'])"; }
run 2 29611
run 2 29612 --deliver root
run 3 29613 --deliver root --frames-per-launch 2
run 2 29614 --shard bands
run 4 29615 --shard bands --orbit
