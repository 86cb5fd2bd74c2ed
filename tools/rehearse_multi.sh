#!/bin/bash
# Rehearsal of bench.py's N > 1 control flow on ONE GPU: bench.py starts the ranks itself, they
# share cuda:0 and talk over gloo (device buffers staged through the host).  Every run ends with
# --check: rank 0's gathered frames must equal single-GPU renders, or bench.py exits non-zero.
# (RCCL itself needs one GPU per rank; the driver runs that at round end.)
set -e
run() { python3 bench.py --gpus $1 --backend gloo --share-device --check --steps 6 --warmup 3 --cpu-seconds 0 "${@:2}" \
        | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['config']['parallelism'], '| scaling', d['scaling'], '| frames/step', d['config']['frames_per_step'], '| check', d.get('gathered_frame_equals_single_gpu_frame'), '| Mpix/s', d['value'], '| per-rank kernel ms', d['per_rank_kernel_ms'], '| secondary', sorted(d.get('secondary', {})))"; }
run 2                                   # the default: stripes, weak scaling, 16 frames per step
run 3 --frames-per-launch 2 --no-secondary
run 4 --scaling strong --no-secondary   # 8 frames per step, a quarter of each per rank
run 2 --root-weight 3 --no-secondary
run 2 --shard bands --no-secondary
run 2 --shard frames --no-secondary
run 3 --shard frames --deliver root --frames-per-launch 2 --no-secondary
run 5 --frames-per-launch 8 --no-secondary   # 40 frames per step: two launches per step (32 + 8)
