#!/bin/bash
# Rehearsal of bench.py's N > 1 control flow on ONE GPU: bench.py starts the ranks itself, they
# share cuda:0 and talk over gloo (device buffers staged through the host).  Every run ends with
# --check: rank 0's gathered frames must equal single-GPU renders, or bench.py exits non-zero.
# (RCCL itself needs one GPU per rank; the driver runs that at round end.)
set -e
run() { python3 bench.py --gpus $1 --backend gloo --share-device --check --steps 6 --warmup 3 --cpu-seconds 0 "${@:2}" \
        | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['config']['parallelism'], '| scaling', d['scaling'], '| frames/step', d['config']['frames_per_step'], '| check', d.get('gathered_frame_equals_single_gpu_frame'), '| Mpix/s', d['value'], '| per-rank kernel ms', d['per_rank_kernel_ms'], '| gather', d['config'].get('gather'), d['config'].get('tiles_sent_fraction'), '| root : peer weight', d['config'].get('root_weight'), ':', d['config'].get('peer_weight'), d['config'].get('root_weight_calibration'), '| secondary', sorted(d.get('secondary', {})))"; }
run 2 --frames-per-launch 8             # the default shape: stripes, weak scaling, rank 0's share calibrated
run 3 --frames-per-launch 2 --no-secondary --root-weight 1
run 4 --scaling strong --frames-per-launch 8 --no-secondary   # 8 frames per step, a quarter of each per rank
run 2 --root-weight 3 --frames-per-launch 8 --no-secondary
run 2 --workload cfg1_julia_256 --no-secondary   # default batch: 96 frames per step, two launches (64 + 32)
run 2 --shard bands --frames-per-launch 8 --no-secondary
run 2 --gather dense --frames-per-launch 8 --no-secondary     # every row as it is (round 2's first form)
run 3 --gather dense --root-weight 2 --frames-per-launch 4 --no-secondary
run 3 --workload cfg3_sierpinski_1080p --frames-per-launch 8 --no-secondary
run 2 --shard frames --frames-per-launch 8 --no-secondary
run 3 --shard frames --deliver root --frames-per-launch 2 --no-secondary
run 5 --frames-per-launch 8 --no-secondary --root-weight 1:2  # 40 frames per step, the peers with twice rank 0's share
