#!/usr/bin/env python3
"""What a rank of N renders per step of the default multi-GPU run (N x 48 frames, its stripes of each), on ONE
GPU: the step's launches at most 64 views each (the kernel argument's room) against one launch with the views in
a device table (KIFS_MAX_BATCH = 512).  Prints ms per step and the rate in whole-frame pixels per second.
    python tools/shard_batch_study.py [workload]"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

import kifs_raymarching_amd as K  # noqa: E402
from kifs_raymarching_amd.configs import WORKLOADS, orbit_camera  # noqa: E402

key = sys.argv[1] if len(sys.argv) > 1 else "cfg2_julia_1080p"
w = WORKLOADS[key]
W, H = w.screen.width, w.screen.height
gs = K.GraphicState(0, screen_data=w.screen, camera_data=w.camera, gui_data=w.gui)
gs.set_iters(*w.iters)
stream = torch.cuda.Stream()
for world in (1, 2, 4, 8):
    frames = 48 * world
    stripes, rows = K.shard_stripes(H, 1 % world, world)
    shards = torch.zeros((frames, rows, W, 4), dtype=torch.uint8, device="cuda:0")
    cams = [orbit_camera(w, k) for k in range(frames)]
    for per_launch in sorted({min(64, frames), min(frames, K.MAX_BATCH)}):
        def step():
            for a in range(0, frames, per_launch):
                gs.render_shard_async([shards[i] for i in range(a, min(frames, a + per_launch))],
                                      cams[a:a + per_launch], stripes, stream=stream)
        for _ in range(6):
            step()
        stream.synchronize()
        t0 = torch.cuda.Event(enable_timing=True)
        t1 = torch.cuda.Event(enable_timing=True)
        t0.record(stream)
        n = 20
        for _ in range(n):
            step()
        t1.record(stream)
        stream.synchronize()
        ms = t0.elapsed_time(t1) / n
        print(f"{key}: rank of {world}: {frames} frames x {rows} rows per step, {per_launch:3d} views per launch: "
              f"{ms:.3f} ms per step, {frames * rows * W / ms / 1e6:.1f} Gpixel/s, kernel shape {gs.debug_last_group_tiles()}",
              flush=True)
    del shards
