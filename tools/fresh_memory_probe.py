#!/usr/bin/env python3
"""Does a render launch run slower into memory that has just come from hipMalloc?  (tools/expected_scaling.py saw a
rank's shard of 384 frames take 2.4 ms the first time and 1.2 ms when the allocator handed the same block back.)
Times rank 0's in-place shard of an N = 8 step (384 frames, 3.2 GB) into: a fresh block, the same block again, a
block the caching allocator returns, a fresh block after empty_cache(), and a fresh block that was written once
by a fill kernel first.   python tools/fresh_memory_probe.py"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

import kifs_raymarching_amd as K  # noqa: E402
from kifs_raymarching_amd.configs import WORKLOADS, orbit_camera  # noqa: E402

w = WORKLOADS["cfg2_julia_1080p"]
W, H = w.screen.width, w.screen.height
N, F = 8, 384
gs = K.GraphicState(0, screen_data=w.screen, camera_data=w.camera, gui_data=w.gui)
gs.set_iters(*w.iters)
a = torch.cuda.Stream()
cams = K.camera_array([orbit_camera(w, k).into_buffer_data() for k in range(F)])
mine, _ = K.shard_stripes(H, 0, N)


def timed(frames, reps=10, label=""):
    outs = K.DevicePointers([frames[i] for i in range(F)])
    out = []
    for _ in range(3):
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record(a)
        for _ in range(reps):
            gs.render_shard_async(outs, cams, mine, in_place=True, stream=a)
        t1.record(a)
        a.synchronize()
        out.append(round(t0.elapsed_time(t1) / reps, 3))
    print(f"{label:60s}: ms per launch, three batches of {reps}: {out}", flush=True)


small = torch.zeros((F, 8, W, 4), dtype=torch.uint8, device="cuda:0")  # warm the kernel itself up on a small target
so = K.DevicePointers([small[i] for i in range(F)])
for _ in range(20):
    gs.render_shard_async(so, cams, [0], stream=a)
a.synchronize()
A = torch.empty((F, H, W, 4), dtype=torch.uint8, device="cuda:0")
timed(A, label="fresh torch.empty block (never written)")
timed(A, label="the same block again")
del A
B = torch.empty((F, H, W, 4), dtype=torch.uint8, device="cuda:0")
timed(B, label="block handed back by the caching allocator")
del B
torch.cuda.empty_cache()
C = torch.zeros((F, H, W, 4), dtype=torch.uint8, device="cuda:0")
timed(C, label="fresh block after empty_cache(), zero-filled by torch")
del C
torch.cuda.empty_cache()
D = torch.empty((F, H, W, 4), dtype=torch.uint8, device="cuda:0")
timed(D, label="fresh block after empty_cache(), never written")
