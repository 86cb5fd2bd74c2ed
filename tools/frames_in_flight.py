#!/usr/bin/env python3
"""Throughput with F independent frames in flight on one GPU (one context + stream each).
Usage: frames_in_flight.py [workload] [steps]"""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

import kifs_raymarching_amd as K  # noqa: E402
from kifs_raymarching_amd.configs import WORKLOADS  # noqa: E402

key = sys.argv[1] if len(sys.argv) > 1 else "cfg2_julia_1080p"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 600
w = WORKLOADS[key]
W, H = w.screen.width, w.screen.height
dev = torch.device("cuda", 0)
for F in (1, 2, 3, 4, 6, 8):
    ctxs, streams, outs = [], [], []
    for f in range(F):
        gs = K.GraphicState(0, screen_data=w.screen, camera_data=w.camera, gui_data=w.gui)
        gs.set_iters(*w.iters)
        if w.extensions:
            gs.set_extensions(**w.extensions)
        ctxs.append(gs)
        streams.append(torch.cuda.Stream(device=dev))
        outs.append(torch.zeros((H, W, 4), dtype=torch.uint8, device=dev))
    for k in range(16 * F):
        ctxs[k % F].render_async(outs[k % F], stream=streams[k % F], y0=0, y1=H, encode=1)
    torch.cuda.synchronize()
    for gs in ctxs:
        gs.set_profiling(8)
    t0 = time.perf_counter()
    for k in range(steps):
        ctxs[k % F].render_async(outs[k % F], stream=streams[k % F], y0=0, y1=H, encode=1)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    kms = [gs.profile_read()[1] for gs in ctxs]
    same = all(bool(torch.equal(outs[0], o)) for o in outs[1:])
    print(f"{key} F={F}: {dt / steps * 1e3:.4f} ms/frame, {W * H * steps / dt / 1e6:.0f} Mpix/s, "
          f"kernel ms {min(kms):.4f}..{max(kms):.4f}, frames identical {same}", flush=True)
    for gs in ctxs:
        gs.close()
