#!/usr/bin/env python3
"""GPU box: does it pay to spread the empty tiles of a big launch over the order instead of leaving them for the end?
render_wave_kernel, 48 orbit frames: the order the feedback found (heavy tiles first, the ~90 % without a ray last: their
350 000 waves are the launch's last 0.1 ms of dispatch) against the same order with the empty tiles dealt in between.
    python tools/order_interleave_probe.py [workload] [frames]"""
import json
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import kifs_raymarching_amd as K  # noqa: E402
from kifs_raymarching_amd.configs import WORKLOADS, orbit_camera  # noqa: E402

key = sys.argv[1] if len(sys.argv) > 1 else "cfg2_julia_1080p"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 48
w = WORKLOADS[key]
W, H = w.screen.width, w.screen.height
gs = K.GraphicState(0, w.screen, w.camera, w.gui)
gs.set_iters(*w.iters)
frames = torch.zeros((B, H, W, 4), dtype=torch.uint8, device="cuda:0")
torch.cuda.synchronize()
st = torch.cuda.Stream()
gs.set_profiling(1)
cams = [orbit_camera(w, k) for k in range(B)]  # the same 48 poses every launch: the orders are compared on equal work


def run(n):
    ms = []
    for _ in range(n):
        if B == 1:
            gs.render_async(frames[0], stream=st)
        else:
            gs.render_batch_async([frames[i] for i in range(B)], cams, stream=st)
        st.synchronize()
        ms.append(gs.profile_read()[1])
    return ms


run(60)  # feedback settles, clocks too
order = np.array(gs.debug_get_tile_order(), dtype=np.uint32)
heavy_n = int(sys.argv[3]) if len(sys.argv) > 3 else 1000  # the order's head: every tile that can hold a ray, with room
head, rest = order[:heavy_n], order[heavy_n:]


def interleaved(every):
    per = int(np.ceil(len(rest) / (len(head) / every)))
    mixed, r = [], 0
    for i in range(0, len(head), every):
        mixed.extend(head[i:i + every])
        mixed.extend(rest[r:r + per])
        r += per
    mixed.extend(rest[r:])
    mixed = np.array(mixed, dtype=np.uint32)
    assert len(mixed) == len(order) and len(np.unique(mixed)) == len(order)
    return mixed


orders = {"heavy_first": order, "empty_first": np.concatenate([rest, head])}
for every in (2, 8, 32, 128):
    orders[f"every_{every}"] = interleaved(every)
res = {k: [] for k in orders}
for rnd in range(4):  # alternate: every order four times, 15 launches each
    for k, o in orders.items():
        gs.debug_set_tile_order(o)
        res[k].append(float(np.median(run(15))))
print(json.dumps({"workload": key, "frames": B, "kernel": gs.debug_last_kernel(), "heavy_n": heavy_n,
                  "ms": {k: [round(x, 4) for x in v] for k, v in res.items()}}))
