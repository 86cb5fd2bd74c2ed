#!/usr/bin/env python3
"""What one host thread pays per step of kifs_multi_render_batch_async when it drives N devices.

A gpurun box has one GPU, so the device is listed N times (peer-copy transport); the frames are made tiny
(256 x 64, the full view count of a real step: 48 N frames) so that the GPU work is a few tens of microseconds
and the step rate is set by the HOST: N render launches (+ view-table uploads beyond 64 views), N - 1 packs with
their count read-backs, N - 1 transfers, the erases and scatters, the event waits.  That figure must stay well
under a real step's ~1.1 ms for one thread to keep 8 GPUs busy.  A second line per N renders the real 1080p
headline the same way (N x the work on the one GPU): its frames must still equal single-device frames.

    python tools/multi_host_cost.py [N ...]
"""
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

import torch  # noqa: E402

import kifs_raymarching_amd as K  # noqa: E402
from kifs_raymarching_amd.configs import WORKLOADS, orbit_camera  # noqa: E402


def run(n, screen, frames_per_step, steps, w, gather="sparse"):
    cams = [K.camera_array([orbit_camera(w, (k * frames_per_step + i) % 120).into_buffer_data() for i in range(frames_per_step)])
            for k in range(4)]
    with K.MultiGraphicState([0] * n, screen, w.camera, w.gui, iters=w.iters) as mg:
        mg.set_gather(gather, "copy")
        bufs = [torch.zeros((frames_per_step, screen.height, screen.width, 4), dtype=torch.uint8, device="cuda:0")
                for _ in range(2)]
        for k in range(6):
            mg.render_batch_async(bufs[k % 2], cams[k % 4], untouched=k >= 2)
        mg.wait_all()
        torch.cuda.synchronize()
        mg.stats(reset=True)
        t0 = time.perf_counter()
        in_call = 0.0
        for k in range(steps):
            t = time.perf_counter()
            mg.render_batch_async(bufs[k % 2], cams[k % 4], untouched=True)
            in_call += time.perf_counter() - t
        mg.wait_all()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        st = mg.stats()
        return {"devices_listed": n, "width": screen.width, "height": screen.height, "frames_per_step": frames_per_step,
                "gather": gather, "steps": steps, "ms_per_step": round(dt / steps * 1e3, 4),
                "ms_in_submit_per_step": round(in_call / steps * 1e3, 4),
                "gpixel_s": round(frames_per_step * screen.width * screen.height * steps / dt / 1e9, 2),
                "bytes_into_root_per_step": int(st["bytes_received"] / max(1, st["steps"])),
                "shard_kernel_ms": [round(s[3], 4) for s in mg.shards()]}


def main():
    ns = [int(a) for a in sys.argv[1:]] or [2, 4, 8]
    w = WORKLOADS["cfg2_julia_1080p"]
    for n in ns:
        tiny = run(n, K.ScreenData(256, 64), min(48 * n, K.MAX_BATCH), 200, w)
        tiny["what"] = "host-bound: tiny frames, a real step's view count"
        print(json.dumps(tiny), flush=True)
        real = run(n, w.screen, min(48 * n, K.MAX_BATCH) // n, 30, w)
        real["what"] = "the 1080p headline, 48 frames per step split over the listed devices (one GPU does all of it)"
        print(json.dumps(real), flush=True)


if __name__ == "__main__":
    main()
