#!/usr/bin/env python3
"""Sweep of frame sizes x camera distances x scenes x launch shapes, looking for performance cliffs
between the points the launch heuristics of kifs_api.cpp were tuned on.

    python tools/cliff_sweep.py [--out FILE] [--tag TAG] [--quick]

For every point it prints one JSON line: kernel time per launch (HIP events inside the library,
mean of the timed launches), ms per frame, the projected-disc pixel count of the scene's bounding
sphere (what the heuristics see) and ns per pixel-in-disc -- the figure that should vary smoothly
from point to point.  `--tag` labels the lines (e.g. the KIFS_GROUP_TILES setting the process was
started with: tuning overrides are read once per process).  GPU only.
"""
import argparse
import json
import math
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

SIZES = [(1280, 720), (1920, 1080), (2560, 1440), (3840, 2160), (1000, 1000)]
DISTANCES = [2.0, 3.0, 5.0, 8.0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--tag", default="default")
    ap.add_argument("--quick", action="store_true", help="fewer launches per point")
    ap.add_argument("--batches", default="1,8,32")
    args = ap.parse_args()
    import torch

    import kifs_raymarching_amd as K
    from kifs_raymarching_amd.configs import JULIA_C

    FG, PS = K.FractalGroup, K.PrimitiveShape
    scenes = {
        "julia": (K.GuiData(fractal_group=FG.JuliaSet, constant=JULIA_C, max_iterations=256), (12, 10, 10), 2.0),
        "sierpinski": (K.GuiData(primitive_shape=PS.SierpinskiTetrahedron, max_iterations=256), (100, 10, 16), 2.0),
    }
    batches = [int(b) for b in args.batches.split(",")]
    stream = torch.cuda.Stream()
    out = open(args.out, "a") if args.out else None
    for name, (gui, iters, bound) in scenes.items():
        for (W, H) in SIZES:
            for d in DISTANCES:
                gs = K.GraphicState(0, screen_data=K.ScreenData(W, H), gui_data=gui,
                                    camera_data=K.CameraData(origin_distance=d))
                gs.set_iters(*iters)
                R = bound + gui.epsilon
                if d * d > R * R * 1.0001:
                    r_px = 0.5 * H * math.sqrt(R * R / (d * d - R * R))
                    disc_px = min(W * H, math.pi * r_px * r_px)
                else:
                    disc_px = W * H
                for B in batches:
                    if B * W * H * 4 > (3 << 30):
                        continue
                    frames = torch.zeros((B, H, W, 4), dtype=torch.uint8, device="cuda:0")
                    cams = [K.CameraData(origin_distance=d, phi=0.05 * i, theta=0.3) for i in range(B)]
                    launches = (6 if args.quick else 16) if B * W * H < 4e7 else 5

                    def launch():
                        if B == 1:
                            gs.set_camera(cams[0])
                            gs.render_async(frames[0], stream=stream)
                        else:
                            gs.render_batch_async([frames[i] for i in range(B)], cams, stream=stream)
                    for _ in range(4):
                        launch()
                    stream.synchronize()
                    gs.set_profiling(1)
                    for _ in range(launches):
                        launch()
                    n, mean, lo, hi = gs.profile_read()
                    gs.set_profiling(0)
                    rec = {"tag": args.tag, "scene": name, "width": W, "height": H, "distance": d, "batch": B,
                           "kernel_ms": round(mean, 5), "kernel_ms_min": round(lo, 5), "ms_per_frame": round(mean / B, 5),
                           "disc_tiles": round(disc_px / 256.0, 1), "ns_per_disc_pixel": round(mean / B * 1e6 / disc_px, 3),
                           "mpix_s": round(B * W * H / mean / 1e3, 1), "round_steps": gs.debug_last_round_steps(),
                           "group_tiles": gs.debug_last_group_tiles()}
                    line = json.dumps(rec)
                    print(line, flush=True)
                    if out:
                        out.write(line + "\n")
                        out.flush()
                    del frames
                gs.close()


if __name__ == "__main__":
    main()
