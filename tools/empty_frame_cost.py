#!/usr/bin/env python3
"""Cost of frames in which (almost) every ray is culled at set-up: what a launch pays for
workgroup dispatch, ray set-up and the pixel store alone.  Usage: empty_frame_cost.py [distance]"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import kifs_raymarching_amd as K  # noqa: E402
from kifs_raymarching_amd.configs import WORKLOADS  # noqa: E402

dist = float(sys.argv[1]) if len(sys.argv) > 1 else 900.0
for key in ["cfg2_julia_1080p", "cfg3_sierpinski_1080p", "n2_bunny_1080p", "n1_genjulia_1080p", "cfg4_julia_4096", "cfg5_sierpinski_8k_orbit"]:
    w = WORKLOADS[key]
    for d in (w.camera.origin_distance, dist):
        cam = K.CameraData(origin_distance=d, phi=w.camera.phi, theta=w.camera.theta)
        with K.GraphicState(0, screen_data=w.screen, camera_data=cam, gui_data=w.gui) as gs:
            gs.set_iters(*w.iters)
            ms = []
            for _ in range(6):
                gs.render_to_device() if hasattr(gs, "render_to_device") else gs.render()
                ms.append(gs.last_kernel_ms())
            tiles = ((w.screen.width + 31) // 32) * ((w.screen.height + 7) // 8)
            best = min(ms[2:])
            print(f"{key:28s} distance {d:6.1f}: kernel {best:.4f} ms, {tiles} tiles, "
                  f"{best * 1e6 / tiles:.1f} ns/tile")
