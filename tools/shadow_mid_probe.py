#!/usr/bin/env python3
"""Mid-size and lone launches WITH the soft-shadow extension (BASELINE names it for config 5 only, which bench.py covers):
ms per launch for 1 / 8 / 16 frames of the 1080p Sierpinski and Julia workloads, orbit cameras.  r03, same-box A/B of the
pooled secondary rays in render_group_kernel: Sierpinski x8 0.514 -> 0.474 ms, x16 0.727 -> 0.683; Julia x8 0.398 -> 0.354."""
import sys
sys.path.insert(0, ".")
import torch
import kifs_raymarching_amd as K
from kifs_raymarching_amd.configs import WORKLOADS, orbit_camera
for key in ("cfg3_sierpinski_1080p", "cfg2_julia_1080p"):
    w = WORKLOADS[key]
    W, H = w.screen.width, w.screen.height
    for B in (1, 8, 16):
        gs = K.GraphicState(0, screen_data=w.screen, camera_data=w.camera, gui_data=w.gui)
        gs.set_iters(*w.iters)
        gs.set_extensions(soft_shadow=True, shadow_steps=64, shadow_k=8.0, shadow_t0=0.02, shadow_max_t=10.0)
        frames = torch.zeros((B, H, W, 4), dtype=torch.uint8, device="cuda:0")
        torch.cuda.synchronize()
        st = torch.cuda.Stream()
        outs = K.DevicePointers([frames[i] for i in range(B)])
        poses = [K.camera_array([orbit_camera(w, r * B + k).into_buffer_data() for k in range(B)]) for r in range(50)]
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for r in range(50):
            if r == 10:
                t0.record(st)
            if B == 1:
                gs.set_raw_uniforms(camera=poses[r][0]); gs.render_async(frames[0], stream=st)
            else:
                gs.render_batch_async(outs, poses[r], stream=st)
        t1.record(st); st.synchronize()
        print(f"{key} shadows x{B}: {gs.debug_last_kernel()} {t0.elapsed_time(t1)/40:.4f} ms per launch", flush=True)
        gs.close()
