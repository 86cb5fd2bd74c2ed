#!/usr/bin/env python3
"""Generalised Julia set: how often does an orbit step fall back to the general elementary functions, and how many lanes
does a step carry?  (diagnosis tool, r03; needs a library built with the counting hooks, as tools/bunny_eval_counts.py:
    make -C kifs_raymarching_amd/csrc -B EXTRA=-DKIFS_EVAL_COUNT OUT=../../build_variants/libkifs_count.so
    (GPU box)  cp build_variants/libkifs_count.so kifs_raymarching_amd/libkifs_hip.so; python tools/genjulia_pow_counts.py 48)
r03, n1_genjulia_1080p: 48 frames per launch 13.1 M wave-level orbit steps, 1.9 % repeated with the general functions (a lane
with an exponent beyond 2^+-128 or a NaN), 34.0 of 64 lanes in a step; one frame: 278 k steps, 2.5 %, 32.3 lanes.
The fallback is not where the time goes; the lanes are (trip counts differ, rays end): tools/genjulia_orbit_study.py."""
import ctypes as C, sys, json
sys.path.insert(0, ".")
import torch
import kifs_raymarching_amd as K
from kifs_raymarching_amd import _lib
from kifs_raymarching_amd.configs import WORKLOADS, orbit_camera
w = WORKLOADS["n1_genjulia_1080p"]
B = int(sys.argv[1])
gs = K.GraphicState(0, screen_data=w.screen, camera_data=w.camera, gui_data=w.gui)
gs.set_iters(*w.iters)
W, H = w.screen.width, w.screen.height
frames = torch.zeros((B, H, W, 4), dtype=torch.uint8, device="cuda:0")
torch.cuda.synchronize()
cams = K.camera_array([orbit_camera(w, k).into_buffer_data() for k in range(B)])
outs = K.DevicePointers([frames[i] for i in range(B)])
st = torch.cuda.Stream()
out = (C.c_ulonglong * 8)()
f = _lib.lib.kifs_debug_pow_counts
f.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
gs.render_batch_async(outs, cams, stream=st); st.synchronize()
f(out, 1)
gs.render_batch_async(outs, cams, stream=st); st.synchronize()
f(out, 1)
print(json.dumps({"B": B, "kernel": gs.debug_last_kernel(), "wave_steps": out[0], "repeated_general": out[1], "lanes_per_step": out[2] / max(1, out[0])}))
