#!/usr/bin/env python3
"""How many network evaluations do the bunny's throughput kernels really run?  (diagnosis tool, r03)

Needs a library built with the counting hooks (kifs_bunny.hpp, #ifdef KIFS_EVAL_COUNT -- not in the shipped build):
    make -C kifs_raymarching_amd/csrc -B EXTRA=-DKIFS_EVAL_COUNT OUT=../../build_variants/libkifs_count.so
    (GPU box)  cp build_variants/libkifs_count.so kifs_raymarching_amd/libkifs_hip.so
               KIFS_TUNING=1 KIFS_BUNNY_COOP=1 python tools/bunny_eval_counts.py 48
Per launch of B frames: wave-level calls of the estimate, how many of them ran the network (some lane inside the unit
ball), and how many rays were inside at those (per ray, not per lane: the four-lanes form counts lanes / 4, the
four-waves form counts wave 0 only).  r03, 48 frames (profiles/r03/bunny_eval_counts.jsonl):
    four waves per 64 rays : 1.869 M wave evaluations, 44.0 of 64 rays inside on average   (69 %)
    four lanes per ray, T=2: 1.336 M wave evaluations, 13.4 of 16                          (84 %)
i.e. the coarser chunk costs 1.40 x the vector instructions for the same rays (PMC: 950 against 696 per pixel) --
part chunks at the end of every round, rays that stopped inside a round, and far rays riding along."""
import ctypes as C, sys, json
sys.path.insert(0, ".")
import torch
import kifs_raymarching_amd as K
from kifs_raymarching_amd import _lib
from kifs_raymarching_amd.configs import WORKLOADS, orbit_camera
w = WORKLOADS["n2_bunny_1080p"]
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
f = _lib.lib.kifs_debug_eval_counts
f.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
gs.render_batch_async(outs, cams, stream=st); st.synchronize()
f(out, 1)
gs.render_batch_async(outs, cams, stream=st); st.synchronize()
f(out, 1)
print(json.dumps({"B": B, "kernel": gs.debug_last_kernel(), "tiles": gs.debug_last_group_tiles(), "rounds": gs.debug_last_round_steps(),
                  "wave_evals_network": out[0], "rays_inside_at_network_evals": out[1], "wave_calls": out[2], "lanes_at_calls": out[3]}))
