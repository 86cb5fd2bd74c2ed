#!/usr/bin/env python3
"""When does every workgroup of a render_group_kernel launch start and end?  (diagnosis tool, r04; needs a library built
with the timeline hook -- profiles/r04/timeline_hook.patch applied, -DKIFS_TIMELINE -- loaded through KIFS_TUNING=1
KIFS_LIB_VARIANT=...; GPU box.  The hook is kept as a patch: in the tree it would change the kernels' source hash.)
    python tools/group_timeline.py WORKLOAD FRAMES > timeline.json"""
import ctypes as C
import json
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import kifs_raymarching_amd as K  # noqa: E402
from kifs_raymarching_amd import _lib  # noqa: E402
from kifs_raymarching_amd.configs import WORKLOADS, orbit_camera  # noqa: E402

key, B = sys.argv[1], int(sys.argv[2])
w = WORKLOADS[key]
gs = K.GraphicState(0, screen_data=w.screen, camera_data=w.camera, gui_data=w.gui)
gs.set_iters(*w.iters)
W, H = w.screen.width, w.screen.height
frames = torch.zeros((B, H, W, 4), dtype=torch.uint8, device="cuda:0")
torch.cuda.synchronize()
cams = [orbit_camera(w, k) for k in range(B)] if B > 1 else None
st = torch.cuda.Stream()
gs.set_profiling(1)
f = _lib.lib.kifs_debug_timeline
f.argtypes = [C.c_void_p, C.c_int]
ms = []
for rep in range(9):
    if B > 1:
        gs.render_batch_async([frames[i] for i in range(B)], cams, stream=st)
    else:
        gs.render_async(frames[0], stream=st)
    st.synchronize()
    ms.append(gs.profile_read()[1])
tiles = ((W + 31) // 32) * ((H + 7) // 8)
T = max(1, gs.debug_last_group_tiles())
n = min(1 << 17, (tiles + T - 1) // T * B)
buf = np.zeros((n, 4), dtype=np.uint64)
assert f(buf.ctypes.data, n) == 0
t0 = buf[:, 0].min()
start, rounds, end = [(buf[:, k] - t0).astype(np.float64) / 100.0 for k in range(3)]  # microseconds
rays = (buf[:, 3] & np.uint64(0xffffffff)).astype(np.int64)
xcc = (buf[:, 3] >> np.uint64(56)).astype(np.int64)
heavy = rays > 0
span = end.max()
out = {"workload": key, "frames": B, "kernel": gs.debug_last_kernel(), "group_tiles": T, "kernel_ms": ms, "workgroups": int(n),
       "with_rays": int(heavy.sum()), "span_us": float(span)}
# the launch in 20 slices: workgroups with rays running, started, all workgroups running
edges = np.linspace(0, span, 21)
out["running_with_rays"] = [int(((start <= e) & (end > e) & heavy).sum()) for e in edges]
out["running_all"] = [int(((start <= e) & (end > e)).sum()) for e in edges]
out["in_rounds_with_rays"] = [int(((start <= e) & (rounds > e) & heavy).sum()) for e in edges]
life = (end - start)[heavy]
out["lifetime_us"] = {"mean": float(life.mean()), "p50": float(np.median(life)), "p90": float(np.percentile(life, 90)), "max": float(life.max())}
order = np.argsort(-life)
hidx = np.nonzero(heavy)[0][order[:12]]
out["longest"] = [{"workgroup": int(i), "start_us": float(start[i]), "rounds_end_us": float(rounds[i]), "end_us": float(end[i]), "rays": int(rays[i]),
                   "xcc": int(xcc[i])} for i in hidx]
late = np.nonzero(heavy)[0][np.argsort(-end[heavy])[:12]]
out["last_to_end"] = [{"workgroup": int(i), "start_us": float(start[i]), "end_us": float(end[i]), "rays": int(rays[i])} for i in late]
out["heavy_start_us_percentiles"] = [float(np.percentile(start[heavy], p)) for p in (0, 10, 50, 90, 100)]
out["per_xcc_last_end_us"] = [float(end[xcc == x].max()) if (xcc == x).any() else None for x in range(8)]
print(json.dumps(out))
