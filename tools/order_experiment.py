"""Feeds the kernel tile orders derived from measured per-tile cost and times them."""
import sys; sys.path.insert(0,'.')
import numpy as np, torch
import kifs_raymarching_amd as K
from kifs_raymarching_amd.configs import WORKLOADS
key = sys.argv[1] if len(sys.argv) > 1 else "cfg2_julia_1080p"
w = WORKLOADS[key]
gs = K.GraphicState(0, screen_data=w.screen, camera_data=w.camera, gui_data=w.gui); gs.set_iters(*w.iters)
dev = torch.zeros((w.screen.height, w.screen.width, 4), dtype=torch.uint8, device="cuda:0")
def timeit(n=30):
    ts = []
    for _ in range(n):
        gs.render(out=dev); ts.append(gs.last_kernel_ms())
    ts.sort(); return ts[len(ts)//2]
gs.render(out=dev)
base_order = gs.debug_get_tile_order()
print(key, "tiles", len(base_order), "centre-first order:", round(timeit()*1e3,1), "us")
gs.debug_counters(True); gs.render(out=dev); rec = gs.debug_wave_records().copy(); gs.debug_counters(False)
n = len(base_order)
cost = rec[:4*n,0].reshape(n,4).max(1).astype(np.int64)       # per dispatched block (in base order)
steps = (rec[:4*n,2] & np.uint64(0xffffffff)).reshape(n,4).max(1).astype(np.int64)
by_cost = np.argsort(-cost, kind="stable")
def try_order(name, perm):
    gs.debug_set_tile_order(base_order[perm]); print(f"  {name:58s} {timeit()*1e3:7.1f} us")
try_order("cost-descending", by_cost)
for nh in (128, 256, 512, 1024):
    heavy, rest = by_cost[:nh], by_cost[nh:]
    try_order(f"{nh} heaviest, then lightest-first", np.concatenate([heavy, rest[::-1]]))
    try_order(f"{nh} heaviest, then rest in centre-first order", np.concatenate([heavy, np.sort(rest)]))
    # heavy tiles spaced out: one heavy followed by 7 light, repeated
    light = rest[::-1]
    k = 7
    inter = []
    li = 0
    for i in range(nh):
        inter.append(heavy[i]); inter.extend(light[li:li+k]); li += k
    inter.extend(light[li:])
    try_order(f"{nh} heaviest interleaved 1:7 with lightest", np.array(inter))
try_order("row-major (no ordering)", np.argsort(base_order, kind="stable"))
gs.debug_set_tile_order(base_order)
print("heavy stats: blocks with >=64 steps:", int((steps>=64).sum()), ">=128:", int((steps>=128).sum()), ">=250:", int((steps>=250).sum()))
