"""Times every 8-row band of a workload on its own (few workgroups -> the slowest wave's
critical path dominates) and a few parameter variations.  Diagnostic tool."""
import sys, json
sys.path.insert(0, '.')
import numpy as np, torch
import kifs_raymarching_amd as K
from kifs_raymarching_amd.configs import WORKLOADS

key = sys.argv[1] if len(sys.argv) > 1 else "cfg2_julia_1080p"
w = WORKLOADS[key]
gs = K.GraphicState(0, screen_data=w.screen, camera_data=w.camera, gui_data=w.gui)
gs.set_iters(*w.iters)
W, H = w.screen.width, w.screen.height
dev = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda:0")

def t_band(y0, y1, reps=3):
    best = 1e9
    for _ in range(reps):
        gs.render(out=dev[y0:y1], y0=y0, y1=y1)
        best = min(best, gs.last_kernel_ms())
    return best

full = t_band(0, H, 5)
bands = [(y, t_band(y, min(y + 8, H))) for y in range(0, H, 8)]
top = sorted(bands, key=lambda b: -b[1])[:6]
print(f"{key}: full frame {full*1e3:.1f} us; sum of 8-row bands {sum(b[1] for b in bands)*1e3:.1f} us; "
      f"max band {top[0][1]*1e3:.1f} us at y={top[0][0]}")
print("top bands:", [(y, round(t*1e3,1)) for y, t in top])
print("background band (y=0):", round(bands[0][1]*1e3, 1), "us")
ymax = top[0][0]
# parameter variations on the slowest band
for sdf in (0, 1, 6, 12, 24):
    gs.set_iters(sdf, w.iters[1], w.iters[2])
    print(f"  sdf_iters={sdf:3d}: slowest band {t_band(ymax, ymax+8)*1e3:8.1f} us, full {t_band(0,H)*1e3:8.1f} us")
gs.set_iters(*w.iters)
for mi in (32, 64, 128, 256, 512):
    g = K.GuiData(**{**w.gui.__dict__, "max_iterations": mi})
    gs.update_options(g)
    print(f"  max_iterations={mi:4d}: slowest band {t_band(ymax, ymax+8)*1e3:8.1f} us, full {t_band(0,H)*1e3:8.1f} us")
