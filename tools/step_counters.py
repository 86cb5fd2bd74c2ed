import sys; sys.path.insert(0,'.')
import numpy as np
import kifs_raymarching_amd as K
from kifs_raymarching_amd.configs import WORKLOADS
for key in sys.argv[1:] or ["cfg2_julia_1080p"]:
    w = WORKLOADS[key]
    gs = K.GraphicState(0, screen_data=w.screen, camera_data=w.camera, gui_data=w.gui); gs.set_iters(*w.iters)
    gs.render()
    gs.debug_counters(True); gs.render(); ms = gs.last_kernel_ms(); rec = gs.debug_wave_records().copy(); c = gs.debug_counters(False)
    rec[:,2] &= np.uint64(0xffffffff)
    print("kernel ms while counting", ms)
    print(key, c, "ticks/fast step", c["fast_ticks"]/max(1,c["fast_steps"]))
    rec = rec[rec[:,0] > 0]
    order = np.argsort(-rec[:,0].astype(np.int64))
    print(" waves recorded", len(rec), "total wave-ticks", int(rec[:,0].sum()), "max", int(rec[:,0].max()))
    print(" slowest waves: total_ticks fast_ticks fast_steps general_steps | general ticks/step")
    for i in order[:12]:
        t, ft, fs, g = [int(v) for v in rec[i]]
        print("  wave", i, t, ft, fs, g, "|", round((t-ft)/max(1,g),1), "fast t/step", round(ft/max(1,fs),1))
    hist = np.histogram(rec[:,0], bins=[0,2e3,1e4,3e4,1e5,2e5,3e5,4e5,1e6])
    print(" histogram of wave ticks:", list(zip(hist[1][1:].astype(int), hist[0])))
