#!/usr/bin/env python3
"""Prices lane occupancy of the generalised-Julia march on a CPU replay: the oracle's
per-step orbit trip counts (kor_march_trace) for the rays of the 32 x 8 tiles that hold the fractal, 64 rays per wave in
the kernels' order (8 x 8 blocks), against three executions of the same per-ray operation sequences:

  lockstep   all lanes take march step i together; the wave pays max_lanes(trips_i) orbit trips + one step tail
  decoupled  every lane runs its own steps inside a round of R march steps: per loop iteration the wave pays one
             orbit trip (if any lane is in its orbit) + one step tail (if any lane finishes or starts a step)
  ideal      sum over lanes / 64

Costs in instructions: T per orbit trip, E per step tail (finish the estimate, hit test, advance, bound test, start
the next orbit), O for a step outside the bounding sphere."""
import ctypes as C
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import oracle as O  # noqa: E402
import kifs_raymarching_amd as K  # noqa: E402
from kifs_raymarching_amd.configs import WORKLOADS, orbit_camera  # noqa: E402


def traces(key, pose, tiles=60, seed=0):
    w = WORKLOADS[key]
    ub = K.uniform_bytes
    s = O.from_bytes(O.Screen, ub(w.screen.into_buffer_data()))
    c = O.from_bytes(O.Camera, ub(orbit_camera(w, pose).into_buffer_data()))
    o = O.from_bytes(O.Options, ub(w.gui.into_buffer_data()))
    it = O.iters(*w.iters)
    calls, inside, inner = O.render_ray_costs(s, c, o, it)
    H, Wd = calls.shape
    # 8 x 8 blocks with any ray that enters the bounding sphere
    by, bx = np.nonzero(inside[:H // 8 * 8, :Wd // 8 * 8].reshape(H // 8, 8, Wd // 8, 8).max(axis=(1, 3)) > 0)
    rng = np.random.default_rng(seed)
    pick = rng.permutation(len(by))[:tiles]
    fn = O.lib().kor_march_trace
    fn.restype = C.c_int
    M = int(w.gui.max_iterations)
    out = []
    for b in pick:
        tr = np.zeros((64, M), dtype=np.uint8)
        n = np.zeros(64, dtype=np.int32)
        for l in range(64):
            x, y = bx[b] * 8 + l % 8, by[b] * 8 + l // 8
            n[l] = fn(C.byref(s), C.byref(c), C.byref(o), C.byref(it), int(x), int(y),
                      tr[l].ctypes.data_as(C.POINTER(C.c_uint8)), M)
        out.append((tr, n, inside[by[b] * 8:by[b] * 8 + 8, bx[b] * 8:bx[b] * 8 + 8].ravel() > 0))
    return out


def price(blocks, T, E, O_, R):
    lock = dec = ideal = 0.0
    for tr, n, _ in blocks:
        steps = int(n.max())
        live = np.arange(tr.shape[1])[None, :] < n[:, None]
        # lockstep: per step the wave pays the longest orbit + the tail (outside-only steps: O)
        for i in range(steps):
            k = tr[live[:, i], i]
            lock += (T * int(k.max()) + E) if k.max() > 0 else O_
        ideal += (T * float(tr[live].sum()) + E * float((tr[live] > 0).sum()) + O_ * float((tr[live] == 0).sum())) / 64.0
        # decoupled inside rounds of R steps
        for r0 in range(0, steps, R):
            seg = [list(tr[l, r0:min(int(n[l]), r0 + R)]) for l in range(64)]
            seg = [sg for sg in seg if sg]
            pos = [0] * len(seg)            # step index inside the round
            left = [sg[0] for sg in seg]    # trips left in the current step
            while any(p < len(sg) for p, sg in zip(pos, seg)):
                orbit = tail = False
                for j, sg in enumerate(seg):
                    if pos[j] >= len(sg):
                        continue
                    if left[j] > 0:
                        orbit = True
                        left[j] -= 1
                    if left[j] == 0:  # finishes (or is outside): tail now, next step starts
                        tail = True
                        pos[j] += 1
                        if pos[j] < len(sg):
                            left[j] = sg[pos[j]]
                dec += (T if orbit else 0) + (E if tail else 0)
    return lock, dec, ideal


if __name__ == "__main__":
    key = sys.argv[1] if len(sys.argv) > 1 else "n1_genjulia_1080p"
    blocks = []
    for pose in (0, 40, 80):
        blocks += traces(key, pose, tiles=40, seed=pose)
    tr_all = np.concatenate([b[0][np.arange(b[0].shape[1])[None, :] < b[1][:, None]] for b in blocks])
    ins = tr_all[tr_all > 0]
    print(f"{key}: {len(blocks)} blocks; steps inside the sphere: mean trips {ins.mean():.2f}, max {ins.max()}, "
          f"share of steps inside {len(ins) / len(tr_all):.3f}")
    for T, E, O_ in ((190, 100, 45), (150, 100, 45), (190, 60, 45)):
        for R in (8, 16, 32):
            lock, dec, ideal = price(blocks, T, E, O_, R)
            print(f"T={T} E={E} O={O_} R={R}: lockstep {lock / ideal:.2f} x ideal, decoupled {dec / ideal:.2f} x ideal, "
                  f"decoupled / lockstep {dec / lock:.3f}")
