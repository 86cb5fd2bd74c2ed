#!/usr/bin/env python3
"""Prices the SHAPE of the re-queuing workgroup on a CPU replay (profiles/r04: SQ_WAIT_ANY is 40-50 % of the wave-cycles
of every render_group_kernel launch -- waves parked at the round's barrier): T tiles of 32 x 8 pixels share one ray
queue, W waves draw chunks of 64 rays from it (tickets), a round is R march steps, survivors are re-queued.  Per chunk
and step the wave pays max_lanes(trips) x I_TRIP + I_TAIL if a lane is inside the bounding sphere, else I_OUT.  Reports,
per (T, W): the share of wave-time the waves are BUSY (the rest is the barrier), lane use while busy, and the product --
useful lane-instructions per wave-slot-instruction -- relative to T = 2, W = 4 (what ships)."""
import ctypes as C
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import oracle as O  # noqa: E402
import kifs_raymarching_amd as K  # noqa: E402
from kifs_raymarching_amd.configs import WORKLOADS, orbit_camera  # noqa: E402


def tile_traces(key, pose, n_groups, tiles_per_group, seed=0):
    """Groups of `tiles_per_group` horizontally adjacent 32 x 8 tiles around the fractal: per ray its per-step trips."""
    w = WORKLOADS[key]
    ub = K.uniform_bytes
    s = O.from_bytes(O.Screen, ub(w.screen.into_buffer_data()))
    c = O.from_bytes(O.Camera, ub(orbit_camera(w, pose).into_buffer_data()))
    o = O.from_bytes(O.Options, ub(w.gui.into_buffer_data()))
    it = O.iters(*w.iters)
    calls, inside, inner = O.render_ray_costs(s, c, o, it)
    H, Wd = calls.shape
    ty, tx = np.nonzero(inside[:H // 8 * 8, :Wd // 32 * 32].reshape(H // 8, 8, Wd // 32, 32).max(axis=(1, 3)) > 0)
    rng = np.random.default_rng(seed)
    fn = O.lib().kor_march_trace
    fn.restype = C.c_int
    M = int(w.gui.max_iterations)
    groups = []
    for b in rng.permutation(len(ty))[:n_groups]:
        rays = []
        x0 = min(int(tx[b]), Wd // 32 - tiles_per_group) * 32
        for t in range(tiles_per_group):
            for ly in range(8):
                for lx in range(32):
                    x, y = x0 + 32 * t + lx, int(ty[b]) * 8 + ly
                    if calls[y, x] <= 1 and inside[y, x] == 0 and False:
                        continue
                    tr = np.zeros(M, dtype=np.uint8)
                    n = fn(C.byref(s), C.byref(c), C.byref(o), C.byref(it), x, y, tr.ctypes.data_as(C.POINTER(C.c_uint8)), M)
                    # rays that can never reach the sphere are culled at set-up by the kernels: approximate by "never inside
                    # and gone within 3 steps"
                    if inside[y, x] == 0 and n <= 3:
                        continue
                    rays.append(tr[:n].copy())
        groups.append(rays)
    return groups


def simulate(rays, tiles, W, R, I_TRIP, I_TAIL, I_OUT, per_chunk=120):
    """One workgroup: returns (sum of round times, busy wave-time, useful lane-instructions)."""
    queue = list(range(len(rays)))
    step0 = 0
    total = busy = useful = 0.0
    while queue:
        chunks = [queue[i:i + 64] for i in range(0, len(queue), 64)]
        costs, survivors = [], []
        for ch in chunks:
            cost = per_chunk
            for i in range(step0, step0 + R):
                k = [int(rays[r][i]) for r in ch if i < len(rays[r])]
                if not k:
                    break
                mx = max(k)
                cost += (I_TRIP * mx + I_TAIL) if mx > 0 else I_OUT
                useful += sum((I_TRIP * v + I_TAIL) if v > 0 else I_OUT for v in k) / 64.0
            costs.append(cost)
            survivors.append([r for r in ch if len(rays[r]) > step0 + R])
        # tickets: the next chunk goes to the wave that is free first
        free = [0.0] * W
        for cst in costs:
            j = int(np.argmin(free))
            free[j] += cst
        total += max(free)
        busy += sum(free)
        queue = [r for sv in survivors for r in sv]
        step0 += R
    return total, busy, useful


if __name__ == "__main__":
    key = sys.argv[1] if len(sys.argv) > 1 else "n1_genjulia_1080p"
    I_TRIP, I_TAIL, I_OUT = (170, 100, 45) if "genjulia" in key else (14, 110, 48)
    R = 16
    print(f"{key}: I_TRIP {I_TRIP} I_TAIL {I_TAIL} I_OUT {I_OUT}, rounds of {R}")
    base = None
    for T in (1, 2, 4, 8):
        groups = []
        for pose in (0, 40, 80):
            groups += tile_traces(key, pose, max(4, 24 // T), T, seed=pose + T)
        for W in (1, 2, 4):
            tot = bus = use = 0.0
            for rays in groups:
                if not rays:
                    continue
                t, b, u = simulate(rays, T, W, R, I_TRIP, I_TAIL, I_OUT)
                tot += t * W
                bus += b
                use += u
            eff = use / tot
            if T == 2 and W == 4:
                base = eff
            print(f"T={T} W={W}: busy {bus / tot:.3f} of the wave-time, lanes while busy {use / bus:.3f}, useful per wave-slot {eff:.3f}", flush=True)
    print("relative to T=2 W=4:", base)
