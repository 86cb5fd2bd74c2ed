#!/usr/bin/env python3
"""KIFS workloads, one wave per tile: rounds with re-queuing (today) against lanes that take the next ray of the
tile's pool as soon as theirs ends.  CPU replay of the march (NumPy, the kernel's culls) -> steps per ray; every
march step of these scenes costs a wave the same (the SDF runs for all its lanes), so the price of a tile is its
wave-steps:
  rounds   : per round of R steps the live rays in queue order, 64 to a chunk; a chunk runs until its longest ray ends
             or the round does (one chunk left: to the end)
  refill   : 64 lanes, rays handed out in pixel order, a lane takes the next one when its ray ends (list scheduling)
  ideal    : sum of steps / 64
    python tools/lane_refill_study.py [workload] [round_steps] [tiles per pool]"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import oracle as O  # noqa: E402
from oracle import kifs_oracle_np as NP  # noqa: E402
import kifs_raymarching_amd as K  # noqa: E402
from kifs_raymarching_amd.configs import WORKLOADS  # noqa: E402

F = np.float32
key = sys.argv[1] if len(sys.argv) > 1 else "cfg3_sierpinski_1080p"
R = int(sys.argv[2]) if len(sys.argv) > 2 else 8
POOL = int(sys.argv[3]) if len(sys.argv) > 3 else 1  # tiles whose rays share a pool (consecutive tiles of a row)
w = WORKLOADS[key]
ub = K.uniform_bytes
s = NP.Scene(O.from_bytes(O.Screen, ub(w.screen.into_buffer_data())), O.from_bytes(O.Camera, ub(w.camera.into_buffer_data())),
             O.from_bytes(O.Options, ub(w.gui.into_buffer_data())), O.iters(*w.iters))
W, H = s.width, s.height
if W * H > 3e6:  # a centred 1920 x 1080 window of a bigger frame
    x0, y0, Wc, Hc = (W - 1920) // 2, (H - 1080) // 2, 1920, 1080
else:
    x0, y0, Wc, Hc = 0, 0, W, H
ys, xs = np.mgrid[y0:y0 + Hc, x0:x0 + Wc]
px, py = xs.ravel().astype(F) + F(0.5), ys.ravel().astype(F) + F(0.5)
uvx, uvy = F(2.0) * px / s.h - s.aspect, F(2.0) * py / s.h - F(1.0)
d = [uvx * s.m[1][k] - uvy * s.m[2][k] - s.m[0][k] for k in range(3)]
dirv = NP._normalize(d)
o = s.origin
B = {0: 1.0, 1: 2.2360680, 2: 1.7320508, 3: 1.3, 4: 2.0, 5: 1.0}.get(int(s.primitive), 2.0) if int(getattr(s, "group", 0)) == 0 and hasattr(s, "primitive") else 2.0
Rr = F(B) + s.epsilon
R2 = F(1.1) * Rr * Rr
oo = sum(c * c for c in o)
b = -(o[0] * dirv[0] + o[1] * dirv[1] + o[2] * dirv[2])
never = np.where(b <= 0, oo > R2, (oo - b * b) > R2)
n = Wc * Hc
live = ~never
t = np.zeros(n, dtype=F)
pos = [np.full(n, o[k], dtype=F) for k in range(3)]
steps = np.zeros(n, dtype=np.int32)
it = 0
while live.any() and it < s.max_iterations:
    idx = np.nonzero(live)[0]
    p = [c[idx] for c in pos]
    leaving = (NP._dot(p, p) > R2) & (NP._dot(p, [dirv[k][idx] for k in range(3)]) > 0)
    live[idx[leaving]] = False
    idx = idx[~leaving]
    p = [c[~leaving] for c in p]
    with np.errstate(all="ignore"):
        dist = NP.scene_sdf(s, p).astype(F)
    steps[idx] += 1
    hit = dist < s.epsilon
    go = idx[~hit]
    t[go] = t[go] + dist[~hit]
    for k in range(3):
        pos[k][go] = o[k] + t[go] * dirv[k][go]
    live[idx[hit]] = False
    with np.errstate(invalid="ignore"):
        live[go] = t[go] < s.max_distance
    it += 1
tile = ((ys.ravel() - y0) // 8) * ((Wc + 31) // 32) + (xs.ravel() - x0) // 32
tile = tile // POOL
order = np.lexsort([np.arange(n), tile])
bounds = np.r_[0, np.nonzero(np.diff(tile[order]))[0] + 1, n]
tot_rounds = tot_refill = tot_ideal = 0.0
for a, b_ in zip(bounds[:-1], bounds[1:]):
    S = steps[order[a:b_]]
    S = S[S > 0]
    if S.size == 0:
        continue
    tot_ideal += S.sum() / 64.0
    # rounds
    rem = S.copy()
    while rem.size:
        if rem.size <= 64:
            tot_rounds += rem.max()
            break
        for c in range(0, rem.size, 64):
            tot_rounds += min(R, rem[c:c + 64].max())
        rem = rem - R
        rem = rem[rem > 0]
    # refill: list scheduling in order on 64 lanes
    lanes = np.zeros(64, dtype=np.int64)
    for v in S:
        j = int(np.argmin(lanes))
        lanes[j] += int(v)
    tot_refill += lanes.max()
print(f"{key}: rays marched {int((steps > 0).sum())}, mean steps {steps[steps > 0].mean():.1f}, max {steps.max()}")
print(f"  wave-steps  ideal {tot_ideal:.0f}   rounds of {R}: {tot_rounds:.0f} ({tot_ideal / tot_rounds:.2f} useful)   "
      f"refill: {tot_refill:.0f} ({tot_ideal / tot_refill:.2f} useful)   refill / rounds = {tot_refill / tot_rounds:.3f}")
