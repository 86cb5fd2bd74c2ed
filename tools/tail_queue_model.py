#!/usr/bin/env python3
"""What would a launch-wide TAIL QUEUE buy the headline kernel?

render_wave_kernel marches a tile's rays in rounds and, once they fit one chunk of 64, to the end: a tile's last
handful of rays then occupies a whole wave for up to 240 more steps (profiles/r03/lane_models.txt: the rounds from
step 64 on are 23 % of the march's VALU cycles at 4-30 % useful lanes).  The idea priced here: at a round boundary a
tile that has few rays left FILES them -- (pixel, t), exactly what its own queue holds -- in a queue of its view and
of that step count and finishes (shades its hits, stores); a second launch marches the filed rays, pooled across the
view's tiles, 256 (or more) to a wave, with the same rounds.  Every ray's own sequence of operations is unchanged.

CPU replay of the headline frame's march (NumPy restatement, as tools/lane_desync_study.py) -> per ray and step
(outside | trips); prices in VALU cycles per wave-step as that tool does (outside part 45 if any lane is outside,
inside part 180 + 34 x the longest orbit among the lanes inside) plus a per-chunk-and-round overhead.

    python tools/tail_queue_model.py [workload] [round_steps]"""
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
TRIP, TAIL, OUT, OVER = 34.0, 180.0, 45.0, 250.0  # (OVER: what a chunk pays per round outside the march)
ENDED, OUTSIDE = -2, -1


def replay(key):
    """(codes[n_rays, max_steps] int8, tile[n_rays], pixel[n_rays]) for the rays that survive the set-up culls."""
    cache = Path("/tmp") / f"tail_queue_model_{key}.npz"
    if cache.exists():
        z = np.load(cache)
        return z["codes"], z["tile"], z["pixel"]
    w = WORKLOADS[key]
    if int(w.gui.fractal_group) != int(K.FractalGroup.JuliaSet):
        raise SystemExit("tools/tail_queue_model.py replays the quaternion Julia set's march only")
    ub = K.uniform_bytes
    s = NP.Scene(O.from_bytes(O.Screen, ub(w.screen.into_buffer_data())), O.from_bytes(O.Camera, ub(w.camera.into_buffer_data())),
                 O.from_bytes(O.Options, ub(w.gui.into_buffer_data())), O.iters(*w.iters))
    W, H = s.width, s.height
    ys, xs = np.mgrid[0:H, 0:W]
    px, py = xs.ravel().astype(F) + F(0.5), ys.ravel().astype(F) + F(0.5)
    uvx, uvy = F(2.0) * px / s.h - s.aspect, F(2.0) * py / s.h - F(1.0)
    d = [uvx * s.m[1][k] - uvy * s.m[2][k] - s.m[0][k] for k in range(3)]
    dirv = NP._normalize(d)
    o = s.origin
    R2 = F(1.1) * (F(2.0) + s.epsilon) ** 2
    oo = sum(c * c for c in o)
    b = -(o[0] * dirv[0] + o[1] * dirv[1] + o[2] * dirv[2])
    never = np.where(b <= 0, oo > R2, (oo - b * b) > R2)
    live = ~never
    n = W * H
    rays = np.nonzero(live)[0]
    row = np.full(n, -1, dtype=np.int64)
    row[rays] = np.arange(rays.size)
    codes = np.full((rays.size, s.max_iterations), ENDED, dtype=np.int8)
    t = np.zeros(n, dtype=F)
    pos = [np.full(n, o[k], dtype=F) for k in range(3)]
    tile_of = (ys.ravel() // 8) * ((W + 31) // 32) + xs.ravel() // 32
    step = 0
    while live.any() and step < s.max_iterations:
        idx = np.nonzero(live)[0]
        p = [c[idx] for c in pos]
        norm = NP._length(p)
        outside = norm > F(2.0) + s.epsilon
        trips = np.zeros(idx.size, dtype=np.int32)
        ins = np.nonzero(~outside)[0]
        q = [p[0][ins], p[1][ins], p[2][ins], np.full(ins.size, 0.1, dtype=F)]
        qs = NP._dot(q, q)
        dqs = np.ones(ins.size, dtype=F)
        alive = np.ones(ins.size, dtype=bool)
        with np.errstate(all="ignore"):
            for _ in range(s.sdf_iters):
                if not alive.any():
                    break
                trips[ins[alive]] += 1
                dqs = np.where(alive, dqs * (F(4.0) * qs), dqs)
                nq = NP.quat_add(NP.quat_sq(q), s.c)
                q = [np.where(alive, a, c) for a, c in zip(nq, q)]
                qs = np.where(alive, NP._dot(q, q), qs)
                alive = alive & ~(qs > s.max_distance)
            dist = norm - F(2.0)
            dist[ins] = (F(0.25) * np.log(qs) * np.sqrt(qs / dqs)).astype(F)
        codes[row[idx], step] = np.where(outside, OUTSIDE, trips).astype(np.int8)
        with np.errstate(invalid="ignore"):
            hit = dist < s.epsilon
        go = idx[~hit]
        t[go] = t[go] + dist[~hit]
        for k in range(3):
            pos[k][go] = o[k] + t[go] * dirv[k][go]
        live[idx[hit]] = False
        pg = [pos[k][go] for k in range(3)]
        leaving = (NP._dot(pg, pg) > R2) & (NP._dot(pg, [dirv[k][go] for k in range(3)]) > 0)
        with np.errstate(invalid="ignore"):
            live[go] = (t[go] < s.max_distance) & ~leaving
        step += 1
    np.savez_compressed(cache, codes=codes, tile=tile_of[rays], pixel=rays)
    return codes, tile_of[rays], rays


def chunk_cost(c):
    """VALU cycles of one wave marching the rays `c` (codes[rays, steps]) in lockstep over those steps."""
    if c.size == 0:
        return 0.0
    act = c != ENDED
    any_out = (c == OUTSIDE).any(0)
    mx = np.where(c > 0, c, 0).max(0)
    any_in = mx > 0
    live_steps = act.any(0)
    return float((np.where(any_in, TRIP * mx + TAIL, 0.0) + np.where(any_out, OUT, 0.0))[live_steps].sum())


def useful_cost(codes):
    return float((np.where(codes == OUTSIDE, OUT, 0.0) + np.where(codes > 0, TRIP * codes + TAIL, 0.0)).sum())


def march_pool(codes, members, start, R, evict=None, filed=None, to_end=64):
    """One wave marches the pool `members` (ray rows, in queue order) from step `start` in rounds of R, 64 to a chunk;
    a pool that fits `to_end` rays is marched to the end.  evict = (first_step, max_rays): at a round boundary at or
    after first_step a pool with <= max_rays rays left appends them to filed[step] and stops.
    Returns (march cycles, overhead cycles)."""
    S = codes.shape[1]
    cost = over = 0.0
    step = start
    m = members
    while m.size and step < S:
        m = m[codes[m, step] != ENDED]
        if m.size == 0:
            break
        if evict is not None and step >= evict[0] and step > start and m.size <= evict[1]:
            filed.setdefault(step, []).append(m)
            return cost, over
        end = S if m.size <= to_end else min(S, step + R)
        for a in range(0, m.size, 64):
            cost += chunk_cost(codes[m[a:a + 64], step:end])
            over += OVER
        step = end
    return cost, over


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    key = args[0] if args else "cfg2_julia_1080p"
    R = int(args[1]) if len(args) > 1 else 16
    codes, tile, pixel = replay(key)
    use = useful_cost(codes)
    order = np.lexsort([pixel, tile])
    bounds = np.r_[0, np.nonzero(np.diff(tile[order]))[0] + 1, order.size]
    pools = [order[a:b] for a, b in zip(bounds[:-1], bounds[1:])]
    print(f"{key}: {codes.shape[0]} rays in {len(pools)} tiles, rounds of {R}; the lanes' own work = 1.00")

    def run(evict, pool_rays, tail_R, label, shuffle=False, tail_to_end=64):
        filed = {}
        c = o = 0.0
        for m in pools:
            a, b = march_pool(codes, m, 0, R, evict, filed)
            c += a
            o += b
        main_c, main_o = c, o
        n_filed = 0
        tail_waves = 0
        for step, lists in sorted(filed.items()):
            q = np.concatenate(lists)
            if shuffle:
                q = np.random.default_rng(1).permutation(q)
            n_filed += q.size
            for a in range(0, q.size, pool_rays):
                x, y = march_pool(codes, q[a:a + pool_rays], step, tail_R, to_end=tail_to_end)
                c += x
                o += y
                tail_waves += 1
        return label, c, o, main_c, main_o, n_filed, tail_waves

    rows = [run(None, 0, R, "today (one wave per tile, rounds, to the end from 64 rays)")]
    for first in (16, 32, 48, 64):
        for mx in (16, 32, 64):
            for pool in (256, 512):
                rows.append(run((first, mx), pool, R, f"file from step {first:3d} when <= {mx:2d} rays; tail pools of {pool}"))
    rows.append(run((32, 32), 256, R, "file from step  32 when <= 32 rays; tail pools of 256, filed in random order", shuffle=True))
    rows.append(run((32, 64), 256, 32, "file from step  32 when <= 64 rays; tail pools of 256, tail rounds of 32"))
    rows.append(run((32, 64), 256, 8, "file from step  32 when <= 64 rays; tail pools of 256, tail rounds of 8"))
    base = rows[0][1] + rows[0][2]
    print(f"{'policy':88s}  march  +rounds  = total x useful | vs today | filed rays, tail waves, main-launch share")
    for label, c, o, mc, mo, nf, tw in rows:
        print(f"{label:88s}  {64 * c / use:5.3f}  {64 * o / use:6.3f}  {64 * (c + o) / use:6.3f}          | {base / (c + o):5.3f}    | "
              f"{nf:6d} {tw:5d} {(mc + mo) / (c + o):5.2f}")


if __name__ == "__main__":
    main()
