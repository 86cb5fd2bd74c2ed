#!/usr/bin/env python3
"""Where do the lanes go?  A CPU model of the headline frame's march, ray by ray and step by step
(NumPy restatement of the shader, instrumented), grouped the way the kernels group rays: per 32x8 tile,
64 rays to a wave, re-packed every step (an upper bound on what re-queuing every 16 steps achieves).
For every wave-step it prices the wave (orbit trips of its slowest lane) and the lanes' own needs, and
reports how much of the vector work is useful, and what regrouping rays could gain at best:
  tile      : waves made of one tile's live rays (what render_wave_kernel does)
  tile+split: the same, rays outside the bounding sphere in waves of their own
  packed    : full waves across tile boundaries (rays in tile order), unsorted
  frame     : waves made of ANY 64 live rays of the frame, sorted by orbit length (no kernel can do better)
CPU only.   python tools/lane_study.py [workload]"""
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
TRIP, TAIL, OUT = 34.0, 180.0, 45.0  # VALU cycles: one orbit trip, the rest of an inside step, an outside step


def main():
    key = sys.argv[1] if len(sys.argv) > 1 else "cfg2_julia_1080p"
    w = WORKLOADS[key]
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
    # the kernels' exact cull: rays whose line stays outside 1.1 R^2 never march
    R2 = F(1.1) * (F(2.0) + s.epsilon) ** 2
    oo = sum(c * c for c in o)
    b = -(o[0] * dirv[0] + o[1] * dirv[1] + o[2] * dirv[2])
    never = np.where(b <= 0, oo > R2, (oo - b * b) > R2)
    live = ~never
    n = W * H
    t = np.zeros(n, dtype=F)
    pos = [np.full(n, o[k], dtype=F) for k in range(3)]
    tile = (ys.ravel() // 8) * ((W + 31) // 32) + xs.ravel() // 32
    tot = dict(useful=0.0, tile=0.0, split=0.0, frame=0.0, packed=0.0, packed2=0.0, steps=0, raysteps=0,
               r_tile=0.0, r_pool=0.0, r_pool_split=0.0, r_g2=0.0, r_g4=0.0, r_g16=0.0, r_tile_split=0.0, r_tile_sorted=0.0, r_tile_full=0.0)
    R = 16
    # round-length schedules for one tile per wave (render_wave_kernel): name -> boundaries; each chunk-round
    # also pays ROUND_OVERHEAD cycles (queue traffic, direction rebuilt, ballots: ~110 instructions)
    ROUND_OVERHEAD = 250.0
    def boundaries(lengths):
        b, s = set(), 0
        for L in lengths:
            b.add(s)
            s += L
        while s < 4096:
            b.add(s)
            s += lengths[-1]
        return b
    SCHEDULES = {"8": boundaries([8]), "12": boundaries([12]), "16": boundaries([16]), "24": boundaries([24]), "32": boundaries([32]),
                 "4,4,8,16..": boundaries([4, 4, 8, 16]), "8,8,16..": boundaries([8, 8, 16]), "8,16,32..": boundaries([8, 16, 32]),
                 "8,8,16,32..": boundaries([8, 8, 16, 32]), "4,8,16,32..": boundaries([4, 8, 16, 32]), "16,16,32..": boundaries([16, 16, 32]),
                 "8,24,32..": boundaries([8, 24, 32]), "8,8,16,32,64..": boundaries([8, 8, 16, 32, 64])}
    sched_wave = {k: np.full(n, -1, dtype=np.int64) for k in SCHEDULES}
    sched_cost = {k: 0.0 for k in SCHEDULES}
    ROUND_KEYS = ("r_tile", "r_tile_split", "r_tile_full", "r_tile_sorted", "r_g2", "r_g4", "r_g16", "r_pool", "r_pool_split")
    wave_of = {k: np.full(n, -1, dtype=np.int64) for k in ROUND_KEYS}
    tiles_x = (W + 31) // 32
    g2 = (ys.ravel() // 16) * tiles_x + xs.ravel() // 32                 # 2 tiles: 32 x 16
    g4 = (ys.ravel() // 16) * ((W + 63) // 64) + xs.ravel() // 64       # 4 tiles: 64 x 16
    g16 = (ys.ravel() // 32) * ((W + 127) // 128) + xs.ravel() // 128   # 16 tiles: 128 x 32
    step = 0
    while live.any() and step < s.max_iterations:
        idx = np.nonzero(live)[0]
        p = [c[idx] for c in pos]
        norm = NP._length(p)
        outside = norm > F(2.0) + s.epsilon
        # orbit trips of the inside rays
        trips = np.zeros(idx.size, dtype=np.int32)
        ins = np.nonzero(~outside)[0]
        q = [p[0][ins], p[1][ins], p[2][ins], np.full(ins.size, 0.1, dtype=F)]
        qs = NP._dot(q, q)
        dqs = np.ones(ins.size, dtype=F)
        alive = np.ones(ins.size, dtype=bool)
        with np.errstate(over="ignore", invalid="ignore"):
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
        with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
            dist[ins] = (F(0.25) * np.log(qs) * np.sqrt(qs / dqs)).astype(F)
        # ---- price this step
        need = np.where(outside, OUT, TRIP * trips + TAIL)
        tot["useful"] += float(need.sum())
        tot["raysteps"] += idx.size

        def waves(keys, need_, trips_, out_):
            """rays sorted by keys, 64 to a wave; wave cost = tail (if any inside) + trips of its slowest lane"""
            order = np.lexsort(keys)
            tr, ou = trips_[order], out_[order]
            grp = np.cumsum(np.r_[True, np.any([k[order][1:] != k[order][:-1] for k in keys[-1:]], axis=0)]) - 1  # group = last key
            # position within group -> wave id
            first = np.r_[0, np.nonzero(np.diff(grp))[0] + 1]
            start = np.zeros(grp.size, dtype=np.int64)
            start[first] = first
            start = np.maximum.accumulate(start)
            wave = grp * 100000 + (np.arange(grp.size) - start) // 64
            _, inv = np.unique(wave, return_inverse=True)
            mt = np.zeros(inv.max() + 1)
            np.maximum.at(mt, inv, tr)
            anyin = np.zeros(inv.max() + 1, dtype=bool)
            np.logical_or.at(anyin, inv, ~ou)
            anyout = np.zeros(inv.max() + 1, dtype=bool)
            np.logical_or.at(anyout, inv, ou)
            return float((64.0 * (np.where(anyin, TRIP * mt + TAIL, 0.0) + np.where(anyout, OUT, 0.0))).sum())
        tl = tile[idx]
        tot["tile"] += waves([trips * 0, tl], need, trips, outside)                       # one tile's rays, unsorted
        tot["split"] += waves([-trips, tl * 2 + outside.astype(np.int64)], need, trips, outside)  # outside rays apart, sorted
        tot["frame"] += waves([-trips, outside.astype(np.int64)], need, trips, outside)   # any 64 rays of the frame
        tot["packed"] += waves([tl, tl * 0], need, trips, outside)                        # full waves in tile order, unsorted
        tot["packed2"] += waves([tl, outside.astype(np.int64)], need, trips, outside)     # ... outside rays apart
        tot["steps"] += 1
        # ---- the same with wave membership fixed for rounds of R steps (what kernels can actually do)
        if step % R == 0:
            def assign(keys):
                order = np.lexsort(keys)
                last = keys[-1][order]
                grp = np.cumsum(np.r_[True, last[1:] != last[:-1]]) - 1
                first = np.r_[0, np.nonzero(np.diff(grp))[0] + 1]
                start = np.zeros(grp.size, dtype=np.int64)
                start[first] = first
                start = np.maximum.accumulate(start)
                wid = grp * 1000000 + (np.arange(grp.size) - start) // 64
                out = np.empty(idx.size, dtype=np.int64)
                out[order] = wid
                return out
            zero = np.zeros(idx.size, dtype=np.int64)
            wave_of["r_tile"][idx] = assign([zero, tl])                               # per tile (render_wave_kernel)
            wave_of["r_tile_split"][idx] = assign([outside.astype(np.int64), tl])     # a tile's rays, outside ones first
            wave_of["r_tile_full"][idx] = assign([(trips < s.sdf_iters).astype(np.int64) + outside, tl])  # never-escaping / escaping / outside
            wave_of["r_tile_sorted"][idx] = assign([-trips, tl])                      # a tile's rays by orbit length at the round's first step
            wave_of["r_g2"][idx] = assign([zero, g2[idx]])                            # pools of 2 / 4 / 16 adjacent tiles
            wave_of["r_g4"][idx] = assign([zero, g4[idx]])
            wave_of["r_g16"][idx] = assign([zero, g16[idx]])
            wave_of["r_pool"][idx] = assign([tl, zero])                               # one pool, tile order
            wave_of["r_pool_split"][idx] = assign([tl, outside.astype(np.int64)])     # inside / outside pools
        for k, bset in SCHEDULES.items():
            if step in bset:
                order = np.lexsort([np.zeros(idx.size, dtype=np.int64), tile[idx]])
                last = tile[idx][order]
                grp = np.cumsum(np.r_[True, last[1:] != last[:-1]]) - 1
                first = np.r_[0, np.nonzero(np.diff(grp))[0] + 1]
                start = np.zeros(grp.size, dtype=np.int64)
                start[first] = first
                start = np.maximum.accumulate(start)
                wid = np.empty(idx.size, dtype=np.int64)
                wid[order] = grp * 1000000 + (np.arange(grp.size) - start) // 64
                sched_wave[k][idx] = wid
                sched_cost[k] += 64.0 * ROUND_OVERHEAD * np.unique(wid).size
            _, inv = np.unique(sched_wave[k][idx], return_inverse=True)
            mt = np.zeros(inv.max() + 1)
            np.maximum.at(mt, inv, trips)
            anyin = np.zeros(inv.max() + 1, dtype=bool)
            np.logical_or.at(anyin, inv, ~outside)
            anyout = np.zeros(inv.max() + 1, dtype=bool)
            np.logical_or.at(anyout, inv, outside)
            sched_cost[k] += float((64.0 * (np.where(anyin, TRIP * mt + TAIL, 0.0) + np.where(anyout, OUT, 0.0))).sum())
        for k in ROUND_KEYS:
            _, inv = np.unique(wave_of[k][idx], return_inverse=True)
            mt = np.zeros(inv.max() + 1)
            np.maximum.at(mt, inv, trips)
            anyin = np.zeros(inv.max() + 1, dtype=bool)
            np.logical_or.at(anyin, inv, ~outside)
            anyout = np.zeros(inv.max() + 1, dtype=bool)
            np.logical_or.at(anyout, inv, outside)
            tot[k] += float((64.0 * (np.where(anyin, TRIP * mt + TAIL, 0.0) + np.where(anyout, OUT, 0.0))).sum())
        # ---- advance (literal algorithm + the kernels' "leaving" cull)
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
    u = tot["useful"]
    print(f"{key}: {tot['raysteps']} ray-steps in {tot['steps']} steps; lanes' own work = 100 %")
    for k, label in (("tile", "waves of one tile's rays"), ("split", "… outside rays apart, sorted by orbit length"),
                     ("packed", "full waves of the frame's rays in tile order"), ("packed2", "… outside rays apart"),
                     ("frame", "waves of any 64 rays of the frame, sorted"),
                     ("r_tile", "ROUNDS of 16: waves of one tile's rays"), ("r_tile_split", "ROUNDS of 16: one tile, outside rays first"),
                     ("r_tile_full", "ROUNDS of 16: one tile, full-orbit / escaping / outside"), ("r_tile_sorted", "ROUNDS of 16: one tile, sorted by orbit length"),
                     ("r_g2", "ROUNDS of 16: pools of 2 tiles (32x16)"),
                     ("r_g4", "ROUNDS of 16: pools of 4 tiles (64x16)"), ("r_g16", "ROUNDS of 16: pools of 16 tiles (128x32)"), ("r_pool", "ROUNDS of 16: one pool of all rays, tile order"),
                     ("r_pool_split", "ROUNDS of 16: inside / outside pools")):
        print(f"  {label:48s}: vector work {tot[k] / u:.2f} x the lanes' own ({100 * u / tot[k]:.0f} % useful)")
    print("  round-length schedules, one tile per wave, with the rounds' own overhead:")
    for k, c in sched_cost.items():
        print(f"    rounds of {k:16s}: {c / u:.3f} x")


if __name__ == "__main__":
    main()
