#!/usr/bin/env python3
"""What would it buy to let the lanes of a wave fall out of step?

render_wave_kernel marches a tile's rays in rounds of 16 steps with a wave-uniform step counter: in wave-step s
the wave pays the OUTSIDE part (|p| - 2, ~45 VALU cycles) if any lane is outside the bounding sphere and the
INSIDE part (orbit + log / divide / sqrt tail, 180 + 34 per trip) if any lane is inside.  Rays enter the sphere
after different numbers of outside steps, so for most of a round both parts run, each for a fraction of the lanes.

This script replays the headline frame's march on the CPU (NumPy restatement, as tools/lane_study.py) and prices,
with the membership of a wave fixed for a round exactly as the kernel fixes it (one tile, queue order, 64 to a wave):
  lockstep      what the kernel does today
  catch-up      per-lane step counters: the outside lanes first run ahead in a cheap loop of their own until each
                is inside, finished or out of steps for this round; then ONE inside step for every inside lane;
                repeat.  A lane's own sequence of operations is unchanged (same pixels).
  inside-run    the same, but the inside phase also keeps going while NO lane is outside (i.e. it stops for a
                catch-up only when a lane has left the sphere again)
CPU only.   python tools/lane_desync_study.py [workload] [round_steps]"""
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
COUNTER = 4.0                        # per step and phase: the per-lane step counter (v_add + v_cmp)


def price_round(seq_out, seq_trips, seq_len):
    """seq_*: (lanes, R) arrays of one wave for one round; seq_len: steps each lane takes in it.
    Returns (lockstep, catch_up, inside_run) VALU cycles of the wave."""
    L, R = seq_out.shape
    steps = np.arange(R)[None, :]
    act = steps < seq_len[:, None]
    # lockstep
    any_in = (act & ~seq_out).any(0)
    any_out = (act & seq_out).any(0)
    max_tr = np.where(act & ~seq_out, seq_trips, 0).max(0)
    lock = float((np.where(any_in, TRIP * max_tr + TAIL, 0.0) + np.where(any_out, OUT, 0.0)).sum())
    res = []
    for keep_going in (False, True):
        ptr = np.zeros(L, dtype=np.int64)
        cost = 0.0
        lanes = np.arange(L)
        while (ptr < seq_len).any():
            # outside phase: every lane advances over its run of outside steps
            run = np.zeros(L, dtype=np.int64)
            moving = (ptr < seq_len) & seq_out[lanes, np.minimum(ptr, R - 1)]
            while moving.any():
                run += moving
                ptr = ptr + moving
                moving = (ptr < seq_len) & seq_out[lanes, np.minimum(ptr, R - 1)]
            if run.max() > 0:
                cost += run.max() * (OUT + COUNTER)
            # inside phase
            while True:
                ins = (ptr < seq_len) & ~seq_out[lanes, np.minimum(ptr, R - 1)]
                if not ins.any():
                    break
                cost += TRIP * seq_trips[lanes, np.minimum(ptr, R - 1)][ins].max() + TAIL + COUNTER
                ptr = ptr + ins
                if not keep_going:
                    break
                if ((ptr < seq_len) & seq_out[lanes, np.minimum(ptr, R - 1)]).any():
                    break
        res.append(cost)
    return lock, res[0], res[1]


SORT = next((a.split("=")[1] for a in sys.argv[1:] if a.startswith("--sort=")), "")


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    key = args[0] if args else "cfg2_julia_1080p"
    R = int(args[1]) if len(args) > 1 else 16
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
    R2 = F(1.1) * (F(2.0) + s.epsilon) ** 2
    oo = sum(c * c for c in o)
    b = -(o[0] * dirv[0] + o[1] * dirv[1] + o[2] * dirv[2])
    never = np.where(b <= 0, oo > R2, (oo - b * b) > R2)
    live = ~never
    n = W * H
    t = np.zeros(n, dtype=F)
    pos = [np.full(n, o[k], dtype=F) for k in range(3)]
    tile = (ys.ravel() // 8) * ((W + 31) // 32) + xs.ravel() // 32
    tot = np.zeros(3)
    useful = 0.0
    per_round = []  # (first step, rays, waves, lockstep cycles, useful cycles)
    round_useful = [0.0]
    step = 0
    # per round: the rays alive at its start, and their per-step (outside, trips) records
    members, rec_out, rec_tr, rec_len = None, None, None, None
    col = {}
    repack = [int(x) for a in sys.argv[1:] if a.startswith("--repack=") for x in a.split("=")[1].split(",")]
    gwave = np.full(n, -1, dtype=np.int64)  # frame-wide wave id of a ray once it has been repacked
    tot_repack = [0.0]

    def close_round():
        nonlocal tot
        if members is None or members.size == 0:
            return
        order = np.lexsort([members, tile[members]])  # tile, then pixel order within the tile (queue order ~ pixel order)
        if SORT:
            # what if a tile's rays were cut into waves by orbit length instead of pixel order?  key: the ray's trip count
            # at its first step of the round ("first": what a predictor run at filing time could know exactly), its
            # mean over the round ("mean": an oracle), or both with outside steps counted as -1
            if SORT == "first":
                key = np.where(rec_out[:, 0], -1, rec_tr[:, 0])
            else:
                w_ = np.arange(rec_tr.shape[1])[None, :] < rec_len[:, None]
                key = (np.where(rec_out, 0, rec_tr) * w_).sum(1) / np.maximum(1, rec_len)
            order = np.lexsort([members, -key, tile[members]])
        tl = tile[members][order]
        first = np.r_[0, np.nonzero(np.diff(tl))[0] + 1]
        start = np.zeros(tl.size, dtype=np.int64)
        start[first] = first
        start = np.maximum.accumulate(start)
        wave = (np.cumsum(np.r_[True, tl[1:] != tl[:-1]]) - 1) * 1000 + (np.arange(tl.size) - start) // 64
        bounds = np.r_[0, np.nonzero(np.diff(wave))[0] + 1, wave.size]
        before = tot[0]
        for a, b_ in zip(bounds[:-1], bounds[1:]):
            sel = order[a:b_]
            tot += np.array(price_round(rec_out[sel], rec_tr[sel], rec_len[sel]))
        if repack:
            # the same round with the waves of the repacked scheme: per-tile waves before the first boundary,
            # afterwards the frame-wide waves formed at the last boundary (a wave keeps its rays between boundaries)
            first_step = step - R if step % R == 0 else step - step % R
            if first_step < repack[0]:
                tot_repack[0] += tot[0] - before
            else:
                if first_step in repack:
                    o2 = np.lexsort([members, tile[members]])
                    gwave[members[o2]] = first_step * 10**7 + np.arange(members.size) // 64
                g = gwave[members]
                o3 = np.argsort(g, kind="stable")
                b3 = np.r_[0, np.nonzero(np.diff(g[o3]))[0] + 1, g.size]
                for a, b_ in zip(b3[:-1], b3[1:]):
                    sel = o3[a:b_]
                    tot_repack[0] += price_round(rec_out[sel], rec_tr[sel], rec_len[sel])[0]
        per_round.append((step - R if step % R == 0 else step - step % R, members.size, bounds.size - 1, tot[0] - before, round_useful[0]))
        round_useful[0] = 0.0

    while live.any() and step < s.max_iterations:
        if step % R == 0:
            close_round()
            members = np.nonzero(live)[0]
            col = {int(m): i for i, m in enumerate(members)} if False else None
            index_of = np.full(n, -1, dtype=np.int64)
            index_of[members] = np.arange(members.size)
            rec_out = np.zeros((members.size, R), dtype=bool)
            rec_tr = np.zeros((members.size, R), dtype=np.int32)
            rec_len = np.zeros(members.size, dtype=np.int64)
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
        useful += float(np.where(outside, OUT, TRIP * trips + TAIL).sum())
        round_useful[0] += float(np.where(outside, OUT, TRIP * trips + TAIL).sum())
        r = index_of[idx]
        rec_out[r, step % R] = outside
        rec_tr[r, step % R] = trips
        rec_len[r] = step % R + 1
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
    close_round()
    global PER_ROUND, TOT, USEFUL
    PER_ROUND, TOT, USEFUL = per_round, tot, useful
    if repack:
        print(f"  frame-wide repack at steps {repack}: {64.0 * tot_repack[0] / useful:.3f} x the lanes' own; "
              f"{tot[0] / tot_repack[0]:.3f} x fewer VALU cycles than today")
    print(f"{key}, rounds of {R}: lanes' own work = 1.00")
    for name, v in zip(("lockstep (today)", "catch-up", "catch-up + inside runs"), tot):
        print(f"  {name:26s}: {64.0 * v / useful:.3f} x the lanes' own   ({100 * useful / (64.0 * v):.0f} % useful)   "
              f"{tot[0] / v:.3f} x fewer VALU cycles than lockstep")


def print_rounds():
    if "--rounds" in sys.argv:
        print("  round  first_step   rays  waves  rays/wave  lockstep_share  useful_share  useful/lockstep")
        for i, (fs, rays, waves, lock, use) in enumerate(PER_ROUND):
            print(f"  {i:5d}  {fs:10d}  {rays:6d} {waves:6d}  {rays / max(1, waves):9.1f}  {lock / TOT[0]:14.3f}  {use / USEFUL:12.3f}  {use / (64 * lock):.2f}")


if __name__ == "__main__":
    main()
    print_rounds()
