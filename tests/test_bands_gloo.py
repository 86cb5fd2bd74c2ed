"""N > 1 path on CPU: world_size 2 and 3 over the gloo backend.  Each rank renders its
row band (the CPU oracle stands in for the GPU kernel -- same band contract: global pixel
coordinates, rows [y0, y1)) into the BandFrame buffers; the grouped point-to-point gather
must assemble, on rank 0, a frame identical to the single-process frame, for several
pipelined frames with double buffering."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, width, height, frames, outdir):
    sys.path.insert(0, str(ROOT))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import oracle as O
        from kifs_raymarching_amd.bands import BandFrame

        sc = O.screen_uniform(width, height)
        opt = O.options_from_gui(fractal_group=1, constant=(-0.2, 0.6, 0.2, 0.2), max_iterations=48)
        it = O.iters(8, 4, 4)
        bf = BandFrame(width, height, rank, world, "cpu")
        assert bf.ranges[0][0] == 0 and bf.ranges[-1][1] == height

        def make_render(k):
            cam = O.camera_uniform(3.0, 0.3 * k, 0.1)

            def render_band(out, y0, y1):
                out.copy_(torch.from_numpy(O.render(sc, cam, opt, it, y0=y0, y1=y1, nthreads=1)))
            return render_band

        results = []
        for k in range(frames):
            bf.step(k, make_render(k))
            if k >= 1:  # consume frame k-1 while frame k is in flight (double buffering)
                bf.wait(k - 1)
                if rank == 0:
                    results.append(bf.frame(k - 1).clone().numpy())
        bf.wait_all()
        if rank == 0:
            results.append(bf.frame(frames - 1).clone().numpy())
            np.save(os.path.join(outdir, "gathered.npy"), np.stack(results))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,height", [(2, 40), (3, 41)])
def test_band_gather_equals_single_frame(world, height, tmp_path, oracle):
    width, frames = 48, 4
    mp.spawn(_worker, args=(world, _free_port(), width, height, frames, str(tmp_path)),
             nprocs=world, join=True)
    got = np.load(tmp_path / "gathered.npy")
    sc = oracle.screen_uniform(width, height)
    opt = oracle.options_from_gui(fractal_group=1, constant=(-0.2, 0.6, 0.2, 0.2), max_iterations=48)
    for k in range(frames):
        want = oracle.render(sc, oracle.camera_uniform(3.0, 0.3 * k, 0.1), opt, oracle.iters(8, 4, 4))
        assert (got[k] == want).all(), f"frame {k} differs"
        assert (want != want[0, 0]).any()


def test_bandframe_single_rank_needs_no_process_group(kifs):
    from kifs_raymarching_amd.bands import BandFrame
    bf = BandFrame(16, 9, 0, 1, "cpu")
    calls = []
    bf.step(0, lambda out, y0, y1: calls.append((tuple(out.shape), y0, y1)))
    bf.wait_all()
    assert calls == [((9, 16, 4), 0, 9)] and bf.frame(0).shape == (9, 16, 4)
    with pytest.raises(ValueError):
        BandFrame(0, 9, 0, 1, "cpu")


def _frames_worker(rank, world, port, width, height, steps, outdir):
    sys.path.insert(0, str(ROOT))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import oracle as O
        from kifs_raymarching_amd.bands import FrameStream

        sc = O.screen_uniform(width, height)
        opt = O.options_from_gui(primitive_shape=4, max_iterations=40)
        fs = FrameStream(width, height, rank, world, "cpu")

        def render_frame(out, index):  # frame `index` of an orbit
            cam = O.camera_uniform(3.0, 0.25 * index, 0.2)
            out.copy_(torch.from_numpy(O.render(sc, cam, opt, O.iters(100, 10, 6), nthreads=1)))

        got = []
        for k in range(steps):
            fs.step(k, render_frame)
            if k >= 1:
                fs.wait(k - 1)
                if rank == 0:
                    got.extend(f.clone().numpy() for f in fs.frames(k - 1))
        fs.wait_all()
        if rank == 0:
            got.extend(f.clone().numpy() for f in fs.frames(steps - 1))
            np.save(os.path.join(outdir, "frames.npy"), np.stack(got))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_frame_stream_delivers_the_sequence_in_order(world, tmp_path, oracle):
    """Frame-parallel mode: step k, rank r renders frame k*world + r; the root ends up with
    the whole orbit in sequence order, identical to rendering it in one process."""
    width, height, steps = 40, 30, 3
    mp.spawn(_frames_worker, args=(world, _free_port(), width, height, steps, str(tmp_path)),
             nprocs=world, join=True)
    got = np.load(tmp_path / "frames.npy")
    assert got.shape[0] == steps * world
    sc = oracle.screen_uniform(width, height)
    opt = oracle.options_from_gui(primitive_shape=4, max_iterations=40)
    for i in range(steps * world):
        want = oracle.render(sc, oracle.camera_uniform(3.0, 0.25 * i, 0.2), opt, oracle.iters(100, 10, 6))
        assert (got[i] == want).all(), f"frame {i}"
    assert (got[0] != got[1]).any()


def _local_frames_worker(rank, world, port, width, height, steps, outdir):
    sys.path.insert(0, str(ROOT))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import oracle as O
        from kifs_raymarching_amd.bands import FrameStream

        sc = O.screen_uniform(width, height)
        opt = O.options_from_gui(primitive_shape=3, max_iterations=40)
        # bench.py's default at N > 1: frames stay where they were rendered, 3 buffers (as with
        # three launches in flight), a step's "frame" is a stack of 2 frames (2 per launch)
        fs = FrameStream(width, 2 * height, rank, world, "cpu", buffers=3, deliver=False)
        assert fs.frames(0) is None or len(fs.frames(0)) == 1

        def render_stack(out, step_index):
            for i in range(2):
                cam = O.camera_uniform(3.0, 0.2 * (2 * step_index + i), 0.1)
                out[i * height:(i + 1) * height].copy_(
                    torch.from_numpy(O.render(sc, cam, opt, O.iters(100, 10, 6), nthreads=1)))

        mine = []
        for k in range(steps):
            fs.step(k, render_stack)
            fs.wait(k)  # no exchange: nothing to wait for, must not block or raise
            mine.append((fs.frame_index(k), fs.target(k).clone().numpy()))
        fs.wait_all()
        np.save(os.path.join(outdir, f"rank{rank}.npy"), np.stack([m[1] for m in mine]))
        np.save(os.path.join(outdir, f"index{rank}.npy"), np.array([m[0] for m in mine]))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_frame_stream_without_delivery_keeps_frames_local(tmp_path, oracle):
    """deliver=False (bench.py's default at N > 1): no exchange step, every rank keeps the frames
    it rendered; together the ranks hold the whole sequence exactly once."""
    world, width, height, steps = 2, 36, 24, 3
    mp.spawn(_local_frames_worker, args=(world, _free_port(), width, height, steps, str(tmp_path)),
             nprocs=world, join=True)
    sc = oracle.screen_uniform(width, height)
    opt = oracle.options_from_gui(primitive_shape=3, max_iterations=40)
    seen = set()
    for rank in range(world):
        stacks = np.load(tmp_path / f"rank{rank}.npy")
        index = np.load(tmp_path / f"index{rank}.npy")
        assert list(index) == [k * world + rank for k in range(steps)]
        for k in range(steps):
            for i in range(2):
                f = 2 * int(index[k]) + i
                seen.add(f)
                want = oracle.render(sc, oracle.camera_uniform(3.0, 0.2 * f, 0.1), opt, oracle.iters(100, 10, 6))
                assert (stacks[k][i * height:(i + 1) * height] == want).all(), (rank, k, i)
    assert seen == set(range(2 * steps * world))


# ---- row shards (interleaved stripes), batches of frames per step ---------------------------
def _oracle_rows(O, sc, cam, opt, it, height, stripes):
    """The rows of a shard, packed: what kifs_render_shard_async writes with in_place = 0."""
    parts = [O.render(sc, cam, opt, it, y0=8 * s, y1=min(height, 8 * s + 8), nthreads=1) for s in stripes]
    return np.concatenate(parts, axis=0)


def _shard_worker(rank, world, port, width, height, steps, count, weights, contiguous, outdir, sparse=False,
                  distance=3.0, hook=False):
    sys.path.insert(0, str(ROOT))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import oracle as O
        from kifs_raymarching_amd import bands
        from kifs_raymarching_amd.bands import ShardFrames, SparseShardFrames

        sc = O.screen_uniform(width, height)
        opt = O.options_from_gui(fractal_group=1, constant=(-0.2, 0.6, 0.2, 0.2), max_iterations=40)
        it = O.iters(8, 4, 4)
        if sparse:
            # the background pixel as the renderer encodes it: the corner of a frame seen from far away
            far = O.render(sc, O.camera_uniform(50.0, 0.0, 0.0), opt, it, y0=0, y1=1, nthreads=1)[0, 0]
            bg = int(far[0]) | int(far[1]) << 8 | int(far[2]) << 16 | int(far[3]) << 24

            def pack(shards, stripes, records):
                r = bands.pack_sparse_torch(shards, stripes, height, bg)
                records[:r.shape[0]].copy_(r)
                return lambda: r.shape[0]
            sf = SparseShardFrames(width, height, rank, world, "cpu", frames_per_step=count, weights=weights,
                                   contiguous=contiguous, pack=pack,
                                   unpack_sparse=lambda frames, records, n, stripes: bands.unpack_sparse_torch(frames, records[:n], stripes),
                                   fill=lambda frames, stripes: bands.fill_stripes_torch(frames, stripes, bg),
                                   erase=(lambda frames, records, n, stripes: bands.erase_sparse_torch(frames, records[:n], stripes, bg))
                                   if world == 3 else None)
        else:
            sf = ShardFrames(width, height, rank, world, "cpu", frames_per_step=count, weights=weights,
                             contiguous=contiguous)
        every = sorted(s for st in sf.stripes for s in st)
        assert every == list(range((height + 7) // 8)), "every stripe is dealt exactly once"

        def render_shard(outs, first_frame, stripes, in_place):
            assert in_place == (rank == 0) and len(outs) == count
            for i, out in enumerate(outs):
                cam = O.camera_uniform(distance, 0.3 * (first_frame + i), 0.1)
                rows = torch.from_numpy(_oracle_rows(O, sc, cam, opt, it, height, stripes))
                if in_place:  # the root writes its rows at their frame positions
                    y = 0
                    for s in stripes:
                        n = min(height, 8 * s + 8) - 8 * s
                        out[8 * s:8 * s + n].copy_(rows[y:y + n])
                        y += n
                else:
                    out.copy_(rows)

        got = []
        if hook:
            # a plain step() loop (what bench.py times): the consumer is handed every step's frames exactly once,
            # complete, before their slot is rendered into again
            handed = {}

            def on_frames(k, frames):
                assert k not in handed
                handed[k] = frames.clone().numpy()
            sf.on_frames = on_frames
            for k in range(steps):
                sf.step(k, render_shard)
                assert sorted(handed) == list(range(max(0, k - 1))) or rank != 0  # two buffers: step k hands off k - 2
            sf.wait_all()
            got = [handed[k] for k in range(steps)] if rank == 0 else []
        else:
            for k in range(steps):
                sf.step(k, render_shard)
                if k >= 1:
                    sf.wait(k - 1)
                    if rank == 0:
                        got.append(sf.frames(k - 1).clone().numpy())
            sf.wait_all()
            if rank == 0:
                got.append(sf.frames(steps - 1).clone().numpy())
        if rank == 0:
            np.save(os.path.join(outdir, "shards.npy"), np.concatenate(got, axis=0))
            if sparse:
                np.save(os.path.join(outdir, "sparse_stats.npy"), np.array([sf.records_sent, sf.tiles_seen]))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,height,count,weights,contiguous,sparse,width,distance,hook", [
    (2, 40, 2, None, False, False, 40, 3.0, True),  # frames handed to a consumer from a plain step() loop: dense
    (3, 43, 2, None, False, True, 72, 3.0, True),   # ... and sparse (the slot's erase follows the hand-off)
])
def test_shard_frames_are_handed_to_a_consumer_before_their_slot_is_reused(world, height, count, weights, contiguous, sparse,
                                                                           width, distance, hook, tmp_path, oracle):
    steps = 6
    mp.spawn(_shard_worker,
             args=(world, _free_port(), width, height, steps, count, weights, contiguous, str(tmp_path), sparse, distance, hook),
             nprocs=world, join=True)
    got = np.load(tmp_path / "shards.npy")
    assert got.shape == (steps * count, height, width, 4)
    sc = oracle.screen_uniform(width, height)
    opt = oracle.options_from_gui(fractal_group=1, constant=(-0.2, 0.6, 0.2, 0.2), max_iterations=40)
    for f in range(steps * count):
        want = oracle.render(sc, oracle.camera_uniform(distance, 0.3 * f, 0.1), opt, oracle.iters(8, 4, 4))
        assert (got[f] == want).all(), f"frame {f} differs"


@pytest.mark.parametrize("world,height,count,weights,contiguous,sparse,width,distance", [
    (2, 40, 2, None, False, False, 40, 3.0),       # 5 stripes over 2 ranks: 3 + 2
    (3, 43, 2, None, False, False, 40, 3.0),       # 6 stripes, the last one 3 rows tall
    (3, 52, 1, [3, 1, 1], False, False, 40, 3.0),  # a root that renders three stripes in five
    (2, 44, 2, None, True, False, 40, 3.0),        # contiguous runs of stripes through the same machinery
    (2, 40, 2, None, False, True, 100, 3.0),       # sparse shards: only the tiles that hold something travel
    (3, 43, 2, None, False, True, 72, 3.0),        # ... ragged tiles at the right and bottom edges
    (3, 52, 1, [1, 2, 2], False, True, 96, 3.0),   # ... a root with the smaller share
    (3, 72, 2, None, False, True, 96, 30.0),      # ... a fractal a few pixels wide: most ranks have nothing to send
])
def test_shard_gather_equals_single_frames(world, height, count, weights, contiguous, sparse, width, distance, tmp_path,
                                            oracle):
    """bench.py's default at N > 1: every rank renders its stripes of the step's frames, one message
    per peer (dense rows, or records of the non-background tiles with the sizes exchanged a step behind),
    the root unpacks; the gathered frames equal single-process frames."""
    steps = 5 if sparse else 3  # (sparse: every buffer slot comes round again, its previous records erased)
    mp.spawn(_shard_worker,
             args=(world, _free_port(), width, height, steps, count, weights, contiguous, str(tmp_path), sparse, distance),
             nprocs=world, join=True)
    got = np.load(tmp_path / "shards.npy")
    assert got.shape == (steps * count, height, width, 4)
    sc = oracle.screen_uniform(width, height)
    opt = oracle.options_from_gui(fractal_group=1, constant=(-0.2, 0.6, 0.2, 0.2), max_iterations=40)
    for f in range(steps * count):
        want = oracle.render(sc, oracle.camera_uniform(distance, 0.3 * f, 0.1), opt, oracle.iters(8, 4, 4))
        assert (got[f] == want).all(), f"frame {f} differs"
    assert (got[0] != got[1]).any()
    if sparse:
        sent, seen = np.load(tmp_path / "sparse_stats.npy")
        assert sent < seen, "the background tiles did not travel"
        assert sent > 0 or distance > 10.0, "some tiles travelled"


def test_sparse_records_round_trip():
    """pack_sparse_torch / unpack_sparse_torch / fill_stripes_torch (the CPU forms of the library's kernels):
    a frame's stripes -> records -> the frame again, for ragged widths and heights; all-background shards make
    no records, a shard without background one per tile."""
    from kifs_raymarching_amd import bands
    rng = np.random.default_rng(5)
    bg = 0xff332211
    bg_px = np.array([0x11, 0x22, 0x33, 0xff], dtype=np.uint8)
    for (W, H, count) in ((100, 43, 2), (64, 16, 1), (33, 9, 3)):
        frames = np.broadcast_to(bg_px, (count, H, W, 4)).copy()
        for _ in range(6):  # a few blobs
            f, y, x = rng.integers(count), rng.integers(H), rng.integers(W)
            frames[f, y:y + rng.integers(1, 12), x:x + rng.integers(1, 40)] = rng.integers(0, 255, 4, dtype=np.uint8)
        n_stripes = (H + 7) // 8
        stripes = [s for s in range(n_stripes) if s % 2 == 0]
        rows = [y for s in stripes for y in range(8 * s, min(H, 8 * s + 8))]
        shards = torch.from_numpy(frames[:, rows])
        rec = bands.pack_sparse_torch(shards, stripes, H, bg)
        tiles = count * len(stripes) * ((W + 31) // 32)
        assert rec.shape[1] == 1040 and 0 < rec.shape[0] <= tiles
        out = torch.zeros((count, H, W, 4), dtype=torch.uint8)
        bands.fill_stripes_torch(out, stripes, bg)
        bands.unpack_sparse_torch(out, rec, stripes)
        assert (out.numpy()[:, rows] == frames[:, rows]).all()
        other = [y for y in range(H) if y not in rows]
        assert (out.numpy()[:, other] == 0).all()          # rows of other stripes untouched
        assert bands.pack_sparse_torch(torch.from_numpy(np.broadcast_to(bg_px, shards.shape).copy()), stripes, H, bg).shape[0] == 0
        full = torch.from_numpy(rng.integers(0, 200, shards.shape, dtype=np.uint8))
        assert bands.pack_sparse_torch(full, stripes, H, bg).shape[0] == tiles


def test_shard_stripes_partition(kifs):
    """kifs_shard_stripes: equal weights deal r, r + world, ..; weights shift shares; every stripe
    goes to exactly one rank; rows add up to the frame; bad arguments are refused."""
    for height, world in [(1080, 8), (1081, 8), (4096, 8), (4320, 8), (7, 3), (8, 1), (100, 5)]:
        all_stripes, total = [], 0
        for r in range(world):
            st, rows = kifs.shard_stripes(height, r, world)
            assert st == list(range(r, (height + 7) // 8, world))
            all_stripes += st
            total += rows
        assert sorted(all_stripes) == list(range((height + 7) // 8)) and total == height
    w = [3, 1, 1, 1, 1, 1, 1, 1]
    shares = [kifs.shard_stripes(1080, r, 8, w) for r in range(8)]
    assert sorted(s for st, _ in shares for s in st) == list(range(135))
    assert sum(rows for _, rows in shares) == 1080
    assert abs(len(shares[0][0]) - 135 * 3 / 10) <= 1 and all(abs(len(st) - 13.5) <= 1 for st, _ in shares[1:])
    gaps = np.diff(shares[0][0])
    assert gaps.max() <= 5, "the root's stripes are spread through the frame, not clumped"
    assert kifs.shard_stripes(64, 1, 2, [1, 0]) == ([], 0)  # a rank of weight 0 renders nothing
    for bad in [dict(height=-1, rank=0, world=2), dict(height=8, rank=2, world=2), dict(height=8, rank=0, world=0),
                dict(height=8, rank=0, world=2, weights=[0, 0]), dict(height=8, rank=0, world=2, weights=[-1, 2])]:
        with pytest.raises(kifs.KifsError):
            kifs.shard_stripes(**bad)
