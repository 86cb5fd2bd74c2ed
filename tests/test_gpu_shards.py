"""Row shards on the GPU: kifs_shard_stripes / kifs_render_shard_async / kifs_unpack_shard_async,
the building blocks of bench.py's N > 1 path, exercised on one device: the shards of all "ranks"
rendered one after another must tile the frame exactly (packed + unpack, and in place)."""
import ctypes as C

import numpy as np
import pytest

from helpers import oracle_frame

pytestmark = pytest.mark.gpu


def _scene(kifs, name):
    FG, PS = kifs.FractalGroup, kifs.PrimitiveShape
    return {
        "julia": (kifs.GuiData(fractal_group=FG.JuliaSet, constant=(-0.2, 0.6, 0.2, 0.2), max_iterations=96),
                  (12, 10, 10), (330, 203)),
        "sierpinski": (kifs.GuiData(primitive_shape=PS.SierpinskiTetrahedron, max_iterations=96),
                       (100, 10, 12), (300, 170)),
        "bunny": (kifs.GuiData(primitive_shape=PS.Bunny, max_iterations=64), (100, 10, 10), (100, 61)),
        "heatmap": (kifs.GuiData(fractal_group=FG.JuliaSet, constant=(-0.2, 0.6, 0.2, 0.2), max_iterations=64,
                                 is_heatmap=True), (12, 10, 10), (160, 99)),
    }[name]


@pytest.mark.parametrize("scene", ["julia", "sierpinski", "bunny", "heatmap"])
@pytest.mark.parametrize("world,weights", [(3, None), (8, None), (4, [3, 1, 1, 1])])
def test_shards_tile_the_frame(scene, world, weights, gs, kifs, oracle):
    import torch
    gui, iters, (W, H) = _scene(kifs, scene)
    screen = kifs.ScreenData(W, H)
    gs.update_screen_data(screen)
    gs.update_options(gui)
    gs.set_iters(*iters)
    cams = [kifs.CameraData(origin_distance=3.0 + 0.3 * k, phi=0.7 * k, theta=0.2 * k - 0.3) for k in range(3)]
    want = [oracle_frame(oracle, kifs, screen, c, gui, iters) for c in cams]
    stream = torch.cuda.Stream()
    n = len(cams)
    frames_packed = torch.zeros((n, H, W, 4), dtype=torch.uint8, device="cuda:0")
    frames_inplace = torch.zeros((n, H, W, 4), dtype=torch.uint8, device="cuda:0")
    total = 0
    for r in range(world):
        stripes, rows = kifs.shard_stripes(H, r, world, weights)
        total += rows
        if not stripes:
            continue
        shard = torch.full((n, rows, W, 4), 0x5A, dtype=torch.uint8, device="cuda:0")
        gs.render_shard_async([shard[i] for i in range(n)], cams, stripes, in_place=False, stream=stream)
        gs.unpack_shard_async(frames_packed, shard, stripes, stream=stream)
        gs.render_shard_async([frames_inplace[i] for i in range(n)], cams, stripes, in_place=True, stream=stream)
        stream.synchronize()
        # the packed shard itself: stripe k at rows [8k, 8k + 8)
        got = shard.cpu().numpy()
        y = 0
        for s in stripes:
            h = min(H, 8 * s + 8) - 8 * s
            for i in range(n):
                assert (got[i, y:y + h] == want[i][8 * s:8 * s + h]).all(), (scene, r, s, i)
            y += h
        assert y == rows
    assert total == H
    for i in range(n):
        assert (frames_packed[i].cpu().numpy() == want[i]).all(), (scene, world, i, "packed + unpack")
        assert (frames_inplace[i].cpu().numpy() == want[i]).all(), (scene, world, i, "in place")


def test_headline_shards_take_the_requeuing_kernel(gs, kifs, oracle):
    """1080p Julia, 8 orbit frames per launch, the 8 shards of an 8-GPU node rendered in turn on
    this device (render_group_kernel: the launches are large enough for the throughput path):
    gathered frames == frames rendered whole; one frame is checked against the oracle."""
    import torch
    from kifs_raymarching_amd.configs import WORKLOADS, orbit_camera
    w = WORKLOADS["cfg2_julia_1080p"]
    W, H = w.screen.width, w.screen.height
    gs.update_screen_data(w.screen)
    gs.update_options(w.gui)
    gs.set_iters(*w.iters)
    cams = [orbit_camera(w, 5 * k) for k in range(8)]
    stream = torch.cuda.Stream()
    whole = torch.zeros((8, H, W, 4), dtype=torch.uint8, device="cuda:0")
    gs.render_batch_async([whole[i] for i in range(8)], cams, stream=stream)
    gathered = torch.zeros_like(whole)
    used_rounds = []
    for r in range(8):
        stripes, rows = kifs.shard_stripes(H, r, 8)
        assert stripes == list(range(r, 135, 8))
        if r == 0:
            gs.render_shard_async([gathered[i] for i in range(8)], cams, stripes, in_place=True, stream=stream)
        else:
            shard = torch.zeros((8, rows, W, 4), dtype=torch.uint8, device="cuda:0")
            gs.render_shard_async([shard[i] for i in range(8)], cams, stripes, stream=stream)
            gs.unpack_shard_async(gathered, shard, stripes, stream=stream)
        used_rounds.append(gs.debug_last_round_steps())
    stream.synchronize()
    assert all(u > 0 for u in used_rounds), used_rounds
    assert torch.equal(gathered, whole)
    want = oracle_frame(oracle, kifs, w.screen, cams[3], w.gui, w.iters)
    assert (gathered[3].cpu().numpy() == want).all()


def test_batch_of_64_frames(gs, kifs, oracle):
    """64 views in one launch: the most that travel in the kernel argument (3.9 KB, limit 4 KB)."""
    import torch
    assert kifs.MAX_BATCH == 512
    gui = kifs.GuiData(fractal_group=kifs.FractalGroup.JuliaSet, constant=(-0.2, 0.6, 0.2, 0.2), max_iterations=64)
    screen = kifs.ScreenData(128, 72)
    gs.update_screen_data(screen)
    gs.update_options(gui)
    gs.set_iters(12, 10, 10)
    cams = [kifs.CameraData(origin_distance=3.0 + 0.03 * k, phi=0.2 * k, theta=0.02 * k - 0.6) for k in range(64)]
    outs = torch.zeros((64, 72, 128, 4), dtype=torch.uint8, device="cuda:0")
    stream = torch.cuda.Stream()
    gs.render_batch_async([outs[i] for i in range(64)], cams, stream=stream)
    stream.synchronize()
    got = outs.cpu().numpy()
    for k in (0, 7, 8, 19, 31, 32, 47, 63):
        assert (got[k] == oracle_frame(oracle, kifs, screen, cams[k], gui, (12, 10, 10))).all(), k
    assert len({got[k].tobytes() for k in range(64)}) == 64
    with pytest.raises(ValueError):
        gs.render_batch_async([outs[0]] * 513, (cams * 9)[:513], stream=stream)


def test_batches_beyond_the_kernel_argument(gs, kifs, oracle):
    """65 .. KIFS_MAX_BATCH = 512 views: the views go through a device table (a ring of four, uploaded before
    the launch).  A 384-frame shard launch -- what a rank of eight renders per step -- and six launches in a
    row on two streams with different cameras (the ring wraps); every checked frame == the oracle."""
    import torch
    gui = kifs.GuiData(fractal_group=kifs.FractalGroup.JuliaSet, constant=(-0.2, 0.6, 0.2, 0.2), max_iterations=64)
    W, H = 256, 144
    screen = kifs.ScreenData(W, H)
    gs.update_screen_data(screen)
    gs.update_options(gui)
    gs.set_iters(12, 10, 10)
    cam = lambda k: kifs.CameraData(origin_distance=3.0 + 0.002 * k, phi=0.05 * k, theta=0.001 * k - 0.4)
    stripes, rows = kifs.shard_stripes(H, 3, 8)
    n = 384
    shards = torch.zeros((n, rows, W, 4), dtype=torch.uint8, device="cuda:0")
    stream = torch.cuda.Stream()
    gs.render_shard_async([shards[i] for i in range(n)], [cam(k) for k in range(n)], stripes, stream=stream)
    stream.synchronize()
    got = shards.cpu().numpy()
    ys = [y for s in stripes for y in range(8 * s, min(H, 8 * s + 8))]
    for k in (0, 63, 64, 65, 200, 383):
        assert (got[k] == oracle_frame(oracle, kifs, screen, cam(k), gui, (12, 10, 10))[ys]).all(), k
    # the ring: six whole-frame launches of 80..85 views back to back, alternating streams
    other = torch.cuda.Stream()
    outs = [torch.zeros((80 + j, H, W, 4), dtype=torch.uint8, device="cuda:0") for j in range(6)]
    for j in range(6):
        gs.render_batch_async([outs[j][i] for i in range(80 + j)], [cam(1000 * j + i) for i in range(80 + j)],
                              stream=(stream, other)[j % 2])
    stream.synchronize()
    other.synchronize()
    for j in range(6):
        for i in (0, 64, 79 + j):
            want = oracle_frame(oracle, kifs, screen, cam(1000 * j + i), gui, (12, 10, 10))
            assert (outs[j][i].cpu().numpy() == want).all(), (j, i)


def test_shard_argument_checks(gs, kifs):
    import torch
    from kifs_raymarching_amd._lib import CameraUniform, lib
    gs.update_screen_data(kifs.ScreenData(64, 40))  # 5 stripes
    gs.update_options(kifs.GuiData())
    out = torch.zeros((2, 40, 64, 4), dtype=torch.uint8, device="cuda:0")
    cams = (CameraUniform * 2)(*[kifs.CameraData().into_buffer_data() for _ in range(2)])
    ptrs = (C.c_void_p * 2)(out[0].data_ptr(), out[1].data_ptr())
    arr = lambda *s: (C.c_int * len(s))(*s)

    def call(st, n=None, count=2, cams_=cams, ptrs_=ptrs, pitch=256, in_place=1, enc=1):
        return lib.kifs_render_shard_async(gs._ctx, None, count, cams_, ptrs_, pitch, st,
                                           len(st) if n is None else n, in_place, enc)
    assert call(arr(0, 2, 4)) == 0
    assert call(arr(0, 2, 4), n=0) == 0           # an empty shard renders nothing
    assert call(arr(2, 0)) == 7                   # not ascending
    assert call(arr(1, 1)) == 7                   # repeated
    assert call(arr(0, 5)) == 7                   # stripe 5 starts at row 40 = H
    assert call(arr(-1)) == 7
    assert lib.kifs_render_shard_async(gs._ctx, None, 2, cams, ptrs, 256, None, 1, 1, 1) == 7  # NULL list
    assert call(arr(0), cams_=None) == 7          # no cameras for two frames
    assert call(arr(0), count=1, cams_=None) == 0  # one frame: the context's camera
    assert call(arr(0), count=513) == 7
    assert call(arr(0), pitch=100) == 3
    assert call(arr(0), enc=9) == 7
    un = lambda st, fp=256, sp=256: lib.kifs_unpack_shard_async(gs._ctx, None, 2, out.data_ptr(), fp, 40 * 256,
                                                                 out.data_ptr(), sp, 8 * 256, st, len(st))
    assert un(arr(4, 2)) == 7 and un(arr(0), fp=100) == 3 and un(arr(0), sp=254) == 3
    gs.synchronize()
    # the Python wrapper checks shapes before any pointer reaches the library
    frames = torch.zeros((2, 40, 64, 4), dtype=torch.uint8, device="cuda:0")
    with pytest.raises(ValueError):
        gs.unpack_shard_async(frames, torch.zeros((2, 16, 64, 4), dtype=torch.uint8, device="cuda:0"), [0, 2, 4])  # 24 rows
    with pytest.raises(ValueError):
        gs.unpack_shard_async(frames[0], torch.zeros((2, 24, 64, 4), dtype=torch.uint8, device="cuda:0"), [0, 2, 4])
    gs.unpack_shard_async(frames, torch.ones((2, 24, 64, 4), dtype=torch.uint8, device="cuda:0"), [0, 2, 4])
    gs.synchronize()
    assert int(frames.sum()) == 2 * 24 * 64 * 4
    # ... and the render: packed shard tensors are too small for in_place, which writes up to frame row H - 1
    packed = torch.zeros((2, 24, 64, 4), dtype=torch.uint8, device="cuda:0")
    two = [kifs.CameraData(), kifs.CameraData(phi=1.0)]
    with pytest.raises(ValueError):
        gs.render_shard_async([packed[0], packed[1]], two, [0, 2, 4], in_place=True)
    with pytest.raises(ValueError):
        gs.render_shard_async(kifs.DevicePointers([packed[0], packed[1]]), two, [0, 1, 2, 4])  # 32 rows needed
    with pytest.raises(ValueError):
        gs.render_shard_async([packed[0], packed[1]], two, [0, 5])  # stripe 5 is outside the 40-row frame
    prepared = kifs.DevicePointers([packed[0], packed[1]])
    gs.render_shard_async(prepared, two, [0, 2, 4])
    gs.render_shard_async(prepared, two, [0, 2, 4])  # (checked once, remembered)
    gs.synchronize()
    # A stripe list is validated against the CURRENT frame height even when its device table is cached: a list
    # accepted for a 2160-row frame must be refused once kifs_set_screen made the frame 1080 rows.
    tall = [0, 100, 200, 269]
    gs.update_screen_data(kifs.ScreenData(64, 2160))
    big = torch.zeros((1, 32, 64, 4), dtype=torch.uint8, device="cuda:0")
    gs.render_shard_async([big[0]], [kifs.CameraData()], tall)
    gs.synchronize()
    gs.update_screen_data(kifs.ScreenData(64, 1080))
    st = arr(*tall)
    one = (C.c_void_p * 1)(big.data_ptr())
    assert lib.kifs_render_shard_async(gs._ctx, None, 1, None, one, 256, st, 4, 0, 1) == 7
    assert lib.kifs_unpack_shard_async(gs._ctx, None, 1, out.data_ptr(), 256, 1080 * 256, big.data_ptr(), 256, 32 * 256, st, 4) == 7
    assert lib.kifs_fill_shard_async(gs._ctx, None, 1, out.data_ptr(), 256, 1080 * 256, st, 4, 1) == 7
    assert lib.kifs_render_shard_async(gs._ctx, None, 1, None, one, 256, arr(0, 100), 2, 0, 1) == 0
    gs.synchronize()


def test_shardframes_on_gpu_world1(gs, kifs, oracle):
    """ShardFrames with one rank: the root renders every stripe in place, no exchange."""
    import torch
    from kifs_raymarching_amd.bands import ShardFrames
    screen = kifs.ScreenData(96, 54)
    gui = kifs.GuiData(fractal_group=kifs.FractalGroup.JuliaSet, constant=(-0.2, 0.6, 0.2, 0.2))
    gs.update_screen_data(screen)
    gs.update_options(gui)
    gs.set_iters(12, 10, 10)
    cams = [kifs.CameraData(phi=0.1 * k) for k in range(6)]
    sf = ShardFrames(96, 54, 0, 1, "cuda:0", frames_per_step=2)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for k in range(3):
            sf.step(k, lambda outs, first, stripes, in_place: gs.render_shard_async(
                outs, cams[first:first + 2], stripes, in_place=in_place, stream=s))
        sf.wait_all()
    s.synchronize()
    for i in range(2):
        want = oracle_frame(oracle, kifs, screen, cams[4 + i], gui, (12, 10, 10))
        assert (sf.frames(2)[i].cpu().numpy() == want).all()


def test_feedback_survives_a_change_of_pipeline(gs, kifs, oracle):
    """Tile-order feedback is on for 1080p Julia frames and off for the bunny: switching options
    Julia -> bunny -> Julia between launches (with a sort possibly still running on the side
    stream) must not lose or duplicate a tile."""
    import torch
    from kifs_raymarching_amd.configs import WORKLOADS, orbit_camera
    w = WORKLOADS["cfg2_julia_1080p"]
    bunny = WORKLOADS["n2_bunny_1080p"]
    W, H = w.screen.width, w.screen.height
    gs.update_screen_data(w.screen)
    gs.set_iters(*w.iters)
    cam = orbit_camera(w, 11)
    gs.set_camera(cam)
    want = oracle_frame(oracle, kifs, w.screen, cam, w.gui, w.iters)
    out = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda:0")
    band = torch.zeros((16, W, 4), dtype=torch.uint8, device="cuda:0")
    stream = torch.cuda.Stream()
    for cycle in range(6):
        gs.update_options(w.gui)
        for _ in range(2 + cycle % 3):  # stop the feedback period at every phase
            gs.render_async(out, stream=stream)
        gs.update_options(bunny.gui)
        gs.render_async(band, stream=stream, y0=536, y1=552)  # the table of another geometry
        gs.render_async(out[:8], stream=stream, y0=0, y1=8) if cycle % 2 else None
        gs.update_options(w.gui)
        out.zero_()
        gs.render_async(out, stream=stream)
        stream.synchronize()
        assert (out.cpu().numpy() == want).all(), cycle
    order = gs.debug_get_tile_order()
    assert len({int(o) for o in order}) == 60 * 135


@pytest.mark.parametrize("prim", ["Sphere", "Box", "SierpinskiTetrahedron"])
def test_wave_kernel_with_every_pixel_live_and_hitting(prim, gs, kifs, oracle):
    """render_wave_kernel keeps its hit list in the same LDS buffer as a ray queue.  The worst case for that
    layout: a camera on the bounding sphere looking at a solid that fills the frame -- all 256 rays of a
    tile live, whole chunks hitting in the same round."""
    import torch
    PS = kifs.PrimitiveShape
    gui = kifs.GuiData(primitive_shape=PS[prim], max_iterations=120, epsilon=1e-3, fractal_color=(250, 180, 90))
    screen = kifs.ScreenData(1024, 528)  # 32 x 66 tiles x 32 views: past the load at which KIFS launches go one wave per tile
    gs.update_screen_data(screen)
    gs.update_options(gui)
    gs.set_iters(100, 10, 6)
    cams = [kifs.CameraData(origin_distance=2.0 if prim != "Sphere" else 1.2, min_distance=1.0, phi=0.21 * k,
                            theta=0.05 * k - 0.7) for k in range(32)]
    outs = torch.zeros((32, 528, 1024, 4), dtype=torch.uint8, device="cuda:0")
    stream = torch.cuda.Stream()
    gs.render_batch_async([outs[i] for i in range(32)], cams, stream=stream)
    stream.synchronize()
    assert gs.debug_last_group_tiles() == 0 and gs.debug_last_round_steps() > 0
    got = outs.cpu().numpy()
    for k in (0, 13, 31):
        want = oracle_frame(oracle, kifs, screen, cams[k], gui, (100, 10, 6))
        assert (want != want[0, 0]).any(-1).mean() > 0.05, "the solid should be in view"
        assert (got[k] == want).all(), (prim, k, int((got[k] != want).any(-1).sum()))


def test_shards_of_a_big_batch_take_the_wave_kernel(gs, kifs, oracle):
    """Two ranks' shards of 64 orbit frames of the 1080p headline: each shard launch is heavy enough for
    render_wave_kernel (one wave per tile), here with stripe rows, packed and in place.  Gathered == the
    same 64 frames rendered whole; two of them against the oracle."""
    import torch
    from kifs_raymarching_amd.configs import WORKLOADS, orbit_camera
    w = WORKLOADS["cfg2_julia_1080p"]
    W, H = w.screen.width, w.screen.height
    gs.update_screen_data(w.screen)
    gs.update_options(w.gui)
    gs.set_iters(*w.iters)
    cams = [orbit_camera(w, k) for k in range(64)]
    stream = torch.cuda.Stream()
    whole = torch.zeros((64, H, W, 4), dtype=torch.uint8, device="cuda:0")
    gs.render_batch_async([whole[i] for i in range(64)], cams, stream=stream)
    assert gs.debug_last_group_tiles() == 0
    gathered = torch.zeros_like(whole)
    for r, weights in ((0, None), (1, None)):
        stripes, rows = kifs.shard_stripes(H, r, 2, weights)
        if r == 0:
            gs.render_shard_async([gathered[i] for i in range(64)], cams, stripes, in_place=True, stream=stream)
        else:
            shard = torch.zeros((64, rows, W, 4), dtype=torch.uint8, device="cuda:0")
            gs.render_shard_async([shard[i] for i in range(64)], cams, stripes, stream=stream)
            gs.unpack_shard_async(gathered, shard, stripes, stream=stream)
        assert gs.debug_last_group_tiles() == 0 and gs.debug_last_round_steps() > 0, r
    stream.synchronize()
    assert torch.equal(gathered, whole)
    for k in (5, 63):
        assert (gathered[k].cpu().numpy() == oracle_frame(oracle, kifs, w.screen, cams[k], w.gui, w.iters)).all(), k


def test_tile_level_exit_with_odd_cameras(gs, kifs, oracle):
    """render_wave_kernel leaves whole tiles after ONE test at the tile's centre (tile_is_culled), whose
    angle bound assumes an orthonormal camera matrix and a camera outside the grown sphere.  One launch of
    48 views whose cameras stress that: just outside and just inside the bounding sphere, very far, matrices
    that are scaled, sheared or mirrored (raw CameraUniform images: the boundary takes any 64 bytes), a camera
    that looks away from the fractal, and one that looks past it.  Every frame == the oracle on the same bytes."""
    import ctypes as C
    import numpy as np
    import torch
    from kifs_raymarching_amd._lib import CameraUniform
    from kifs_raymarching_amd.configs import WORKLOADS, orbit_camera
    w = WORKLOADS["cfg2_julia_1080p"]
    screen = kifs.ScreenData(1600, 900)  # x 48 views: past the load at which Julia launches go one wave per tile
    gs.update_screen_data(screen)
    gs.update_options(w.gui)
    gs.set_iters(*w.iters)

    def raw(cam, edit=None):
        u = cam.into_buffer_data()
        if edit:
            edit(u)
        return u

    def scale_row(r, f):
        def e(u):
            for k in range(3):
                u.matrix[r][k] *= f
        return e

    def shear(u):
        for k in range(3):
            u.matrix[1][k] += 0.4 * u.matrix[2][k]

    def look_away(u):
        for k in range(3):
            u.matrix[0][k] = -u.matrix[0][k]

    def look_past(u):  # the forward axis tilted by about 40 degrees: the fractal sits at the frame's edge
        for k in range(3):
            u.matrix[0][k] = 0.77 * u.matrix[0][k] + 0.64 * u.matrix[1][k]

    CD = kifs.CameraData
    special = [raw(CD(origin_distance=2.0 + 1e-3, min_distance=1.0)), raw(CD(origin_distance=2.2, min_distance=1.0, phi=1.0)),
               raw(CD(origin_distance=2.32, min_distance=1.0, theta=0.4)), raw(CD(origin_distance=2.45, phi=2.0)),
               raw(CD(origin_distance=40.0)), raw(CD(origin_distance=5.0), scale_row(1, 1.5)),
               raw(CD(origin_distance=5.0), scale_row(2, 0.5)), raw(CD(origin_distance=4.0), scale_row(0, 2.0)),
               raw(CD(origin_distance=5.0, phi=0.7), shear), raw(CD(origin_distance=5.0, theta=-0.5), scale_row(1, -1.0)),
               raw(CD(origin_distance=5.0), look_away), raw(CD(origin_distance=3.0, phi=0.3), look_past)]
    cams = special + [raw(orbit_camera(w, k)) for k in range(48 - len(special))]
    outs = torch.zeros((48, 900, 1600, 4), dtype=torch.uint8, device="cuda:0")
    stream = torch.cuda.Stream()
    gs.render_batch_async([outs[i] for i in range(48)], cams, stream=stream)
    stream.synchronize()
    assert gs.debug_last_group_tiles() == 0 and gs.debug_last_round_steps() > 0
    got = outs.cpu().numpy()
    s = oracle.from_bytes(oracle.Screen, kifs.uniform_bytes(screen.into_buffer_data()))
    o = oracle.from_bytes(oracle.Options, kifs.uniform_bytes(w.gui.into_buffer_data()))
    seen_fractal = 0
    for k in list(range(len(special))) + [20, 47]:
        c = oracle.from_bytes(oracle.Camera, kifs.uniform_bytes(cams[k]))
        want = oracle.render(s, c, o, oracle.iters(*w.iters), encode=1)
        seen_fractal += int((want != want[0, 0]).any())
        assert (got[k] == want).all(), (k, int((got[k] != want).any(-1).sum()))
    assert seen_fractal >= 10
    # orthonormal views alone take the tile-level exit; the launch above had it switched off by the odd ones
    cams2 = [raw(orbit_camera(w, k)) for k in range(44)] + special[:4]
    outs.zero_()
    gs.render_batch_async([outs[i] for i in range(48)], cams2, stream=stream)
    stream.synchronize()
    assert gs.debug_last_group_tiles() == 0
    got = outs.cpu().numpy()
    for k in (0, 30, 44, 45, 46, 47):
        c = oracle.from_bytes(oracle.Camera, kifs.uniform_bytes(cams2[k]))
        want = oracle.render(s, c, o, oracle.iters(*w.iters), encode=1)
        assert (got[k] == want).all(), (k, int((got[k] != want).any(-1).sum()))


@pytest.mark.parametrize("size", [(1920, 1080), (1000, 530)])
def test_sparse_shards_on_the_gpu(size, gs, kifs, oracle):
    """kifs_pack_sparse_async / kifs_unpack_sparse_async / kifs_fill_shard_async against their CPU forms
    (bands.pack_sparse_torch ..): rank 1 of 3's shards of four orbit frames -> records (the same set, in any
    order) -> background-filled frames == the frames' rows; ragged tiles at the right and bottom edges; a
    frame with no background at all; argument checks."""
    import torch
    from kifs_raymarching_amd import bands
    from kifs_raymarching_amd.configs import WORKLOADS, orbit_camera
    w = WORKLOADS["cfg2_julia_1080p"]
    W, H = size
    gs.update_screen_data(kifs.ScreenData(W, H))
    gs.update_options(w.gui)
    gs.set_iters(*w.iters)
    stripes, rows = kifs.shard_stripes(H, 1, 3)
    cams = [orbit_camera(w, 5 * k) for k in range(4)]
    stream = torch.cuda.Stream()
    shards = torch.zeros((4, rows, W, 4), dtype=torch.uint8, device="cuda:0")
    gs.render_shard_async([shards[i] for i in range(4)], cams, stripes, stream=stream)
    cap = gs.sparse_capacity(4, stripes)
    assert cap == 4 * len(stripes) * ((W + 31) // 32)
    records = torch.zeros((cap, 1040), dtype=torch.uint8, device="cuda:0")
    n_dev = torch.full((1,), 77, dtype=torch.int32, device="cuda:0")
    n_host = torch.zeros(1, dtype=torch.int32).pin_memory()
    gs.pack_sparse_async(shards, stripes, records, n_dev, n_host, stream=stream)
    stream.synchronize()
    n = int(n_host[0])
    assert n == int(n_dev.cpu()[0]) and 0 < n < cap // 3, "most of these frames is background"
    far = oracle_frame(oracle, kifs, kifs.ScreenData(64, 8), kifs.CameraData(origin_distance=50.0), w.gui, w.iters)[0, 0]
    bg = int(far[0]) | int(far[1]) << 8 | int(far[2]) << 16 | int(far[3]) << 24
    want = bands.pack_sparse_torch(shards.cpu(), stripes, H, bg)
    got = records[:n].cpu()
    order = torch.argsort(got[:, :4].contiguous().view(torch.int32).flatten())
    assert want.shape[0] == n and (got[order] == want).all()
    # the root's side: background under the stripes, records over it
    frames = torch.zeros((4, H, W, 4), dtype=torch.uint8, device="cuda:0")
    gs.fill_shard_async(frames, stripes, stream=stream)
    gs.unpack_sparse_async(frames, records, n, stripes, stream=stream)
    stream.synchronize()
    ys = [y for s in stripes for y in range(8 * s, min(H, 8 * s + 8))]
    assert (frames[:, ys] == shards).all()
    others = [y for y in range(H) if y not in set(ys)]
    assert int(frames[:, others].max()) == 0, "rows of other ranks' stripes are not touched"
    # a buffer that comes round again: the background back under these records only
    gs.erase_sparse_async(frames, records, n, stripes, stream=stream)
    stream.synchronize()
    bgt = torch.tensor([(bg >> s) & 255 for s in (0, 8, 16, 24)], dtype=torch.uint8, device="cuda:0")
    assert (frames[:, ys] == bgt).all() and int(frames[:, others].max()) == 0
    # a shard with no background: one record per tile; an all-background one: none
    noise = torch.randint(0, 200, shards.shape, dtype=torch.uint8, device="cuda:0")
    gs.pack_sparse_async(noise, stripes, records, n_dev, n_host, stream=stream)
    stream.synchronize()
    assert int(n_host[0]) == cap
    frames.zero_()
    gs.unpack_sparse_async(frames, records, cap, stripes, stream=stream)
    stream.synchronize()
    assert (frames[:, ys] == noise).all()
    blank = torch.zeros_like(shards)
    blank[...] = torch.tensor([(bg >> s) & 255 for s in (0, 8, 16, 24)], dtype=torch.uint8, device="cuda:0")
    gs.pack_sparse_async(blank, stripes, records, n_dev, n_host, stream=stream)
    stream.synchronize()
    assert int(n_host[0]) == 0
    gs.unpack_sparse_async(frames, records, 0, stripes, stream=stream)  # nothing to do is fine
    # argument checks
    with pytest.raises(ValueError):
        gs.pack_sparse_async(shards, stripes, records[:cap - 1], n_dev, n_host, stream=stream)   # no room for every tile
    with pytest.raises(ValueError):
        gs.pack_sparse_async(shards[:, :-1], stripes, records, n_dev, n_host, stream=stream)     # rows do not match the stripes
    with pytest.raises(ValueError):
        gs.pack_sparse_async(shards, stripes, records, n_dev, torch.zeros(1, dtype=torch.int32), stream=stream)  # not pinned
    with pytest.raises(ValueError):
        gs.unpack_sparse_async(frames, records, cap + 1, stripes, stream=stream)
    with pytest.raises(ValueError):
        gs.fill_shard_async(frames[:, :-1], stripes, stream=stream)
    # ids that do not belong to the shard are skipped, not written somewhere
    bad = records[:2].clone()
    bad[:, :4] = 255
    frames.zero_()
    gs.unpack_sparse_async(frames, bad, 2, stripes, stream=stream)
    stream.synchronize()
    assert int(frames.max()) == 0
