"""GPU behaviour of the C ABI itself: status codes, host/device destinations, pitch,
bands, async launches on a caller stream, and the BASELINE configurations at full size."""
import ctypes as C

import numpy as np
import pytest

from helpers import diff_report, gpu_frame, oracle_frame

pytestmark = pytest.mark.gpu


def test_render_before_configuration_is_an_error(kifs):
    from kifs_raymarching_amd._lib import lib
    st = C.c_int()
    ctx = lib.kifs_create(0, C.byref(st))
    assert ctx and st.value == 0
    try:
        buf = np.zeros(64, dtype=np.uint8)
        assert lib.kifs_render(ctx, buf.ctypes.data, 16, 0, 1, 1) == 4  # UNCONFIGURED
        s = kifs.ScreenData(4, 4).into_buffer_data()
        assert lib.kifs_set_screen(ctx, C.byref(s)) == 0
        assert lib.kifs_render(ctx, buf.ctypes.data, 16, 0, 1, 1) == 4  # camera/options missing
        bad = kifs.ScreenData(4, 4).into_buffer_data()
        bad.width = 0.0
        assert lib.kifs_set_screen(ctx, C.byref(bad)) == 3  # BAD_SIZE
        bad.width = 4.5
        assert lib.kifs_set_screen(ctx, C.byref(bad)) == 3
        o = kifs.GuiData().into_buffer_data()
        o.fractal_group_id = 3
        assert lib.kifs_set_options(ctx, C.byref(o)) == 7  # FractalGroup::from_id -> None
        assert lib.kifs_set_iters(ctx, -1, 0, 0) == 7
        assert lib.kifs_last_kernel_ms(ctx) < 0
    finally:
        lib.kifs_destroy(ctx)


def test_argument_checks_on_render(gs, kifs):
    from kifs_raymarching_amd._lib import lib
    gs.update_screen_data(kifs.ScreenData(32, 16))
    gs.set_camera(kifs.CameraData())
    gs.update_options(kifs.GuiData())
    buf = np.zeros(32 * 16 * 4, dtype=np.uint8)
    ctx = gs._ctx
    assert lib.kifs_render(ctx, buf.ctypes.data, 32 * 4, 0, 16, 1) == 0
    assert lib.kifs_render(ctx, buf.ctypes.data, 32 * 4, 0, 17, 1) == 7   # y1 > H
    assert lib.kifs_render(ctx, buf.ctypes.data, 32 * 4, 5, 4, 1) == 7    # y0 > y1
    assert lib.kifs_render(ctx, buf.ctypes.data, 32 * 4 - 4, 0, 16, 1) == 3  # pitch < 4W
    assert lib.kifs_render(ctx, buf.ctypes.data, 32 * 4, 0, 16, 2) == 7   # unknown encode
    assert lib.kifs_render(ctx, buf.ctypes.data, 32 * 4, 7, 7, 1) == 0    # empty band is fine
    assert gs.last_kernel_ms() > 0


def test_host_pitch_and_ragged_sizes(gs, kifs, oracle):
    """Widths/heights that are not multiples of the 32x8 tile, and a padded host pitch."""
    cam, gui = kifs.CameraData(origin_distance=3.0), kifs.GuiData(primitive_shape=kifs.PrimitiveShape.Torus)
    for w, h in ((1, 1), (33, 9), (31, 7), (100, 3), (65, 130)):
        screen = kifs.ScreenData(w, h)
        want = oracle_frame(oracle, kifs, screen, cam, gui, (100, 10, 10))
        got = gpu_frame(gs, screen, cam, gui, (100, 10, 10))
        assert (got == want).all(), (w, h)
        pitch = w * 4 + 24
        raw = np.full(pitch * h, 0xAB, dtype=np.uint8)
        gs.render(out=raw, pitch_bytes=pitch)
        rows = raw.reshape(h, pitch)
        assert (rows[:, :w * 4].reshape(h, w, 4) == want).all()
        assert (rows[:, w * 4:] == 0xAB).all()  # padding untouched
    # a host destination too small for its pitch is refused before the library copies into it
    gs.update_screen_data(kifs.ScreenData(16, 4))
    with pytest.raises(ValueError):
        gs.render(out=np.zeros(16 * 4 * 4, dtype=np.uint8), pitch_bytes=16 * 4 + 32)
    with pytest.raises(ValueError):
        gs.render(out=np.zeros(16 * 4 * 4, dtype=np.uint8), pitch_bytes=60)


def test_bands_tile_the_frame(gs, kifs, oracle):
    """SURVEY 8e on one GPU: N bands rendered one after another == the single frame."""
    screen, cam = kifs.ScreenData(200, 135), kifs.CameraData(origin_distance=3.2, phi=0.4)
    gui = kifs.GuiData(fractal_group=kifs.FractalGroup.JuliaSet, constant=(-0.2, 0.6, 0.2, 0.2))
    full = gpu_frame(gs, screen, cam, gui, (12, 10, 10))
    assert (full == oracle_frame(oracle, kifs, screen, cam, gui, (12, 10, 10))).all()
    for world in (2, 3, 8):
        parts = [gs.render(y0=a, y1=b) for a, b in (kifs.band_range(135, r, world) for r in range(world))]
        assert (np.concatenate(parts) == full).all(), world


def test_device_destination_and_async_stream(gs, kifs, oracle):
    import torch
    screen, cam = kifs.ScreenData(160, 90), kifs.CameraData()
    gui = kifs.GuiData(primitive_shape=kifs.PrimitiveShape.SierpinskiTetrahedron)
    want = oracle_frame(oracle, kifs, screen, cam, gui, (100, 10, 16))
    host = gpu_frame(gs, screen, cam, gui, (100, 10, 16))
    assert (host == want).all()
    dev = torch.zeros((90, 160, 4), dtype=torch.uint8, device="cuda:0")
    gs.render(out=dev)  # synchronous, device destination
    assert (dev.cpu().numpy() == want).all()
    s = torch.cuda.Stream()
    dev2 = torch.zeros_like(dev)
    with torch.cuda.stream(s):
        gs.render_async(dev2, stream=s)
        gs.render_async(dev2[45:], stream=s, y0=45, y1=90)  # overwrite the lower half again
    s.synchronize()
    assert (dev2.cpu().numpy() == want).all()
    dev3 = torch.zeros_like(dev)
    gs.render_async(dev3)  # context stream
    gs.synchronize()
    assert (dev3.cpu().numpy() == want).all()
    with pytest.raises(ValueError):
        gs.render_async(dev3, stream=torch.cuda.default_stream())


def test_bandframe_on_gpu_world1(gs, kifs, oracle):
    import torch
    from kifs_raymarching_amd.bands import BandFrame
    screen, cam = kifs.ScreenData(96, 54), kifs.CameraData()
    gui = kifs.GuiData(fractal_group=kifs.FractalGroup.JuliaSet, constant=(-0.2, 0.6, 0.2, 0.2))
    gpu_frame(gs, screen, cam, gui, (12, 10, 10))
    bf = BandFrame(96, 54, 0, 1, "cuda:0")
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for k in range(3):
            bf.step(k, lambda out, y0, y1: gs.render_async(out, stream=s, y0=y0, y1=y1))
        bf.wait_all()
    s.synchronize()
    want = oracle_frame(oracle, kifs, screen, cam, gui, (12, 10, 10))
    assert (bf.frame(2).cpu().numpy() == want).all()


FULL = ["cfg2_julia_1080p", "cfg3_sierpinski_1080p", "ref_julia_1080p"]


@pytest.mark.parametrize("key", FULL)
def test_baseline_configs_full_size(key, gs, kifs, oracle):
    """The 1080p BASELINE workloads, every pixel, against the oracle (multi-threaded)."""
    from kifs_raymarching_amd.configs import WORKLOADS
    w = WORKLOADS[key]
    got = gpu_frame(gs, w.screen, w.camera, w.gui, w.iters)
    want = oracle_frame(oracle, kifs, w.screen, w.camera, w.gui, w.iters)
    rep = diff_report(got, want)
    assert rep["mismatched_pixels"] == 0, rep
    assert (want[..., 0] != want[0, 0, 0]).sum() > 10000


def test_cfg4_4096_bands_and_symmetry_properties(gs, kifs, oracle):
    """4096x4096 (config 4): eight 512-row bands equal the full frame; two bands are
    checked against the oracle; untouched corners are exact background."""
    from kifs_raymarching_amd.configs import WORKLOADS
    w = WORKLOADS["cfg4_julia_4096"]
    full = gpu_frame(gs, w.screen, w.camera, w.gui, w.iters)
    assert full.shape == (4096, 4096, 4)
    for r in range(8):
        y0, y1 = kifs.band_range(4096, r, 8)
        assert (y0, y1) == (512 * r, 512 * (r + 1))
        assert (gs.render(y0=y0, y1=y1) == full[y0:y1]).all(), r
    for y0, y1 in ((2040, 2056), (1400, 1408)):
        want = oracle_frame(oracle, kifs, w.screen, w.camera, w.gui, w.iters, y0=y0, y1=y1)
        assert (full[y0:y1] == want).all()
    assert (full[:64, :64] == full[0, 0]).all() and (full[..., 3] == 255).all()


def test_orbit_frames_match_oracle(gs, kifs, oracle):
    """Config-5 style orbit (host camera scripting per graphics.rs:280-302), thumbnails."""
    from kifs_raymarching_amd.configs import WORKLOADS, orbit_camera
    w = WORKLOADS["cfg5_sierpinski_8k_orbit"]
    screen = kifs.ScreenData(192, 108)
    for k in (0, 17, 60, 119):
        cam = orbit_camera(w, k)
        got = gpu_frame(gs, screen, cam, w.gui, w.iters)
        want = oracle_frame(oracle, kifs, screen, cam, w.gui, w.iters)
        assert (got == want).all(), k


def test_single_process_multi_device_render(kifs, oracle):
    """kifs_multi_*: one context per listed device, stripe shards collected on the root by peer copies
    and unpacked.  On a one-GPU box the device is listed several times; the frame must equal the
    oracle's, with equal and with unequal shares."""
    import torch
    screen, cam = kifs.ScreenData(200, 135), kifs.CameraData(origin_distance=3.2, phi=0.4)
    gui = kifs.GuiData(fractal_group=kifs.FractalGroup.JuliaSet, constant=(-0.2, 0.6, 0.2, 0.2))
    want = oracle_frame(oracle, kifs, screen, cam, gui, (12, 10, 10))
    for n in (1, 2, 5):
        with kifs.MultiGraphicState([0] * n, screen, cam, gui, iters=(12, 10, 10)) as mg:
            assert (mg.render() == want).all(), n                       # host destination
            dev = torch.zeros((135, 200, 4), dtype=torch.uint8, device="cuda:0")
            mg.render(out=dev)                                          # root-device destination
            assert (dev.cpu().numpy() == want).all(), n
            padded = torch.full((135, 200 * 4 + 64), 7, dtype=torch.uint8, device="cuda:0")
            mg.render(out=padded, pitch_bytes=200 * 4 + 64)             # padded rows
            got = padded.cpu().numpy()
            assert (got[:, :800].reshape(135, 200, 4) == want).all() and (got[:, 800:] == 7).all()
            shards = mg.shards()
            assert sum(sh[1] for sh in shards) == 17 and sum(sh[2] for sh in shards) == 135  # 17 stripes, 135 rows
            assert all(sh[3] >= 0 for sh in shards) and shards[0][3] > 0
            if n > 1:
                mg.set_weights([3] + [1] * (n - 1))
                assert (mg.render() == want).all(), (n, "weighted")
                assert mg.shards()[0][1] > mg.shards()[1][1]
                with pytest.raises(kifs.KifsError):
                    mg.set_weights([0] * n)
    with pytest.raises(kifs.KifsError):
        kifs.MultiGraphicState([0, 4096], screen)


def test_multi_device_render_on_distinct_devices(kifs, oracle):
    """The same on as many DISTINCT devices as the box has (skipped on a one-GPU box): real peer copies
    over xGMI, and every device's shard must report its kernel time."""
    import torch
    n = torch.cuda.device_count()
    if n < 2:
        pytest.skip("needs at least two GPUs")
    screen, cam = kifs.ScreenData(640, 360), kifs.CameraData(origin_distance=3.0, phi=0.4)
    gui = kifs.GuiData(fractal_group=kifs.FractalGroup.JuliaSet, constant=(-0.2, 0.6, 0.2, 0.2))
    want = oracle_frame(oracle, kifs, screen, cam, gui, (12, 10, 10))
    with kifs.MultiGraphicState(list(range(n)), screen, cam, gui, iters=(12, 10, 10)) as mg:
        assert (mg.render() == want).all()
        shards = mg.shards()
        assert [sh[0] for sh in shards] == list(range(n)) and all(sh[3] > 0 for sh in shards)
        assert sum(sh[2] for sh in shards) == 360


def test_tile_order_feedback_never_changes_pixels(gs, kifs, oracle):
    """The tile order is re-derived every fourth launch from an earlier launch's per-tile costs
    by a sort running on a side stream (Julia frames of >= 2048 tiles).  Order may only affect
    speed: 30 launches alternating between the full frame, a band, two caller streams and the
    context stream, with the camera moving, must all equal the oracle, and the order table
    must stay a permutation of the frame's tiles."""
    import torch
    from kifs_raymarching_amd.configs import WORKLOADS, orbit_camera
    w = WORKLOADS["cfg2_julia_1080p"]
    W, H = w.screen.width, w.screen.height
    gs.update_screen_data(w.screen)
    gs.update_options(w.gui)
    gs.set_iters(*w.iters)
    streams = [torch.cuda.Stream(), torch.cuda.Stream(), None]
    bufs = [torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda:0") for _ in range(3)]
    cams, pending = {}, []
    for k in range(30):
        cam = orbit_camera(w, (k // 3) * 4)  # the view changes every third launch
        gs.set_camera(cam)
        s = streams[k % 3]
        band = (k % 5 == 4)
        y0, y1 = (H // 4, H // 4 + 600) if band else (0, H)
        out = bufs[k % 3]
        # make sure the buffer is not being written by an earlier launch on another stream
        torch.cuda.synchronize() if k % 3 == 0 else None
        gs.render_async(out[y0:y1], stream=s, y0=y0, y1=y1)
        pending.append((out, cam, y0, y1))
        if k % 3 == 2:
            torch.cuda.synchronize()
            gs.synchronize()
            for out_, cam_, a, b in pending:
                key = (round(cam_.phi, 6), a, b)
                if key not in cams:
                    cams[key] = oracle_frame(oracle, kifs, w.screen, cam_, w.gui, w.iters, y0=a, y1=b)
                assert (out_[a:b].cpu().numpy() == cams[key]).all(), (k, key)
            pending = []
    order = gs.debug_get_tile_order()
    tiles = {(int(o) & 0xffff, int(o) >> 16) for o in order}
    assert len(order) == 60 * 135 and len(tiles) == len(order)
    assert all(x < 60 and y < 135 for x, y in tiles)
    # the feedback really reordered the table: centre-first would start with the four tiles
    # around the frame centre (x in {29, 30}, y in {66, 67, 68})
    first = {(int(o) & 0xffff, int(o) >> 16) for o in order[:4]}
    assert not first <= {(x, y) for x in (29, 30) for y in (66, 67, 68)}


@pytest.mark.parametrize("scene", ["julia", "sierpinski", "bunny", "genjulia"])
def test_batch_launch_equals_single_frames(scene, gs, kifs, oracle):
    """kifs_render_batch_async: several frames per launch, one camera each, interleaved workgroups.
    Every frame must equal the same frame rendered alone (and the oracle's), for full frames and
    for a band that starts and ends inside a tile; the context's own camera is left alone."""
    import torch
    FG, PS = kifs.FractalGroup, kifs.PrimitiveShape
    gui, iters, size = {
        "julia": (kifs.GuiData(fractal_group=FG.JuliaSet, constant=(-0.2, 0.6, 0.2, 0.2), max_iterations=96),
                  (12, 10, 10), (330, 200)),
        "sierpinski": (kifs.GuiData(primitive_shape=PS.SierpinskiTetrahedron, max_iterations=96),
                       (100, 10, 12), (300, 170)),
        "bunny": (kifs.GuiData(primitive_shape=PS.Bunny, max_iterations=64), (100, 10, 10), (100, 60)),
        "genjulia": (kifs.GuiData(fractal_group=FG.GeneralizedJuliaSet, power=3.0, max_iterations=48),
                     (8, 4, 10), (120, 90)),
    }[scene]
    screen = kifs.ScreenData(*size)
    W, H = size
    own = kifs.CameraData(origin_distance=4.0, phi=0.3, theta=0.1)
    gs.update_screen_data(screen)
    gs.update_options(gui)
    gs.set_iters(*iters)
    gs.set_camera(own)
    cams = [kifs.CameraData(origin_distance=3.0 + 0.2 * k, phi=0.5 * k, theta=0.15 * k - 0.4) for k in range(8)]
    stream = torch.cuda.Stream()
    for n, (y0, y1) in [(1, (0, H)), (3, (0, H)), (8, (0, H)), (5, (5, H - 11))]:
        outs = [torch.zeros((y1 - y0, W, 4), dtype=torch.uint8, device="cuda:0") for _ in range(n)]
        gs.render_batch_async(outs, cams[:n], stream=stream, y0=y0, y1=y1)
        stream.synchronize()
        for k in range(n):
            want = oracle_frame(oracle, kifs, screen, cams[k], gui, iters, y0=y0, y1=y1)
            assert (outs[k].cpu().numpy() == want).all(), (scene, n, k)
    # the context's camera was not touched
    assert (gs.render() == oracle_frame(oracle, kifs, screen, own, gui, iters)).all()
    with pytest.raises(ValueError):
        gs.render_batch_async([], [], stream=stream)


def test_batch_launch_argument_checks(gs, kifs):
    import torch
    from kifs_raymarching_amd._lib import CameraUniform, lib
    gs.update_screen_data(kifs.ScreenData(64, 40))
    gs.update_options(kifs.GuiData())
    out = torch.zeros((40, 64, 4), dtype=torch.uint8, device="cuda:0")
    cams = (CameraUniform * 513)(*[kifs.CameraData().into_buffer_data() for _ in range(513)])
    ptrs = (C.c_void_p * 513)(*[out.data_ptr()] * 513)
    call = lambda n, cams_, ptrs_, pitch=256, y0=0, y1=40, enc=1: lib.kifs_render_batch_async(
        gs._ctx, None, n, cams_, ptrs_, pitch, y0, y1, enc)
    assert call(0, cams, ptrs) == 7 and call(513, cams, ptrs) == 7 and call(-1, cams, ptrs) == 7  # BAD_ARG
    assert call(2, None, ptrs) == 7 and call(2, cams, None) == 7
    null_second = (C.c_void_p * 2)(out.data_ptr(), None)
    assert call(2, cams, null_second) == 7
    odd = (C.c_void_p * 1)(out.data_ptr() + 2)
    assert call(1, cams, odd) == 7                       # unaligned destination
    assert call(2, cams, ptrs, pitch=100) == 3           # BAD_SIZE: pitch < 4 * width
    assert call(2, cams, ptrs, y0=5, y1=41) == 7 and call(2, cams, ptrs, enc=5) == 7
    assert call(2, cams, ptrs, y0=7, y1=7) == 0          # empty band: nothing to do
    assert call(8, cams, ptrs) == 0 and call(64, cams, ptrs) == 0
    gs.synchronize()


def test_batch_launch_of_the_headline_workload(gs, kifs, oracle):
    """Four 1080p Julia frames of an orbit in one launch (bench.py's default step) over several
    launches, so that the tile-order feedback also runs on batched launches."""
    import torch
    from kifs_raymarching_amd.configs import WORKLOADS, orbit_camera
    w = WORKLOADS["cfg2_julia_1080p"]
    W, H = w.screen.width, w.screen.height
    gs.update_screen_data(w.screen)
    gs.update_options(w.gui)
    gs.set_iters(*w.iters)
    stream = torch.cuda.Stream()
    cams = [orbit_camera(w, 7 * k) for k in range(4)]
    want = [oracle_frame(oracle, kifs, w.screen, c, w.gui, w.iters) for c in cams]
    outs = [torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda:0") for _ in range(4)]
    for launch in range(7):
        for o in outs:
            o.zero_()
        gs.render_batch_async(outs, cams, stream=stream)
        stream.synchronize()
        for k in range(4):
            assert (outs[k].cpu().numpy() == want[k]).all(), (launch, k)


def test_frames_in_flight_on_one_device(kifs, oracle):
    """Three contexts with a stream each render an orbit concurrently (kifs_set_frames_in_flight:
    no residency cap, kernels of different frames share the CUs); every frame equals the oracle's
    and the hint rejects n < 1."""
    import torch
    from kifs_raymarching_amd._lib import lib
    from kifs_raymarching_amd.configs import WORKLOADS, orbit_camera
    w = WORKLOADS["cfg2_julia_1080p"]
    W, H = w.screen.width, w.screen.height
    F = 3
    ctxs = [kifs.GraphicState(0, screen_data=w.screen, camera_data=w.camera, gui_data=w.gui)
            for _ in range(F)]
    try:
        assert lib.kifs_set_frames_in_flight(ctxs[0]._ctx, 0) == 7  # BAD_ARG
        streams = [torch.cuda.Stream() for _ in range(F)]
        outs = [torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda:0") for _ in range(4 * F)]
        for g in ctxs:
            g.set_iters(*w.iters)
            g.set_frames_in_flight(F)
        poses = [orbit_camera(w, 10 * k) for k in range(4)]
        for k in range(4 * F):  # pose k % 4 on context k % 3: every context sees every pose
            ctxs[k % F].set_camera(poses[k % 4])
            ctxs[k % F].render_async(outs[k], stream=streams[k % F], y0=0, y1=H)
        torch.cuda.synchronize()
        want = [oracle_frame(oracle, kifs, w.screen, cam, w.gui, w.iters) for cam in poses]
        for k in range(4 * F):
            assert (outs[k].cpu().numpy() == want[k % 4]).all(), k
    finally:
        for g in ctxs:
            g.close()


@pytest.mark.parametrize("scene", ["sierpinski", "torus", "julia", "bunny"])
def test_soft_shadow_extension_matches_oracle(scene, gs, kifs, oracle):
    """Soft shadows are an extension with no reference counterpart: the contract is the
    oracle's soft_shadow(), and the HIP path must match it bit for bit; switched off, the
    frame is the reference frame again."""
    FG, PS = kifs.FractalGroup, kifs.PrimitiveShape
    cfg = {
        "sierpinski": (kifs.CameraData(origin_distance=3.5, phi=0.6, theta=0.5),
                       kifs.GuiData(primitive_shape=PS.SierpinskiTetrahedron, fractal_color=(250, 200, 160)), (100, 10, 10)),
        "torus": (kifs.CameraData(origin_distance=3.5, phi=0.9, theta=-0.4),
                  kifs.GuiData(primitive_shape=PS.Torus, fractal_color=(120, 220, 250)), (100, 10, 10)),
        "julia": (kifs.CameraData(origin_distance=3.0, phi=0.3, theta=0.2),
                  kifs.GuiData(fractal_group=FG.JuliaSet, constant=(-0.2, 0.6, 0.2, 0.2)), (12, 10, 10)),
        "bunny": (kifs.CameraData(origin_distance=2.2, phi=0.6, theta=0.3),
                  kifs.GuiData(primitive_shape=PS.Bunny, max_iterations=96), (100, 10, 10)),
    }[scene]
    cam, gui, iters = cfg
    screen = kifs.ScreenData(160, 120) if scene != "bunny" else kifs.ScreenData(80, 60)
    plain = gpu_frame(gs, screen, cam, gui, iters)
    ext = oracle.Ext(1, 48, 8.0, 0.02, 10.0)
    from helpers import oracle_uniforms
    s, c, o = oracle_uniforms(oracle, kifs, (screen, cam, gui))
    want = oracle.render(s, c, o, oracle.iters(*iters), ext=ext)
    gs.set_extensions(soft_shadow=True, shadow_steps=48, shadow_k=8.0, shadow_t0=0.02, shadow_max_t=10.0)
    try:
        got = gs.render()
    finally:
        gs.set_extensions(soft_shadow=False)
    assert (got == want).all(), int((got != want).any(-1).sum())
    assert (want != plain).any(), "the shadow pass changed nothing"
    assert (gpu_frame(gs, screen, cam, gui, iters) == plain).all()


def test_cfg5_8k_frame_bands_shards_and_shadows(gs, kifs, oracle):
    """BASELINE config 5 at its real size, 7680x4320 KIFS Sierpinski (16 folds, orbit pose 17): the lone
    frame takes render_wave_kernel; eight 540-row bands == the frame; the eight stripe shards gathered
    == the frame; two 8-row bands against the oracle; corners are background; and one 8-row band of the
    soft-shadow workload against the oracle's definition of the extension."""
    import torch
    from helpers import oracle_uniforms
    from kifs_raymarching_amd.configs import WORKLOADS, orbit_camera
    w = WORKLOADS["cfg5_sierpinski_8k_orbit"]
    W, H = w.screen.width, w.screen.height
    assert (W, H) == (7680, 4320)
    cam = orbit_camera(w, 17)
    gs.update_screen_data(w.screen)
    gs.update_options(w.gui)
    gs.set_iters(*w.iters)
    gs.set_camera(cam)
    full = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda:0")
    gs.render(out=full)
    assert gs.debug_last_round_steps() > 0 and gs.debug_last_group_tiles() == 0  # one wave per tile
    band = torch.zeros((540, W, 4), dtype=torch.uint8, device="cuda:0")
    for r in range(8):
        y0, y1 = kifs.band_range(H, r, 8)
        assert (y0, y1) == (540 * r, 540 * (r + 1))
        band.zero_()
        gs.render(out=band, y0=y0, y1=y1)
        assert torch.equal(band, full[y0:y1]), r
    del band
    gathered = torch.zeros((1, H, W, 4), dtype=torch.uint8, device="cuda:0")
    stream = torch.cuda.Stream()
    for r in range(8):
        stripes, rows = kifs.shard_stripes(H, r, 8)
        shard = torch.zeros((1, rows, W, 4), dtype=torch.uint8, device="cuda:0")
        gs.render_shard_async([shard[0]], [cam], stripes, stream=stream)
        gs.unpack_shard_async(gathered, shard, stripes, stream=stream)
        stream.synchronize()
        del shard
    assert torch.equal(gathered[0], full)
    del gathered
    host = full.cpu().numpy()
    for y0 in (2160, 1400):
        want = oracle_frame(oracle, kifs, w.screen, cam, w.gui, w.iters, y0=y0, y1=y0 + 8)
        assert (host[y0:y0 + 8] == want).all(), y0
        assert (want != want[0, 0]).any()
    assert (host[:64, :64] == host[0, 0]).all() and (host[-64:, -64:] == host[0, 0]).all()
    assert (host[..., 3] == 255).all()
    # the soft-shadow workload (an extension: the oracle's definition is the contract)
    ws = WORKLOADS["cfg5_sierpinski_8k_orbit_shadows"]
    gs.set_extensions(**ws.extensions)
    try:
        got = gs.render(y0=2160, y1=2168)
        two = torch.zeros((2, H, W, 4), dtype=torch.uint8, device="cuda:0")  # and through a batched launch
        gs.render_batch_async([two[0], two[1]], [cam, orbit_camera(w, 18)], stream=stream)
        stream.synchronize()
        batched = two[0, 2160:2168].cpu().numpy()
        del two
    finally:
        gs.set_extensions(soft_shadow=False)
    s, c, o = oracle_uniforms(oracle, kifs, (ws.screen, cam, ws.gui))
    e = ws.extensions
    ext = oracle.Ext(1, e["shadow_steps"], e["shadow_k"], e["shadow_t0"], e["shadow_max_t"])
    want = oracle.render(s, c, o, oracle.iters(*ws.iters), y0=2160, y1=2168, ext=ext)
    assert (got == want).all() and (batched == want).all()
    assert (want != host[2160:2168]).any(), "the shadow pass changed nothing"
