"""GPU parity: full frames from the HIP library (through the C ABI) against the CPU
oracle on identical uniform bytes.  Bar: bit-exact RGBA8 (north_star tolerance is <=1
per channel; by construction the two perform the same binary32 operation sequence, so
any difference is a bug)."""
import numpy as np
import pytest

from helpers import diff_report, gpu_frame, oracle_frame

pytestmark = pytest.mark.gpu


def _cases(K):
    from kifs_raymarching_amd.configs import JULIA_C
    S, Cam, G = K.ScreenData, K.CameraData, K.GuiData
    FG, PS = K.FractalGroup, K.PrimitiveShape
    cases = {
        "cfg1_julia_256": (S(256, 256), Cam(), G(max_iterations=64, fractal_group=FG.JuliaSet,
                                                 constant=JULIA_C), (8, 10, 10)),
        "julia_cfg2_small": (S(480, 270), Cam(), G(max_iterations=256, fractal_group=FG.JuliaSet,
                                                   constant=JULIA_C), (12, 10, 10)),
        "julia_refconst_close": (S(320, 200), Cam(origin_distance=2.5, phi=0.7, theta=0.4),
                                 G(fractal_group=FG.JuliaSet), (100, 10, 10)),
        "julia_heatmap": (S(200, 120), Cam(origin_distance=3.0),
                          G(fractal_group=FG.JuliaSet, is_heatmap=True, constant=JULIA_C,
                            fractal_color=(255, 128, 30)), (12, 10, 10)),
        "sierpinski_cfg3_small": (S(480, 270), Cam(), G(fractal_group=FG.KaleidoscopicIFS,
                                                        primitive_shape=PS.SierpinskiTetrahedron),
                                  (100, 10, 16)),
        "sierpinski_close": (S(256, 256), Cam(origin_distance=3.0, phi=1.0, theta=0.3),
                             G(primitive_shape=PS.SierpinskiTetrahedron,
                               background_color=(10, 40, 90)), (100, 10, 10)),
        "genjulia_p2": (S(160, 120), Cam(origin_distance=3.0), G(max_iterations=64,
                        fractal_group=FG.GeneralizedJuliaSet, constant=JULIA_C), (8, 4, 10)),
        "genjulia_p3.5": (S(160, 120), Cam(origin_distance=3.0, phi=0.5),
                          G(max_iterations=64, fractal_group=FG.GeneralizedJuliaSet, power=3.5),
                          (8, 4, 10)),
    }
    for prim in (PS.Sphere, PS.Cylinder, PS.Box, PS.Torus, PS.Bunny):
        cases[f"prim_{prim.name}"] = (S(192, 160), Cam(origin_distance=3.5, phi=0.6, theta=0.5),
                                      G(primitive_shape=prim, fractal_color=(250, 120, 60),
                                        background_color=(5, 5, 30)), (100, 10, 10))
    return cases


CASE_NAMES = ["cfg1_julia_256", "julia_cfg2_small", "julia_refconst_close", "julia_heatmap",
              "sierpinski_cfg3_small", "sierpinski_close", "genjulia_p2", "genjulia_p3.5",
              "prim_Sphere", "prim_Cylinder", "prim_Box", "prim_Torus", "prim_Bunny"]


@pytest.mark.parametrize("name", CASE_NAMES)
@pytest.mark.parametrize("encode", [1, 0])
def test_frame_bit_exact(name, encode, gs, kifs, oracle):
    screen, cam, gui, iters = _cases(kifs)[name]
    want = oracle_frame(oracle, kifs, screen, cam, gui, iters, encode=encode)
    got = gpu_frame(gs, screen, cam, gui, iters, encode=encode)
    rep = diff_report(got, want)
    assert got.shape == want.shape
    assert rep["mismatched_pixels"] == 0, (name, rep)
    # sanity: the frame is not trivially empty
    if "heatmap" not in name:
        assert (want[..., :3] != want[0, 0, :3]).any()


@pytest.mark.parametrize("what", ["huge_max_distance", "far_origin", "far_origin_batched"])
def test_scenes_outside_the_cull_conditions(what, gs, kifs, oracle):
    """The bounding-sphere culls -- and with them the short square root of the long-ray loop's outside-the-sphere
    steps -- are on only for sane scenes: max_distance below 1e15 and every view's origin within 1e15 of the scene.
    Beyond that the general forms run; frames must equal the oracle's either way (they are not empty: from 20 units
    away the fractal is a few pixels, and a ray that overshoots by a rounding error of its huge t lands anywhere)."""
    import torch
    from kifs_raymarching_amd.configs import JULIA_C
    screen = kifs.ScreenData(160, 96)
    iters = (12, 10, 10)
    if what == "huge_max_distance":
        gui = kifs.GuiData(max_iterations=96, max_distance=3.0e16, fractal_group=kifs.FractalGroup.JuliaSet, constant=JULIA_C)
        cams = [kifs.CameraData(origin_distance=20.0, phi=0.3)]
    else:
        gui = kifs.GuiData(max_iterations=96, max_distance=1.0e4, fractal_group=kifs.FractalGroup.JuliaSet, constant=JULIA_C)
        cams = [kifs.CameraData(origin_distance=4.0e15, phi=0.3)]
        if what == "far_origin_batched":  # one far view switches the culls off for the whole launch
            cams = [kifs.CameraData(origin_distance=3.5, phi=0.3), kifs.CameraData(origin_distance=4.0e15, phi=0.3),
                    kifs.CameraData(origin_distance=5.0, theta=0.4)]
    wants = [oracle_frame(oracle, kifs, screen, cam, gui, iters) for cam in cams]
    if len(cams) == 1:
        got = [gpu_frame(gs, screen, cams[0], gui, iters)]
    else:
        gs.update_screen_data(screen)
        gs.update_options(gui)
        gs.set_iters(*iters)
        outs = [torch.zeros((96, 160, 4), dtype=torch.uint8, device="cuda:0") for _ in cams]
        stream = torch.cuda.Stream()
        gs.render_batch_async(outs, cams, stream=stream)
        stream.synchronize()
        got = [o.cpu().numpy() for o in outs]
    for g, w_ in zip(got, wants):
        assert diff_report(g, w_)["mismatched_pixels"] == 0, what
    if what == "far_origin_batched":
        assert (wants[0][..., :3] != wants[0][0, 0, :3]).any()


def test_unknown_primitive_is_all_background(gs, kifs, oracle):
    gui = kifs.GuiData(background_color=(30, 60, 90))
    u = gui.into_buffer_data()
    u.primitive_id = 17  # kifs.wgsl:154 returns 1.0: never hits
    gs.update_screen_data(kifs.ScreenData(64, 48))
    gs.set_camera(kifs.CameraData())
    gs.update_options(u)
    img = gs.render()
    assert (img == img[0, 0]).all() and img[0, 0, 3] == 255


@pytest.mark.parametrize("size,band,heatmap", [((77, 45), None, False), ((130, 61), (7, 38), False),
                                               ((96, 50), (25, 26), False), ((64, 40), None, True)])
def test_bunny_quad_kernel_ragged_frames_and_bands(size, band, heatmap, gs, kifs, oracle):
    """The bunny runs four lanes per pixel on quarter tiles (render_bunny_quad_kernel): frames
    that end inside a tile, bands that start and end inside a quarter tile, heatmap mode (no
    culls, so every wave loads the weights)."""
    screen = kifs.ScreenData(*size)
    cam = kifs.CameraData(origin_distance=2.6, phi=2.1, theta=-0.4)
    gui = kifs.GuiData(primitive_shape=kifs.PrimitiveShape.Bunny, max_iterations=80, is_heatmap=heatmap,
                       fractal_color=(90, 220, 140), background_color=(12, 0, 40))
    y0, y1 = band if band else (0, size[1])
    want = oracle_frame(oracle, kifs, screen, cam, gui, (100, 10, 10), y0=y0, y1=y1)
    got = gpu_frame(gs, screen, cam, gui, (100, 10, 10), y0=y0, y1=y1)
    assert got.shape == want.shape == (y1 - y0, size[0], 4)
    assert diff_report(got, want)["mismatched_pixels"] == 0
    assert (want[..., :3] != want[0, 0, :3]).any()


@pytest.mark.parametrize("scene", ["julia", "julia_ref", "sierpinski", "torus", "genjulia", "sierpinski_shadow",
                                   "bunny", "bunny_shadow"])
def test_requeued_march_equals_oracle(scene, gs, kifs, oracle):
    """render_group_kernel re-queues a workgroup's surviving rays into full waves every 8 / 16 march
    steps (marches of at least two rounds; not heatmap, not residency-capped lone Julia frames).
    Long marches on a ragged frame of more than 4096 tiles (smaller launches keep one wave per
    block) and on a band that cuts tiles: every pixel equals the oracle's, i.e. the
    one-ray-at-a-time march."""
    FG, PS = kifs.FractalGroup, kifs.PrimitiveShape
    cam = kifs.CameraData(origin_distance=2.6, phi=0.9, theta=0.35)
    gui, iters, size = {
        "julia": (kifs.GuiData(fractal_group=FG.JuliaSet, constant=(-0.2, 0.6, 0.2, 0.2), max_iterations=200),
                  (12, 10, 10), (1030, 1032)),  # 33 x 129 tiles: an odd count, the last pair is half empty
        "julia_ref": (kifs.GuiData(fractal_group=FG.JuliaSet, max_iterations=128), (100, 10, 10), (1030, 1040)),
        "sierpinski": (kifs.GuiData(primitive_shape=PS.SierpinskiTetrahedron, max_iterations=200,
                                    background_color=(3, 20, 60)), (100, 10, 14), (1030, 1040)),
        "torus": (kifs.GuiData(primitive_shape=PS.Torus, max_iterations=150), (100, 10, 10), (1030, 1040)),
        "genjulia": (kifs.GuiData(fractal_group=FG.GeneralizedJuliaSet, power=4.0, max_iterations=64),
                     (8, 4, 10), (1030, 1040)),
        "sierpinski_shadow": (kifs.GuiData(primitive_shape=PS.SierpinskiTetrahedron, max_iterations=96),
                              (100, 10, 10), (1030, 1040)),
        # the bunny takes the re-queuing path on batched launches only, four lanes per ray
        "bunny": (kifs.GuiData(primitive_shape=PS.Bunny, max_iterations=64, fractal_color=(240, 200, 90)),
                  (100, 10, 10), (1030, 1032)),
        "bunny_shadow": (kifs.GuiData(primitive_shape=PS.Bunny, max_iterations=48), (100, 10, 10), (1030, 1040)),
    }[scene]
    screen = kifs.ScreenData(*size)
    W, H = size
    gs.update_screen_data(screen)
    gs.set_camera(cam)
    gs.update_options(gui)
    gs.set_iters(*iters)
    ext = None
    if scene.endswith("_shadow"):
        gs.set_extensions(soft_shadow=True, shadow_steps=32, shadow_k=8.0, shadow_t0=0.02, shadow_max_t=6.0)
        ext = oracle.Ext(1, 32, 8.0, 0.02, 6.0)
    import torch
    julia = scene.startswith("julia")
    batched = julia or scene.startswith("bunny")
    if scene.startswith("bunny"):
        cam = kifs.CameraData(origin_distance=2.2, phi=0.9, theta=0.35)
        gs.set_camera(cam)
    try:
        for y0, y1 in [(0, H), (13, H - 21)]:
            s, c, o = __import__("helpers").oracle_uniforms(oracle, kifs, (screen, cam, gui))
            want = oracle.render(s, c, o, oracle.iters(*iters), y0=y0, y1=y1, ext=ext)
            if scene.startswith("bunny"):
                # the bunny's throughput path has four forms, chosen by the launch's load (kifs_schedule.cpp, rules::BUNNY_*:
                # disc tiles x views; ~850 a view here): four lanes per ray with one tile per workgroup (a farther camera:
                # ~700 heavy tiles in two views) or two (from 1600), the same with layer 2 of the network in LDS and three
                # waves per SIMD (from 2600), four waves per 64 rays (from 5000)
                far = kifs.CameraData(origin_distance=3.2, phi=0.9, theta=0.35)
                s_f, c_f, o_f = __import__("helpers").oracle_uniforms(oracle, kifs, (screen, far, gui))
                want_far = oracle.render(s_f, c_f, o_f, oracle.iters(*iters), y0=y0, y1=y1, ext=ext)
                for view_cam, view_want, views, kernel, tiles, form in ((far, want_far, 2, "render_group_kernel", 1, 0),
                                                                        (cam, want, 2, "render_group_kernel", 2, 0),
                                                                        (cam, want, 4, "render_group_kernel", 2, 2),
                                                                        (cam, want, 7, "render_bunny_coop_kernel", 2, 1)):
                    outs = [torch.zeros((y1 - y0, W, 4), dtype=torch.uint8, device="cuda:0") for _ in range(views)]
                    st = torch.cuda.Stream()
                    gs.render_batch_async(outs, [view_cam] * views, stream=st, y0=y0, y1=y1)
                    st.synchronize()
                    assert (gs.debug_last_kernel(), gs.debug_last_group_tiles(), gs.debug_last_bunny_form()) == (kernel, tiles, form), (scene, views)
                    got = outs[-1].cpu().numpy()
                    for o in outs[:-1]:
                        assert (o.cpu().numpy() == got).all()
                    assert diff_report(got, view_want)["mismatched_pixels"] == 0, (scene, y0, y1, views)
            elif batched:  # a batch of two is never residency-capped
                outs = [torch.zeros((y1 - y0, W, 4), dtype=torch.uint8, device="cuda:0") for _ in range(2)]
                st = torch.cuda.Stream()
                gs.render_batch_async(outs, [cam, cam], stream=st, y0=y0, y1=y1)
                st.synchronize()
                got = outs[1].cpu().numpy()
                assert (outs[0].cpu().numpy() == got).all()
            else:
                got = gs.render(y0=y0, y1=y1)
            # (the bunny's last launch is the one with four waves per 64 rays: rounds of 4)
            assert gs.debug_last_round_steps() == (16 if julia or scene == "genjulia" else 4 if scene.startswith("bunny") else 8), scene
            assert diff_report(got, want)["mismatched_pixels"] == 0, (scene, y0, y1)
            assert (want[..., :3] != want[0, 0, :3]).any()
    finally:
        gs.set_extensions(soft_shadow=False)


@pytest.mark.parametrize("encode", [0, 1])
def test_bunny_four_waves_per_ray_chunk_equals_oracle(encode, gs, kifs, oracle):
    """render_bunny_coop_kernel: the four waves of a workgroup march the same 64 rays, one column group of the
    network each, activations through LDS.  Eight DIFFERENT views of a ragged frame (19 x 42 tiles, the last
    column 24 pixels wide, the last row 5 high), cameras from inside the unit sphere (every tile heavy, rays that
    start inside the network's domain) to 1.6 away, a short epsilon and a long one: three views per launch
    against the oracle, pixel for pixel."""
    import torch
    PS = kifs.PrimitiveShape
    W, H = 600, 333
    screen = kifs.ScreenData(W, H)
    gs.update_screen_data(screen)
    gs.set_iters(100, 10, 10)
    for eps, colour in ((1e-4, (240, 200, 90)), (2e-2, (20, 250, 130))):
        gui = kifs.GuiData(primitive_shape=PS.Bunny, max_iterations=72, epsilon=eps, fractal_color=colour,
                           background_color=(9, 30, 66))
        gs.update_options(gui)
        cams = [kifs.CameraData(origin_distance=d, min_distance=0.5, phi=0.7 * k, theta=0.25 * (k % 3) - 0.2)
                for k, d in enumerate((0.9, 1.3, 1.45, 1.2, 1.6, 1.1, 1.3, 1.25))]  # (view 0 prices the launch: inside the sphere,
        # every tile heavy -- 8 x 798 tiles is past rules::BUNNY_COOP_FROM)
        outs = torch.zeros((len(cams), H, W, 4), dtype=torch.uint8, device="cuda:0")
        torch.cuda.synchronize()
        st = torch.cuda.Stream()
        gs.render_batch_async([outs[i] for i in range(len(cams))], cams, stream=st, encode=encode)
        st.synchronize()
        assert gs.debug_last_kernel() == "render_bunny_coop_kernel" and gs.debug_last_round_steps() == 4
        got = outs.cpu().numpy()
        # ... and the first six of the same views in one launch: 6 x 798 heavy tiles (past the 4096 workgroups from which
        # launches re-queue at all), the load at which the four-lanes form keeps layer 2 of the network in LDS
        # (render_group_kernel<KIFS, BUNNY, 2, true>: round 4) -- the same pixels
        outs6 = torch.zeros((6, H, W, 4), dtype=torch.uint8, device="cuda:0")
        gs.render_batch_async([outs6[i] for i in range(6)], cams[:6], stream=st, encode=encode)
        st.synchronize()
        assert (gs.debug_last_kernel(), gs.debug_last_group_tiles(), gs.debug_last_bunny_form()) == ("render_group_kernel", 2, 2)
        assert (outs6.cpu().numpy() == got[:6]).all()
        for k in (1, 4, 7) if eps < 1e-3 else (0, 1):
            s, c, o = __import__("helpers").oracle_uniforms(oracle, kifs, (screen, cams[k], gui))
            want = oracle.render(s, c, o, oracle.iters(100, 10, 10), encode=encode)
            assert diff_report(got[k], want)["mismatched_pixels"] == 0, (eps, k)
            assert (want[..., :3] != want[0, 0, :3]).any()


@pytest.mark.parametrize("prim,eps,dist", [("SierpinskiTetrahedron", 1e-3, 2.0), ("Torus", 2e-4, 1.3), ("Box", 5e-3, 1.73),
                                           ("julia", 1e-3, 2.0)])
def test_pooled_secondary_rays_in_the_wave_kernel_equal_the_oracle(prim, eps, dist, gs, kifs, oracle):
    """render_wave_kernel shades the hits of a tile with the soft-shadow extension in three passes: normals and direct
    terms, the secondary rays from a pool (a lane that finishes its ray takes the next one), the colours.  68 views
    from ON the bounding sphere (radius `dist`: every tile counts as heavy and the launch takes the one-wave-per-tile
    shape, while the camera stays outside the primitive) of a
    ragged frame whose tiles hold anything from no hit to 256; UNORM and sRGB; three views against the oracle."""
    import torch
    PS = kifs.PrimitiveShape
    W, H = 424, 300  # 14 x 38 tiles, the last column 8 wide, the last row 4 high
    screen = kifs.ScreenData(W, H)
    if prim == "julia":  # the Julia pipeline: its own SDF and analytic normal, the same secondary rays
        gui = kifs.GuiData(fractal_group=kifs.FractalGroup.JuliaSet, constant=(-0.2, 0.6, 0.2, 0.2), max_iterations=96, epsilon=eps,
                           fractal_color=(230, 180, 60), background_color=(10, 30, 70))
    else:
        gui = kifs.GuiData(primitive_shape=getattr(PS, prim), max_iterations=96, epsilon=eps, fractal_color=(230, 180, 60),
                           background_color=(10, 30, 70))
    iters = (12, 10, 9)
    gs.update_screen_data(screen)
    gs.update_options(gui)
    gs.set_iters(*iters)
    gs.set_extensions(soft_shadow=True, shadow_steps=24, shadow_k=8.0, shadow_t0=0.02, shadow_max_t=6.0)
    ext = oracle.Ext(1, 24, 8.0, 0.02, 6.0)
    cams = [kifs.CameraData(origin_distance=dist - 0.002 * (k % 3), min_distance=0.5, phi=0.39 * k, theta=0.21 * (k % 9) - 0.8)
            for k in range(68)]  # (127 200 pixels = 497 tiles' worth x 68 views: past the 32 000 of the one-wave-per-tile shape,
    shown = []                    #  and past the 64 views that travel in the kernel argument)
    try:
        for encode in (1, 0):
            outs = torch.zeros((len(cams), H, W, 4), dtype=torch.uint8, device="cuda:0")
            torch.cuda.synchronize()
            st = torch.cuda.Stream()
            gs.render_batch_async([outs[i] for i in range(len(cams))], cams, stream=st, encode=encode)
            st.synchronize()
            assert gs.debug_last_kernel() == "render_wave_kernel", gs.debug_last_kernel()
            got = outs.cpu().numpy()
            for k in (0, 29, 67) if encode else (17,):
                s, c, o = __import__("helpers").oracle_uniforms(oracle, kifs, (screen, cams[k], gui))
                want = oracle.render(s, c, o, oracle.iters(*iters), encode=encode, ext=ext)
                assert diff_report(got[k], want)["mismatched_pixels"] == 0, (prim, encode, k)
                shown.append(float((want[..., :3] != want[0, 0, :3]).any(-1).mean()))
    finally:
        gs.set_extensions(soft_shadow=False)
    assert max(shown) > 0.05, ("the checked views should show the primitive", shown)
