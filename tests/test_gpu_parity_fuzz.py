"""Randomised GPU-vs-oracle parity: scenes drawn from the GUI's parameter ranges
(render/gui.rs:24-105: max_iterations 1..=1000, max_distance 10..=10000, epsilon 1e-6..=1,
power 1..=10, constant in [-1,1]^4) plus random camera poses, every pipeline and primitive,
both colour targets, ragged frame sizes; every fifth camera sits ON the bounding sphere
(origin_distance = min_distance = 2, data.rs:105-113).  Seeded: the same scenes every run.
A second family renders BATCHES of views large enough to take the throughput kernels
(render_group_kernel, render_wave_kernel), which the small frames of the first never reach; a third renders
batches from OUTSIDE the bounding sphere on render_wave_kernel, where whole tiles leave after one test at
their centre (three more seeds of it, 518 frames, were compared once by hand: none differed)."""
import numpy as np
import pytest

from helpers import diff_report, gpu_frame, oracle_frame

pytestmark = pytest.mark.gpu


def scenes(K, n=40, seed=20250904):
    rng = np.random.default_rng(seed)
    FG, PS = K.FractalGroup, K.PrimitiveShape
    out = []
    for i in range(n):
        group = [FG.JuliaSet, FG.KaleidoscopicIFS, FG.GeneralizedJuliaSet, FG.JuliaSet][i % 4]
        prim = PS(int(rng.integers(0, 6)))
        heavy = group == FG.GeneralizedJuliaSet or (group == FG.KaleidoscopicIFS and prim == PS.Bunny)
        w = int(rng.integers(17, 120 if not heavy else 64))
        h = int(rng.integers(9, 90 if not heavy else 48))
        gui = K.GuiData(
            # (log-uniform over the GUI's 1..=1000, capped where the oracle would take minutes)
            max_iterations=int(10 ** rng.uniform(0, 3.0 if not heavy else 1.7)),
            max_distance=float(10 ** rng.uniform(1, 4)),
            epsilon=float(10 ** rng.uniform(-6, 0)),
            fractal_color=tuple(int(v) for v in rng.integers(0, 256, 3)),
            background_color=tuple(int(v) for v in rng.integers(0, 256, 3)),
            is_heatmap=bool(rng.integers(0, 4) == 0),
            fractal_group=group, primitive_shape=prim,
            power=float(rng.uniform(1, 10)),
            constant=tuple(float(v) for v in rng.uniform(-1, 1, 4)))
        cam = K.CameraData(origin_distance=2.0 if i % 5 == 4 else float(rng.uniform(2.0, 8.0)),
                           phi=float(rng.uniform(0, 2 * np.pi)), theta=float(rng.uniform(-1.5, 1.5)))
        iters = (int(rng.integers(0, 40 if not heavy else 8)), int(rng.integers(0, 12 if not heavy else 4)),
                 int(rng.integers(0, 24)))
        out.append((K.ScreenData(w, h), cam, gui, iters, int(rng.integers(0, 2))))
    return out


def big_scenes(K, n=12, seed=20261004):
    """Batched launches of >= 4096 workgroups: (screen, cameras, gui, iters, encode, expected shape) with
    shape 0 = render_wave_kernel (many views, cameras on the bounding sphere: every tile is heavy), 1 / 2 =
    tiles per workgroup of render_group_kernel."""
    rng = np.random.default_rng(seed)
    FG, PS = K.FractalGroup, K.PrimitiveShape
    out = []
    for i in range(n):
        group = [FG.JuliaSet, FG.KaleidoscopicIFS, FG.JuliaSet, FG.KaleidoscopicIFS, FG.GeneralizedJuliaSet][i % 5]
        prim = [PS.SierpinskiTetrahedron, PS.Torus, PS.Box, PS.Sphere, PS.Cylinder][int(rng.integers(0, 5))]
        wave = i % 3 == 0 and group != FG.GeneralizedJuliaSet
        w, h = int(rng.integers(600, 700)), int(rng.integers(400, 450))  # 19..22 x 50..57 tiles >= 1000
        if wave:  # comfortably past the load at which the library switches to one wave per tile
            w, h = int(rng.integers(800, 900)), int(rng.integers(500, 560))
        julia = group == FG.JuliaSet
        views = (16 if julia else 32) if wave else int(rng.integers(5, 9))
        gui = K.GuiData(
            max_iterations=int(rng.integers(40, 300)) if group != FG.GeneralizedJuliaSet else int(rng.integers(20, 40)),
            max_distance=float(10 ** rng.uniform(1, 4)),
            epsilon=float(10 ** rng.uniform(-5, -1)),
            fractal_color=tuple(int(v) for v in rng.integers(0, 256, 3)),
            background_color=tuple(int(v) for v in rng.integers(0, 256, 3)),
            fractal_group=group, primitive_shape=prim, power=float(rng.uniform(1, 10)),
            constant=tuple(float(v) for v in rng.uniform(-1, 1, 4)))
        # wave scenes: a camera on (Julia, B = 2) or inside (the primitives' B is 1 .. 2.24) the bounding
        # sphere, so that every tile of the frame counts as heavy
        d = (2.0 if julia else 1.2) if wave else float(rng.uniform(2.5, 6.0))
        cams = [K.CameraData(origin_distance=d, min_distance=1.0, phi=float(rng.uniform(0, 2 * np.pi)), theta=float(rng.uniform(-1.2, 1.2)))
                for _ in range(views)]
        iters = (int(rng.integers(1, 30)) if group != FG.GeneralizedJuliaSet else int(rng.integers(1, 6)),
                 int(rng.integers(0, 12)) if group != FG.GeneralizedJuliaSet else int(rng.integers(0, 3)),
                 int(rng.integers(0, 20)))
        load = w * h // 256 * views  # cameras on the sphere: every tile counts
        shape = 0 if wave else (2 if group != FG.GeneralizedJuliaSet else 1)
        assert not wave or load >= 32000 or (julia and load >= 16000)
        out.append((K.ScreenData(w, h), cams, gui, iters, int(rng.integers(0, 2)), shape))
    return out


@pytest.mark.parametrize("index", range(12))
def test_random_batch_bit_exact_on_the_throughput_kernels(index, gs, kifs, oracle):
    import torch
    screen, cams, gui, iters, encode, shape = big_scenes(kifs)[index]
    W, H = screen.width, screen.height
    gs.update_screen_data(screen)
    gs.update_options(gui)
    gs.set_iters(*iters)
    outs = torch.zeros((len(cams), H, W, 4), dtype=torch.uint8, device="cuda:0")
    stream = torch.cuda.Stream()
    gs.render_batch_async([outs[i] for i in range(len(cams))], cams, stream=stream, encode=encode)
    stream.synchronize()
    assert gs.debug_last_round_steps() > 0, "the launch was meant to re-queue its rays"
    if shape == 0 or gs.debug_last_group_tiles() == 0:
        assert gs.debug_last_group_tiles() == shape, (gs.debug_last_group_tiles(), shape)
    got = outs.cpu().numpy()
    for k in sorted({0, len(cams) // 2, len(cams) - 1, int(index) % len(cams)}):
        want = oracle_frame(oracle, kifs, screen, cams[k], gui, iters, encode=encode)
        rep = diff_report(got[k], want)
        assert rep["mismatched_pixels"] == 0, (index, k, gui, cams[k], iters, rep)


@pytest.mark.parametrize("index", range(40))
def test_random_scene_bit_exact(index, gs, kifs, oracle):
    screen, cam, gui, iters, encode = scenes(kifs)[index]
    want = oracle_frame(oracle, kifs, screen, cam, gui, iters, encode=encode)
    got = gpu_frame(gs, screen, cam, gui, iters, encode=encode)
    rep = diff_report(got, want)
    assert rep["mismatched_pixels"] == 0, (index, gui, cam, iters, rep)


def test_fuzz_scenes_are_not_trivial(kifs, oracle):
    """At least half of the random scenes show the fractal (not only background)."""
    shown = 0
    for screen, cam, gui, iters, encode in scenes(kifs):
        f = oracle_frame(oracle, kifs, screen, cam, gui, iters, encode=encode)
        shown += int((f != f[0, 0]).any())
    assert shown >= 20


def outside_scenes(K, n=10, seed=20261005):
    """Batches of 64 views from OUTSIDE the bounding sphere, large enough for render_wave_kernel, where whole
    tiles leave after one test at their centre (tile_is_culled): cameras at mixed distances just outside the
    sphere (a tile-sized ring around the projected sphere is what the test must not cut into), random
    directions, epsilon up to 0.3 (the sphere's radius is B + epsilon), every primitive's B."""
    rng = np.random.default_rng(seed)
    FG, PS = K.FractalGroup, K.PrimitiveShape
    out = []
    for i in range(n):
        julia = i % 2 == 0
        prim = [PS.SierpinskiTetrahedron, PS.Torus, PS.Box, PS.Sphere, PS.Cylinder][i % 5]
        bound = 2.0 if julia else {PS.SierpinskiTetrahedron: 2.0, PS.Torus: 1.3, PS.Box: 1.7320508, PS.Sphere: 1.0,
                                   PS.Cylinder: 2.2360680}[prim]
        eps = float(10 ** rng.uniform(-4, -0.5))
        gui = K.GuiData(
            max_iterations=int(rng.integers(40, 200)), max_distance=float(10 ** rng.uniform(1.5, 4)), epsilon=eps,
            fractal_color=tuple(int(v) for v in rng.integers(0, 256, 3)),
            background_color=tuple(int(v) for v in rng.integers(0, 256, 3)),
            fractal_group=FG.JuliaSet if julia else FG.KaleidoscopicIFS, primitive_shape=prim,
            constant=tuple(float(v) for v in rng.uniform(-1, 1, 4)))
        R = bound + eps
        # from 1 % outside the sphere of the quick test (sqrt(1.2) R) to 1.6 x: the projected sphere fills the frame
        # at the near end and covers about a third of it at the far end
        cams = [K.CameraData(origin_distance=float(R * 1.0955 * rng.uniform(1.01, 1.6)), min_distance=0.5,
                             phi=float(rng.uniform(0, 2 * np.pi)), theta=float(rng.uniform(-1.3, 1.3))) for _ in range(64)]
        iters = (int(rng.integers(4, 24)), int(rng.integers(0, 10)), int(rng.integers(2, 14)))
        out.append((K.ScreenData(int(rng.integers(36, 44)) * 32 - int(rng.integers(0, 2)) * 7, int(rng.integers(85, 100)) * 8 - int(rng.integers(0, 2)) * 3),
                    cams, gui, iters, int(rng.integers(0, 2))))
    return out


@pytest.mark.parametrize("index", range(10))
def test_random_batch_with_whole_tiles_leaving(index, gs, kifs, oracle):
    import torch
    screen, cams, gui, iters, encode = outside_scenes(kifs)[index]
    W, H = screen.width, screen.height
    gs.update_screen_data(screen)
    gs.update_options(gui)
    gs.set_iters(*iters)
    outs = torch.zeros((len(cams), H, W, 4), dtype=torch.uint8, device="cuda:0")
    stream = torch.cuda.Stream()
    gs.render_batch_async([outs[i] for i in range(len(cams))], cams, stream=stream, encode=encode)
    stream.synchronize()
    assert gs.debug_last_group_tiles() == 0 and gs.debug_last_round_steps() > 0, "meant for render_wave_kernel"
    got = outs.cpu().numpy()
    far = max(range(len(cams)), key=lambda k: cams[k].origin_distance)
    near = min(range(len(cams)), key=lambda k: cams[k].origin_distance)
    some_background = False
    for k in sorted({0, far, near, (7 * index) % len(cams)}):
        want = oracle_frame(oracle, kifs, screen, cams[k], gui, iters, encode=encode)
        rep = diff_report(got[k], want)
        assert rep["mismatched_pixels"] == 0, (index, k, gui, cams[k], iters, rep)
        some_background |= bool((want == want[0, 0]).all(-1).mean() > 0.3)
    assert some_background, "a view with room around the sphere belongs to every batch"
