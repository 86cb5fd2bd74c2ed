"""Randomised GPU-vs-oracle parity: scenes drawn from the GUI's parameter ranges
(render/gui.rs:24-105: max_iterations 1..=1000, max_distance 10..=10000, epsilon 1e-6..=1,
power 1..=10, constant in [-1,1]^4) plus random camera poses, every pipeline and primitive,
both colour targets, ragged frame sizes.  Seeded: the same 40 scenes every run."""
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
            max_iterations=int(rng.integers(1, 200 if not heavy else 48)),
            max_distance=float(10 ** rng.uniform(1, 4)),
            epsilon=float(10 ** rng.uniform(-6, -1.5)),
            fractal_color=tuple(int(v) for v in rng.integers(0, 256, 3)),
            background_color=tuple(int(v) for v in rng.integers(0, 256, 3)),
            is_heatmap=bool(rng.integers(0, 4) == 0),
            fractal_group=group, primitive_shape=prim,
            power=float(rng.uniform(1, 10)),
            constant=tuple(float(v) for v in rng.uniform(-1, 1, 4)))
        cam = K.CameraData(origin_distance=float(rng.uniform(2.0, 8.0)),
                           phi=float(rng.uniform(0, 2 * np.pi)), theta=float(rng.uniform(-1.5, 1.5)))
        iters = (int(rng.integers(0, 40 if not heavy else 8)), int(rng.integers(0, 12 if not heavy else 4)),
                 int(rng.integers(0, 24)))
        out.append((K.ScreenData(w, h), cam, gui, iters, int(rng.integers(0, 2))))
    return out


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
