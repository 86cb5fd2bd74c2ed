"""Analytic known answers for the oracle's shader half, plus the
cross-check against the independent NumPy restatement.  The reference pins nothing here
("parity unpinned"), so these are the anchors that keep the oracle honest."""
import ctypes as C
import math

import numpy as np
import pytest

F = np.float32


def _opts(O, **kw):
    return O.options_from_gui(**kw)


# ---- SDF known answers ---------------------------------------------------------------
def test_julia_outside_bounding_sphere_is_norm_minus_two(oracle):  # julia.wgsl:7-10
    o = _opts(oracle, fractal_group=1)
    it = oracle.iters()
    rng = np.random.default_rng(1)
    for _ in range(200):
        v = rng.normal(size=3)
        v = v / np.linalg.norm(v) * rng.uniform(2.01, 50.0)
        p = v.astype(F)
        n = F(np.sqrt(F(F(p[2] * p[2]) + F(F(p[1] * p[1]) + F(p[0] * p[0])))))
        got = oracle.scene_sdf(o, it, p)
        assert abs(got - (float(np.linalg.norm(p.astype(np.float64))) - 2.0)) < 2e-6 * max(1.0, float(n))


def test_julia_c0_interior_semantics(oracle):
    """c = 0, |q| < 1: the orbit collapses towards 0.  With few iterations log(qs) < 0 gives a
    negative estimate (a hit); with the reference's 100 iterations qs and dqs both underflow
    to 0, so the estimate is log(0) * sqrt(0/0) = NaN -- `NaN < epsilon` is false and
    `t += NaN` ends the march as a miss.  Both are the literal f32 semantics of julia.wgsl:15-26
    and must be preserved (no fast-math, no flush assumptions)."""
    o = _opts(oracle, fractal_group=1, constant=(0, 0, 0, 0))
    assert oracle.scene_sdf(o, oracle.iters(5, 1, 1), (0.3, 0.2, 0.1)) < 0.0
    assert math.isnan(oracle.scene_sdf(o, oracle.iters(), (0.3, 0.2, 0.1)))
    # a NaN estimate on the first step: camera inside the ball -> background, i stays 0
    cam = oracle.Camera()
    cam.origin[:] = [0.3, 0.2, 0.1]
    cam.matrix[0][:] = [1, 0, 0, 0]; cam.matrix[1][:] = [0, 1, 0, 0]; cam.matrix[2][:] = [0, 0, 1, 0]
    ob = _opts(oracle, fractal_group=1, constant=(0, 0, 0, 0), background_color=(50, 60, 70))
    i, rgba = oracle.shade_pixel(oracle.screen_uniform(4, 4), cam, ob, oracle.iters(), 1, 1)
    assert i == 1 and np.allclose(rgba[:3], [ob.background_color[k] for k in range(3)])


def test_julia_distance_estimate_for_c0(oracle):
    """For c = 0 the Douady-Hubbard estimate is exact-ish: q -> q^2 gives |q_n| = |q|^(2^n),
    d = 0.5 |q| ln|q| (in 4-D: |q|^2 = |p|^2 + w^2 with w = 0.1)."""
    o = _opts(oracle, fractal_group=1, constant=(0, 0, 0, 0), max_distance=1000.0)
    it = oracle.iters()
    for r in (1.2, 1.5, 1.9):
        p = (r, 0.0, 0.0)
        q = math.sqrt(r * r + 0.01)
        want = 0.5 * q * math.log(q)
        assert abs(oracle.scene_sdf(o, it, p) - want) < 2e-5 * want + 1e-6


def test_quat_sq_equals_quat_mul(oracle):  # quaternions.wgsl:30-50
    rng = np.random.default_rng(2)
    L = oracle.lib()
    for _ in range(100):
        q = rng.uniform(-2, 2, size=4).astype(F)
        a, b = (C.c_float * 4)(), (C.c_float * 4)()
        qc = (C.c_float * 4)(*q)
        L.kor_quat_sq(qc, a)
        L.kor_quat_mul(qc, qc, b)
        assert np.allclose(a[:], b[:], rtol=0, atol=2e-6)


def test_genjulia_power2_tracks_julia(oracle):
    """gen_julia.wgsl:16 with power = 2 has the same recurrence as julia.wgsl:16."""
    c = (-0.2, 0.6, 0.2, 0.2)
    oj = _opts(oracle, fractal_group=1, constant=c)
    og = _opts(oracle, fractal_group=2, constant=c, power=2.0)
    it = oracle.iters(8, 10, 10)
    rng = np.random.default_rng(3)
    for _ in range(100):
        p = rng.uniform(-1.2, 1.2, size=3)
        a, b = oracle.scene_sdf(oj, it, p), oracle.scene_sdf(og, it, p)
        assert abs(a - b) <= 2e-3 * max(abs(a), 1e-3)


def test_sierpinski_known_points(oracle):  # kifs.wgsl:68-81
    o = _opts(oracle, fractal_group=0, primitive_shape=4, max_distance=1000.0)
    # fixed point (1,1,1): fold is the identity there and 2p-1 = p, r = sqrt(3) every time
    for folds in (1, 4, 10, 16):
        d = oracle.scene_sdf(o, oracle.iters(100, 10, folds), (1.0, 1.0, 1.0))
        assert abs(d - (math.sqrt(3.0) - 2.0) / 2 ** folds) < 1e-7
    # a point at r >= max_distance never enters the loop: (r - 2) / 1
    d = oracle.scene_sdf(o, oracle.iters(), (2000.0, 0.0, 0.0))
    assert d == 1998.0
    # zero folds: plain (|p| - 2)
    assert oracle.scene_sdf(o, oracle.iters(100, 10, 0), (3.0, 4.0, 0.0)) == 3.0


def test_tetrahedral_fold_properties(oracle):  # kifs.wgsl:1-14, 56-66
    L = oracle.lib()
    rng = np.random.default_rng(4)
    for _ in range(200):
        p = rng.uniform(-3, 3, size=3).astype(F)
        out = (C.c_float * 3)()
        L.kor_tetrahedral_fold((C.c_float * 3)(*p), out)
        q = np.array(out[:], dtype=np.float64)
        assert abs(np.linalg.norm(q) - np.linalg.norm(p.astype(np.float64))) < 1e-5  # isometry
        assert q[0] + q[2] >= -1e-5  # ends on the positive side of the last mirror plane
    # points already on the positive side of all three planes are untouched
    out = (C.c_float * 3)()
    L.kor_tetrahedral_fold((C.c_float * 3)(1.0, 2.0, 3.0), out)
    assert out[:] == [1.0, 2.0, 3.0]


@pytest.mark.parametrize("prim,p,want", [
    (0, (3.0, 0.0, 0.0), 2.0),            # sphere r = 1
    (0, (0.0, 0.0, 0.5), -0.5),
    (1, (3.0, 0.0, 0.0), 2.0),            # cylinder r = 1, half-height 2 along z
    (1, (0.0, 0.0, 5.0), 3.0),
    (1, (0.0, 0.0, 0.0), -1.0),
    (2, (3.0, 0.0, 0.0), 2.0),            # box half-extent 1
    (2, (2.0, 2.0, 2.0), math.sqrt(3.0)),
    (2, (0.0, 0.0, 0.0), -1.0),
    (3, (1.0, 0.0, 0.0), -0.3),           # torus R = 1, r = 0.3 in the xy-plane
    (3, (0.0, 0.0, 0.0), 0.7),
    (3, (1.0, 0.0, 1.0), 0.7),
    (6, (0.1, 0.2, 0.3), 1.0),            # unknown id: kifs.wgsl:154
])
def test_primitive_sdfs(oracle, prim, p, want):  # kifs.wgsl:16-53
    o = _opts(oracle, fractal_group=0, primitive_shape=prim)
    assert abs(oracle.scene_sdf(o, oracle.iters(), p) - want) < 1e-6


def test_bunny_bounding_patch(oracle):  # kifs.wgsl:85-87
    o = _opts(oracle, fractal_group=0, primitive_shape=5)
    assert abs(oracle.scene_sdf(o, oracle.iters(), (3.0, 0.0, 4.0)) - 4.2) < 1e-6
    inside = oracle.scene_sdf(o, oracle.iters(), (0.0, 0.0, 0.0))
    assert -1.0 < inside < 0.0  # the origin is inside the bunny


# ---- rays, frames ----------------------------------------------------------------------
def test_centre_ray_points_at_origin(oracle):  # entry.wgsl:51-55
    sc, cam = oracle.screen_uniform(640, 480), oracle.camera_uniform()
    a, b = oracle.ray_direction(sc, cam, 319, 239), oracle.ray_direction(sc, cam, 320, 240)
    mid = (a + b) / 2
    assert abs(mid[0] + 1.0) < 1e-5 and abs(mid[1]) < 1e-6 and abs(mid[2]) < 1e-6
    top = oracle.ray_direction(sc, cam, 320, 0)
    assert top[2] > 0.69  # y = 0 is the top row: +z (90 degree vertical field of view)
    right = oracle.ray_direction(sc, cam, 639, 240)
    assert right[1] > 0.0


def test_sphere_frame_closed_form(oracle):
    """primitive 0 from (5,0,0): a ray hits iff its closest approach to the origin is < 1
    (up to epsilon); hit pixels are lit by diffuse = 0.1 + 0.9 clamp(n.(1,1,1))."""
    W, H = 160, 120
    sc, cam = oracle.screen_uniform(W, H), oracle.camera_uniform()
    o = _opts(oracle, fractal_group=0, primitive_shape=0, fractal_color=(255, 255, 255))
    img = oracle.render(sc, cam, o, oracle.iters(), encode=0)
    fc = o.fractal_color[0]
    n_checked = 0
    for y in range(0, H, 3):
        for x in range(0, W, 3):
            d = oracle.ray_direction(sc, cam, x, y).astype(np.float64)
            b = 5.0 * d[0]
            disc = b * b - 24.0  # |o + t d|^2 = 1
            if disc > 1e-2:
                t = -b - math.sqrt(disc)
                n = np.array([5.0, 0, 0]) + t * d
                diffuse = 0.1 + 0.9 * min(max(n.sum(), 0.0), 1.0)
                want = diffuse * fc * 255.0
                assert abs(float(img[y, x, 0]) - want) <= 1.5, (x, y)
                n_checked += 1
            elif disc < -1e-2:
                assert tuple(img[y, x]) == (0, 0, 0, 255)
    assert n_checked > 20


def test_background_pixels_are_exact(oracle):
    o = _opts(oracle, fractal_group=1, background_color=(12, 99, 240))
    img = oracle.render(oracle.screen_uniform(64, 48), oracle.camera_uniform(), o, oracle.iters(4, 2, 2))
    bg = [oracle.lib().kor_encode_channel(o.background_color[i], 1) for i in range(3)] + [255]
    assert list(img[0, 0]) == bg and list(img[-1, -1]) == bg
    # sRGB target inverts the host's sRGB->linear conversion up to the /256 quirk
    assert abs(int(bg[1]) - round(99 / 256 * 255)) <= 1


def test_heatmap_counts_steps(oracle):
    """All-miss frame (unknown primitive: sdf = 1): t grows by 1 per step, so i = number of
    steps until t >= max_distance or i == max_iterations (entry.wgsl:12,27)."""
    o = _opts(oracle, fractal_group=0, primitive_shape=6, is_heatmap=True, max_iterations=50,
              max_distance=20.0, fractal_color=(255, 255, 255))
    sc, cam = oracle.screen_uniform(16, 16), oracle.camera_uniform()
    i, rgba = oracle.shade_pixel(sc, cam, o, oracle.iters(), 3, 4)
    assert i == 20 and abs(rgba[0] - 20 / 50 * o.fractal_color[0]) < 1e-6
    o2 = _opts(oracle, fractal_group=0, primitive_shape=6, is_heatmap=True, max_iterations=10,
               max_distance=1000.0)
    i, _ = oracle.shade_pixel(sc, cam, o2, oracle.iters(), 0, 0)
    assert i == 10


def test_break_leaves_counter_unincremented(oracle):
    """A hit on the k-th SDF call leaves i = k-1 (entry.wgsl:15-21): a sphere seen from
    distance 5 is hit on the second evaluation for the centre ray (t: 0 -> 4 -> hit)."""
    o = _opts(oracle, fractal_group=0, primitive_shape=0, is_heatmap=True)
    sc, cam = oracle.screen_uniform(2, 2), oracle.camera_uniform()
    sc1 = oracle.screen_uniform(1, 1)
    i, _ = oracle.shade_pixel(sc1, cam, o, oracle.iters(), 0, 0)
    assert i == 1


def test_stats_are_consistent(oracle):
    sc, cam = oracle.screen_uniform(96, 64), oracle.camera_uniform(3.0)
    o = _opts(oracle, fractal_group=1, constant=(-0.2, 0.6, 0.2, 0.2), max_iterations=64)
    img, steps, st = oracle.render_stats(sc, cam, o, oracle.iters(8, 10, 10))
    assert (img == oracle.render(sc, cam, o, oracle.iters(8, 10, 10))).all()
    assert st.pixels == 96 * 64 and st.max_steps <= 64 and st.hits > 0
    assert st.march_steps == int(steps.astype(np.int64).sum()) + st.hits
    assert st.sdf_calls == st.march_steps  # Julia normals are analytic: no extra SDF calls


def test_render_rejects_bad_arguments(oracle):
    sc, cam, o = oracle.screen_uniform(8, 8), oracle.camera_uniform(), _opts(oracle)
    with pytest.raises(ValueError):
        oracle.render(sc, cam, o, y0=0, y1=9)
    assert oracle.render(sc, cam, o, y0=3, y1=3).shape == (0, 8, 4)


def test_bands_equal_full_frame(oracle):
    sc, cam = oracle.screen_uniform(50, 37), oracle.camera_uniform(3.0, 0.4, 0.2)
    o = _opts(oracle, fractal_group=0, primitive_shape=4)
    full = oracle.render(sc, cam, o)
    parts = [oracle.render(sc, cam, o, y0=a, y1=b) for a, b in ((0, 5), (5, 20), (20, 37))]
    assert (np.concatenate(parts) == full).all()
    assert (oracle.render(sc, cam, o, nthreads=1) == oracle.render(sc, cam, o, nthreads=5)).all()


# ---- colour target ---------------------------------------------------------------------
def test_srgb_thresholds(oracle):
    from oracle import kifs_oracle_np as NP
    t = oracle.srgb_thresholds()
    assert t[0] == 0.0 and (np.diff(t) > 0).all() and t[255] < 1.0
    # at each threshold the ideal encoder steps from k-1 to k
    k = np.arange(1, 256)
    assert (NP.srgb_encode_ideal(t[1:]) == k).all()
    assert (NP.srgb_encode_ideal(np.nextafter(t[1:], F(-1))) == k - 1).all()
    L = oracle.lib()
    rng = np.random.default_rng(5)
    xs = np.concatenate([rng.uniform(0, 1, 20000), rng.uniform(0, 0.01, 5000), [0, 1, 2, -1, 0.5]]).astype(F)
    got = np.array([L.kor_encode_channel(float(x), 1) for x in xs])
    assert (got == NP.srgb_encode_ideal(xs)).all()
    assert L.kor_encode_channel(float("nan"), 1) == 0 and L.kor_encode_channel(float("inf"), 1) == 255
    assert L.kor_encode_channel(float("nan"), 0) == 0 and L.kor_encode_channel(-0.0, 0) == 0
    assert L.kor_encode_channel(0.5, 0) == 128 and L.kor_encode_channel(1.0, 0) == 255


def test_product_srgb_table_matches_oracle(oracle, kifs):
    """The library builds its own table (kifs_api.cpp); both must be the same 256 floats."""
    import hashlib
    t = oracle.srgb_thresholds()
    golden = np.load(str(__import__("pathlib").Path(__file__).parent / "golden" / "srgb_thresholds.npy"))
    assert t.tobytes() == golden.tobytes()


# ---- independent restatement ------------------------------------------------------------
@pytest.mark.parametrize("case", ["julia", "julia_ref", "sierpinski", "torus", "heatmap", "genjulia_p8", "genjulia_p3",
                                  "genjulia_p2", "genjulia_p8_n3", "bunny", "bunny_close"])
def test_c_oracle_agrees_with_numpy_restatement(oracle, case):
    """Two independent readings of the WGSL (C with pinned op order vs NumPy with libm and
    no fma) must agree except where a 1-ulp difference flips `distance < epsilon`."""
    from oracle import kifs_oracle_np as NP
    sc = oracle.screen_uniform(128, 96)
    cfg = {
        "julia": (oracle.camera_uniform(), _opts(oracle, max_iterations=64, fractal_group=1,
                                                 constant=(-0.2, 0.6, 0.2, 0.2)), oracle.iters(8, 10, 10)),
        "julia_ref": (oracle.camera_uniform(2.5, 0.7, 0.4), _opts(oracle, fractal_group=1), oracle.iters()),
        "sierpinski": (oracle.camera_uniform(3.0, 1.0, 0.3), _opts(oracle, primitive_shape=4), oracle.iters()),
        "torus": (oracle.camera_uniform(3.5, 0.6, 0.5), _opts(oracle, primitive_shape=3,
                                                               fractal_color=(250, 120, 60)), oracle.iters()),
        "heatmap": (oracle.camera_uniform(3.0), _opts(oracle, fractal_group=1, is_heatmap=True,
                                                      constant=(-0.2, 0.6, 0.2, 0.2)), oracle.iters(12, 10, 10)),
        # N1 / N2: the NumPy side reads quat_pow (quaternions.wgsl:57-63), gen_julia.wgsl:16 and the bunny network
        # (kifs.wgsl:84-137) LITERALLY -- two lengths, true divisions, pow, separate log2 -- against the C oracle's
        # contract form (shared log2, reciprocal products, pinned polynomials)
        "genjulia_p8": (oracle.camera_uniform(3.0, 0.5, 0.3), _opts(oracle, fractal_group=2, power=8.0), oracle.iters()),
        "genjulia_p3": (oracle.camera_uniform(3.0, 1.5, -0.3), _opts(oracle, fractal_group=2, power=3.0, max_iterations=128),
                        oracle.iters(40, 10, 10)),
        "genjulia_p2": (oracle.camera_uniform(3.0), _opts(oracle, fractal_group=2, power=2.0, constant=(-0.2, 0.6, 0.2, 0.2),
                                                           max_iterations=64), oracle.iters(12, 10, 10)),
        "genjulia_p8_n3": (oracle.camera_uniform(2.6, 2.0, 0.1), _opts(oracle, fractal_group=2, power=8.0), oracle.iters(100, 3, 10)),
        "bunny": (oracle.camera_uniform(2.2, 0.8, 0.2), _opts(oracle, primitive_shape=5), oracle.iters()),
        "bunny_close": (oracle.camera_uniform(1.6, 3.8, -0.4), _opts(oracle, primitive_shape=5), oracle.iters()),
    }[case]
    cam, o, it = cfg
    a = oracle.render(sc, cam, o, it)
    b, _, hit = NP.render(sc, cam, o, it)
    d = np.abs(a.astype(int) - b.astype(int)).max(-1)
    assert hit.sum() > 50
    assert (d > 0).mean() < 0.003, f"{(d > 0).sum()} pixels differ"
    assert (d > 1).sum() <= 4, f"{(d > 1).sum()} pixels differ by more than 1"
