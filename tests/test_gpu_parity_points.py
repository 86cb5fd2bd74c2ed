"""GPU parity below the frame level: scene_SDF / get_normal at arbitrary points and the
pinned elementary functions, HIP (through the C ABI) vs oracle, compared as bit patterns
(NaNs compare equal to NaNs)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
F = np.float32


def same_bits(a, b):
    a, b = np.asarray(a, dtype=F), np.asarray(b, dtype=F)
    nan = np.isnan(a) & np.isnan(b)
    return (a.view(np.uint32) == b.view(np.uint32)) | nan


def points(n, seed, radius=2.5):
    rng = np.random.default_rng(seed)
    p = rng.uniform(-radius, radius, size=(n, 3)).astype(F)
    p[:8] = [[0, 0, 0], [1, 1, 1], [5, 0, 0], [0.3, 0.2, 0.1], [1e-30, 0, 0], [-0.0, 0.0, 2.0],
             [1e20, 1e20, 0], [0.5, -0.25, 0.125]]
    # inputs that leave the verified range of the constant-division shortcut in the fold
    # (exact cancellation, tiny and denormal sums, infinities, NaN) and must take the exact path
    p[8:20] = [[0.25, -0.25, 0.7], [0.7, 0.25, -0.25], [-0.25, 0.7, 0.25], [1e-35, 1e-36, 0.4],
               [-1e-40, 0.0, 0.0], [1e-45, -1e-45, 1e-45], [np.inf, 0.0, 0.0], [0.0, -np.inf, 1.0],
               [np.nan, 0.5, 0.5], [0.5, 0.5, np.nan], [-0.0, -0.0, -0.0], [3e38, 3e38, 0.0]]
    return p


CASES = [
    ("julia12", dict(fractal_group=1, constant=(-0.2, 0.6, 0.2, 0.2)), (12, 10, 10)),
    ("julia100", dict(fractal_group=1), (100, 10, 10)),
    ("julia_c0", dict(fractal_group=1, constant=(0, 0, 0, 0)), (100, 10, 10)),  # NaN / underflow paths
    ("genjulia2", dict(fractal_group=2, constant=(-0.2, 0.6, 0.2, 0.2), power=2.0), (8, 4, 10)),
    ("genjulia7.3", dict(fractal_group=2, power=7.3), (6, 3, 10)),
    ("sphere", dict(primitive_shape=0), (100, 10, 10)),
    ("cylinder", dict(primitive_shape=1), (100, 10, 10)),
    ("box", dict(primitive_shape=2), (100, 10, 10)),
    ("torus", dict(primitive_shape=3), (100, 10, 10)),
    ("sierpinski10", dict(primitive_shape=4), (100, 10, 10)),
    ("sierpinski16", dict(primitive_shape=4), (100, 10, 16)),
    ("sierpinski200", dict(primitive_shape=4, max_distance=1e30), (100, 10, 200)),  # scale -> inf
    ("bunny", dict(primitive_shape=5), (100, 10, 10)),
    ("other", dict(primitive_shape=9), (100, 10, 10)),
]


@pytest.mark.parametrize("name,gui,iters", CASES, ids=[c[0] for c in CASES])
def test_sdf_and_normal_bit_exact(name, gui, iters, gs, kifs, oracle):
    g = kifs.GuiData(**{k: (kifs.FractalGroup(v) if k == "fractal_group" else
                            v if k != "primitive_shape" else v) for k, v in gui.items()})
    u = g.into_buffer_data() if gui.get("primitive_shape", 0) <= 5 else None
    if u is None:  # out-of-enum primitive id: patch the packed image directly
        gg = dict(gui); gg["primitive_shape"] = 0
        u = kifs.GuiData(**gg).into_buffer_data()
        u.primitive_id = gui["primitive_shape"]
    gs.update_options(u)
    gs.set_iters(*iters)
    n = 4096 if name not in ("bunny", "genjulia7.3") else 1024
    pts = points(n, seed=hash(name) % 1000, radius=1.2 if name == "bunny" else 2.5)
    sdf, nrm = gs.eval_points(pts)
    o = oracle.from_bytes(oracle.Options, kifs.uniform_bytes(u))
    it = oracle.iters(*iters)
    L = oracle.lib()
    want_sdf = np.empty(n, dtype=F)
    want_nrm = np.empty((n, 3), dtype=F)
    buf = (C.c_float * 3)()
    for i in range(n):
        pc = (C.c_float * 3)(*pts[i])
        want_sdf[i] = L.kor_scene_sdf(C.byref(o), C.byref(it), pc)
        L.kor_get_normal(C.byref(o), C.byref(it), pc, buf)
        want_nrm[i] = buf[:]
    bad = ~same_bits(sdf, want_sdf)
    assert not bad.any(), (name, int(bad.sum()), pts[bad][:3], sdf[bad][:3], want_sdf[bad][:3])
    badn = ~same_bits(nrm, want_nrm).all(-1)
    assert not badn.any(), (name, int(badn.sum()), pts[badn][:3], nrm[badn][:3], want_nrm[badn][:3])


MATH = [(0, "kor_logf"), (1, "kor_log2f"), (2, "kor_exp2f"), (3, "kor_sinf"), (4, "kor_cosf"),
        (5, "kor_acosf"), (11, "kor_sinf")]  # 11: sin_flat, the branch-free form in the bunny network


def math_inputs(fn):
    rng = np.random.default_rng(100 + fn)
    special = np.array([0.0, -0.0, 1.0, -1.0, 0.5, 2.0, np.inf, -np.inf, np.nan, 1e-45, 1e-38,
                        1.17549435e-38, 3.4028235e38, 127.99999, 128.0, -149.0, -150.0, -151.0,
                        1048576.0, 1048577.0, 1e20, 0.70710677, 0.70710678, 0.7071068], dtype=F)
    if fn in (0, 1):
        body = np.exp(rng.uniform(-100, 88, 200000)).astype(F)
        body = np.concatenate([body, rng.uniform(0.4, 2.2, 100000).astype(F),
                               (rng.integers(1, 0x00800000, 2000, dtype=np.uint32)).view(F)])
    elif fn == 2:
        body = rng.uniform(-160, 135, 300000).astype(F)
    elif fn in (3, 4, 11):
        body = np.concatenate([rng.uniform(-40, 40, 200000), rng.uniform(-1e5, 1e5, 50000),
                               rng.uniform(-4e6, 4e6, 5000)]).astype(F)
    else:
        body = rng.uniform(-1.05, 1.05, 300000).astype(F)
    return np.concatenate([special, body])


@pytest.mark.parametrize("fn,oname", MATH, ids=[f"{m[1]}_{m[0]}" for m in MATH])
def test_elementary_functions_bit_exact(fn, oname, gs, oracle):
    xs = math_inputs(fn)
    got = gs.eval_math(fn, xs)
    f = getattr(oracle.lib(), oname)
    want = np.array([f(float(x)) for x in xs], dtype=F)
    bad = ~same_bits(got, want)
    assert not bad.any(), (oname, int(bad.sum()), xs[bad][:5], got[bad][:5], want[bad][:5])


def test_straight_line_cores_equal_the_oracle_on_ordinary_arguments(gs, oracle):
    """The cores of the generalised-Julia orbit step (kifs_device_math.hpp: exp2_core scales with one v_ldexp_f32, log2_core
    skips the denormal pre-scaling, the sin / cos quadrant is int(n) & 3) against the oracle's full functions wherever
    quat_pow_step lets a core's result through: |e| < 127.99999 for exp2 (every integer and half-integer neighbourhood,
    the denormal results below -126, the overflow at 128), 2^-60 <= x < 2^60 for log2, |x| <= 10 pi and out to 2^20 for
    sin / cos, |x| <= 1 for acos."""
    rng = np.random.default_rng(21)
    L = oracle.lib()
    halves = np.arange(-128, 129, 0.5, dtype=F)
    near = np.concatenate([np.nextafter(halves, F(-1e9)), np.nextafter(halves, F(1e9)), halves])
    ex = np.concatenate([near, rng.uniform(-127.99999, 127.99999, 400000).astype(F),
                         rng.uniform(-127.99999, -125.0, 50000).astype(F), rng.uniform(126.0, 127.99999, 50000).astype(F)])
    ex = ex[np.abs(ex) < F(127.99999)]
    lg = np.concatenate([np.exp2(rng.uniform(-60, 60, 300000)).astype(F), rng.uniform(0.4, 2.2, 100000).astype(F),
                         np.array([2.0 ** -60, 1.0, 0.70710677, 0.70710678, 0.7071068], dtype=F)])
    sc = np.concatenate([rng.uniform(-32, 32, 300000), rng.uniform(-1048576, 1048576, 100000),
                         np.arange(-40, 41) * (np.pi / 2)]).astype(F)
    ac = np.concatenate([rng.uniform(-1, 1, 300000).astype(F), np.array([1.0, -1.0, 0.5, -0.5, 0.0, -0.0], dtype=F),
                         np.nextafter(F(0.5), F(1)).reshape(1), np.nextafter(F(-0.5), F(-1)).reshape(1)])
    for fn, oname, xs in ((12, "kor_exp2f", ex), (13, "kor_log2f", lg), (14, "kor_sinf", sc), (15, "kor_cosf", sc),
                          (16, "kor_acosf", ac)):
        f = getattr(L, oname)
        got = gs.eval_math(fn, xs)
        want = np.array([f(float(x)) for x in xs], dtype=F)
        bad = ~same_bits(got, want)
        assert not bad.any(), (oname, int(bad.sum()), xs[bad][:5], got[bad][:5], want[bad][:5])


def test_pow_bit_exact(gs, oracle):
    rng = np.random.default_rng(7)
    xs = np.concatenate([rng.uniform(0, 40, 100000), [0.0, 1.0, np.inf, np.nan, -1.0]]).astype(F)
    f = oracle.lib().kor_powf
    for y in (1.0, 2.0, 3.5, 9.99, 0.5):
        got = gs.eval_math(6, xs, param=y)
        want = np.array([f(float(x), y) for x in xs], dtype=F)
        assert same_bits(got, want).all(), y


def test_encoders_bit_exact(gs, oracle):
    rng = np.random.default_rng(8)
    t = oracle.srgb_thresholds()
    xs = np.concatenate([rng.uniform(-0.1, 1.1, 200000).astype(F), t, np.nextafter(t, F(-1)),
                         np.nextafter(t, F(2)), np.array([np.nan, np.inf, -np.inf, -0.0], dtype=F)])
    enc = oracle.lib().kor_encode_channel
    for fn, mode in ((7, 1), (8, 0)):
        got = gs.eval_math(fn, xs)
        want = np.array([enc(float(x), mode) for x in xs], dtype=F)
        assert (got == want).all(), mode


def test_mid_range_reciprocal_and_sqrt_are_correctly_rounded_exhaustively(gs):
    """rcp_mid / sqrt_mid (the generalised-Julia step's shortened 1/x and sqrt) against IEEE division and
    square root for EVERY mantissa: a reciprocal's rounding depends on the mantissa only (scaling by a power
    of two is exact in the normal range), a square root's on the mantissa and the exponent's parity; the
    range's first and last binades are swept whole as well, and 0 for the square root (the acos tail)."""
    mant = np.arange(1 << 23, dtype=np.uint32)
    for fn, exps in ((9, (127, 127 - 60, 127 + 59)), (10, (127, 128, 127 - 60, 127 + 59))):
        for e in exps:
            xs = (mant | np.uint32(e << 23)).view(F)
            got = gs.eval_math(fn, xs)
            with np.errstate(all="ignore"):
                want = (F(1.0) / xs) if fn == 9 else np.sqrt(xs)
            bad = got.view(np.uint32) != want.astype(F).view(np.uint32)
            assert not bad.any(), (fn, e, int(bad.sum()), xs[bad][:4], got[bad][:4], want[bad][:4])
    assert gs.eval_math(10, np.array([0.0], dtype=F)).view(np.uint32)[0] == 0
    rng = np.random.default_rng(11)
    xs = np.exp2(rng.uniform(-60, 60, 2_000_000)).astype(F)
    assert (gs.eval_math(9, xs).view(np.uint32) == (F(1.0) / xs).view(np.uint32)).all()
    assert (gs.eval_math(10, xs).view(np.uint32) == np.sqrt(xs).view(np.uint32)).all()
