"""ctypes binding of the CPU ORACLE (test infrastructure only).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package; the product (kifs_raymarching_amd/) never does.  See
oracle/kifs_oracle.h for the contract and the "parity unpinned" statement.
"""
import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent


class Screen(C.Structure):
    _fields_ = [("width", C.c_float), ("height", C.c_float), ("aspect_ratio", C.c_float)]


class Camera(C.Structure):
    _fields_ = [("origin", C.c_float * 3), ("_padding", C.c_uint32),
                ("matrix", (C.c_float * 4) * 3)]


class Options(C.Structure):
    _fields_ = [("max_iterations", C.c_int32), ("max_distance", C.c_float),
                ("epsilon", C.c_float), ("_padding1", C.c_uint32),
                ("fractal_color", C.c_float * 3), ("_padding2", C.c_uint32),
                ("background_color", C.c_float * 3), ("is_heatmap", C.c_uint32),
                ("fractal_group_id", C.c_uint32), ("primitive_id", C.c_uint32),
                ("power", C.c_float), ("_padding3", C.c_uint32),
                ("constant", C.c_float * 4)]


class Iters(C.Structure):
    _fields_ = [("sdf_iters", C.c_int32), ("normal_iters", C.c_int32),
                ("fold_iters", C.c_int32)]


class Ext(C.Structure):
    _fields_ = [("soft_shadow", C.c_uint32), ("shadow_steps", C.c_int32), ("shadow_k", C.c_float),
                ("shadow_t0", C.c_float), ("shadow_max_t", C.c_float)]


class Stats(C.Structure):
    _fields_ = [("pixels", C.c_uint64), ("march_steps", C.c_uint64),
                ("inner_iters", C.c_uint64), ("hits", C.c_uint64),
                ("sdf_calls", C.c_uint64), ("max_steps", C.c_uint32)]


assert C.sizeof(Screen) == 12 and C.sizeof(Camera) == 64 and C.sizeof(Options) == 80


def build(force=False):
    """Compile the oracle (gcc, seconds).  Called by __graft_entry__.build()."""
    if force or not (_HERE / "libkifs_oracle.so").exists() or \
            not (_HERE / "libkifs_oracle_fma.so").exists():
        subprocess.run(["make", "-C", str(_HERE), "-s"] + (["-B"] if force else []), check=True)


def _cpu_has_fma():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("flags"):
                    flags = line.split()
                    return "fma" in flags and "sse4_1" in flags
    except OSError:
        pass
    return False


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    name = "libkifs_oracle_fma.so" if _cpu_has_fma() and not os.environ.get("KIFS_ORACLE_GENERIC") \
        else "libkifs_oracle.so"
    path = _HERE / name
    if not path.exists():
        build()
    L = C.CDLL(str(path))
    P = C.POINTER
    f32, i32, u32, u8p = C.c_float, C.c_int, C.c_uint32, P(C.c_uint8)
    sigs = {
        "kor_render": (C.c_int, [P(Screen), P(Camera), P(Options), P(Iters), i32, i32, i32, u8p,
                                 C.c_size_t, i32]),
        "kor_render_ext": (C.c_int, [P(Screen), P(Camera), P(Options), P(Iters), P(Ext), i32, i32, i32,
                                     u8p, C.c_size_t, i32]),
        "kor_render_stats": (C.c_int, [P(Screen), P(Camera), P(Options), P(Iters), i32, i32, i32,
                                       u8p, C.c_size_t, P(Stats), P(C.c_uint16)]),
        "kor_shade_pixel": (C.c_int, [P(Screen), P(Camera), P(Options), P(Iters), i32, i32,
                                      P(f32)]),
        "kor_scene_sdf": (f32, [P(Options), P(Iters), P(f32)]),
        "kor_get_normal": (None, [P(Options), P(Iters), P(f32), P(f32)]),
        "kor_ray_direction": (None, [P(Screen), P(Camera), i32, i32, P(f32)]),
        "kor_quat_sq": (None, [P(f32), P(f32)]),
        "kor_quat_mul": (None, [P(f32), P(f32), P(f32)]),
        "kor_tetrahedral_fold": (None, [P(f32), P(f32)]),
        "kor_logf": (f32, [f32]), "kor_sinf": (f32, [f32]), "kor_cosf": (f32, [f32]),
        "kor_acosf": (f32, [f32]), "kor_exp2f": (f32, [f32]), "kor_log2f": (f32, [f32]),
        "kor_powf": (f32, [f32, f32]),
        "kor_srgb_thresholds": (None, [P(f32)]),
        "kor_encode_channel": (C.c_uint8, [f32, i32]),
        "kor_screen_uniform": (None, [u32, u32, P(Screen)]),
        "kor_camera_matrix": (None, [f32, f32, P(f32)]),
        "kor_camera_uniform": (None, [f32, f32, f32, P(Camera)]),
        "kor_linear_from_srgb_u8": (f32, [C.c_uint8]),
        "kor_options_from_gui": (None, [u32, f32, f32, u8p, u8p, i32, u32, u32, f32, P(f32),
                                        P(Options)]),
        "kor_rotation_matrix": (None, [i32, f32, P(f32)]),
        "kor_mat3_mul": (None, [P(f32), P(f32), P(f32)]),
        "kor_mat3_vec": (None, [P(f32), P(f32), P(f32)]),
        "kor_radians_from_degrees": (f32, [f32]),
        "kor_radians_standardize": (f32, [f32]),
        "kor_rotate_camera": (None, [P(f32), P(f32), f32, f32]),
        "kor_zoom_camera": (f32, [f32, f32, f32]),
    }
    for name_, (res, args) in sigs.items():
        fn = getattr(L, name_)
        fn.restype = res
        fn.argtypes = args
    L._variant = name
    _lib = L
    return L


# ---------------------------------------------------------------- helpers ----

def _fv(vals):
    arr = (C.c_float * len(vals))(*[float(v) for v in vals])
    return arr


def from_bytes(cls, raw):
    """Uniform struct from its byte image (e.g. produced by the product host code)."""
    raw = bytes(raw)
    assert len(raw) == C.sizeof(cls), (len(raw), C.sizeof(cls))
    return cls.from_buffer_copy(raw)


def screen_uniform(width, height):
    s = Screen()
    lib().kor_screen_uniform(width, height, C.byref(s))
    return s


def camera_uniform(origin_distance=5.0, phi=0.0, theta=0.0):
    c = Camera()
    lib().kor_camera_uniform(origin_distance, phi, theta, C.byref(c))
    return c


def options_from_gui(max_iterations=256, max_distance=1000.0, epsilon=1e-4,
                     fractal_color=(200, 200, 200), background_color=(0, 0, 0),
                     is_heatmap=False, fractal_group=0, primitive_shape=0, power=2.0,
                     constant=(-0.1, 0.6, 0.9, -0.3)):
    """Defaults are GuiData::default() (data.rs:145-160)."""
    o = Options()
    fc = (C.c_uint8 * 3)(*fractal_color)
    bc = (C.c_uint8 * 3)(*background_color)
    lib().kor_options_from_gui(max_iterations, max_distance, epsilon, fc, bc, int(is_heatmap),
                               fractal_group, primitive_shape, power, _fv(constant), C.byref(o))
    return o


def iters(sdf_iters=100, normal_iters=10, fold_iters=10):
    return Iters(sdf_iters, normal_iters, fold_iters)


def render(screen, camera, options, it=None, encode=1, y0=0, y1=None, nthreads=0, ext=None):
    """RGBA8 rows [y0,y1) as a (rows, W, 4) uint8 array.  `ext`: Ext (soft-shadow extension)."""
    w, h = int(screen.width), int(screen.height)
    y1 = h if y1 is None else y1
    it = it or iters()
    out = np.zeros((max(y1 - y0, 0), w, 4), dtype=np.uint8)
    rc = lib().kor_render_ext(C.byref(screen), C.byref(camera), C.byref(options), C.byref(it),
                              C.byref(ext) if ext is not None else None, encode,
                              y0, y1, out.ctypes.data_as(C.POINTER(C.c_uint8)), w * 4, nthreads)
    if rc != 0:
        raise ValueError("kor_render: bad arguments")
    return out


def render_ray_costs(screen, camera, options, it=None, y0=0, y1=None):
    """Per pixel of rows [y0, y1): (scene_SDF calls of the march, calls past the bounding-sphere early-out, inner
    iterations) as three (rows, W) arrays -- kor_render_ray_costs; what shading adds after the march is not counted."""
    w, h = int(screen.width), int(screen.height)
    y1 = h if y1 is None else y1
    it = it or iters()
    calls = np.zeros((y1 - y0, w), dtype=np.uint16)
    inside = np.zeros((y1 - y0, w), dtype=np.uint16)
    inner = np.zeros((y1 - y0, w), dtype=np.uint32)
    fn = lib().kor_render_ray_costs
    fn.restype = C.c_int
    rc = fn(C.byref(screen), C.byref(camera), C.byref(options), C.byref(it), C.c_int(y0), C.c_int(y1),
            calls.ctypes.data_as(C.POINTER(C.c_uint16)), inside.ctypes.data_as(C.POINTER(C.c_uint16)),
            inner.ctypes.data_as(C.POINTER(C.c_uint32)))
    if rc != 0:
        raise ValueError("kor_render_ray_costs: bad arguments")
    return calls, inside, inner


def render_stats(screen, camera, options, it=None, encode=1, y0=0, y1=None):
    w, h = int(screen.width), int(screen.height)
    y1 = h if y1 is None else y1
    it = it or iters()
    out = np.zeros((y1 - y0, w, 4), dtype=np.uint8)
    steps = np.zeros((y1 - y0, w), dtype=np.uint16)
    st = Stats()
    rc = lib().kor_render_stats(C.byref(screen), C.byref(camera), C.byref(options), C.byref(it),
                                encode, y0, y1, out.ctypes.data_as(C.POINTER(C.c_uint8)), w * 4,
                                C.byref(st), steps.ctypes.data_as(C.POINTER(C.c_uint16)))
    if rc != 0:
        raise ValueError("kor_render_stats: bad arguments")
    return out, steps, st


def shade_pixel(screen, camera, options, it, x, y):
    rgba = (C.c_float * 4)()
    i = lib().kor_shade_pixel(C.byref(screen), C.byref(camera), C.byref(options), C.byref(it),
                              x, y, rgba)
    return i, np.array(rgba[:], dtype=np.float32)


def scene_sdf(options, it, p):
    return float(lib().kor_scene_sdf(C.byref(options), C.byref(it), _fv(p)))


def get_normal(options, it, p):
    n = (C.c_float * 3)()
    lib().kor_get_normal(C.byref(options), C.byref(it), _fv(p), n)
    return np.array(n[:], dtype=np.float32)


def ray_direction(screen, camera, x, y):
    d = (C.c_float * 3)()
    lib().kor_ray_direction(C.byref(screen), C.byref(camera), x, y, d)
    return np.array(d[:], dtype=np.float32)


def srgb_thresholds():
    t = np.zeros(256, dtype=np.float32)
    lib().kor_srgb_thresholds(t.ctypes.data_as(C.POINTER(C.c_float)))
    return t
