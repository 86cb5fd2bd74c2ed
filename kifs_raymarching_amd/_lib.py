"""ctypes binding of libkifs_hip.so (include/kifs_hip.h).

There is no fallback: if the shared library is missing or does not load, importing
this module raises.  Build it with `python -c "import __graft_entry__ as g; g.build()"`
or `make -C kifs_raymarching_amd/csrc`.
"""
import ctypes as C
from pathlib import Path

PKG_DIR = Path(__file__).resolve().parent
LIB_PATH = PKG_DIR / "libkifs_hip.so"

KIFS_OK = 0
STATUS_NAMES = {0: "OK", 1: "NO_DEVICE", 2: "DEVICE_INIT", 3: "BAD_SIZE", 4: "UNCONFIGURED",
                5: "RUNTIME", 6: "COMM", 7: "BAD_ARG"}

ENCODE_UNORM, ENCODE_SRGB = 0, 1


class ScreenUniform(C.Structure):  # data.rs:17-23
    _fields_ = [("width", C.c_float), ("height", C.c_float), ("aspect_ratio", C.c_float)]


class CameraUniform(C.Structure):  # data.rs:25-31
    _fields_ = [("origin", C.c_float * 3), ("_padding", C.c_uint32),
                ("matrix", (C.c_float * 4) * 3)]


class OptionsUniform(C.Structure):  # data.rs:33-49
    _fields_ = [("max_iterations", C.c_int32), ("max_distance", C.c_float),
                ("epsilon", C.c_float), ("_padding1", C.c_uint32),
                ("fractal_color", C.c_float * 3), ("_padding2", C.c_uint32),
                ("background_color", C.c_float * 3), ("is_heatmap", C.c_uint32),
                ("fractal_group_id", C.c_uint32), ("primitive_id", C.c_uint32),
                ("power", C.c_float), ("_padding3", C.c_uint32),
                ("constant", C.c_float * 4)]


class ExtensionsC(C.Structure):  # KifsExtensions
    _fields_ = [("soft_shadow", C.c_uint32), ("shadow_steps", C.c_int32), ("shadow_k", C.c_float),
                ("shadow_t0", C.c_float), ("shadow_max_t", C.c_float)]


class GuiDataC(C.Structure):  # KifsGuiData
    _fields_ = [("max_iterations", C.c_uint32), ("max_distance", C.c_float),
                ("epsilon", C.c_float), ("fractal_color", C.c_uint8 * 3),
                ("background_color", C.c_uint8 * 3), ("is_heatmap", C.c_uint8),
                ("_reserved", C.c_uint8), ("fractal_group", C.c_uint32),
                ("primitive_shape", C.c_uint32), ("power", C.c_float),
                ("constant", C.c_float * 4)]


class CameraDataC(C.Structure):  # KifsCameraData
    _fields_ = [("origin_distance", C.c_float), ("min_distance", C.c_float),
                ("phi", C.c_float), ("theta", C.c_float)]


class MultiStatsC(C.Structure):  # KifsMultiStats
    _fields_ = [("steps", C.c_uint64), ("records_received", C.c_uint64), ("tiles_covered", C.c_uint64),
                ("bytes_received", C.c_uint64), ("transport", C.c_int32), ("gather", C.c_int32),
                ("rccl_version", C.c_int32), ("comm_ranks", C.c_int32)]


GATHER_SPARSE, GATHER_DENSE = 0, 1
TRANSPORT_AUTO, TRANSPORT_RCCL, TRANSPORT_COPY = 0, 1, 2
MULTI_FRAMES_UNTOUCHED = 1

assert C.sizeof(ScreenUniform) == 12
assert C.sizeof(CameraUniform) == 64
assert C.sizeof(OptionsUniform) == 80

# every symbol include/kifs_hip.h declares: name -> (restype, argtypes)
_P = C.POINTER
_ctx = C.c_void_p
_f32p = _P(C.c_float)
SIGNATURES = {
    "kifs_create": (_ctx, [C.c_int, _P(C.c_int)]),
    "kifs_destroy": (None, [_ctx]),
    "kifs_set_screen": (C.c_int, [_ctx, _P(ScreenUniform)]),
    "kifs_set_camera": (C.c_int, [_ctx, _P(CameraUniform)]),
    "kifs_set_options": (C.c_int, [_ctx, _P(OptionsUniform)]),
    "kifs_set_iters": (C.c_int, [_ctx, C.c_int, C.c_int, C.c_int]),
    "kifs_set_extensions": (C.c_int, [_ctx, _P(ExtensionsC)]),
    "kifs_multi_set_extensions": (C.c_int, [_ctx, _P(ExtensionsC)]),
    "kifs_render": (C.c_int, [_ctx, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int]),
    "kifs_render_async": (C.c_int, [_ctx, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int,
                                    C.c_int]),
    "kifs_order_after": (C.c_int, [_ctx, C.c_void_p, C.c_void_p]),
    "kifs_multi_order_after": (C.c_int, [_ctx, C.c_void_p]),
    "kifs_render_batch_async": (C.c_int, [_ctx, C.c_void_p, C.c_int, _P(CameraUniform),
                                          _P(C.c_void_p), C.c_size_t, C.c_int, C.c_int, C.c_int]),
    "kifs_band_range": (C.c_int, [C.c_int, C.c_int, C.c_int, _P(C.c_int), _P(C.c_int)]),
    "kifs_shard_stripes": (C.c_int, [C.c_int, C.c_int, _P(C.c_int), C.c_int, _P(C.c_int), C.c_int,
                                     _P(C.c_int), _P(C.c_int)]),
    "kifs_render_shard_async": (C.c_int, [_ctx, C.c_void_p, C.c_int, _P(CameraUniform), _P(C.c_void_p),
                                          C.c_size_t, _P(C.c_int), C.c_int, C.c_int, C.c_int]),
    "kifs_unpack_shard_async": (C.c_int, [_ctx, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_size_t,
                                          C.c_void_p, C.c_size_t, C.c_size_t, _P(C.c_int), C.c_int]),
    "kifs_pack_sparse_async": (C.c_int, [_ctx, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_size_t, _P(C.c_int),
                                         C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "kifs_unpack_sparse_async": (C.c_int, [_ctx, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p,
                                           C.c_size_t, _P(C.c_int), C.c_int]),
    "kifs_fill_shard_async": (C.c_int, [_ctx, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_size_t, _P(C.c_int),
                                        C.c_int, C.c_int]),
    "kifs_erase_sparse_async": (C.c_int, [_ctx, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p,
                                          C.c_size_t, _P(C.c_int), C.c_int, C.c_int]),
    "kifs_multi_create": (_ctx, [_P(C.c_int), C.c_int, _P(C.c_int)]),
    "kifs_multi_destroy": (None, [_ctx]),
    "kifs_multi_set_screen": (C.c_int, [_ctx, _P(ScreenUniform)]),
    "kifs_multi_set_camera": (C.c_int, [_ctx, _P(CameraUniform)]),
    "kifs_multi_set_options": (C.c_int, [_ctx, _P(OptionsUniform)]),
    "kifs_multi_set_iters": (C.c_int, [_ctx, C.c_int, C.c_int, C.c_int]),
    "kifs_multi_render": (C.c_int, [_ctx, C.c_void_p, C.c_size_t, C.c_int]),
    "kifs_multi_set_weights": (C.c_int, [_ctx, _P(C.c_int)]),
    "kifs_multi_shard": (C.c_int, [_ctx, C.c_int, _P(C.c_int), _P(C.c_int), _P(C.c_int)]),
    "kifs_multi_shard_ms": (C.c_double, [_ctx, C.c_int]),
    "kifs_multi_set_gather": (C.c_int, [_ctx, C.c_int, C.c_int]),
    "kifs_multi_render_batch_async": (C.c_int, [_ctx, C.c_int, _P(CameraUniform), C.c_void_p, C.c_size_t, C.c_size_t,
                                                C.c_int, C.c_int, _P(C.c_uint64)]),
    "kifs_multi_wait": (C.c_int, [_ctx, C.c_uint64]),
    "kifs_multi_wait_all": (C.c_int, [_ctx]),
    "kifs_multi_stream_wait": (C.c_int, [_ctx, C.c_uint64, C.c_void_p]),
    "kifs_multi_render_batch": (C.c_int, [_ctx, C.c_int, _P(CameraUniform), C.c_void_p, C.c_size_t, C.c_size_t, C.c_int]),
    "kifs_multi_stats": (C.c_int, [_ctx, _P(MultiStatsC), C.c_int]),
    "kifs_multi_comm_selftest": (C.c_int, [_ctx, C.c_size_t]),
    "kifs_last_kernel_ms": (C.c_double, [_ctx]),
    "kifs_synchronize": (C.c_int, [_ctx]),
    "kifs_set_profiling": (C.c_int, [_ctx, C.c_int]),
    "kifs_set_frames_in_flight": (C.c_int, [_ctx, C.c_int]),
    "kifs_debug_last_round_steps": (C.c_int, [_ctx]),
    "kifs_debug_last_group_tiles": (C.c_int, [_ctx]),
    "kifs_debug_last_kernel": (C.c_int, [_ctx]),
    "kifs_debug_last_bunny_form": (C.c_int, [_ctx]),
    "kifs_profile_read": (C.c_int, [_ctx, _P(C.c_int), _P(C.c_double), _P(C.c_double), _P(C.c_double)]),
    "kifs_strerror": (C.c_char_p, [C.c_int]),
    "kifs_abi_version": (C.c_int, []),
    "kifs_eval_points": (C.c_int, [_ctx, _f32p, C.c_int, _f32p, _f32p]),
    "kifs_debug_counters": (C.c_int, [_ctx, C.c_int, _P(C.c_uint64)]),
    "kifs_debug_get_tile_order": (C.c_int, [_ctx, _P(C.c_uint32), C.c_size_t, _P(C.c_size_t)]),
    "kifs_debug_set_tile_order": (C.c_int, [_ctx, _P(C.c_uint32), C.c_size_t]),
    "kifs_debug_wave_records": (C.c_int, [_ctx, _P(C.c_uint64), C.c_size_t, _P(C.c_size_t)]),
    "kifs_eval_math": (C.c_int, [_ctx, C.c_int, _f32p, C.c_float, _f32p, C.c_int]),
    "kifs_host_gui_default": (None, [_P(GuiDataC)]),
    "kifs_host_camera_default": (None, [_P(CameraDataC)]),
    "kifs_host_screen": (C.c_int, [C.c_uint32, C.c_uint32, _P(ScreenUniform)]),
    "kifs_host_camera": (C.c_int, [_P(CameraDataC), _P(CameraUniform)]),
    "kifs_host_options": (C.c_int, [_P(GuiDataC), _P(OptionsUniform)]),
    "kifs_host_rotate": (C.c_int, [_P(CameraDataC), C.c_float, C.c_float]),
    "kifs_host_zoom": (C.c_int, [_P(CameraDataC), C.c_float]),
    "kifs_host_mouse_motion": (C.c_int, [_P(CameraDataC), C.c_double, C.c_double]),
    "kifs_host_radians_from_degrees": (C.c_float, [C.c_float]),
    "kifs_host_camera_matrix": (None, [_P(CameraDataC), _f32p]),
    "kifs_host_rotation_matrix": (None, [C.c_int, C.c_float, _f32p]),
    "kifs_host_mat3_mul": (None, [_f32p, _f32p, _f32p]),
    "kifs_host_mat3_vec": (None, [_f32p, _f32p, _f32p]),
}


class KifsError(RuntimeError):
    def __init__(self, status, what=""):
        self.status = status
        name = STATUS_NAMES.get(status, str(status))
        super().__init__(f"kifs: {what + ': ' if what else ''}{name} ({status})")


def _share_hip_runtime_with_torch():
    """One process, one HIP runtime.  PyTorch ships its own libamdhip64.so.7; libkifs_hip.so
    names the same soname and would otherwise pull in the system copy, after which torch
    (whichever is imported second) cannot see the GPU.  If PyTorch is installed, load its copy
    first -- without importing torch -- so both resolve to it regardless of import order."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    cand = Path(list(spec.submodule_search_locations)[0]) / "lib" / "libamdhip64.so"
    if cand.exists():
        try:
            C.CDLL(str(cand), mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def _load():
    _share_hip_runtime_with_torch()
    import os
    variant = os.environ.get("KIFS_LIB_VARIANT")
    if variant and os.environ.get("KIFS_TUNING") == "1":
        # a sweep's prebuilt variant (tools/sweep_*_variants.sh), loaded from where it lies: the library in the tree
        # and its stamp stay what the sources say.  Said on stderr, so that no measurement can pass for the tree's.
        import sys
        print(f"kifs: loading the library VARIANT {variant} (KIFS_LIB_VARIANT under KIFS_TUNING=1), not the tree's",
              file=sys.stderr)
        return _bind(C.CDLL(variant))
    if not LIB_PATH.exists():
        raise ImportError(
            f"{LIB_PATH} is missing: the HIP library has not been built "
            "(run __graft_entry__.build() or make -C kifs_raymarching_amd/csrc). "
            "There is no CPU fallback.")
    # a library older than its sources (edited, not rebuilt) is refused, not used: measurements and parity
    # claims must belong to the code in the tree
    import importlib.util
    spec = importlib.util.spec_from_file_location("_kifs_build_check", PKG_DIR / "build.py")
    kb = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(kb)
    if kb.STAMP.exists() and not kb.is_current():
        src = kb.recorded()[0]
        why = ("was built from other sources than the ones in csrc/" if src != kb.source_hash()
               else "is not the file the build produced (replaced after the build?)")
        raise ImportError(f"{LIB_PATH} {why} (hash mismatch): rebuild it "
                          "(python -c 'import __graft_entry__ as g; g.build()')")
    return _bind(C.CDLL(str(LIB_PATH)))


def kernel_hash_of_loaded_library() -> str:
    """The kernel hash (build.kernel_hash()) recorded when the loaded library was built; "" for a variant or an
    unstamped library.  profiles/pmc_traffic.json and profiles/lone_frame_floor.json entries carry the hash of the
    kernels they were measured on or counted from; bench.py drops the ones that are not this library's."""
    import importlib.util
    import os
    if os.environ.get("KIFS_LIB_VARIANT") and os.environ.get("KIFS_TUNING") == "1":
        return ""
    spec = importlib.util.spec_from_file_location("_kifs_build_check", PKG_DIR / "build.py")
    kb = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(kb)
    return kb.recorded()[2]


def _bind(lib):
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the library lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    if lib.kifs_abi_version() != 4:
        raise ImportError("libkifs_hip.so: ABI version mismatch")
    return lib


lib = _load()


def check(status, what=""):
    if status != KIFS_OK:
        raise KifsError(status, what)
