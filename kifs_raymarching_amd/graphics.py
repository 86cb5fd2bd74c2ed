"""Host-side mirror of the reference's scene model and `GraphicState`, over the C ABI.

Same names, argument meaning and error behaviour as the reference so that tests read
like the reference's own (paths under src/ of LesbianLemon/kifs-raymarching):

  FractalGroup, PrimitiveShape          data/scene.rs:4-45
  ScreenData, CameraData, GuiData       data.rs:51-160   (+ .into_buffer_data())
  GraphicState                          render/graphics.rs:25-326
      update_screen_data / zoom_camera / rotate_camera / update_options / render

All arithmetic happens in the library (C++ host model, HIP kernels); this file only
moves bytes.  PyTorch, when used, is plumbing: tensors give device memory and streams.
"""
import ctypes as C
from dataclasses import dataclass, field
from enum import IntEnum

import numpy as np

from . import _lib

MAX_BATCH = 512  # KIFS_MAX_BATCH of include/kifs_hip.h
SPARSE_RECORD_BYTES = 1040  # KIFS_SPARSE_RECORD_BYTES
STRIPE_ROWS = 8  # KIFS_STRIPE_ROWS
from ._lib import (CameraDataC, CameraUniform, ExtensionsC, GuiDataC, KifsError, OptionsUniform,
                   ScreenUniform, check, lib)

ENCODE_UNORM = _lib.ENCODE_UNORM
ENCODE_SRGB = _lib.ENCODE_SRGB


class FractalGroup(IntEnum):  # data/scene.rs:4-11
    KaleidoscopicIFS = 0
    JuliaSet = 1
    GeneralizedJuliaSet = 2

    @classmethod
    def from_id(cls, i):
        try:
            return cls(i)
        except ValueError:
            return None


class PrimitiveShape(IntEnum):  # data/scene.rs:35-45
    Sphere = 0
    Cylinder = 1
    Box = 2
    Torus = 3
    SierpinskiTetrahedron = 4
    Bunny = 5

    @classmethod
    def from_id(cls, i):
        try:
            return cls(i)
        except ValueError:
            return None


@dataclass
class ScreenData:  # data.rs:51-55
    width: int = 0
    height: int = 0

    def into_buffer_data(self) -> ScreenUniform:  # data.rs:66-81
        u = ScreenUniform()
        check(lib.kifs_host_screen(self.width, self.height, C.byref(u)), "ScreenData")
        return u


@dataclass
class CameraData:  # data.rs:83-113
    origin_distance: float = 5.0
    min_distance: float = 2.0
    phi: float = 0.0    # angles.0
    theta: float = 0.0  # angles.1

    def _c(self) -> CameraDataC:
        return CameraDataC(self.origin_distance, self.min_distance, self.phi, self.theta)

    def _take(self, c: CameraDataC):
        self.origin_distance, self.min_distance = c.origin_distance, c.min_distance
        self.phi, self.theta = c.phi, c.theta

    def camera_matrix(self) -> np.ndarray:
        """3x3, element [r, c]; columns are the camera basis (data.rs:91-98)."""
        m = (C.c_float * 9)()
        c = self._c()
        lib.kifs_host_camera_matrix(C.byref(c), m)
        return np.array(m[:], dtype=np.float32).reshape(3, 3).T.copy()

    def into_buffer_data(self) -> CameraUniform:  # data.rs:115-129
        u = CameraUniform()
        c = self._c()
        check(lib.kifs_host_camera(C.byref(c), C.byref(u)), "CameraData")
        return u


@dataclass
class GuiData:  # data.rs:131-160 (defaults = GuiData::default)
    max_iterations: int = 256
    max_distance: float = 1000.0
    epsilon: float = 0.0001
    fractal_color: tuple = (200, 200, 200)
    background_color: tuple = (0, 0, 0)
    is_heatmap: bool = False
    fractal_group: FractalGroup = FractalGroup.KaleidoscopicIFS
    primitive_shape: PrimitiveShape = PrimitiveShape.Sphere
    power: float = 2.0
    constant: tuple = (-0.1, 0.6, 0.9, -0.3)

    def into_buffer_data(self) -> OptionsUniform:
        """OptionsData::from(GuiData).into_buffer_data() (data.rs:176-220)."""
        g = GuiDataC()
        g.max_iterations = int(self.max_iterations)
        g.max_distance = self.max_distance
        g.epsilon = self.epsilon
        g.fractal_color = (C.c_uint8 * 3)(*self.fractal_color)
        g.background_color = (C.c_uint8 * 3)(*self.background_color)
        g.is_heatmap = 1 if self.is_heatmap else 0
        g.fractal_group = int(self.fractal_group)
        g.primitive_shape = int(self.primitive_shape)
        g.power = self.power
        g.constant = (C.c_float * 4)(*self.constant)
        u = OptionsUniform()
        check(lib.kifs_host_options(C.byref(g), C.byref(u)), "GuiData")
        return u


def uniform_bytes(u) -> bytes:
    """`bytemuck::bytes_of` of a uniform struct."""
    return bytes(memoryview(u).cast("B"))


class DevicePointers:
    """The destinations of a batch, prepared once: a C array of device pointers (and the tensors, kept alive).
    render_batch_async / render_shard_async take it in place of the list of tensors -- a step of a few hundred
    views otherwise spends as long collecting pointers on the host as the GPU spends rendering."""

    def __init__(self, outs):
        self.tensors = list(outs)
        self.array = (C.c_void_p * len(self.tensors))(*[_device_pointer(o) for o in self.tensors])

    def __len__(self):
        return len(self.tensors)

    def __getitem__(self, i):
        return self.tensors[i]


def camera_array(cameras):
    """A C array of CameraUniform images from CameraData objects / uniform images (prepared once per
    sequence of poses; the render calls take it in place of the list)."""
    if isinstance(cameras, C.Array):
        return cameras
    return (CameraUniform * len(cameras))(*[c.into_buffer_data() if hasattr(c, "into_buffer_data") else c
                                            for c in cameras])


def _device_pointer(obj):
    if hasattr(obj, "data_ptr"):  # torch.Tensor
        return int(obj.data_ptr())
    return int(obj)


def _producer_stream(dest):
    """hipStream_t of torch's CURRENT stream on the device of `dest` (0 = the legacy default stream) when `dest` is
    a torch CUDA tensor -- the stream whatever produced the tensor (torch.zeros' fill kernel, a consumer's last read)
    was enqueued on, as far as this wrapper can know; None for raw pointers and host arrays."""
    if not getattr(dest, "is_cuda", False):
        return None
    import torch
    return int(torch.cuda.current_stream(dest.device).cuda_stream)


class GraphicState:
    """render/graphics.rs:25-37.  Owns one library context bound to one HIP device."""

    def __init__(self, device: int = 0, screen_data: ScreenData = None,
                 camera_data: CameraData = None, gui_data: GuiData = None):
        st = C.c_int(0)
        self._ctx = lib.kifs_create(device, C.byref(st))
        if not self._ctx:
            raise KifsError(st.value, f"kifs_create(device={device})")
        self.device = device
        self.screen_data = ScreenData()
        self.camera_data = camera_data or CameraData()   # graphics.rs:194 CameraData::default
        self.gui_data = gui_data or GuiData()            # graphics.rs:200 GuiData::default
        self.camera_rotatable = False                    # graphics.rs:229
        self.iters = (100, 10, 10)
        self._upload_camera()
        self.update_options(self.gui_data)
        if screen_data is not None:
            self.update_screen_data(screen_data)

    # -- lifetime -------------------------------------------------------------------
    def close(self):
        if getattr(self, "_ctx", None):
            lib.kifs_destroy(self._ctx)
            self._ctx = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- uniform updates (graphics.rs:262-308) ---------------------------------------
    def update_screen_data(self, new_screen_data: ScreenData):
        u = new_screen_data.into_buffer_data()
        check(lib.kifs_set_screen(self._ctx, C.byref(u)), "update_screen_data")
        self.screen_data = new_screen_data

    def _upload_camera(self):
        u = self.camera_data.into_buffer_data()
        check(lib.kifs_set_camera(self._ctx, C.byref(u)), "camera")

    def zoom_camera(self, distance: float):  # graphics.rs:268-278
        c = self.camera_data._c()
        check(lib.kifs_host_zoom(C.byref(c), distance))
        self.camera_data._take(c)
        self._upload_camera()

    def rotate_camera(self, delta_phi: float, delta_theta: float):  # graphics.rs:280-302
        if not self.camera_rotatable:
            return
        c = self.camera_data._c()
        check(lib.kifs_host_rotate(C.byref(c), delta_phi, delta_theta))
        self.camera_data._take(c)
        self._upload_camera()

    def mouse_motion(self, dx: float, dy: float):  # render.rs:255-270
        if not self.camera_rotatable:
            return
        c = self.camera_data._c()
        check(lib.kifs_host_mouse_motion(C.byref(c), dx, dy))
        self.camera_data._take(c)
        self._upload_camera()

    def enable_camera_rotation(self):
        self.camera_rotatable = True

    def disable_camera_rotation(self):
        self.camera_rotatable = False

    def is_camera_rotatable(self):
        return self.camera_rotatable

    def set_camera(self, camera_data: CameraData):
        self.camera_data = camera_data
        self._upload_camera()

    def update_options(self, new_options):
        """Accepts GuiData (converted like render.rs:320-321) or a packed OptionsUniform."""
        if isinstance(new_options, GuiData):
            self.gui_data = new_options
            u = new_options.into_buffer_data()
        else:
            u = new_options
        check(lib.kifs_set_options(self._ctx, C.byref(u)), "update_options")
        self._options_uniform = u

    def set_raw_uniforms(self, screen: ScreenUniform = None, camera: CameraUniform = None,
                         options: OptionsUniform = None):
        """Upload pre-packed byte images (what a Rust host would pass)."""
        if screen is not None:
            check(lib.kifs_set_screen(self._ctx, C.byref(screen)), "set_screen")
            self.screen_data = ScreenData(int(screen.width), int(screen.height))
        if camera is not None:
            check(lib.kifs_set_camera(self._ctx, C.byref(camera)), "set_camera")
        if options is not None:
            check(lib.kifs_set_options(self._ctx, C.byref(options)), "set_options")

    def set_iters(self, sdf_iters=100, normal_iters=10, fold_iters=10):
        check(lib.kifs_set_iters(self._ctx, sdf_iters, normal_iters, fold_iters), "set_iters")
        self.iters = (sdf_iters, normal_iters, fold_iters)

    def set_extensions(self, soft_shadow=False, shadow_steps=64, shadow_k=8.0, shadow_t0=0.02,
                       shadow_max_t=10.0):
        """Opt-in soft shadows (an extension; absent from the reference).  Off = reference shading."""
        e = ExtensionsC(1 if soft_shadow else 0, shadow_steps, shadow_k, shadow_t0, shadow_max_t)
        check(lib.kifs_set_extensions(self._ctx, C.byref(e)), "set_extensions")

    # -- render (graphics.rs:310-325) -----------------------------------------------------
    def _order_after_producer(self, dest, launch_stream):
        """The wrapper's stream rule for torch destinations: the library launches on non-blocking streams, which
        nothing orders after torch's current stream -- where the destination's fill, or its previous reader, sits.
        Every render entry point below therefore makes its launch stream (None = the context's) wait for an event
        recorded on torch's current stream of the destination's device (kifs_order_after: two HIP calls, nothing
        blocks the host; the same stream on both sides is a no-op).  A C caller does the same with kifs_order_after
        or orders its streams itself (include/kifs_hip.h); the reference has one queue (util/uniform.rs:23-35)."""
        producer = _producer_stream(dest)
        if producer is None or (launch_stream and producer == launch_stream):
            return
        check(lib.kifs_order_after(self._ctx, launch_stream, producer or None), "order_after")

    def render(self, out=None, y0: int = 0, y1: int = None, encode: int = ENCODE_SRGB,
               pitch_bytes: int = None):
        """Synchronous frame.  `out` None -> returns a (rows, W, 4) uint8 ndarray; an
        ndarray -> filled in place; a torch CUDA tensor (uint8, rows*W*4) -> written on
        the device."""
        w, h = self.screen_data.width, self.screen_data.height
        y1 = h if y1 is None else y1
        rows = max(y1 - y0, 0)
        if out is None:
            out = np.empty((rows, w, 4), dtype=np.uint8)
        pitch = pitch_bytes if pitch_bytes is not None else w * 4
        if isinstance(out, np.ndarray):
            _check_host_destination(out, rows, w, pitch)
            ptr = out.ctypes.data
        else:
            ptr = _device_pointer(out)
            self._order_after_producer(out, None)
        check(lib.kifs_render(self._ctx, ptr, pitch, y0, y1, encode), "render")
        return out

    def render_async(self, out, stream=None, y0: int = 0, y1: int = None,
                     encode: int = ENCODE_SRGB, pitch_bytes: int = None):
        """Enqueue on `stream` (int hipStream_t / torch stream / None = context stream)."""
        w, h = self.screen_data.width, self.screen_data.height
        y1 = h if y1 is None else y1
        pitch = pitch_bytes if pitch_bytes is not None else w * 4
        if stream is not None and hasattr(stream, "cuda_stream"):
            stream = stream.cuda_stream
            if not stream:
                # the C ABI reads a null hipStream_t as "the context's own stream"; the legacy
                # default stream has handle 0 and would silently end up there
                raise ValueError("render_async: pass a non-default torch.cuda.Stream "
                                 "(the null stream handle selects the context's stream)")
        self._order_after_producer(out, stream)
        check(lib.kifs_render_async(self._ctx, stream, _device_pointer(out), pitch, y0, y1,
                                    encode), "render_async")

    def render_batch_async(self, outs, cameras, stream=None, y0: int = 0, y1: int = None,
                           encode: int = ENCODE_SRGB, pitch_bytes: int = None):
        """One launch for len(outs) <= MAX_BATCH frames: frame i uses cameras[i] (CameraData or a
        CameraUniform image) and goes to the device tensor outs[i]; screen, options, iteration
        counts and extensions are the context's.  Enqueued on `stream` like render_async."""
        if len(outs) != len(cameras) or not (1 <= len(outs) <= MAX_BATCH):
            raise ValueError(f"render_batch_async: 1..{MAX_BATCH} frames, one camera each")
        w, h = self.screen_data.width, self.screen_data.height
        y1 = h if y1 is None else y1
        pitch = pitch_bytes if pitch_bytes is not None else w * 4
        if stream is not None and hasattr(stream, "cuda_stream"):
            stream = stream.cuda_stream
            if not stream:
                raise ValueError("render_batch_async: pass a non-default torch.cuda.Stream")
        n = len(outs)
        cams = camera_array(cameras)
        ptrs = outs.array if isinstance(outs, DevicePointers) else (C.c_void_p * n)(*[_device_pointer(o) for o in outs])
        self._order_after_producer(outs[0], stream)
        check(lib.kifs_render_batch_async(self._ctx, stream, n, cams, ptrs, pitch, y0, y1, encode),
              "render_batch_async")

    def render_shard_async(self, outs, cameras, stripes, in_place: bool = False, stream=None,
                           encode: int = ENCODE_SRGB, pitch_bytes: int = None):
        """render_batch_async for a row shard (kifs_render_shard_async): `stripes` is the list of
        8-row stripe indices (shard_stripes); outs[i] is a packed (rows, W, 4) device tensor, or
        with in_place a whole (H, W, 4) frame whose shard rows are written where they belong."""
        if len(outs) != len(cameras) or not (1 <= len(outs) <= MAX_BATCH):
            raise ValueError(f"render_shard_async: 1..{MAX_BATCH} frames, one camera each")
        w = self.screen_data.width
        pitch = pitch_bytes if pitch_bytes is not None else w * 4
        if stream is not None and hasattr(stream, "cuda_stream"):
            stream = stream.cuda_stream
            if not stream:
                raise ValueError("render_shard_async: pass a non-default torch.cuda.Stream")
        n = len(outs)
        cams = camera_array(cameras)
        # The library knows nothing about the size of the destinations: a packed shard needs rows * pitch bytes, a
        # shard rendered in place a whole frame.  (A DevicePointers remembers what it has been checked against: a
        # step of a few hundred views must not pay for the check every time.)
        h = self.screen_data.height
        if any(s < 0 or s * STRIPE_ROWS >= h for s in stripes):
            raise ValueError("render_shard_async: stripe outside the frame")
        need_rows = h if in_place else sum(min(STRIPE_ROWS, h - s * STRIPE_ROWS) for s in stripes)
        need = ((need_rows - 1) * pitch + w * 4) if need_rows else 0
        checked = getattr(outs, "checked_bytes", None)
        if checked is None or checked < need:
            for o in (outs.tensors if isinstance(outs, DevicePointers) else outs):
                if hasattr(o, "element_size"):
                    if not o.is_contiguous() or o.element_size() != 1 or o.numel() < need:
                        raise ValueError(f"render_shard_async: every destination must be a contiguous uint8 tensor of at "
                                         f"least {need} bytes ({need_rows} rows of pitch {pitch}"
                                         f"{', a whole frame: in_place' if in_place else ''}); got {tuple(o.shape)}")
            if isinstance(outs, DevicePointers):
                outs.checked_bytes = need
        ptrs = outs.array if isinstance(outs, DevicePointers) else (C.c_void_p * n)(*[_device_pointer(o) for o in outs])
        st = _stripe_array(stripes)
        self._order_after_producer(outs[0], stream)
        check(lib.kifs_render_shard_async(self._ctx, stream, n, cams, ptrs, pitch, st, len(st),
                                          1 if in_place else 0, encode), "render_shard_async")

    def unpack_shard_async(self, frames, shards, stripes, stream=None):
        """Root side of the gather (kifs_unpack_shard_async): `shards` (count, rows, W, 4) packed,
        `frames` (count, H, W, 4); stripe k of every shard goes to frame rows 8 * stripes[k]...
        Both are contiguous device tensors."""
        w, h = self.screen_data.width, self.screen_data.height
        if frames.dim() != 4 or shards.dim() != 4 or not (frames.is_contiguous() and shards.is_contiguous()):
            raise ValueError("unpack_shard_async: frames (count, H, W, 4) and shards (count, rows, W, 4), contiguous")
        count = int(frames.shape[0])
        rows = int(shards.shape[1])
        want_rows = sum(min(STRIPE_ROWS, h - s * STRIPE_ROWS) for s in stripes)
        if (tuple(frames.shape[1:]) != (h, w, 4) or tuple(shards.shape) != (count, rows, w, 4) or rows != want_rows
                or frames.element_size() != 1 or shards.element_size() != 1):
            raise ValueError(f"unpack_shard_async: shapes {tuple(frames.shape)} / {tuple(shards.shape)} do not match "
                             f"{count} frames of {h}x{w} and a shard of {want_rows} rows")
        if stream is not None and hasattr(stream, "cuda_stream"):
            stream = stream.cuda_stream
            if not stream:
                raise ValueError("unpack_shard_async: pass a non-default torch.cuda.Stream")
        st = _stripe_array(stripes)
        self._order_after_producer(frames, stream)
        check(lib.kifs_unpack_shard_async(self._ctx, stream, count, _device_pointer(frames), w * 4, h * w * 4,
                                          _device_pointer(shards), w * 4, rows * w * 4, st, len(st)),
              "unpack_shard_async")

    # ---- sparse shards (kifs_pack_sparse_async / kifs_unpack_sparse_async / kifs_fill_shard_async)
    def _stream_handle(self, stream, what):
        if stream is not None and hasattr(stream, "cuda_stream"):
            stream = stream.cuda_stream
            if not stream:
                raise ValueError(f"{what}: pass a non-default torch.cuda.Stream")
        return stream

    def sparse_capacity(self, count: int, stripes) -> int:
        """Records a payload must have room for: every tile of `count` shards of these stripes."""
        return count * len(stripes) * ((self.screen_data.width + 31) // 32)

    def pack_sparse_async(self, shards, stripes, records, n_records, host_n_records=None, stream=None,
                          encode: int = ENCODE_SRGB):
        """Peer side: `shards` (count, rows, W, 4) packed -> `records` (capacity, 1040) uint8 device tensor,
        their number in `n_records` (int32 device tensor of one element) and, asynchronously, in the pinned
        host tensor `host_n_records` (valid once the stream has passed this point)."""
        w, h = self.screen_data.width, self.screen_data.height
        if shards.dim() != 4 or not shards.is_contiguous() or shards.element_size() != 1:
            raise ValueError("pack_sparse_async: shards (count, rows, W, 4) uint8, contiguous")
        count, rows = int(shards.shape[0]), int(shards.shape[1])
        want_rows = sum(min(STRIPE_ROWS, h - s * STRIPE_ROWS) for s in stripes)
        if tuple(shards.shape) != (count, rows, w, 4) or rows != want_rows:
            raise ValueError(f"pack_sparse_async: shape {tuple(shards.shape)} is not {count} shards of {want_rows} x {w}")
        if (records.dim() != 2 or int(records.shape[1]) != SPARSE_RECORD_BYTES or records.element_size() != 1
                or not records.is_contiguous() or int(records.shape[0]) < self.sparse_capacity(count, stripes)):
            raise ValueError(f"pack_sparse_async: records must be (>= {self.sparse_capacity(count, stripes)}, "
                             f"{SPARSE_RECORD_BYTES}) uint8, contiguous")
        if n_records.numel() != 1 or n_records.element_size() != 4:
            raise ValueError("pack_sparse_async: n_records is one 32-bit integer on the device")
        if host_n_records is not None and (host_n_records.numel() != 1 or host_n_records.element_size() != 4
                                           or not host_n_records.is_pinned()):
            raise ValueError("pack_sparse_async: host_n_records is one 32-bit integer in pinned host memory")
        st = _stripe_array(stripes)
        stream = self._stream_handle(stream, "pack_sparse_async")
        self._order_after_producer(records, stream)
        check(lib.kifs_pack_sparse_async(self._ctx, stream, count,
                                         _device_pointer(shards), w * 4, rows * w * 4, st, len(st), encode,
                                         _device_pointer(records), int(records.shape[0]), _device_pointer(n_records),
                                         _device_pointer(host_n_records) if host_n_records is not None else None),
              "pack_sparse_async")

    def erase_sparse_async(self, frames, records, n_records: int, stripes, stream=None, encode: int = ENCODE_SRGB):
        """The background over the tiles of the first `n_records` records (kifs_erase_sparse_async)."""
        self.unpack_sparse_async(frames, records, n_records, stripes, stream=stream, _erase_encode=encode)

    def unpack_sparse_async(self, frames, records, n_records: int, stripes, stream=None, _erase_encode=None):
        """Root side: the first `n_records` records of `records` -> their rows of `frames` (count, H, W, 4)."""
        w, h = self.screen_data.width, self.screen_data.height
        if frames.dim() != 4 or tuple(frames.shape[1:]) != (h, w, 4) or not frames.is_contiguous() or frames.element_size() != 1:
            raise ValueError(f"unpack_sparse_async: frames (count, {h}, {w}, 4) uint8, contiguous")
        count = int(frames.shape[0])
        if n_records < 0 or n_records > self.sparse_capacity(count, stripes) or (n_records and (
                records.dim() != 2 or int(records.shape[1]) != SPARSE_RECORD_BYTES or int(records.shape[0]) < n_records
                or not records.is_contiguous() or records.element_size() != 1)):
            raise ValueError("unpack_sparse_async: n_records records of 1040 bytes, at most one per tile of the shards")
        st = _stripe_array(stripes)
        stream = self._stream_handle(stream, "erase_sparse_async" if _erase_encode is not None else "unpack_sparse_async")
        self._order_after_producer(frames, stream)
        if _erase_encode is not None:
            check(lib.kifs_erase_sparse_async(self._ctx, stream, count,
                                              _device_pointer(frames), w * 4, h * w * 4,
                                              _device_pointer(records) if n_records else None, n_records, st, len(st),
                                              _erase_encode), "erase_sparse_async")
            return
        check(lib.kifs_unpack_sparse_async(self._ctx, stream, count,
                                           _device_pointer(frames), w * 4, h * w * 4,
                                           _device_pointer(records) if n_records else None, n_records, st, len(st)),
              "unpack_sparse_async")

    def fill_shard_async(self, frames, stripes, stream=None, encode: int = ENCODE_SRGB):
        """The background over the rows of `stripes` of `frames` (count, H, W, 4)."""
        w, h = self.screen_data.width, self.screen_data.height
        if frames.dim() != 4 or tuple(frames.shape[1:]) != (h, w, 4) or not frames.is_contiguous() or frames.element_size() != 1:
            raise ValueError(f"fill_shard_async: frames (count, {h}, {w}, 4) uint8, contiguous")
        st = _stripe_array(stripes)
        stream = self._stream_handle(stream, "fill_shard_async")
        self._order_after_producer(frames, stream)
        check(lib.kifs_fill_shard_async(self._ctx, stream, int(frames.shape[0]),
                                        _device_pointer(frames), w * 4, h * w * 4, st, len(st), encode),
              "fill_shard_async")

    def debug_last_round_steps(self) -> int:
        """Round length of the ray re-queuing in the latest launch (0: one wave per block)."""
        return int(lib.kifs_debug_last_round_steps(self._ctx))

    def debug_last_group_tiles(self) -> int:
        """Latest launch: 0 = one wave per tile (render_wave_kernel), 1 / 2 = tiles per 256-thread
        workgroup (render_group_kernel), -1 = no ray re-queuing."""
        return int(lib.kifs_debug_last_group_tiles(self._ctx))

    KERNEL_NAMES = ("render_kernel", "render_group_kernel", "render_wave_kernel", "render_bunny_quad_kernel",
                    "render_bunny_coop_kernel")

    def debug_last_kernel(self) -> str:
        """Name of the render kernel the latest launch used ("" before the first)."""
        k = int(lib.kifs_debug_last_kernel(self._ctx))
        return self.KERNEL_NAMES[k] if 0 <= k < len(self.KERNEL_NAMES) else ""

    def debug_last_bunny_form(self) -> int:
        """The bunny's throughput form in the latest launch: 0 four lanes per ray (weights in VGPRs), 1 four waves per
        64 rays, 2 four lanes per ray with layer 2 in LDS; -1 = not a re-queued bunny launch."""
        return int(lib.kifs_debug_last_bunny_form(self._ctx))

    def set_frames_in_flight(self, n: int):
        """Scheduling hint: the caller keeps n frames in flight on this device (one context and
        stream each).  n > 1 trades the lone-frame residency cap for throughput."""
        check(lib.kifs_set_frames_in_flight(self._ctx, int(n)), "set_frames_in_flight")

    def set_profiling(self, every: int = 1):
        """every = 0: off; 1: time every launch; n: every n-th launch."""
        check(lib.kifs_set_profiling(self._ctx, int(every)), "set_profiling")

    def profile_read(self):
        """(launches, mean_ms, min_ms, max_ms) of the render kernel since the last read."""
        n, mean, lo, hi = C.c_int(), C.c_double(), C.c_double(), C.c_double()
        check(lib.kifs_profile_read(self._ctx, C.byref(n), C.byref(mean), C.byref(lo), C.byref(hi)),
              "profile_read")
        return n.value, mean.value, lo.value, hi.value

    def last_kernel_ms(self) -> float:
        return float(lib.kifs_last_kernel_ms(self._ctx))

    def synchronize(self):
        check(lib.kifs_synchronize(self._ctx), "synchronize")

    # -- point evaluation (parity tooling) -------------------------------------------------
    def eval_points(self, points, want_normals=True):
        pts = np.ascontiguousarray(points, dtype=np.float32).reshape(-1, 3)
        n = pts.shape[0]
        sdf = np.empty(n, dtype=np.float32)
        nrm = np.empty((n, 3), dtype=np.float32) if want_normals else None
        fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float)) if a is not None else None
        check(lib.kifs_eval_points(self._ctx, fp(pts), n, fp(sdf), fp(nrm)), "eval_points")
        return sdf, nrm

    def debug_counters(self, enable: bool = True):
        """Switch per-wave diagnostics on/off (and reset them); returns the sums of the records
        written since the last call (see kifs_debug_wave_records)."""
        try:
            rec = self.debug_wave_records()
        except KifsError:
            rec = np.zeros((0, 4), dtype=np.uint64)
        out = (C.c_uint64 * 8)()
        check(lib.kifs_debug_counters(self._ctx, 1 if enable else 0, out), "debug_counters")
        rec = rec[rec[:, 0] > 0]
        return dict(fast_steps=int((rec[:, 2] & np.uint64(0xffffffff)).sum()),
                    fast_entries=int((rec[:, 2] >> np.uint64(32)).sum()),
                    general_steps=int(rec[:, 3].sum()), waves=len(rec),
                    fast_ticks=int(rec[:, 1].sum()), wave_ticks=int(rec[:, 0].sum()))

    def debug_get_tile_order(self):
        buf = np.zeros(1 << 22, dtype=np.uint32)
        n = C.c_size_t(0)
        check(lib.kifs_debug_get_tile_order(self._ctx, buf.ctypes.data_as(C.POINTER(C.c_uint32)),
                                            buf.size, C.byref(n)), "get_tile_order")
        return buf[:n.value].copy()

    def debug_set_tile_order(self, order):
        o = np.ascontiguousarray(order, dtype=np.uint32)
        check(lib.kifs_debug_set_tile_order(self._ctx, o.ctypes.data_as(C.POINTER(C.c_uint32)),
                                            o.size), "set_tile_order")

    def debug_wave_records(self, max_waves: int = 1 << 20):
        """(n_waves, 4) uint64: total ticks, long-ray-loop ticks, long-ray steps, general steps."""
        buf = np.zeros((max_waves, 4), dtype=np.uint64)
        n = C.c_size_t(0)
        check(lib.kifs_debug_wave_records(self._ctx, buf.ctypes.data_as(C.POINTER(C.c_uint64)),
                                          max_waves, C.byref(n)), "debug_wave_records")
        return buf[:n.value]

    def eval_math(self, fn: int, x, param: float = 0.0):
        xs = np.ascontiguousarray(x, dtype=np.float32).ravel()
        out = np.empty_like(xs)
        fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
        check(lib.kifs_eval_math(self._ctx, fn, fp(xs), param, fp(out), xs.size), "eval_math")
        return out


class MultiGraphicState:
    """Single-process multi-GPU renderer over kifs_multi_* -- what the reference's one-process host would hold
    in place of its GraphicState (graphics.rs:25-37) on a node with several GPUs: one context per device, 8-row
    stripes dealt to the devices, shards collected on the first device.  render(): one frame, dense, synchronous.
    render_batch_async() / wait(): steps of up to 512 frames, sparse records over RCCL grouped send/recv inside
    the library, two steps in flight."""

    def __init__(self, devices, screen_data: ScreenData, camera_data: CameraData = None,
                 gui_data: GuiData = None, iters=(100, 10, 10)):
        arr = (C.c_int * len(devices))(*devices)
        st = C.c_int(0)
        self._m = lib.kifs_multi_create(arr, len(devices), C.byref(st))
        if not self._m:
            raise KifsError(st.value, f"kifs_multi_create({list(devices)})")
        self.devices = list(devices)
        self.screen_data = screen_data
        u = screen_data.into_buffer_data()
        check(lib.kifs_multi_set_screen(self._m, C.byref(u)), "multi set_screen")
        self.set_camera(camera_data or CameraData())
        self.update_options(gui_data or GuiData())
        check(lib.kifs_multi_set_iters(self._m, *iters), "multi set_iters")

    def set_camera(self, camera_data: CameraData):
        u = camera_data.into_buffer_data()
        check(lib.kifs_multi_set_camera(self._m, C.byref(u)), "multi set_camera")

    def update_options(self, gui_data: GuiData):
        u = gui_data.into_buffer_data()
        check(lib.kifs_multi_set_options(self._m, C.byref(u)), "multi set_options")

    def render(self, out=None, encode: int = ENCODE_SRGB, pitch_bytes: int = None):
        w, h = self.screen_data.width, self.screen_data.height
        if out is None:
            out = np.empty((h, w, 4), dtype=np.uint8)
        if isinstance(out, np.ndarray):
            _check_host_destination(out, h, w, pitch_bytes or w * 4)
        ptr = out.ctypes.data if isinstance(out, np.ndarray) else _device_pointer(out)
        self._order_after_producer(out)
        check(lib.kifs_multi_render(self._m, ptr, pitch_bytes or w * 4, encode), "multi render")
        return out

    def _order_after_producer(self, dest):
        """GraphicState._order_after_producer for the root device's streams (kifs_multi_order_after): what this object
        enqueues on the root from now on follows torch's current stream there."""
        producer = _producer_stream(dest)
        if producer is not None:
            check(lib.kifs_multi_order_after(self._m, producer or None), "multi order_after")

    # ---- batches of frames, gathered on the first device (kifs_multi_render_batch_async) ----------------
    def set_extensions(self, soft_shadow=False, shadow_steps=0, shadow_k=0.0, shadow_t0=0.0, shadow_max_t=0.0):
        e = ExtensionsC(1 if soft_shadow else 0, int(shadow_steps), float(shadow_k), float(shadow_t0), float(shadow_max_t))
        check(lib.kifs_multi_set_extensions(self._m, C.byref(e)), "multi set_extensions")

    def set_gather(self, gather: str = "sparse", transport: str = "auto"):
        """gather: 'sparse' (the other devices send only the tiles that hold something) or 'dense';
        transport: 'auto', 'rccl' (grouped ncclSend / ncclRecv inside the library) or 'copy' (peer copies)."""
        g = {"sparse": _lib.GATHER_SPARSE, "dense": _lib.GATHER_DENSE}[gather]
        t = {"auto": _lib.TRANSPORT_AUTO, "rccl": _lib.TRANSPORT_RCCL, "copy": _lib.TRANSPORT_COPY}[transport]
        check(lib.kifs_multi_set_gather(self._m, g, t), f"multi set_gather({gather}, {transport})")

    def _frames_args(self, frames, cameras):
        w, h = self.screen_data.width, self.screen_data.height
        if (frames.dim() != 4 or tuple(frames.shape[1:]) != (h, w, 4) or frames.element_size() != 1
                or not frames.is_contiguous()):
            raise ValueError(f"render_batch: frames (count, {h}, {w}, 4) uint8, contiguous, on the first listed device")
        cams = camera_array(cameras)
        if len(cams) != int(frames.shape[0]) or not 1 <= len(cams) <= MAX_BATCH:
            raise ValueError(f"render_batch: {int(frames.shape[0])} frames for {len(cams)} cameras (1..{MAX_BATCH})")
        if frames.is_cuda and frames.device.index not in (None, self.devices[0]):
            raise ValueError("render_batch: frames must live on the first listed device (the root of the gather)")
        return len(cams), cams, _device_pointer(frames), w * 4, h * w * 4

    def render_batch_async(self, frames, cameras, encode: int = ENCODE_SRGB, untouched: bool = False) -> int:
        """Submits one step: frames[i] (a (count, H, W, 4) uint8 tensor on the first listed device) <- cameras[i].
        Returns the step number for wait() / stream_wait().  untouched=True: `frames` is the buffer passed two
        steps ago and nothing else wrote to it since (KIFS_MULTI_FRAMES_UNTOUCHED)."""
        n, cams, ptr, pitch, stride = self._frames_args(frames, cameras)
        self._order_after_producer(frames)
        step = C.c_uint64(0)
        check(lib.kifs_multi_render_batch_async(self._m, n, cams, ptr, pitch, stride, encode,
                                                _lib.MULTI_FRAMES_UNTOUCHED if untouched else 0, C.byref(step)),
              "multi render_batch_async")
        return int(step.value)

    def render_batch(self, frames, cameras, encode: int = ENCODE_SRGB):
        n, cams, ptr, pitch, stride = self._frames_args(frames, cameras)
        self._order_after_producer(frames)
        check(lib.kifs_multi_render_batch(self._m, n, cams, ptr, pitch, stride, encode), "multi render_batch")
        return frames

    def wait(self, step: int):
        check(lib.kifs_multi_wait(self._m, int(step)), f"multi wait({step})")

    def wait_all(self):
        check(lib.kifs_multi_wait_all(self._m), "multi wait_all")

    def stream_wait(self, step: int, stream):
        handle = stream.cuda_stream if hasattr(stream, "cuda_stream") else stream
        if not handle:
            raise ValueError("stream_wait: pass a non-default torch.cuda.Stream")
        check(lib.kifs_multi_stream_wait(self._m, int(step), handle), f"multi stream_wait({step})")

    def stats(self, reset: bool = False) -> dict:
        st = _lib.MultiStatsC()
        check(lib.kifs_multi_stats(self._m, C.byref(st), 1 if reset else 0), "multi stats")
        return {"steps": int(st.steps), "records_received": int(st.records_received), "tiles_covered": int(st.tiles_covered),
                "bytes_received": int(st.bytes_received),
                "transport": {0: "auto", 1: "rccl", 2: "copy"}[int(st.transport)],
                "gather": {0: "sparse", 1: "dense"}[int(st.gather)],
                "rccl_version": int(st.rccl_version), "comm_ranks": int(st.comm_ranks)}

    def comm_selftest(self, nbytes: int = 1 << 20):
        check(lib.kifs_multi_comm_selftest(self._m, int(nbytes)), "multi comm_selftest")

    def set_weights(self, weights=None):
        """Shares of the devices (one integer each, None = equal)."""
        arr = None if weights is None else (C.c_int * len(self.devices))(*[int(w) for w in weights])
        check(lib.kifs_multi_set_weights(self._m, arr), "multi set_weights")

    def shards(self):
        """[(device, n_stripes, rows, kernel ms of the last render)] per listed device."""
        res = []
        for i in range(len(self.devices)):
            d, n, rows = C.c_int(), C.c_int(), C.c_int()
            check(lib.kifs_multi_shard(self._m, i, C.byref(d), C.byref(n), C.byref(rows)))
            res.append((d.value, n.value, rows.value, float(lib.kifs_multi_shard_ms(self._m, i))))
        return res

    def close(self):
        if getattr(self, "_m", None):
            lib.kifs_multi_destroy(self._m)
            self._m = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _check_host_destination(out: np.ndarray, rows: int, width: int, pitch: int):
    """A host destination must hold rows of `pitch` bytes: the library copies (rows-1)*pitch + 4*width
    bytes into it and knows nothing about the array's size."""
    if out.dtype != np.uint8 or not out.flags.c_contiguous:
        raise ValueError("render: host destination must be a C-contiguous uint8 array")
    if pitch < width * 4 or pitch % 4:
        raise ValueError("render: pitch_bytes must be a multiple of 4 and at least 4 * width")
    if rows > 0 and out.nbytes < (rows - 1) * pitch + width * 4:
        raise ValueError(f"render: host destination of {out.nbytes} bytes is too small for {rows} rows "
                         f"of pitch {pitch}")


def _stripe_array(stripes):
    return (C.c_int * len(stripes))(*[int(s) for s in stripes])


def shard_stripes(height: int, rank: int, world: int, weights=None):
    """kifs_shard_stripes: (stripe indices of `rank`, total rows) when the frame's 8-row stripes
    are dealt to `world` ranks (weights None: equal shares = stripes rank, rank + world, ...)."""
    cap = (height + STRIPE_ROWS - 1) // STRIPE_ROWS
    buf = (C.c_int * max(cap, 1))()
    n, rows = C.c_int(), C.c_int()
    wts = None if weights is None else (C.c_int * world)(*[int(x) for x in weights])
    check(lib.kifs_shard_stripes(height, world, wts, rank, buf, cap, C.byref(n), C.byref(rows)), "shard_stripes")
    return list(buf[:n.value]), rows.value


def band_range(height: int, rank: int, world: int):
    y0, y1 = C.c_int(), C.c_int()
    check(lib.kifs_band_range(height, rank, world, C.byref(y0), C.byref(y1)), "band_range")
    return y0.value, y1.value
