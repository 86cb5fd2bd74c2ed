"""Presentation side of the path: write an RGBA8 frame to disk.

The reference presents frames on a window surface (render.rs:354-356) and has no read-back;
a headless harness needs files.  PNG (stdlib zlib, no dependencies) and binary PPM.  Frames
produced with the sRGB colour target (the reference's preferred surface format,
render.rs:72-80) are already display-encoded, which is what both formats expect.
"""
import struct
import zlib

import numpy as np


def _chunk(tag: bytes, data: bytes) -> bytes:
    return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xffffffff)


def png_bytes(rgba: np.ndarray, keep_alpha: bool = False, level: int = 6) -> bytes:
    """`rgba`: (H, W, 4) uint8, rows top to bottom (y = 0 is the top row, entry.wgsl:54) -> a PNG file image."""
    a = np.ascontiguousarray(rgba, dtype=np.uint8)
    if a.ndim != 3 or a.shape[2] != 4:
        raise ValueError("png_bytes: expected an (H, W, 4) uint8 array")
    h, w, _ = a.shape
    body = a if keep_alpha else a[..., :3]
    raw = np.concatenate([np.zeros((h, 1), dtype=np.uint8), body.reshape(h, -1)], axis=1).tobytes()
    ihdr = struct.pack(">IIBBBBB", w, h, 8, 6 if keep_alpha else 2, 0, 0, 0)
    return (b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", ihdr) + _chunk(b"IDAT", zlib.compress(raw, level))
            + _chunk(b"IEND", b""))


def write_png(path, rgba: np.ndarray, keep_alpha: bool = False):
    with open(path, "wb") as f:
        f.write(png_bytes(rgba, keep_alpha))


def write_ppm(path, rgba: np.ndarray):
    a = np.ascontiguousarray(rgba, dtype=np.uint8)
    h, w, _ = a.shape
    with open(path, "wb") as f:
        f.write(b"P6\n%d %d\n255\n" % (w, h))
        f.write(a[..., :3].tobytes())


def read_png_rgb(path) -> np.ndarray:
    """Minimal reader for files written by write_png (filter type 0 rows only); `path` may also be the
    file's bytes."""
    data = path if isinstance(path, (bytes, bytearray)) else open(path, "rb").read()
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, w = 8, b"", 0
    h = channels = 0
    while pos < len(data):
        n, tag = struct.unpack(">I4s", data[pos:pos + 8])
        payload = data[pos + 8:pos + 8 + n]
        if tag == b"IHDR":
            w, h, _, ctype = struct.unpack(">IIBB", payload[:10])
            channels = 4 if ctype == 6 else 3
        elif tag == b"IDAT":
            idat += payload
        pos += 12 + n
    rows = np.frombuffer(zlib.decompress(idat), dtype=np.uint8).reshape(h, 1 + w * channels)
    assert (rows[:, 0] == 0).all()
    return rows[:, 1:].reshape(h, w, channels)
