"""Shared helpers for the parity tests: drive the oracle with the product's uniform bytes."""
import ctypes as C

import numpy as np


def oracle_uniforms(O, K, workload_or_parts):
    """(screen, camera, options) oracle structs built from the PRODUCT's byte images, so a
    parity test compares renderers on identical 156 bytes."""
    screen, camera, gui = workload_or_parts
    ub = K.uniform_bytes
    return (O.from_bytes(O.Screen, ub(screen.into_buffer_data())),
            O.from_bytes(O.Camera, ub(camera.into_buffer_data())),
            O.from_bytes(O.Options, ub(gui.into_buffer_data())))


def oracle_frame(O, K, screen, camera, gui, iters, encode=1, y0=0, y1=None, nthreads=0):
    s, c, o = oracle_uniforms(O, K, (screen, camera, gui))
    return O.render(s, c, o, O.iters(*iters), encode=encode, y0=y0, y1=y1, nthreads=nthreads)


def gpu_frame(gs, screen, camera, gui, iters, encode=1, y0=0, y1=None):
    gs.update_screen_data(screen)
    gs.set_camera(camera)
    gs.update_options(gui)
    gs.set_iters(*iters)
    return gs.render(y0=y0, y1=y1, encode=encode)


def diff_report(a, b):
    d = np.abs(a.astype(np.int16) - b.astype(np.int16))
    bad = (d > 0).any(-1)
    return {"max_abs": int(d.max()) if d.size else 0, "mismatched_pixels": int(bad.sum()),
            "pixels": int(bad.size)}
