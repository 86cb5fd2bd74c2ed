#!/usr/bin/env python3
"""Generates the golden fixtures under tests/golden/ from the CPU oracle.

The reference cannot run here and pins no shader output, so these vectors
are the oracle's own output, frozen: they detect drift of the oracle (CPU test) and are a
second anchor for the HIP kernels (GPU test).  Inputs are stored as the exact uniform
bytes, so a fixture is self-contained data: (156 uniform bytes, iteration counts,
encode) -> RGBA8 frame.  Re-run only when the arithmetic contract changes on purpose.
"""
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parent.parent))
import oracle as O  # noqa: E402

JULIA_C = (-0.2, 0.6, 0.2, 0.2)
CASES = {
    # name: (W, H, camera(d, phi, theta), gui kwargs, iters, encode)
    "cfg1_julia_256": (256, 256, (5.0, 0.0, 0.0), dict(max_iterations=64, fractal_group=1, constant=JULIA_C), (8, 10, 10), 1),
    "cfg2_julia_thumb": (96, 54, (5.0, 0.0, 0.0), dict(max_iterations=256, fractal_group=1, constant=JULIA_C), (12, 10, 10), 1),
    "julia_reference_constants": (96, 64, (2.6, 0.7, 0.4), dict(fractal_group=1), (100, 10, 10), 1),
    "julia_heatmap_unorm": (80, 60, (3.0, 0.0, 0.0), dict(fractal_group=1, is_heatmap=True, constant=JULIA_C, fractal_color=(255, 128, 30)), (12, 10, 10), 0),
    "cfg3_sierpinski_thumb": (96, 54, (5.0, 0.0, 0.0), dict(primitive_shape=4), (100, 10, 16), 1),
    "sierpinski_close": (96, 96, (3.0, 1.0, 0.3), dict(primitive_shape=4, background_color=(10, 40, 90)), (100, 10, 10), 1),
    "genjulia_p3": (64, 48, (3.0, 0.5, 0.0), dict(max_iterations=64, fractal_group=2, power=3.0), (8, 4, 10), 1),
    "sphere": (64, 48, (3.5, 0.6, 0.5), dict(primitive_shape=0, fractal_color=(250, 120, 60), background_color=(5, 5, 30)), (100, 10, 10), 1),
    "cylinder": (64, 48, (3.5, 0.6, 0.5), dict(primitive_shape=1, fractal_color=(250, 120, 60)), (100, 10, 10), 1),
    "box": (64, 48, (3.5, 0.6, 0.5), dict(primitive_shape=2, fractal_color=(250, 120, 60)), (100, 10, 10), 0),
    "torus": (64, 48, (3.5, 0.6, 0.5), dict(primitive_shape=3, fractal_color=(250, 120, 60)), (100, 10, 10), 1),
    "bunny": (64, 48, (2.2, 0.6, 0.3), dict(primitive_shape=5, fractal_color=(230, 200, 160)), (100, 10, 10), 1),
}


def build(name):
    w, h, cam, gui, iters, encode = CASES[name]
    sc, ca, op = O.screen_uniform(w, h), O.camera_uniform(*cam), O.options_from_gui(**gui)
    frame = O.render(sc, ca, op, O.iters(*iters), encode=encode)
    raw = lambda u: np.frombuffer(bytes(memoryview(u).cast("B")), dtype=np.uint8)
    return dict(screen=raw(sc), camera=raw(ca), options=raw(op),
                iters=np.array(iters, dtype=np.int32), encode=np.array(encode, dtype=np.int32),
                frame=frame)


if __name__ == "__main__":
    for name in CASES:
        np.savez_compressed(HERE / f"{name}.npz", **build(name))
        print("wrote", name)
