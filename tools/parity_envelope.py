#!/usr/bin/env python3
"""Quantifies "parity unpinned": how far two LEGAL evaluations of the reference shader differ.

The reference (WGSL through naga + a driver) leaves builtin precision, fma contraction and
the order of dot/length to the implementation, and it cannot be run here.
The C oracle fixes one legal evaluation (fma chains, pinned polynomial log); the NumPy
restatement (oracle/kifs_oracle_np.py) is another (no fma, libm log/sqrt).  This script
renders BASELINE-sized frames with both and reports, per workload,

  differ        fraction of pixels whose RGBA8 differs at all
  differ_gt1    fraction differing by more than 1 in some channel (north_star's tolerance)
  flips         fraction whose hit/miss decision differs (a 1-ulp change of the estimate
                crossing `distance < epsilon`): these are the > 1 differences
  max_nonflip   largest channel difference among pixels with the same hit/miss decision

That is the envelope to expect against any real driver; the HIP kernels are held to the C
oracle bit for bit.  CPU only (build container); writes profiles/<round>/parity_envelope.json.
The NumPy side renders big frames in bands of rows (memory), the C oracle whole frames.

    python tools/parity_envelope.py [--round r03] [workload ...]
"""
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

import oracle as O  # noqa: E402
from oracle import kifs_oracle_np as NP  # noqa: E402
import kifs_raymarching_amd as K  # noqa: E402  (host packing only: no GPU call is made)
from kifs_raymarching_amd.configs import WORKLOADS  # noqa: E402

DEFAULT = ["cfg2_julia_1080p", "cfg3_sierpinski_1080p", "ref_julia_1080p", "cfg1_julia_256", "n1_genjulia_1080p",
           "n1_genjulia_1080p_heatmap", "n1_genjulia_p3_1080p", "n1_genjulia_p8_n3_1080p",
           "n2_bunny_1080p", "cfg4_julia_4096", "cfg5_sierpinski_8k_orbit"]
BAND_ROWS = 256


def _variants():
    """Envelope-only views of the generalised Julia set.  With the reference's constants (power 8, ten normal
    iterations WITHOUT an escape test, gen_julia.wgsl:41-48) every orbit of the six offset points overflows, the
    normal is inf - inf = NaN and the hit pixel is black in BOTH readings -- so the as-shipped workload says nothing
    about the orbit arithmetic.  These do: the march alone (heatmap), a power whose normals stay finite, and power 8
    with three normal iterations."""
    import dataclasses
    base = WORKLOADS["n1_genjulia_1080p"]
    gui = lambda **kw: dataclasses.replace(base.gui, **kw)
    return {
        "n1_genjulia_1080p_heatmap": dataclasses.replace(base, name=base.name + ", heatmap", gui=gui(is_heatmap=True)),
        "n1_genjulia_p3_1080p": dataclasses.replace(base, name="1920x1080 generalised Julia, power 3, reference constants (100/10)",
                                                    gui=gui(power=3.0)),
        "n1_genjulia_p8_n3_1080p": dataclasses.replace(base, name="1920x1080 generalised Julia, power 8, 100 SDF / 3 normal iterations",
                                                       iters=(100, 3, 10)),
    }


def envelope(key, rows=None):
    """`rows` = (y0, y1): only that band of the frame (the 8K frames take minutes whole; a band through the fractal
    is what tests/test_parity_envelope.py re-renders)."""
    w = WORKLOADS.get(key) or _variants()[key]
    ub = K.uniform_bytes
    s = O.from_bytes(O.Screen, ub(w.screen.into_buffer_data()))
    c = O.from_bytes(O.Camera, ub(w.camera.into_buffer_data()))
    o = O.from_bytes(O.Options, ub(w.gui.into_buffer_data()))
    it = O.iters(*w.iters)
    t0 = time.perf_counter()
    H = w.screen.height
    ya, yb = rows if rows else (0, H)
    a = O.render(s, c, o, it, y0=ya, y1=yb)
    t1 = time.perf_counter()
    _, steps_c, _ = O.render_stats(s, c, o, it, y0=ya, y1=yb)
    parts = [NP.render(s, c, o, it, y0=y, y1=min(yb, y + BAND_ROWS)) for y in range(ya, yb, BAND_ROWS)]
    b = np.concatenate([p[0] for p in parts], axis=0)
    hit_np = np.concatenate([p[2] for p in parts], axis=0)
    steps_np = np.concatenate([p[1] for p in parts], axis=0)
    t2 = time.perf_counter()
    bg = O.render(s, c, o, it, y0=0, y1=1)[0, 0].copy()  # the corner is background in every BASELINE view
    hit_c = (a != bg).any(-1)
    d = np.abs(a.astype(np.int16) - b.astype(np.int16)).max(-1)
    flip = hit_c != hit_np
    n = d.size
    return {
        "workload": key + (f"@rows{ya}-{yb}" if rows else ""), "description": w.name, "pixels": int(n),
        "hit_pixels_c_oracle": int(hit_c.sum()), "hit_pixels_numpy": int(hit_np.sum()),
        "differ": float((d > 0).sum() / n), "differ_pixels": int((d > 0).sum()),
        "differ_gt1": float((d > 1).sum() / n), "differ_gt1_pixels": int((d > 1).sum()),
        "flips": float(flip.sum() / n), "flip_pixels": int(flip.sum()),
        "max_nonflip": int(d[~flip].max()) if (~flip).any() else 0,
        "nonflip_differ_pixels": int(((d > 0) & ~flip).sum()),
        "max_any": int(d.max()),
        # the march itself, whatever the shading makes of it: per-pixel loop counters of entry.wgsl:12-25
        "march_steps_differ_pixels": int((steps_c.astype(np.int64) != steps_np.astype(np.int64)).sum()),
        "c_oracle_s": round(t1 - t0, 1), "numpy_s": round(t2 - t1, 1),
    }


def parse_key(k):
    """'workload' or 'workload@rowsA-B' -> (workload, rows or None)."""
    if "@rows" in k:
        k, r = k.split("@rows")
        a, b = r.split("-")
        return k, (int(a), int(b))
    return k, None


def main():
    args = sys.argv[1:]
    rnd = "r03"
    if args[:1] == ["--round"]:
        rnd, args = args[1], args[2:]
    keys = args or DEFAULT
    out = ROOT / "profiles" / rnd / "parity_envelope.json"
    out.parent.mkdir(parents=True, exist_ok=True)
    res = json.loads(out.read_text()) if (out.exists() and args) else []  # named workloads: add to the file
    res = [r for r in res if r["workload"] not in keys]
    for k in keys:
        r = envelope(*parse_key(k))
        print(json.dumps(r), flush=True)
        res.append(r)
        out.write_text(json.dumps(res, indent=1) + "\n")


if __name__ == "__main__":
    main()
