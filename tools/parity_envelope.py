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
oracle bit for bit.  CPU only (build container); writes profiles/r02/parity_envelope.json.

    python tools/parity_envelope.py [workload ...]
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

DEFAULT = ["cfg2_julia_1080p", "cfg3_sierpinski_1080p", "ref_julia_1080p"]


def envelope(key):
    w = WORKLOADS[key]
    ub = K.uniform_bytes
    s = O.from_bytes(O.Screen, ub(w.screen.into_buffer_data()))
    c = O.from_bytes(O.Camera, ub(w.camera.into_buffer_data()))
    o = O.from_bytes(O.Options, ub(w.gui.into_buffer_data()))
    it = O.iters(*w.iters)
    t0 = time.perf_counter()
    a = O.render(s, c, o, it)
    t1 = time.perf_counter()
    b, _, hit_np = NP.render(s, c, o, it)
    t2 = time.perf_counter()
    bg = a[0, 0].copy()  # the corner is background in every BASELINE view
    hit_c = (a != bg).any(-1)
    d = np.abs(a.astype(np.int16) - b.astype(np.int16)).max(-1)
    flip = hit_c != hit_np
    n = d.size
    return {
        "workload": key, "description": w.name, "pixels": int(n),
        "hit_pixels_c_oracle": int(hit_c.sum()), "hit_pixels_numpy": int(hit_np.sum()),
        "differ": float((d > 0).sum() / n), "differ_pixels": int((d > 0).sum()),
        "differ_gt1": float((d > 1).sum() / n), "differ_gt1_pixels": int((d > 1).sum()),
        "flips": float(flip.sum() / n), "flip_pixels": int(flip.sum()),
        "max_nonflip": int(d[~flip].max()) if (~flip).any() else 0,
        "nonflip_differ_pixels": int(((d > 0) & ~flip).sum()),
        "max_any": int(d.max()),
        "c_oracle_s": round(t1 - t0, 1), "numpy_s": round(t2 - t1, 1),
    }


def main():
    keys = sys.argv[1:] or DEFAULT
    out = ROOT / "profiles" / "r02" / "parity_envelope.json"
    out.parent.mkdir(parents=True, exist_ok=True)
    res = []
    for k in keys:
        r = envelope(k)
        print(json.dumps(r), flush=True)
        res.append(r)
        out.write_text(json.dumps(res, indent=1) + "\n")


if __name__ == "__main__":
    main()
