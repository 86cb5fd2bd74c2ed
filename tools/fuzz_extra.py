#!/usr/bin/env python3
"""Longer random parity run for the round's new code paths (beyond tests/test_gpu_parity_fuzz.py): batches big enough for
render_wave_kernel and mid-size ones for render_group_kernel with the soft-shadow extension on (pooled secondary rays), and
bunny batches through render_bunny_coop_kernel, random primitives / cameras / epsilon / extension parameters, two views per launch against the
oracle.  Prints one line per launch and a summary; exit code 1 on any mismatch.
    python tools/fuzz_extra.py [launches] [seed]"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import torch  # noqa: E402

import kifs_raymarching_amd as K  # noqa: E402
import oracle as O  # noqa: E402
from helpers import diff_report, oracle_uniforms  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 20261006)
FG, PS = K.FractalGroup, K.PrimitiveShape
BOUND = {PS.Sphere: 1.0, PS.Cylinder: 2.236, PS.Box: 1.732, PS.Torus: 1.3, PS.SierpinskiTetrahedron: 2.0, PS.Bunny: 1.0}
bad = 0
gs = K.GraphicState(0, screen_data=K.ScreenData(64, 64), camera_data=K.CameraData(), gui_data=K.GuiData())
for it in range(N):
    kind = rng.choice(["kifs", "kifs", "julia", "bunny", "genjulia"])
    W, H = int(rng.integers(300, 520)), int(rng.integers(200, 330))
    tiles = ((W + 31) // 32) * ((H + 7) // 8)
    eps = float(10 ** rng.uniform(-4, -2))
    colours = dict(fractal_color=tuple(int(v) for v in rng.integers(0, 256, 3)), background_color=tuple(int(v) for v in rng.integers(0, 256, 3)))
    if kind == "julia":
        gui = K.GuiData(fractal_group=FG.JuliaSet, constant=tuple(float(v) for v in rng.uniform(-0.8, 0.8, 4)),
                        max_iterations=int(rng.integers(40, 160)), epsilon=eps, **colours)
        B, need = 2.0, 12500
    elif kind == "genjulia":  # (round 4: the orbit step's trimmed cores, chunks drawn by ticket in render_group_kernel)
        gui = K.GuiData(fractal_group=FG.GeneralizedJuliaSet, constant=tuple(float(v) for v in rng.uniform(-0.8, 0.8, 4)),
                        power=float(rng.choice([2.0, 3.0, 3.5, 4.0, 7.3, 8.0, float(rng.uniform(1.5, 9.0))])),
                        max_iterations=int(rng.integers(24, 72)), epsilon=eps, **colours)
        B, need = 2.0, int(rng.choice([1500, 6000]))  # one tile per workgroup / pairs (rules::PAIR_FROM_GENJULIA)
    else:
        prim = PS.Bunny if kind == "bunny" else rng.choice([PS.Sphere, PS.Cylinder, PS.Box, PS.Torus, PS.SierpinskiTetrahedron])
        gui = K.GuiData(primitive_shape=prim, max_iterations=int(rng.integers(40, 130)), epsilon=eps, **colours)
        # (the bunny's three re-queued forms by load, rules::BUNNY_*: pairs with every weight in VGPRs, with layer 2 in LDS, four waves per 64 rays)
        B, need = BOUND[prim], (int(rng.choice([1800, 3200, 5500])) if kind == "bunny" else 32000)
    iters = (int(rng.integers(4, 16)), int(rng.integers(1, 11)), int(rng.integers(2, 14)))
    d = float(B * rng.uniform(0.97, 1.0))  # on the bounding sphere: every tile counts as heavy
    px_tiles = W * H / 256.0
    views = int(min(120, np.ceil(need / px_tiles) + 1))
    if kind not in ("bunny", "genjulia") and rng.integers(0, 3) == 0:
        # a mid-size launch instead: a bigger frame seen from outside, few views -> render_group_kernel (rays re-queued by a
        # 256-thread workgroup; with the extension on: its pooled secondary rays)
        W, H = int(rng.integers(780, 900)), int(rng.integers(480, 560))
        d = float(B + rng.uniform(0.3, 2.5))
        views = int(rng.integers(4, 9))
    cams = [K.CameraData(origin_distance=d, min_distance=0.3, phi=float(rng.uniform(0, 6.28)), theta=float(rng.uniform(-1.2, 1.2))) for _ in range(views)]
    shadows = bool(rng.integers(0, 4) != 0)
    ext_args = dict(soft_shadow=shadows, shadow_steps=int(rng.integers(1, 48)), shadow_k=float(rng.uniform(1, 16)),
                    shadow_t0=float(rng.uniform(0.005, 0.1)), shadow_max_t=float(rng.uniform(0.5, 8)))
    screen = K.ScreenData(W, H)
    gs.update_screen_data(screen)
    gs.update_options(gui)
    gs.set_iters(*iters)
    gs.set_extensions(**ext_args)
    ext = O.Ext(1 if shadows else 0, ext_args["shadow_steps"], ext_args["shadow_k"], ext_args["shadow_t0"], ext_args["shadow_max_t"])
    encode = int(rng.integers(0, 2))
    outs = torch.zeros((views, H, W, 4), dtype=torch.uint8, device="cuda:0")
    torch.cuda.synchronize()
    st = torch.cuda.Stream()
    gs.render_batch_async([outs[i] for i in range(views)], cams, stream=st, encode=encode)
    st.synchronize()
    kernel = gs.debug_last_kernel() + (f"/form{gs.debug_last_bunny_form()}" if kind == "bunny" else f"/T{gs.debug_last_group_tiles()}")
    got = outs.cpu().numpy()
    worst = 0
    for k in sorted({int(rng.integers(0, views)), views - 1}):
        s, c, o = oracle_uniforms(O, K, (screen, cams[k], gui))
        want = O.render(s, c, o, O.iters(*iters), encode=encode, ext=ext)
        worst = max(worst, diff_report(got[k], want)["mismatched_pixels"])
    bad += worst > 0
    print(f"{it:3d} {kind:6s} {W}x{H} x{views:<3d} {kernel:32s} shadows={int(shadows)} steps={ext_args['shadow_steps']:2d} eps={eps:.1e} "
          f"iters={iters} mismatched={worst}", flush=True)
    del outs
gs.close()
print(f"{N} launches, {bad} with mismatches")
sys.exit(1 if bad else 0)
