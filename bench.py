#!/usr/bin/env python3
"""bench.py -- headline benchmark of the raymarching hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME]

A "step" is one launch of the render path per rank over its batch of synthetic input:
`--frames-per-launch` frames of the workload's sequence (default 8, one camera and one
destination each; 1 = the lone-frame latency path), rendered by one kifs_render_batch_async
call.  At N > 1 (one process per GPU, launched by torch.distributed.run) the default is
frame-parallel: in step k rank r renders frames (k N + r) B .. + B - 1 of the sequence.
Frames are independent units, so there is no exchange step: each frame stays in the HBM of
the GPU that rendered it (weak scaling: per-GPU work is fixed, value = N x B x pixels per
step / time); `--deliver root` ships the finished frames to rank 0 by grouped RCCL
point-to-point transfers over xGMI instead.  `--shard bands` splits every frame into row
bands gathered into rank 0's frame (strong scaling of one frame: the low-latency mode; it
cannot raise throughput much because every band still contains the frame's longest rays --
DESIGN.md section 6).  The default workload is BASELINE.json's metric configuration:
1920x1080 quaternion-Julia, 256 march steps, 12 SDF iterations.  Rank 0 prints ONE JSON line.

There are no HBM-resident inputs beyond the 156 uniform bytes; the output frame lives in
HBM (torch tensor) and is written by the kernel.  `roofline` prices the dominant kernel
against the HBM-write roofline the north star mandates (4 algorithmic bytes per pixel);
`cpu_baseline` times the CPU oracle (the stand-in for the reference's wgpu path, which
cannot run here) on the host cores -- a reported baseline, not the target.
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
METRIC = "Mpixels/s at 1920x1080, 256 march steps, 12 SDF iters; %HBM-peak"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--workload", default=None, help="name in kifs_raymarching_amd.configs.WORKLOADS")
    ap.add_argument("--cpu-seconds", type=float, default=10.0,
                    help="wall-clock budget of the cpu_baseline sample (0 disables it)")
    ap.add_argument("--encode", type=int, default=1, help="1 = sRGB target (reference default), 0 = UNORM")
    ap.add_argument("--orbit", action="store_true",
                    help="move the camera every frame (phi = 2*pi*frame/frames of the workload, "
                         "host camera model + 64-byte uniform update per step) instead of re-rendering one view")
    ap.add_argument("--shard", default="frames", choices=["frames", "bands"],
                    help="N > 1: whole frames per rank (throughput, default) or row bands of each frame (latency)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend; gloo is only for rehearsing N > 1 on one GPU")
    ap.add_argument("--share-device", action="store_true",
                    help="rehearsal: every rank uses cuda:0 (needs --backend gloo)")
    ap.add_argument("--frames-per-launch", type=int, default=8,
                    help="frames of the sequence rendered by one launch (kifs_render_batch_async, 1..8); "
                         "1 = one frame per launch, the lone-frame latency path")
    ap.add_argument("--frames-in-flight", type=int, default=1,
                    help="launches each rank keeps in flight on separate contexts and streams")
    ap.add_argument("--deliver", default="none", choices=["none", "root"],
                    help="N > 1, --shard frames: leave every frame in the HBM of the GPU that rendered it "
                         "(frames are independent units: no exchange step), or ship them to rank 0 by "
                         "grouped RCCL p2p")
    ap.add_argument("--check", action="store_true",
                    help="after the timed region, compare rank 0's gathered frame with a single-GPU render")
    return ap.parse_args()


def cpu_baseline(w, seconds):
    """Times the CPU oracle on whole frames of the same workload for ~`seconds`."""
    import oracle as O
    import kifs_raymarching_amd as K

    ub = K.uniform_bytes
    s = O.from_bytes(O.Screen, ub(w.screen.into_buffer_data()))
    c = O.from_bytes(O.Camera, ub(w.camera.into_buffer_data()))
    o = O.from_bytes(O.Options, ub(w.gui.into_buffer_data()))
    it = O.iters(*w.iters)
    cores = len(os.sched_getaffinity(0))
    # bound the sample: time a slice first, then whole frames while the budget lasts
    rows = max(8, w.screen.height // 16)
    y0 = (w.screen.height - rows) // 2
    t = time.perf_counter()
    O.render(s, c, o, it, y0=y0, y1=y0 + rows, nthreads=cores)
    slice_s = time.perf_counter() - t
    est_frame = slice_s * w.screen.height / rows  # centre rows are the heavy ones: upper bound
    if est_frame > seconds:  # a frame does not fit the budget: sample centred bands only
        frames, px, t0 = 0, 0, time.perf_counter()
        while time.perf_counter() - t0 < seconds:
            O.render(s, c, o, it, y0=y0, y1=y0 + rows, nthreads=cores)
            px += rows * w.screen.width
            frames += 1
        dt = time.perf_counter() - t0
        sample = f"{frames} x centre band of {rows} rows of the {w.screen.width}x{w.screen.height} frame"
    else:
        frames, t0 = 0, time.perf_counter()
        while frames < 1 or time.perf_counter() - t0 < seconds:
            O.render(s, c, o, it, nthreads=cores)
            frames += 1
        dt = time.perf_counter() - t0
        px = frames * w.pixels
        sample = f"{frames} full {w.screen.width}x{w.screen.height} frames in {dt:.1f} s"
    return {"value": round(px / dt / 1e6, 3), "unit": "Mpixels/s", "cores": cores, "kind": "port",
            "sample": sample + f", C oracle ({O.lib()._variant}), {cores} threads"}


VALU_F32_PEAK_TFLOPS = 157.3  # MI355X vector f32 (MI355X_MICROARCH.md); the path has no MFMA work


def work_count(w, kernel_s):
    """Secondary figure of SURVEY 8(d): algorithmic flops of one frame, counted by the
    instrumented CPU oracle (march steps, Julia iterations / Sierpinski folds, hits), against
    the f32 vector peak.  Flop model: 15 per march step, 24 per Julia
    iteration or 40 per fold, 4 for the log/sqrt/divide tail of a Julia step, and per hit
    40 * normal_iters (Julia Jacobian) -- the six extra SDF calls of a KIFS normal are already
    in the step and fold counts."""
    import oracle as O
    import kifs_raymarching_amd as K

    ub = K.uniform_bytes
    s = O.from_bytes(O.Screen, ub(w.screen.into_buffer_data()))
    c = O.from_bytes(O.Camera, ub(w.camera.into_buffer_data()))
    o = O.from_bytes(O.Options, ub(w.gui.into_buffer_data()))
    it = O.iters(*w.iters)
    W, H = w.screen.width, w.screen.height
    stride = max(1, (W * H + (1 << 22) - 1) >> 22)  # count every stride-th row of big frames
    steps = inner = hits = calls = rows = 0
    bands = [(0, H)] if stride == 1 else [(y, y + 1) for y in range(stride // 2, H, stride)]
    for y0, y1 in bands:
        _, _, st = O.render_stats(s, c, o, it, y0=y0, y1=y1)
        steps += st.march_steps
        inner += st.inner_iters
        hits += st.hits
        calls += st.sdf_calls
        rows += y1 - y0
    scale = H / rows
    julia = int(w.gui.fractal_group) != 0
    flops = (15.0 * calls + (24.0 if julia else 40.0) * inner + (4.0 * steps if julia else 0.0)
             + (40.0 * w.iters[1] * hits if julia else 0.0)) * scale
    tflops = flops / kernel_s / 1e12
    return {"bound": "valu-f32", "achieved": round(tflops, 3), "peak": VALU_F32_PEAK_TFLOPS,
            "unit": "TFLOP/s", "frac": round(tflops / VALU_F32_PEAK_TFLOPS, 5),
            "flops_per_pixel": round(flops / (W * H), 1),
            "march_steps_per_pixel": round(steps * scale / (W * H), 2),
            "inner_iterations_per_step": round(inner / max(calls, 1), 3),
            "hit_fraction": round(hits * scale / (W * H), 5),
            "sample": "whole frame" if stride == 1 else f"every {stride}th row, scaled"}


def pmc_traffic(workload_key, frames_per_launch=1):
    """HBM bytes per launch from the committed rocprofv3 PMC pass, if one exists (keyed by
    workload, and workload@B for launches of B frames)."""
    p = ROOT / "profiles" / "pmc_traffic.json"
    if frames_per_launch > 1:
        workload_key = f"{workload_key}@{frames_per_launch}"
    try:
        rec = json.loads(p.read_text()).get(workload_key)
        return rec["hbm_bytes_per_launch"] if rec else None
    except (OSError, ValueError, KeyError):
        return None


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    import kifs_raymarching_amd as K
    from kifs_raymarching_amd.bands import BandFrame, FrameStream
    from kifs_raymarching_amd.configs import HEADLINE, WORKLOADS, orbit_camera

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py: --gpus N > 1 must be launched with torch.distributed.run "
                     "(one process per GPU)")
        args.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py: no GPU visible; the render path has no CPU fallback")
    if args.share_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    key = args.workload or HEADLINE
    w = WORKLOADS[key]
    W, H = w.screen.width, w.screen.height
    # F frames in flight: F contexts, each with its own stream, tile tables and output buffer;
    # step k uses context k % F.  Kernel launches, events and the RCCL gather's stream
    # dependencies of a step all sit on that context's dedicated non-default stream.
    F = max(1, args.frames_in_flight)
    gss, streams = [], []
    for _ in range(F):
        g = K.GraphicState(local_rank, screen_data=w.screen, camera_data=w.camera, gui_data=w.gui)
        g.set_iters(*w.iters)
        if w.extensions:
            g.set_extensions(**w.extensions)
        g.set_frames_in_flight(F)
        gss.append(g)
        streams.append(torch.cuda.Stream(device=device))
    gs, stream = gss[0], streams[0]
    torch.cuda.set_stream(stream)
    buffers = F if F >= 2 else 2  # buffer slot k % buffers always belongs to stream k % F
    bands_mode = world > 1 and args.shard == "bands"
    frames_mode = world > 1 and not bands_mode
    B = 1 if bands_mode else max(1, min(args.frames_per_launch, K.MAX_BATCH))
    cur = [0]  # index of the context / stream of the step being enqueued
    if not bands_mode:
        # a step's B frames are stacked in one (B*H, W, 4) buffer
        bf = FrameStream(W, B * H, rank, world, device, buffers=buffers,
                         deliver=(args.deliver == "root"))
        rows0 = H

        def render_band(out, step_index):  # without --orbit every frame is the same view
            g, st = gss[cur[0]], streams[cur[0]]
            first = step_index * B
            cams = [orbit_camera(w, (first + i) % max(w.frames, 120)) if args.orbit else w.camera
                    for i in range(B)]
            if B == 1:
                if args.orbit:
                    g.set_camera(cams[0])
                g.render_async(out, stream=st, y0=0, y1=H, encode=args.encode)
            else:
                g.render_batch_async([out[i * H:(i + 1) * H] for i in range(B)], cams, stream=st,
                                     y0=0, y1=H, encode=args.encode)
    else:
        bf = BandFrame(W, H, rank, world, device, buffers=buffers)
        rows0 = bf.y1 - bf.y0

        orbit_frame = [0]

        def render_band(out, y0, y1):
            if args.orbit:
                gss[cur[0]].set_camera(orbit_camera(w, orbit_frame[0] % max(w.frames, 120)))
                orbit_frame[0] += 1
            gss[cur[0]].render_async(out, stream=streams[cur[0]], y0=y0, y1=y1, encode=args.encode)

    def step(k):
        cur[0] = k % F
        with torch.cuda.stream(streams[cur[0]]):
            bf.step(k, render_band)

    def barrier():
        if world > 1:
            if args.backend == "nccl":
                dist.barrier(device_ids=[local_rank])
            else:
                dist.barrier()

    for k in range(args.warmup):
        step(k)
    bf.wait_all()
    torch.cuda.synchronize()

    # HIP events around the render kernel itself (recorded by the library on the launch stream,
    # one launch in eight): the roofline's "average launch duration"
    for g in gss:
        g.set_profiling(8)  # every 8th launch: an event pair costs a few microseconds
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(args.warmup + k)
    bf.wait_all()
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    dev_ms = elapsed * 1e3  # this rank's wall time between the two synchronisation points
    # which kernel the launches used: the re-queuing throughput kernel or the one-wave-per-block one
    bunny = int(w.gui.fractal_group) == 0 and int(w.gui.primitive_shape) == 5
    kernel_name = ("render_group_kernel" if gss[0].debug_last_round_steps() > 0 else
                   "render_bunny_quad_kernel" if bunny else "render_kernel")
    reads = [g.profile_read() for g in gss]
    timed_launches = sum(r[0] for r in reads)
    kernel_mean_ms = sum(r[0] * r[1] for r in reads) / max(timed_launches, 1)
    kernel_min_ms = min((r[2] for r in reads if r[0] > 0), default=0.0)
    kernel_max_ms = max((r[3] for r in reads if r[0] > 0), default=0.0)
    for g in gss:
        g.set_profiling(0)

    if world > 1:
        red_dev = device if args.backend == "nccl" else torch.device("cpu")
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        tk = torch.tensor([dev_ms], dtype=torch.float64, device=red_dev)
        gathered = [torch.zeros_like(tk) for _ in range(world)]
        dist.all_gather(gathered, tk)
        per_rank_ms = [float(x.item()) / args.steps for x in gathered]
    else:
        per_rank_ms = [dev_ms / args.steps]

    check = None
    if args.check and rank == 0:
        last = args.warmup + args.steps - 1
        ref = torch.zeros((H, W, 4), dtype=torch.uint8, device=device)
        gs.render(out=ref, encode=args.encode)
        if not bands_mode:  # every frame of the step's batch, from every rank that delivered one
            check = all(bool(torch.equal(f[i * H:(i + 1) * H], ref))
                        for f in bf.frames(last) for i in range(B))
        else:
            check = bool(torch.equal(bf.frame(last), ref))

    if rank == 0:
        frames_per_step = (world if frames_mode else 1) * B
        mpix = frames_per_step * W * H * args.steps / elapsed / 1e6
        # average duration of the dominant kernel over the timed region (per-launch event pairs);
        # dev_ms / steps additionally contains the inter-launch gaps
        launch_s = (kernel_mean_ms if timed_launches > 0 else dev_ms / args.steps) / 1e3
        alg_bytes = 4.0 * W * rows0 * B  # 4 B written per pixel, 0 read; B frames per launch
        achieved = alg_bytes / launch_s / 1e9
        out = {
            "metric": METRIC,
            "value": round(mpix, 2),
            "unit": "Mpixels/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 5),
            "higher_is_better": True,
            "scaling": "weak" if (frames_mode or world == 1) else "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": key, "description": w.name, "width": W, "height": H,
                       "max_iterations": w.gui.max_iterations, "sdf_iters": w.iters[0],
                       "normal_iters": w.iters[1], "fold_iters": w.iters[2],
                       "encode": "srgb8" if args.encode else "unorm8",
                       "camera": "orbit, one pose per frame" if args.orbit else "fixed",
                       "frames_per_launch": B, "launches_in_flight": F,
                       "parallelism": (
                           f"1 GPU, {B} frame(s) of the sequence per launch"
                           + (f", {F} launches in flight" if F > 1 else "") if world == 1 else
                           (f"{world} GPUs x whole frames (frame-parallel, {B} per launch), one process per GPU, "
                            + ("finished frames sent to rank 0 by grouped RCCL p2p" if args.deliver == "root"
                               else "frames stay on the GPU that rendered them (no exchange step)")
                            if frames_mode else
                            f"{world} row bands per frame, one process per GPU, RCCL p2p gather to rank 0"))},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6),
                         "traffic": pmc_traffic(key, B) if world == 1 else None,
                         "kernel": kernel_name, "kernel_ms": round(launch_s * 1e3, 5),
                         "kernel_ms_min": round(kernel_min_ms, 5), "kernel_ms_max": round(kernel_max_ms, 5),
                         "launches_timed": timed_launches,
                         "concurrent_launches": F,
                         "algorithmic_bytes_per_launch": int(alg_bytes)},
            "per_rank_kernel_ms": [round(x, 5) for x in per_rank_ms],
        }
        if check is not None:
            out["gathered_frame_equals_single_gpu_frame"] = check
        if world == 1 and args.cpu_seconds > 0:
            out["cpu_baseline"] = cpu_baseline(w, args.cpu_seconds)
            out["roofline"]["secondary"] = work_count(w, launch_s / B)
        print(json.dumps(out), flush=True)

    for g in gss:
        g.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
