#!/usr/bin/env python3
"""bench.py -- headline benchmark of the raymarching hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME]

A "step" is one pass of the render path over one batch of synthetic input: the next
`--frames-per-launch` (default 48) frames of the workload's ORBIT -- one camera pose and one
destination per frame, packed by the host camera model exactly as the reference does per frame
(render.rs:320-345) -- rendered by one kifs_render_batch_async launch.  The default workload is
BASELINE.json's metric configuration: 1920x1080 quaternion-Julia, 256 march steps, 12 SDF
iterations.  Rank 0 prints ONE JSON line; besides the headline it carries, measured in the same
process after the timed region, `secondary.lone_frame` (one frame per launch: the latency path),
`secondary.orbit_x8` (8 frames per launch), `secondary.fixed_camera` (the same view in all 8 slots
of a batch: round 1's headline, the most favourable case for the tile-order feedback) and, for the headline
workload, the north star's other sizes: `secondary.cfg4_julia_4096` (4096 x 4096, 512 / 16; batched and lone) and
`secondary.ref_constants_1080p` (the reference's hard-coded 100 / 10 iterations and GUI-default constant), plus
BASELINE configs 3 and 5: `secondary.cfg3_sierpinski_1080p` and `secondary.cfg5_whole_orbit` (the 120-frame 7680 x 4320
orbit with soft-shadow secondary rays, all frames resident, timed in a child process under a time limit;
`secondary.cfg5_whole_orbit_reference_shading`: the same orbit without the extension).

N > 1 (one process per GPU; `python bench.py --gpus N` launches the N ranks itself, or it runs
under torch.distributed.run): the north star's path.  Every frame is split into ROW SHARDS --
its 8-row stripes dealt to the ranks in turn, so that every rank gets its share of the expensive
middle rows -- each rank renders its stripes of all the step's frames with one launch, sends them
to rank 0 in ONE message (grouped RCCL point-to-point: every sender -> root pair rides its own
xGMI link), and rank 0 moves the received stripes to their rows of the final frames.  Weak
scaling by default: a step has N x frames-per-launch frames, so every GPU keeps rendering
frames-per-launch frames' worth of pixels per step (`--scaling strong` fixes the step at
frames-per-launch frames instead).  `--shard bands` uses contiguous runs of stripes,
`--shard frames` whole frames per rank (no exchange unless `--deliver root`); the latter is also
measured after the timed region and reported as `secondary.frame_parallel`.

`--host one-process` (N > 1): the form the reference's own host would use -- ONE process and one thread drive all N
devices through the C ABI's kifs_multi_render_batch_async (sparse records gathered by RCCL grouped send/recv inside
the library, two steps in flight); the JSON line has the same shape.  `--whole-orbit`: a step is the workload's whole
orbit (120 frames for cfg5) in launches of --frames-per-launch frames, every frame resident in HBM.

There are no HBM-resident inputs beyond the 156 uniform bytes; the output frames live in HBM
(torch tensors) and are written by the kernel.  `roofline` prices the dominant kernel against
the HBM-write roofline the north star mandates (4 algorithmic bytes per pixel); `cpu_baseline`
times the CPU oracle (the stand-in for the reference's wgpu path, which cannot run here) on the
host cores -- a reported baseline, not the target.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
VALU_F32_PEAK_TFLOPS = 157.3  # MI355X vector f32 (same guide); the path has no MFMA work
METRIC = "Mpixels/s at 1920x1080, 256 march steps, 12 SDF iters; %HBM-peak"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--workload", default=None, help="name in kifs_raymarching_amd.configs.WORKLOADS")
    ap.add_argument("--cpu-seconds", type=float, default=10.0,
                    help="wall-clock budget of the cpu_baseline sample (0 disables it)")
    ap.add_argument("--encode", type=int, default=1, help="1 = sRGB target (reference default), 0 = UNORM")
    ap.add_argument("--camera", default="orbit", choices=["orbit", "fixed"],
                    help="orbit (default): a distinct pose per frame, phi = 2*pi*frame/120 (host camera model + "
                         "64-byte uniform per frame); fixed: every frame is the workload's view")
    ap.add_argument("--orbit", action="store_true", help="same as --camera orbit (kept for older scripts)")
    ap.add_argument("--frames-per-launch", type=int, default=48,
                    help="frames of the sequence per launch (1..64, default 48); 1 = the lone-frame latency path")
    ap.add_argument("--frames-in-flight", type=int, default=1,
                    help="N = 1 only: launches kept in flight on separate contexts and streams")
    ap.add_argument("--shard", default="stripes", choices=["stripes", "bands", "frames"],
                    help="N > 1: interleaved 8-row stripes of every frame gathered to rank 0 (default), "
                         "contiguous row bands gathered to rank 0, or whole frames per rank")
    ap.add_argument("--gather", default="sparse", choices=["sparse", "dense"],
                    help="N > 1, row shards: peers send only the 32 x 8 tiles that hold something and rank 0 fills in "
                         "the background (default), or every row as it is")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="N > 1, row shards: frames per step = N x frames-per-launch (weak) or frames-per-launch")
    ap.add_argument("--root-weight", default="auto",
                    help="N > 1, stripes: W or W:P -- rank 0 renders W stripes for every P (default 1) of a peer.  "
                         "Rank 0 also takes every peer's pixels in (each over ONE point-to-point xGMI link) and writes "
                         "the background under their rows: 'auto' (default) tries a few shares on the real pipeline "
                         "before the warm-up steps and keeps the fastest")
    ap.add_argument("--settle-ms", type=int, default=60,
                    help="untimed rendering before the warm-up steps, about this many milliseconds of it (clocks and "
                         "tile order in steady state; 0 = none)")
    ap.add_argument("--deliver", default="none", choices=["none", "root"],
                    help="N > 1, --shard frames: leave frames where they were rendered, or send them to rank 0")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend; gloo is only for rehearsing N > 1 on one GPU")
    ap.add_argument("--share-device", action="store_true",
                    help="rehearsal: every rank uses cuda:0 (needs --backend gloo)")
    ap.add_argument("--check", action="store_true",
                    help="after the timed region, compare rank 0's gathered frames with single-GPU renders")
    ap.add_argument("--no-check", action="store_true",
                    help="N > 1: skip the default comparison of the last step's gathered frames with single-GPU renders")
    ap.add_argument("--host", default="per-gpu", choices=["per-gpu", "one-process"],
                    help="N > 1: one process per GPU over torch.distributed (default, the driver's launch), or ONE process "
                         "driving all N devices through kifs_multi_render_batch_async (RCCL inside the library)")
    ap.add_argument("--transport", default="auto", choices=["auto", "rccl", "copy"],
                    help="--host one-process: the library's transport (auto = RCCL on distinct devices, peer copies otherwise)")
    ap.add_argument("--one-process-secondary", default="auto", choices=["auto", "off"],
                    help="N > 1, per-GPU host, row shards: once every collective of this job is done, rank 0 also runs the "
                         "ONE-process form (kifs_multi_render_batch_async: RCCL inside the library) in a child process with "
                         "a time limit and reports it as secondary.one_process_c_abi; 'off' skips it")
    ap.add_argument("--whole-orbit", action="store_true",
                    help="N = 1: a step is the workload's WHOLE orbit (max(frames, 120) poses, every frame resident in HBM) "
                         "in launches of --frames-per-launch frames")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary measurements")
    ap.add_argument("--calibration-note", default=None, help=argparse.SUPPRESS)  # set by self_launch()'s second attempt
    ap.add_argument("--dist-at-one", action="store_true",
                    help="--gpus 1 only: run the N > 1 code path with a world of one rank -- process group (backend as given), "
                         "its barrier / all_reduce / all_gather, the CPU side group, the sharded pipeline with no peers -- the "
                         "half of a multi-GPU run that a one-GPU box can prove (tests/test_gpu_bench.py)")
    ap.add_argument("--cfg5-secondary", default="auto", choices=["auto", "off"],
                    help="headline workload at N = 1: 'auto' also times BASELINE config 5 as it is named -- the whole 120-frame "
                         "7680x4320 Sierpinski orbit, every frame resident (15.9 GB) -- in a child process under a time limit "
                         "(secondary.cfg5_whole_orbit); 'off' skips it")
    return ap.parse_args()


# ---- stages: where a run is, said on stderr by every rank, with a deadline ------------------------
# The first multi-GPU run is also the first contact of this code with RCCL between two devices: a rank that
# stalls in a collective must cost a clear message and a non-zero exit within minutes, not the driver's whole
# time limit.  Every rank announces each stage it enters (one stderr line: rank, stage, deadline) and a
# watchdog thread in the rank gives up -- os._exit(3), after saying where -- when a stage outlives its
# deadline; under torch.distributed.run that ends the job, under self_launch() the parent (which follows the
# same lines through the children's stderr) stops the other ranks and reports every rank's last stage.
_T0 = time.time()
_STAGE = {"name": "start", "since": _T0, "deadline": 600.0}
STAGE_DEADLINES = (  # first prefix that matches; seconds
    ("import", 420.0),            # the first `import torch` of a fresh box pages the image in: 1-2 minutes
    ("init process group", 300.0),
    ("gloo side group", 120.0),
    ("calibration", 150.0),       # per candidate: ten steps of the real pipeline
    ("settle", 180.0), ("warmup", 180.0), ("check", 300.0), ("reduce", 120.0),
    ("one-process child", 220.0), ("cfg5", 200.0), ("cpu baseline", 180.0), ("secondary", 240.0),
)


def _deadline_scale():
    try:
        return max(0.01, float(os.environ.get("KIFS_BENCH_DEADLINE_SCALE", "1")))
    except ValueError:
        return 1.0


def stage(name, deadline_s=None):
    """Announces the stage this rank enters.  KIFS_BENCH_STALL=<stage prefix>:<rank or *> makes that rank sleep
    there (the rehearsal of a hung collective: tests/test_bench_stages.py, tools/rehearse_multi.sh)."""
    if deadline_s is None:
        deadline_s = next((d for p, d in STAGE_DEADLINES if name.startswith(p)), 240.0)
    deadline_s *= _deadline_scale()
    now = time.time()
    _STAGE.update(name=name, since=now, deadline=deadline_s)
    rank = os.environ.get("RANK", "0")
    print(f"bench.py[rank {rank}] stage: {name} | deadline {deadline_s:.0f} s | t=+{now - _T0:.1f} s", file=sys.stderr, flush=True)
    stall = os.environ.get("KIFS_BENCH_STALL", "")
    if stall and ":" in stall:
        prefix, who = stall.rsplit(":", 1)
        if name.startswith(prefix) and who in ("*", rank):
            time.sleep(1e6)


def start_watchdog():
    import threading
    if os.environ.get("KIFS_BENCH_NO_RANK_WATCHDOG") == "1":  # tests of self_launch()'s own backstop
        return

    def watch():
        while True:
            time.sleep(0.25)
            st = dict(_STAGE)
            if st["name"] == "done":
                return
            waited = time.time() - st["since"]
            if waited > st["deadline"]:
                rank = os.environ.get("RANK", "0")
                print(f"bench.py[rank {rank}] STALLED in stage '{st['name']}' for {waited:.0f} s (deadline "
                      f"{st['deadline']:.0f} s): giving up", file=sys.stderr, flush=True)
                os._exit(3)
    threading.Thread(target=watch, daemon=True, name="bench-stage-watchdog").start()


# ---- N > 1 without a launcher: start the ranks ------------------------------------------------
def self_launch(args, attempt=0, extra=()):
    """Starts one fresh child process per GPU and relays rank 0's JSON line.  Runs BEFORE anything
    in this process touches the GPU (no torch import, no HIP call): a process that has initialised
    the GPU must never be replaced or forked into ranks.  The children's stderr passes through this process, which
    follows their stage lines: a rank that exits non-zero, or a stage that outlives its deadline by more than the
    grace the rank's own watchdog gets, stops exactly the children started here; the last stage of every rank is
    printed and the exit code is non-zero.  The ranks are started a second time in ONE case only: when what failed was the
    calibration of rank 0's share (see the end of this function) -- fresh children at --root-weight 1:1, once."""
    import tempfile
    import threading
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    last = {}  # rank -> (stage, entered at, deadline)
    lock = threading.Lock()

    def follow(r, pipe):
        for raw in iter(pipe.readline, b""):
            line = raw.decode("utf-8", "replace")
            sys.stderr.write(line)
            sys.stderr.flush()
            tag = f"bench.py[rank {r}] stage: "
            if line.startswith(tag):
                parts = line[len(tag):].split(" | ")
                try:
                    dl = float(parts[1].split()[1])
                except (IndexError, ValueError):
                    dl = 240.0
                with lock:
                    last[r] = (parts[0].strip(), time.time(), dl)
        pipe.close()

    def report(why):
        with lock:
            where = {r: f"'{v[0]}' for {time.time() - v[1]:.0f} s (deadline {v[2]:.0f} s)" for r, v in sorted(last.items())}
        print(f"bench.py: {why}; last stage of every rank: " + json.dumps(where), file=sys.stderr, flush=True)

    with tempfile.TemporaryFile("w+") as out0:
        readers = []
        for r in range(args.gpus):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus),
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
            p = subprocess.Popen([sys.executable, str(Path(__file__).resolve())] + sys.argv[1:] + list(extra), env=env,
                                 stdout=out0 if r == 0 else subprocess.DEVNULL, stderr=subprocess.PIPE)
            procs.append(p)
            last[r] = ("(not started)", time.time(), 420.0 * _deadline_scale())
            t = threading.Thread(target=follow, args=(r, p.stderr), daemon=True)
            t.start()
            readers.append(t)

        def stop_children():
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            t_end = time.time() + 10
            for p in procs:
                try:
                    p.wait(timeout=max(0.1, t_end - time.time()))
                except subprocess.TimeoutExpired:
                    p.kill()
                    p.wait()

        def exited_badly():
            with lock:
                where = {r: last[r][0] for r, p in enumerate(procs) if p.returncode not in (None, 0)}
            return (f"rank(s) {sorted(where)} exited with code(s) {[procs[r].returncode for r in sorted(where)]} in stage(s) "
                    + ", ".join(f"'{where[r]}'" for r in sorted(where)))

        failed = None
        # the rank's own watchdog fires first and says where; this is the backstop for a wedged interpreter
        grace = max(1.0, 20.0 * min(1.0, _deadline_scale()))
        while any(p.poll() is None for p in procs):
            if any(p.returncode not in (None, 0) for p in procs):
                # a rank that dies leaves the others waiting in a collective: stop them (exactly the
                # children started above) instead of waiting for a communicator time-out
                failed = exited_badly()
                stop_children()
                break
            now = time.time()
            with lock:
                late = [(r, v) for r, v in last.items() if procs[r].poll() is None and now - v[1] > v[2] + grace]
            if late:
                r, v = late[0]
                failed = f"rank {r} stalled in stage '{v[0]}' for {now - v[1]:.0f} s (deadline {v[2]:.0f} s)"
                stop_children()
                break
            time.sleep(0.05)
        for t in readers:
            t.join(timeout=5)
        if failed is None and any(p.returncode != 0 for p in procs):
            failed = exited_badly()
        out0.seek(0)
        sys.stdout.write(out0.read())
        sys.stdout.flush()
    if failed:
        report(failed)
        with lock:
            stalled_in_calibration = any(v[0].startswith("calibration") for v in last.values())
        if stalled_in_calibration and attempt == 0 and args.root_weight == "auto":
            # The trial of rank 0's share is an optimisation, not the measurement: if it is what failed, the job runs once
            # more WITHOUT it -- fresh children of this process, which has never touched a GPU -- at 1:1, and the line says
            # so (config.root_weight_calibration).  Anything that fails there fails the run.
            print("bench.py: the calibration of rank 0's share failed; starting the ranks once more with --root-weight 1:1",
                  file=sys.stderr, flush=True)
            return self_launch(args, attempt=1, extra=["--root-weight", "1:1", "--calibration-note",
                                                       "first attempt: " + failed])
        sys.exit(f"bench.py: {failed} (exit codes {[p.returncode for p in procs]})")


# ---- CPU-side figures ----------------------------------------------------------------------------
def _oracle_uniforms(w, camera=None):
    import oracle as O
    import kifs_raymarching_amd as K

    ub = K.uniform_bytes
    return (O.from_bytes(O.Screen, ub(w.screen.into_buffer_data())),
            O.from_bytes(O.Camera, ub((camera or w.camera).into_buffer_data())),
            O.from_bytes(O.Options, ub(w.gui.into_buffer_data())), O.iters(*w.iters))


def cpu_baseline(w, seconds):
    """Times the CPU oracle on whole frames of the same workload for ~`seconds`."""
    import oracle as O

    s, c, o, it = _oracle_uniforms(w)
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


def work_count(w, kernel_s):
    """Secondary figure of SURVEY 8(d): algorithmic flops of one frame, counted by the
    instrumented CPU oracle (march steps, Julia iterations / Sierpinski folds, hits), against
    the f32 vector peak.  Flop model: 15 per march step, 24 per Julia
    iteration or 40 per fold, 4 for the log/sqrt/divide tail of a Julia step, and per hit
    40 * normal_iters (Julia Jacobian) -- the six extra SDF calls of a KIFS normal are already
    in the step and fold counts."""
    import oracle as O

    s, c, o, it = _oracle_uniforms(w)
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
            "sample": ("whole frame" if stride == 1 else f"every {stride}th row, scaled") + ", the workload's own view"}


def pmc_traffic(workload_key, frames_per_launch=1):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (counters cannot be read from inside the process),
    keyed by workload and workload@B: (bytes or None, source, record or None, note or None).  Every entry carries the
    kernel hash of the library it was measured on (kifs_raymarching_amd/build.py: kernel_hash()); an entry that belongs
    to other kernels than the ones loaded is NOT reported -- new kernels must not carry old counters."""
    import kifs_raymarching_amd as K
    p = ROOT / "profiles" / "pmc_traffic.json"
    if frames_per_launch > 1:
        workload_key = f"{workload_key}@{frames_per_launch}"
    try:
        rec = json.loads(p.read_text()).get(workload_key)
    except (OSError, ValueError):
        return None, None, None, None
    if not rec:
        return None, "profiles/pmc_traffic.json", None, f"no counter pass recorded for {workload_key}"
    loaded = K._lib.kernel_hash_of_loaded_library()
    if not loaded or rec.get("kernel_hash") != loaded:
        return (None, "profiles/pmc_traffic.json", None,
                f"the recorded counter pass ({rec.get('round', '?')}) belongs to other kernels than the loaded library's "
                f"(kernel hash {str(rec.get('kernel_hash'))[:12]} != {loaded[:12] or 'unstamped'}): re-run tools/profile_round.sh")
    return rec.get("hbm_bytes_per_launch"), "profiles/pmc_traffic.json", rec, None


def lone_frame_floor(workload_key, camera_mode):
    """The committed floor under a lone frame (tools/lone_frame_floor.py: the critical ray's instructions, counted from
    the source of the long-ray loop and the instrumented oracle, x a lone wave's issue interval), if it belongs to the
    loaded kernels: (floor_ms or None, note)."""
    import kifs_raymarching_amd as K
    try:
        rec = json.loads((ROOT / "profiles" / "lone_frame_floor.json").read_text()).get(workload_key)
    except (OSError, ValueError):
        rec = None
    if not rec:
        return None, "no floor recorded for this workload (tools/lone_frame_floor.py)"
    loaded = K._lib.kernel_hash_of_loaded_library()
    if not loaded or rec.get("kernel_hash") != loaded:
        return None, "the recorded floor was counted from other kernel sources than the loaded library's: re-run tools/lone_frame_floor.py"
    return rec["floor_ms_orbit_mean" if camera_mode == "orbit" else "floor_ms_fixed_view"], (
        f"critical ray's instructions ({rec['instructions_per_step']['inside_fixed']} per march step inside the bounding "
        f"sphere + {rec['instructions_per_step']['per_orbit_trip']} per orbit trip, counted from kifs_julia_march_asm.hpp; the "
        f"ray from the instrumented oracle, {'mean over sampled orbit poses' if camera_mode == 'orbit' else 'the fixed view'}) x "
        f"{rec['issue_cycles']} cycles per instruction of a lone wave (tools/microbench/issue_cost) at {rec['clock_hz'] / 1e9:.1f} GHz; "
        "taken branches, wait states, set-up, shading and the store are not in it")


def measure_workload(name, local_rank, B, steps, warmup, encode, camera_mode="orbit", settle_ms=30.0):
    """One more workload in the same process, after the timed region: its own context, stream and buffers; `steps`
    launches of B frames of its orbit; wall clock around the loop and the library's per-launch event pairs."""
    import torch

    import kifs_raymarching_amd as K
    from kifs_raymarching_amd.configs import WORKLOADS, orbit_camera

    w = WORKLOADS[name]
    W, H = w.screen.width, w.screen.height
    device = torch.device("cuda", local_rank)
    g = K.GraphicState(local_rank, screen_data=w.screen, camera_data=w.camera, gui_data=w.gui)
    g.set_iters(*w.iters)
    if w.extensions:
        g.set_extensions(**w.extensions)
    stream = torch.cuda.Stream(device=device)
    bufs = [torch.zeros((B * H, W, 4), dtype=torch.uint8, device=device) for _ in range(2)]
    n_poses = max(w.frames, 120)
    poses = ([orbit_camera(w, i).into_buffer_data() for i in range(n_poses)] if camera_mode == "orbit"
             else [w.camera.into_buffer_data()])
    arrays = {}

    def cams(first):
        key_ = first % len(poses)
        if key_ not in arrays:
            arrays[key_] = K.camera_array([poses[(first + i) % len(poses)] for i in range(B)])
        return arrays[key_]

    def step(k):
        out = bufs[k % 2]
        with torch.cuda.stream(stream):
            if B == 1:
                g.set_raw_uniforms(camera=cams(k)[0])
                g.render_async(out, stream=stream, y0=0, y1=H, encode=encode)
            else:
                g.render_batch_async([out[i * H:(i + 1) * H] for i in range(B)], cams(k * B), stream=stream,
                                     y0=0, y1=H, encode=encode)
    t_settle = time.perf_counter()
    k = 0
    while k < warmup or (time.perf_counter() - t_settle) * 1e3 < settle_ms:
        step(k)
        k += 1
        if k % 8 == 0:
            stream.synchronize()
    stream.synchronize()
    g.set_profiling(2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for j in range(steps):
        step(k + j)
    stream.synchronize()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    n, mean_ms, lo, hi = g.profile_read()
    g.set_profiling(0)
    shape = g.debug_last_kernel()
    g.close()
    del bufs
    torch.cuda.empty_cache()
    ms = elapsed / steps * 1e3
    k_ms = mean_ms if n else ms
    return {"workload": name, "width": W, "height": H, "frames_per_launch": B, "steps": steps,
            "ms_per_step": round(ms, 5), "ms_per_frame": round(ms / B, 5),
            "mpix_s": round(B * W * H / (ms * 1e-3) / 1e6, 2), "kernel": shape, "kernel_ms": round(k_ms, 5),
            "hbm_frac": round(B * 4.0 * W * H / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 6)}


def child_line(flags, limit_s, pick):
    """Runs `bench.py <flags>` as a child under a time limit and returns pick(its JSON line), or {"error": ...}: a child
    that fails or hangs costs a field of the line, not the line."""
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK", "ROLE_RANK",
                        "LOCAL_WORLD_SIZE", "ROLE_WORLD_SIZE", "TORCHELASTIC_RUN_ID")}
    try:
        cp = subprocess.run([sys.executable, str(Path(__file__).resolve())] + flags, env=env, capture_output=True, text=True,
                            timeout=limit_s)
        lines = [l for l in cp.stdout.splitlines() if l.startswith("{")]
        if cp.returncode == 0 and len(lines) == 1:
            return pick(json.loads(lines[0]))
        return {"error": f"exit code {cp.returncode}", "stderr_tail": cp.stderr[-400:]}
    except subprocess.TimeoutExpired:
        return {"error": f"no result within {limit_s} s (the child was stopped)"}
    except Exception as e:  # the main line must not depend on this
        return {"error": repr(e)[:300]}


def one_process(args):
    """--host one-process: this process alone drives all N devices through kifs_multi_render_batch_async -- what the
    reference's single-process host (application.rs:37-48) would call.  Same protocol and JSON shape as the
    per-GPU form: settle, W warm-up steps, exactly K timed steps between synchronisations, frames gathered on device 0."""
    import torch

    import kifs_raymarching_amd as K
    from kifs_raymarching_amd.configs import HEADLINE, WORKLOADS, orbit_camera

    if not torch.cuda.is_available():
        sys.exit("bench.py: no GPU visible; the render path has no CPU fallback")
    N = args.gpus
    stage("one-process: contexts on every device")
    have = torch.cuda.device_count()
    devices = [0] * N if args.share_device else list(range(N))
    if not args.share_device and have < N:
        sys.exit(f"bench.py: --host one-process --gpus {N} needs {N} devices, {have} visible (--share-device rehearses on one)")
    key = args.workload or HEADLINE
    w = WORKLOADS[key]
    W, H = w.screen.width, w.screen.height
    B = max(1, min(args.frames_per_launch, K.MAX_BATCH))
    fps = B * N if args.scaling == "weak" else B
    fps = max(1, min(fps, K.MAX_BATCH, int(24e9 // (2 * 4 * W * H))))
    torch.cuda.set_device(0)
    dev0 = torch.device("cuda", 0)
    mg = K.MultiGraphicState(devices, w.screen, w.camera, w.gui, iters=w.iters)
    if w.extensions:
        mg.set_extensions(**w.extensions)
    mg.set_gather(args.gather, args.transport)
    weights = None
    if args.root_weight not in ("auto", "1", "1:1"):
        parts = [max(1, int(x)) for x in str(args.root_weight).split(":")]
        weights = [parts[0]] + [parts[1] if len(parts) > 1 else 1] * (N - 1)
        mg.set_weights(weights)
    frames = [torch.zeros((fps, H, W, 4), dtype=torch.uint8, device=dev0) for _ in range(2)]
    n_poses = max(w.frames, 120)
    poses = ([orbit_camera(w, i).into_buffer_data() for i in range(n_poses)] if args.camera == "orbit"
             else [w.camera.into_buffer_data()])
    arrays = {}

    def cams(first):
        key_ = first % len(poses)
        if key_ not in arrays:
            if len(arrays) > 64:
                arrays.clear()
            arrays[key_] = K.camera_array([poses[(first + i) % len(poses)] for i in range(fps)])
        return arrays[key_]

    def step(k):
        return mg.render_batch_async(frames[k % 2], cams(k * fps), encode=args.encode, untouched=k >= 2)

    step_ms = max(0.05, fps * W * H / N / 60.0e6)
    settle = min(400, int(-(-max(0, args.settle_ms) // step_ms)))
    settle += settle % 2
    stage(f"one-process: first step (transport set-up: ncclCommInitAll over {len(set(devices))} device(s))")
    step(0)
    mg.wait_all()
    seen = mg.stats()
    if seen["transport"] == "rccl" and seen["comm_ranks"] != N:
        sys.exit(f"bench.py: the library's communicator has {seen['comm_ranks']} ranks, --gpus says {N}")
    stage(f"one-process: settle + warmup ({settle + args.warmup} steps)")
    for k in range(1, settle + args.warmup):
        step(k)
    mg.wait_all()
    for d in set(devices):
        torch.cuda.synchronize(d)
    mg.stats(reset=True)
    base = settle + args.warmup
    stage(f"one-process: timed ({args.steps} steps)", deadline_s=120.0 + 0.5 * args.steps)
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(base + k)
    mg.wait_all()
    for d in set(devices):
        torch.cuda.synchronize(d)
    elapsed = time.perf_counter() - t0
    stats = mg.stats()
    shards = mg.shards()
    check = None
    if not args.no_check:
        stage("one-process: check")
        last = base + args.steps - 1
        got = frames[last % 2]
        ref = torch.zeros((H, W, 4), dtype=torch.uint8, device=dev0)
        with K.GraphicState(0, screen_data=w.screen, camera_data=w.camera, gui_data=w.gui) as g1:
            g1.set_iters(*w.iters)
            if w.extensions:
                g1.set_extensions(**w.extensions)
            arr = cams(last * fps)
            check = True
            for i in range(fps):
                g1.set_raw_uniforms(camera=arr[i])
                g1.render(out=ref, encode=args.encode)
                if not bool(torch.equal(got[i], ref)):
                    check = False
                    break
    mpix = fps * W * H * args.steps / elapsed / 1e6
    rows0 = shards[0][2]
    k_ms = shards[0][3] if shards[0][3] > 0 else elapsed / args.steps * 1e3
    alg_bytes = 4.0 * W * rows0 * fps
    achieved = alg_bytes / (k_ms * 1e-3) / 1e9
    out = {
        "metric": METRIC, "value": round(mpix, 2), "unit": "Mpixels/s", "n_gpus": N, "steps": args.steps,
        "warmup": args.warmup, "settle_steps": settle, "ms_per_step": round(elapsed / args.steps * 1e3, 5),
        "higher_is_better": True, "scaling": "strong" if args.scaling == "strong" else "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": key, "description": w.name, "width": W, "height": H,
                   "max_iterations": w.gui.max_iterations, "sdf_iters": w.iters[0], "normal_iters": w.iters[1],
                   "fold_iters": w.iters[2], "encode": "srgb8" if args.encode else "unorm8",
                   "camera": "orbit, one pose per frame" if args.camera == "orbit" else "fixed",
                   "frames_per_step": fps, "frames_per_launch": fps, "launches_in_flight": 1,
                   "host": "one process, one thread, kifs_multi_render_batch_async (C ABI)",
                   "parallelism": (f"{N} devices {devices} driven by ONE process: row shards (8-row stripes dealt round-robin"
                                   + (f", weights {weights}" if weights else "") + f") of {fps} frames per step, one launch "
                                   "per device, gathered on device 0 " +
                                   ("as sparse records of the 32x8 tiles that hold something" if args.gather == "sparse" else "dense")
                                   + f" by {'RCCL grouped ncclSend/ncclRecv inside the library' if stats['transport'] == 'rccl' else 'peer copies (hipMemcpyPeerAsync)'}"),
                   "settle_steps_before_warmup": settle, "gather": args.gather,
                   "rows_per_rank": [sh[2] for sh in shards]},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": None, "traffic_source": None,
                     "kernel": "render_wave_kernel", "kernel_ms": round(k_ms, 5), "launches_timed": 1,
                     "algorithmic_bytes_per_launch": int(alg_bytes),
                     "note": "device 0's launch of the last step (event pair around its render kernel)"},
        "per_rank_kernel_ms": [round(sh[3], 5) for sh in shards],
        "comm": {"backend": "rccl (in-library, ncclCommInitAll)" if stats["transport"] == "rccl" else "hip peer copies",
                 "world_size_seen": stats["comm_ranks"] if stats["transport"] == "rccl" else len(set(devices)),
                 "rccl_version": stats["rccl_version"], "gather": stats["gather"], "check": check,
                 "bytes_into_root_per_step": int(stats["bytes_received"] / max(1, stats["steps"])),
                 "tiles_sent_fraction": (round(stats["records_received"] / stats["tiles_covered"], 4)
                                         if stats["tiles_covered"] else None)},
    }
    if check is not None:
        out["gathered_frame_equals_single_gpu_frame"] = check
    print(json.dumps(out), flush=True)
    mg.close()
    if check is False:
        sys.exit("bench.py: gathered frames differ from single-GPU frames")


def main():
    args = parse()
    if args.orbit:
        args.camera = "orbit"
    if args.gpus > 1 and args.host == "one-process":
        # one process for the whole node: under a launcher every rank but the first has nothing to do
        if int(os.environ.get("RANK", "0")) == 0:
            start_watchdog()
            one_process(args)
            _STAGE["name"] = "done"
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args)

    start_watchdog()
    stage("import")
    import torch
    import torch.distributed as dist

    import kifs_raymarching_amd as K
    from kifs_raymarching_amd.bands import FrameStream, ShardFrames, SparseShardFrames
    from kifs_raymarching_amd.configs import HEADLINE, WORKLOADS, orbit_camera

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        # the driver's command names --gpus N and launches N ranks: anything else is a launch that lost ranks (or a
        # stale environment), and a line computed from it would be quoted for the wrong N
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: refusing to report a line for the wrong number of GPUs")
    if not torch.cuda.is_available():
        sys.exit("bench.py: no GPU visible; the render path has no CPU fallback")
    if args.share_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist_on = world > 1 or args.dist_at_one
    if dist_on:
        stage(f"init process group ({args.backend})")
        if "MASTER_ADDR" not in os.environ:  # --dist-at-one without a launcher
            with socket.socket() as s_:
                s_.bind(("127.0.0.1", 0))
                os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(s_.getsockname()[1]))
        # One node, rendezvous on the loopback address: gloo (the rehearsal backend, and the CPU side group of the RCCL
        # run) would otherwise pick its interface by resolving the hostname, which a container need not be able to do.
        if os.environ.get("MASTER_ADDR") in ("127.0.0.1", "localhost") and os.path.exists("/sys/class/net/lo"):
            os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        if dist.get_world_size() != args.gpus:
            sys.exit(f"bench.py: the process group has {dist.get_world_size()} ranks, --gpus says {args.gpus}")

    key = args.workload or HEADLINE
    w = WORKLOADS[key]
    W, H = w.screen.width, w.screen.height
    orbit_len = max(w.frames, 120)
    # F launches in flight (N = 1): F contexts, each with its own stream, tile tables and output
    # buffers; step k uses context k % F.  Launches, events and the RCCL stream dependencies of a
    # step all sit on that context's dedicated non-default stream.
    F = max(1, args.frames_in_flight) if world == 1 else 1
    gss, streams = [], []
    for _ in range(F):
        g = K.GraphicState(local_rank, screen_data=w.screen, camera_data=w.camera, gui_data=w.gui)
        g.set_iters(*w.iters)
        if w.extensions:
            g.set_extensions(**w.extensions)
        g.set_frames_in_flight(F)
        gss.append(g)
        streams.append(torch.cuda.Stream(device=device))
    gs = gss[0]
    torch.cuda.set_stream(streams[0])
    buffers = F if F >= 2 else 2  # buffer slot k % buffers always belongs to stream k % F
    B = max(1, min(args.frames_per_launch, K.MAX_BATCH))

    # The orbit's poses as uniform images, made once; a step's camera array is cached by where in the orbit it
    # starts (a step of a few hundred views would otherwise spend a GPU step's time preparing them on the host).
    pose_images = {"orbit": [orbit_camera(w, i).into_buffer_data() for i in range(orbit_len)],
                   "fixed": [w.camera.into_buffer_data()]}
    camera_arrays = {}

    def cameras(first, n, mode):
        poses = pose_images["orbit" if mode == "orbit" else "fixed"]
        key_ = (mode, first % len(poses), n)
        if key_ not in camera_arrays:
            if len(camera_arrays) > 64:
                camera_arrays.clear()
            camera_arrays[key_] = K.camera_array([poses[(first + i) % len(poses)] for i in range(n)])
        return camera_arrays[key_]

    class Scene:
        """What a pipeline renders: a workload, its context on this rank's device and its orbit's camera images."""

        def __init__(self, key_, w_, g_, cameras_):
            self.key, self.w, self.gs, self.cameras = key_, w_, g_, cameras_
            self.W, self.H = w_.screen.width, w_.screen.height

    main_scene = Scene(key, w, gs, cameras)

    def other_scene(key_):
        """A second workload beside the headline's (its own context; the same streams)."""
        w_ = WORKLOADS[key_]
        g_ = K.GraphicState(local_rank, screen_data=w_.screen, camera_data=w_.camera, gui_data=w_.gui)
        g_.set_iters(*w_.iters)
        if w_.extensions:
            g_.set_extensions(**w_.extensions)
        images = [orbit_camera(w_, i).into_buffer_data() for i in range(max(w_.frames, 120))]
        cache = {}

        def cams(first, n, mode):
            k_ = (first % len(images), n)
            if k_ not in cache:
                if len(cache) > 16:
                    cache.clear()
                cache[k_] = K.camera_array([images[(first + i) % len(images)] for i in range(n)])
            return cache[k_]
        return Scene(key_, w_, g_, cams)

    def barrier():
        if dist_on:
            dist.barrier(device_ids=[local_rank]) if args.backend == "nccl" else dist.barrier()

    def max_over_ranks(x):
        """MAX of a host float over the ranks (on the device for RCCL, on the CPU for gloo)."""
        if not dist_on:
            return float(x)
        t = torch.tensor([x], dtype=torch.float64, device=device if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def run(pipeline, step, steps, warmup, announce=False, scene=None):
        """warmup + timed loop of `step(k)`; returns wall seconds and the per-launch kernel times
        (HIP event pairs recorded by the library on the launch stream around the render kernel)."""
        gss_ = [scene.gs] if scene else gss
        if announce:
            stage(f"warmup ({warmup} steps)")
        for k in range(warmup):
            step(k)
        pipeline.wait_all()
        torch.cuda.synchronize()
        for g in gss_:
            # every 4th launch: an event pair costs a few microseconds (short runs: every 2nd)
            g.set_profiling(4 if steps >= 40 else 2)
        if announce:
            stage(f"timed ({steps} steps)", deadline_s=120.0 + 0.5 * steps)
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(steps):
            step(warmup + k)
        pipeline.wait_all()
        torch.cuda.synchronize()
        barrier()
        elapsed = time.perf_counter() - t0
        reads = [g.profile_read() for g in gss_]
        for g in gss_:
            g.set_profiling(0)
        n = sum(r[0] for r in reads)
        return {"elapsed": elapsed, "launches_timed": n,
                "kernel_ms": sum(r[0] * r[1] for r in reads) / max(n, 1),
                "kernel_ms_min": min((r[2] for r in reads if r[0] > 0), default=0.0),
                "kernel_ms_max": max((r[3] for r in reads if r[0] > 0), default=0.0),
                "round_steps": gss_[0].debug_last_round_steps(), "group_tiles": gss_[0].debug_last_group_tiles(),
                "kernel": gss_[0].debug_last_kernel()}

    def whole_frames(frames_per_launch, camera_mode, deliver=False):
        """Every rank renders whole frames: step k, rank r: frames (k N + r) b .. + b - 1 (at N = 1
        simply the next b frames), stacked in one (b*H, W, 4) buffer."""
        b = frames_per_launch
        fs = FrameStream(W, b * H, rank, world, device, buffers=buffers, deliver=deliver)

        cur = [0]

        def render(out, step_index):
            g, st = gss[cur[0]], streams[cur[0]]
            cams = cameras(step_index * b, b, camera_mode)
            if b == 1:
                g.set_raw_uniforms(camera=cams[0])  # the reference's per-frame uniform upload (render.rs:320-321)
                g.render_async(out, stream=st, y0=0, y1=H, encode=args.encode)
            else:
                g.render_batch_async([out[i * H:(i + 1) * H] for i in range(b)], cams, stream=st,
                                     y0=0, y1=H, encode=args.encode)

        def step(k):
            cur[0] = k % F
            with torch.cuda.stream(streams[cur[0]]):
                fs.step(k, render)
        return fs, step

    def whole_orbit(frames_per_launch):
        """A step = every pose of the workload's orbit once (BASELINE cfg 5: the 120-frame orbit), each frame in its
        own resident buffer, in launches of `frames_per_launch` frames on one stream."""
        b = frames_per_launch
        buf = torch.zeros((orbit_len * H, W, 4), dtype=torch.uint8, device=device)

        class Resident:
            frames = buf

            @staticmethod
            def wait_all():
                pass

        def step(k):
            with torch.cuda.stream(streams[0]):
                for a in range(0, orbit_len, b):
                    m_ = min(b, orbit_len - a)
                    if m_ == 1:
                        gs.set_raw_uniforms(camera=cameras(a, 1, "orbit")[0])
                        gs.render_async(buf[a * H:(a + 1) * H], stream=streams[0], y0=0, y1=H, encode=args.encode)
                    else:
                        gs.render_batch_async([buf[(a + i) * H:(a + i + 1) * H] for i in range(m_)],
                                              cameras(a, m_, "orbit"), stream=streams[0], y0=0, y1=H, encode=args.encode)
        return Resident, step

    def row_shards(frames_per_step, camera_mode, contiguous, root_weight=(1, 1), scene=None):
        """The north star's path: every rank renders its row shard of the step's frames (launches of
        at most MAX_BATCH frames) and rank 0 gathers them.  `scene`: another workload than the headline's."""
        sc = scene or main_scene
        W, H, gs, cameras = sc.W, sc.H, sc.gs, sc.cameras  # (shadow the headline's: everything below is the scene's)
        weights = None
        if not contiguous and root_weight[0] != root_weight[1]:
            weights = [root_weight[0]] + [root_weight[1]] * (world - 1)
        if args.gather == "sparse":
            slots = {}  # per payload buffer: the record count on the device, in pinned host memory, and its event

            def pack(shards, stripes, records):
                key_ = records.data_ptr()
                if key_ not in slots:
                    slots[key_] = (torch.zeros(1, dtype=torch.int32, device=device),
                                   torch.zeros(1, dtype=torch.int32).pin_memory(), torch.cuda.Event())
                n_dev, n_host, ev = slots[key_]
                gs.pack_sparse_async(shards, stripes, records, n_dev, n_host, stream=streams[0], encode=args.encode)
                ev.record(streams[0])

                def count():
                    ev.synchronize()
                    return int(n_host[0])
                return count
            sf = SparseShardFrames(
                W, H, rank, world, device, frames_per_step=frames_per_step, buffers=2, weights=weights,
                contiguous=contiguous, pack=pack, count_group=count_group, fill_stream=fill_stream,
                unpack_sparse=lambda frames, records, n, stripes: gs.unpack_sparse_async(
                    frames, records, n, stripes, stream=streams[0]),
                fill=lambda frames, stripes: gs.fill_shard_async(
                    frames, stripes, stream=torch.cuda.current_stream(), encode=args.encode),
                erase=lambda frames, records, n, stripes: gs.erase_sparse_async(
                    frames, records, n, stripes, stream=torch.cuda.current_stream(), encode=args.encode))
        else:
            sf = ShardFrames(W, H, rank, world, device, frames_per_step=frames_per_step, buffers=2,
                             weights=weights, contiguous=contiguous,
                             unpack=lambda frames, shards, stripes: gs.unpack_shard_async(
                                 frames, shards, stripes, stream=streams[0]))

        def render(outs, first_frame, stripes, in_place):
            cams = cameras(first_frame, len(outs), camera_mode)
            if len(outs) <= K.MAX_BATCH:  # one launch: prepared pointer and camera arrays as they are
                gs.render_shard_async(outs, cams, stripes, in_place=in_place, stream=streams[0], encode=args.encode)
                return
            for a in range(0, len(outs), K.MAX_BATCH):
                gs.render_shard_async(outs[a:a + K.MAX_BATCH], cams[a:a + K.MAX_BATCH], stripes,
                                      in_place=in_place, stream=streams[0], encode=args.encode)

        if frames_per_step <= K.MAX_BATCH:
            sf.wrap_targets = K.DevicePointers  # a slot's destination pointers, collected once
        if rank == 0:
            # The hand-off of a step's gathered frames is part of the timed pipeline: a consumer stream takes them when
            # they are complete (a no-op consumer: it waits for the frames and is done), and the slot's next fill and
            # render are ordered after it.
            def hand_off(k, frames):
                consumer_stream.wait_stream(torch.cuda.current_stream())
                done = torch.cuda.Event()
                done.record(consumer_stream)
                return done
            sf.on_frames = hand_off

        def step(k):
            with torch.cuda.stream(streams[0]):
                sf.step(k, render)
        sf.render = render
        return sf, step

    def settle_count(frames_per_step):
        """Steps worth about --settle-ms of rendering: from the workload's constants, the same on every rank (a
        step is roughly its pixels at 60 Gpixel/s per GPU); an even number, so that the double buffering's slots
        line up as without it."""
        step_ms = max(0.05, frames_per_step * W * H / max(1, world) / 60.0e6)
        n = min(400, int(-(-max(0, args.settle_ms) // step_ms)))
        return n + n % 2

    def calibrate_root_weight(frames_per_step):
        """Rank 0 : peer shares by trial, during warm-up: for a few candidate shares the real pipeline runs ten
        steps (the last six timed: barrier, steps, wait_all, synchronise, barrier; the slowest rank counts) and
        the fastest wins, the more even one among those within 2 %.  No model: rank 0 also takes every peer's
        pixels in and puts the background under their rows, a peer also packs its shard, transfers overlap the
        next step's rendering -- what that adds up to is what the trial measures.  With every row crossing the
        links (--gather dense) rank 0 wants several times a peer's share; with sparse shards about the same."""
        candidates = ([(1, 1), (3, 4), (4, 3), (2, 3), (3, 2), (1, 2), (2, 1)] if args.gather == "sparse"
                      else [(1, 1), (2, 1), (3, 1), (4, 1), (6, 1), (8, 1), (12, 1), (16, 1)])
        trial = {}
        note = None
        t_begin = time.perf_counter()
        budget_s = 90.0 * _deadline_scale()  # the whole calibration; a candidate takes well under a second
        for i, cand in enumerate(candidates):
            stage(f"calibration {i + 1}/{len(candidates)} (rank 0 : peer = {cand[0]} : {cand[1]})")
            error = None
            try:  # building a candidate's pipeline allocates (rank 0: two buffers of the step's frames) and talks to nobody
                sf, step = row_shards(frames_per_step, args.camera, contiguous=False, root_weight=cand)
            except Exception as e:
                error, sf, step = repr(e)[:200], None, None
            # every rank agrees whether everybody could build it BEFORE anybody enters a collective of the trial: a rank
            # that cannot must not leave the others waiting for it (a collective that hangs later is the watchdog's)
            if max_over_ranks(1.0 if error else 0.0) > 0.0:
                note = f"candidate {cand[0]}:{cand[1]} could not be set up on a rank" + (f" (here: {error})" if error else "")
                print(f"bench.py[rank {rank}] calibration: {note}; falling back to 1:1", file=sys.stderr, flush=True)
                trial = {}
                del sf, step
                break
            # (the first trial also brings the device's clocks up: it gets the settle phase's worth of steps)
            warm = 4 + (settle_count(frames_per_step) if i == 0 else 0)
            for k in range(warm):
                step(k)
            sf.wait_all()
            torch.cuda.synchronize()
            barrier()
            t0 = time.perf_counter()
            for k in range(warm, warm + 6):
                step(k)
            sf.wait_all()
            torch.cuda.synchronize()
            barrier()
            seconds = (time.perf_counter() - t0) / 6
            del sf, step
            # (no empty_cache() between trials: rank 0's frame buffers have the same size in every trial, so the
            # caching allocator hands them straight back; returning gigabytes to the driver instead makes it scrub
            # them in the background, and whatever is timed in the next second reads up to twice too high --
            # tools/fresh_memory_probe.py, profiles/r03/README.md)
            trial[cand] = max_over_ranks(seconds)
            if args.gather == "dense" and len(trial) >= 3:
                times = list(trial.values())
                if times[-1] > times[-2] > times[-3]:  # past the minimum: larger shares only get slower
                    break
            if max_over_ranks(time.perf_counter() - t_begin) > budget_s:
                note = f"stopped after {len(trial)} of {len(candidates)} candidates: the calibration's {budget_s:.0f} s were spent"
                break
        torch.cuda.empty_cache()  # once, before the real pipeline is built (the settle phase follows)
        if not trial:
            return (1, 1), {"fallback": "1:1", "reason": note or "no candidate completed"}
        floor = min(trial.values())
        best = min((c for c in trial if trial[c] <= 1.02 * floor), key=lambda c: (max(c) / min(c), trial[c]))
        out = {"ms_per_step_by_share": {f"{c[0]}:{c[1]}": round(v * 1e3, 4) for c, v in trial.items()}}
        if note:
            out["note"] = note
        return best, out

    # ---- the headline sequence
    sharded = dist_on and args.shard in ("stripes", "bands")
    count_group = fill_stream = None
    consumer_stream = torch.cuda.Stream(device=device) if sharded else None
    if sharded and args.gather == "sparse":
        # message sizes travel between the hosts over a CPU group, beside the RCCL transfers
        if args.backend == "nccl":
            stage("gloo side group")
            try:
                count_group = dist.new_group(backend="gloo")
            except Exception as e:  # every rank sees the same environment: all fall back together
                print(f"bench.py: no CPU group for the message sizes ({e}); they will travel through device memory",
                      file=sys.stderr)
                count_group = None
        fill_stream = torch.cuda.Stream(device=device)
    root_weight, calibration = (1, 1), None
    if sharded:
        frames_per_step = B * world if args.scaling == "weak" else B
        # rank 0 keeps two buffers of the step's finished frames: at most 24 GB of them (8K frames: 90)
        frames_per_step = max(1, min(frames_per_step, int(24e9 // (2 * 4 * W * H))))
        if args.shard == "stripes":
            if args.root_weight == "auto" and world > 1:
                root_weight, calibration = calibrate_root_weight(frames_per_step)
            elif args.root_weight != "auto":
                parts = [max(1, int(x)) for x in str(args.root_weight).split(":")]
                root_weight = (parts[0], parts[1] if len(parts) > 1 else 1)
                if args.calibration_note:  # a relaunch by self_launch() after the calibration of the first attempt stalled
                    calibration = {"fallback": f"{root_weight[0]}:{root_weight[1]}", "reason": args.calibration_note}
        stage("build pipeline")
        pipe, step = row_shards(frames_per_step, args.camera, contiguous=(args.shard == "bands"),
                                root_weight=root_weight)
        rows0 = pipe.rows[rank]
        frames_per_launch = min(frames_per_step, K.MAX_BATCH)
        launches_per_step = -(-frames_per_step // K.MAX_BATCH)
    elif args.whole_orbit and world == 1:
        pipe, step = whole_orbit(B)
        frames_per_step = orbit_len
        rows0, frames_per_launch, launches_per_step = H, B, -(-orbit_len // B)
    else:
        pipe, step = whole_frames(B, args.camera, deliver=(args.deliver == "root"))
        frames_per_step = B * world
        rows0, frames_per_launch, launches_per_step = H, B, 1
    # Settle (untimed, before the W warm-up steps): the device's clocks and the tile order of this geometry
    # need some tens of milliseconds of work to reach their steady state -- more than the W = 5 steps a
    # short run asks for (5 ms of GPU time; 20 timed steps then read 3 % low).  Steps settle_first ..: the
    # step numbering of warm-up and timed steps continues after them.
    settle_steps = settle_count(frames_per_step) if not args.whole_orbit else 2
    stage(f"settle ({settle_steps} steps)")
    for k in range(settle_steps):
        step(k)
    pipe.wait_all()
    torch.cuda.synchronize()
    base_step = [settle_steps]
    m = run(pipe, lambda k: step(k + base_step[0]), args.steps, args.warmup, announce=True)
    elapsed = m["elapsed"]

    check = None
    if args.check or (dist_on and not args.no_check):
        stage("check (gathered frames against single-GPU renders)")
        last = settle_steps + args.warmup + args.steps - 1
        if rank == 0:
            ref = torch.zeros((H, W, 4), dtype=torch.uint8, device=device)

            def single(frame_index):
                gs.set_raw_uniforms(camera=cameras(frame_index, 1, args.camera)[0])
                gs.render(out=ref, encode=args.encode)
                return ref
            if sharded:
                got = pipe.frames(last)
                check = all(bool(torch.equal(got[i], single(last * frames_per_step + i)))
                            for i in range(frames_per_step))
            else:  # every frame of the step's batch, from every rank that delivered one
                delivered = pipe.frames(last)
                check = all(bool(torch.equal(f[i * H:(i + 1) * H], single((last * world + r) * B + i)))
                            for r, f in enumerate(delivered) for i in range(B))
            gs.set_camera(w.camera)

    # ---- secondary measurements, same process, after the timed region
    secondary = {}
    alg_frame = 4.0 * W * H  # 4 B written per pixel, 0 read

    def summarise(mm, frames, steps, note):
        ms = mm["elapsed"] / steps * 1e3
        k_ms = mm["kernel_ms"] if mm["launches_timed"] else ms
        return {"frames_per_launch": frames, "steps": steps, "ms_per_step": round(ms, 5),
                "mpix_s": round(frames * W * H / (ms * 1e-3) / 1e6, 2), "kernel_ms": round(k_ms, 5),
                "hbm_frac": round(frames * alg_frame / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 6), "note": note}

    if not args.no_secondary:
        sec_steps = max(20, min(args.steps, 200))
        sec_warm = max(6, min(args.warmup, 20))
        if not dist_on:
            stage("secondary lone_frame / orbit_x8 / fixed_camera")
            if B != 1:
                p2, s2 = whole_frames(1, "orbit")
                secondary["lone_frame"] = summarise(
                    run(p2, s2, sec_steps, sec_warm), 1, sec_steps,
                    "one frame per launch, a new orbit pose and a 64-byte camera upload per frame: the latency path "
                    "(the reference's own call pattern, graphics.rs:324)")
                floor_ms, floor_note = lone_frame_floor(key, "orbit")
                lf = secondary["lone_frame"]
                lf["floor_ms"] = floor_ms
                lf["frac_of_floor"] = round(floor_ms / lf["kernel_ms"], 4) if floor_ms and lf["kernel_ms"] > 0 else None
                lf["floor_note"] = floor_note
            if B != 8:
                p5, s5 = whole_frames(8, "orbit")
                secondary["orbit_x8"] = summarise(
                    run(p5, s5, sec_steps, sec_warm), 8, sec_steps,
                    "8 orbit frames per launch: a quarter of the default batch, a quarter of its latency")
            if not (B == 8 and args.camera == "fixed"):
                p3, s3 = whole_frames(8, "fixed")
                secondary["fixed_camera"] = summarise(
                    run(p3, s3, sec_steps, sec_warm), 8, sec_steps,
                    "8 copies of the workload's view per launch (round 1's headline): the best case of the tile-order feedback")
            if key == HEADLINE and not args.whole_orbit:
                # the north star also asks for 4096 x 4096 (BASELINE cfg 4: 512 steps / 16 iterations) and every report
                # carries the reference's own constants (100 / 10, GUI-default c): measured here, in the driver's run
                torch.cuda.empty_cache()
                stage("secondary cfg4_julia_4096 / ref_constants / cfg3")
                short = max(8, min(args.steps, 24))
                secondary["cfg4_julia_4096"] = {
                    "batched": measure_workload("cfg4_julia_4096", local_rank, 16, short, 4, args.encode),
                    "lone_frame": measure_workload("cfg4_julia_4096", local_rank, 1, max(20, min(args.steps, 60)), 6, args.encode),
                    "note": "4096x4096 quaternion-Julia, 512 march steps, 16 SDF iterations, orbit camera; hbm_frac = 4 B per "
                            "pixel / kernel time / 8 TB/s"}
                secondary["ref_constants_1080p"] = measure_workload("ref_julia_1080p", local_rank, 48, sec_steps, sec_warm,
                                                                    args.encode)
                # BASELINE configs 3 and 5: the Sierpinski pipeline at 1080p, and the 8K orbit as a whole
                secondary["cfg3_sierpinski_1080p"] = measure_workload("cfg3_sierpinski_1080p", local_rank, 48, sec_steps, sec_warm,
                                                                      args.encode)
                if args.cfg5_secondary == "auto":
                    torch.cuda.empty_cache()
                    stage("cfg5 whole orbit (two child processes)", deadline_s=2 * 170.0)
                    for name, wl, what in (("cfg5_whole_orbit", "cfg5_sierpinski_8k_orbit_shadows",
                                            "with the soft-shadow secondary rays BASELINE config 5 names (an extension: the reference has none)"),
                                           ("cfg5_whole_orbit_reference_shading", "cfg5_sierpinski_8k_orbit",
                                            "the reference's own shading (no secondary rays)")):
                        secondary[name] = child_line(
                            ["--workload", wl, "--whole-orbit", "--steps", "2", "--warmup", "1", "--frames-per-launch", "24",
                             "--encode", str(args.encode), "--cpu-seconds", "0", "--no-secondary"],
                            150, lambda d, what=what: {
                                "orbit_ms": d["ms_per_step"], "mpix_s": d["value"], "frames": d["config"]["whole_orbit"]["frames"],
                                "frames_per_launch": d["config"]["frames_per_launch"], "kernel": d["roofline"]["kernel"],
                                "note": "all 120 frames of the 7680x4320 Sierpinski orbit (256 march steps, 16 folds), " + what +
                                        "; every frame resident in HBM; one step = the whole orbit; run in a child process"})
        elif args.shard != "frames":
            stage("secondary frame_parallel")
            p4, s4 = whole_frames(B, args.camera, deliver=False)
            mm = run(p4, s4, sec_steps, sec_warm)
            mm["elapsed"] = max_over_ranks(mm["elapsed"])
            sec = summarise(mm, B, sec_steps, f"frame-parallel: every rank renders whole frames ({B} per launch) and "
                            "keeps them in its own HBM; no exchange step, so NOT the north star's gathered frame")
            sec["mpix_s"] = round(sec["mpix_s"] * world, 2)  # whole job: N ranks x B frames per step
            secondary["frame_parallel"] = sec
            del p4, s4
            if sharded and key == HEADLINE:
                # the north star's other size at every N: 4096 x 4096 (BASELINE cfg 4: 512 steps / 16 iterations) through the
                # SAME sharded pipeline -- row shards, sparse records gathered on rank 0, rank 0 : peer as calibrated above
                stage("secondary cfg4_julia_4096 through the sharded pipeline")
                torch.cuda.empty_cache()
                sc4 = other_scene("cfg4_julia_4096")
                f4 = 16 * world if args.scaling == "weak" else 16
                f4 = max(1, min(f4, K.MAX_BATCH, int(24e9 // (2 * 4 * sc4.W * sc4.H))))
                p6, s6 = row_shards(f4, "orbit", contiguous=(args.shard == "bands"), root_weight=root_weight, scene=sc4)
                steps4 = max(6, min(args.steps, 16))
                mm = run(p6, s6, steps4, 3, scene=sc4)
                ms4 = max_over_ranks(mm["elapsed"]) / steps4 * 1e3
                check4 = None
                if rank == 0 and not args.no_check:  # first and last frame of the last step against single-GPU renders
                    last4 = 3 + steps4 - 1
                    ref4 = torch.empty((sc4.H, sc4.W, 4), dtype=torch.uint8, device=device)
                    check4 = True
                    for i in (0, f4 - 1):
                        sc4.gs.set_raw_uniforms(camera=sc4.cameras(last4 * f4 + i, 1, "orbit")[0])
                        sc4.gs.render(out=ref4, encode=args.encode)
                        check4 = check4 and bool(torch.equal(p6.frames(last4)[i], ref4))
                    del ref4
                k4 = mm["kernel_ms"] if mm["launches_timed"] else ms4
                rows4 = p6.rows[rank]
                secondary["cfg4_julia_4096"] = {
                    "workload": "cfg4_julia_4096", "width": sc4.W, "height": sc4.H, "frames_per_step": f4, "steps": steps4,
                    "ms_per_step": round(ms4, 5), "mpix_s": round(f4 * sc4.W * sc4.H / (ms4 * 1e-3) / 1e6, 2),
                    "kernel": mm["kernel"], "kernel_ms": round(k4, 5),
                    "hbm_frac": round(4.0 * sc4.W * rows4 * min(f4, K.MAX_BATCH) / (k4 * 1e-3) / 1e9 / HBM_PEAK_GBS, 6),
                    "rows_per_rank": list(p6.rows), "check": check4,
                    "note": f"4096x4096 quaternion-Julia, 512 march steps, 16 SDF iterations, orbit camera, {world} GPU(s): row "
                            "shards of every frame gathered on rank 0 like the headline; mpix_s is the whole job's; hbm_frac = "
                            "rank 0's rows x 4 B / its kernel time / 8 TB/s"}
                del p6, s6
                sc4.gs.close()
                torch.cuda.empty_cache()
                if check4 is False:
                    check = False

    # ---- reduce over ranks
    if dist_on:
        stage("reduce over ranks")
        red_dev = device if args.backend == "nccl" else torch.device("cpu")
        elapsed = max_over_ranks(elapsed)
        mine = torch.tensor([m["kernel_ms"] * launches_per_step, m["elapsed"] / args.steps * 1e3],
                            dtype=torch.float64, device=red_dev)
        gathered = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine)
        per_rank_kernel_ms = [float(x[0].item()) for x in gathered]
        per_rank_step_ms = [float(x[1].item()) for x in gathered]
    else:
        per_rank_kernel_ms = [m["kernel_ms"]]
        per_rank_step_ms = [m["elapsed"] / args.steps * 1e3]

    # ---- N > 1: the same node once more through the C ABI's one-process path.  Every collective of this job is
    # behind us: the other ranks are finished (they release their process group and leave), rank 0 frees its
    # buffers and starts ONE child that drives all N devices itself -- kifs_multi_render_batch_async with RCCL inside
    # the library -- under a time limit: a child that fails or hangs costs a field of the line, not the line.
    one_proc = None
    rows_per_rank = list(pipe.rows) if sharded else None
    tiles_sent_fraction = (round(pipe.records_sent / pipe.tiles_seen, 4)
                           if sharded and args.gather == "sparse" and getattr(pipe, "tiles_seen", 0) else None)
    if world > 1 and sharded and args.one_process_secondary == "auto" and not args.no_secondary:
        if rank != 0:
            for g in gss:
                g.close()
            dist.destroy_process_group()
            stage("done")
            _STAGE["name"] = "done"
            return
        del pipe, step
        torch.cuda.empty_cache()
        stage("one-process child (kifs_multi_render_batch_async, RCCL inside the library)")
        one_proc = child_line(
            ["--gpus", str(world), "--host", "one-process", "--steps", str(min(args.steps, 40)), "--warmup", str(min(args.warmup, 8)),
             "--workload", key, "--frames-per-launch", str(B), "--gather", args.gather, "--scaling", args.scaling,
             "--camera", args.camera, "--encode", str(args.encode), "--cpu-seconds", "0", "--no-secondary"]
            + (["--share-device"] if args.share_device else []), 180,
            lambda d: {"value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"],
                       "frames_per_step": d["config"]["frames_per_step"], "comm": d.get("comm"),
                       "per_rank_kernel_ms": d.get("per_rank_kernel_ms"),
                       "note": "ONE process drives all the devices through kifs_multi_render_batch_async (C ABI); "
                               "run in a child process after this job's own collectives"})

    if rank == 0:
        mpix = frames_per_step * W * H * args.steps / elapsed / 1e6
        kernel_name = m["kernel"]
        # average duration of the dominant kernel over the timed region (per-launch event pairs);
        # ms_per_step additionally contains the inter-launch gaps and, at N > 1, the exchange
        launch_s = (m["kernel_ms"] if m["launches_timed"] else elapsed / args.steps * 1e3) / 1e3
        alg_bytes = 4.0 * W * rows0 * frames_per_launch  # rank 0's rows of the launch's frames
        achieved = alg_bytes / launch_s / 1e9
        traffic, traffic_source, pmc, traffic_note = pmc_traffic(key, frames_per_launch) if not dist_on else (None, None, None, None)
        if world == 1 and not sharded:
            parallelism = (f"1 GPU, {B} frame(s) of the sequence per launch"
                           + (f", {F} launches in flight" if F > 1 else ""))
        elif sharded:
            parallelism = (f"{world} GPUs, one process each: row shards ("
                           + ("contiguous bands of 8-row stripes" if args.shard == "bands" else
                              "8-row stripes dealt round-robin"
                              + (f", rank 0 : peer = {root_weight[0]} : {root_weight[1]}" if root_weight[0] != root_weight[1] else ""))
                           + f") of {frames_per_step} frames per step, gathered into rank 0's frames by grouped "
                           + ("RCCL point-to-point over xGMI" if args.backend == "nccl" else
                              "point-to-point transfers (REHEARSAL: gloo, staged through the host; RCCL needs a GPU per rank)")
                           + (": peers send the 32x8 tiles that hold something, rank 0 fills in the background"
                              if args.gather == "sparse" else " + stripe unpack"))
        else:
            parallelism = (f"{world} GPUs x whole frames (frame-parallel, {B} per launch), one process per GPU, "
                           + (("finished frames sent to rank 0 by grouped RCCL p2p" if args.backend == "nccl" else
                               "finished frames sent to rank 0 (rehearsal: gloo)") if args.deliver == "root"
                              else "frames stay on the GPU that rendered them (no exchange step)"))
        out = {
            "metric": METRIC,
            "value": round(mpix, 2),
            "unit": "Mpixels/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "settle_steps": settle_steps,
            "ms_per_step": round(elapsed / args.steps * 1e3, 5),
            "higher_is_better": True,
            "scaling": "strong" if (sharded and args.scaling == "strong") else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": key, "description": w.name, "width": W, "height": H,
                       "max_iterations": w.gui.max_iterations, "sdf_iters": w.iters[0],
                       "normal_iters": w.iters[1], "fold_iters": w.iters[2],
                       "encode": "srgb8" if args.encode else "unorm8",
                       "camera": "orbit, one pose per frame" if args.camera == "orbit" else "fixed",
                       "frames_per_step": frames_per_step, "frames_per_launch": frames_per_launch,
                       "launches_in_flight": F, "parallelism": parallelism,
                       "settle_steps_before_warmup": settle_steps},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6),
                         "traffic": traffic, "traffic_source": traffic_source,
                         "kernel": kernel_name, "kernel_ms": round(launch_s * 1e3, 5),
                         "kernel_ms_min": round(m["kernel_ms_min"], 5), "kernel_ms_max": round(m["kernel_ms_max"], 5),
                         "launches_timed": m["launches_timed"],
                         "concurrent_launches": F,
                         "algorithmic_bytes_per_launch": int(alg_bytes)},
            "per_rank_kernel_ms": [round(x, 5) for x in per_rank_kernel_ms],
            "per_rank_step_ms": [round(x, 5) for x in per_rank_step_ms],
        }
        if dist_on:
            out["comm"] = {"backend": ("rccl (torch.distributed backend 'nccl')" if args.backend == "nccl" else "gloo (rehearsal)"),
                           "world_size_seen": dist.get_world_size(),
                           "count_channel": (None if not (sharded and args.gather == "sparse") else
                                             "gloo side group" if count_group is not None else "device all_gather"),
                           "gather": args.gather if sharded else ("frames to root" if args.deliver == "root" else "none"),
                           "check": check}
        if args.whole_orbit:
            out["config"]["whole_orbit"] = {"frames": orbit_len, "launches_per_step": launches_per_step,
                                            "total_ms": round(elapsed / args.steps * 1e3, 4),
                                            "resident_bytes": int(4 * W * H * orbit_len)}
        if sharded:
            out["config"]["root_weight"] = root_weight[0]
            out["config"]["peer_weight"] = root_weight[1]
            out["config"]["rows_per_rank"] = rows_per_rank
            out["config"]["gather"] = args.gather
            if tiles_sent_fraction is not None:
                out["config"]["tiles_sent_fraction"] = tiles_sent_fraction
            if calibration:
                out["config"]["root_weight_calibration"] = calibration
        if traffic_note:
            out["roofline"]["traffic_note"] = traffic_note
        if pmc and pmc.get("valu_instructions_per_launch") and pmc.get("kernel_cycles"):
            # what actually bounds the kernel: the vector pipes.  From the same committed PMC passes: VALU
            # wave-instructions per launch x 2.25 pipe cycles each (tools/microbench/valu_rate: a full chip
            # sustains one plain wave64 VALU instruction per 2.25 cycles per SIMD; compares, packed and
            # transcendental ones take 4 .. 8, so this is a lower bound of the pipes' busy time)
            busy = pmc["valu_instructions_per_launch"] * 2.25 / (1024.0 * pmc["kernel_cycles"])
            out["roofline"]["valu_issue"] = {"bound": "valu-issue", "frac_at_plain_rate": round(busy, 4),
                                             "valu_instructions_per_launch": pmc["valu_instructions_per_launch"],
                                             "kernel_cycles": pmc["kernel_cycles"], "simds": 1024,
                                             "cycles_per_plain_instruction": 2.25,
                                             "source": "profiles/pmc_traffic.json (rocprofv3 --pmc), tools/microbench/valu_rate.hip"}
        if one_proc is not None:
            secondary["one_process_c_abi"] = one_proc
        if secondary:
            out["secondary"] = secondary
        if check is not None:
            out["gathered_frame_equals_single_gpu_frame"] = check
        if world == 1 and args.cpu_seconds > 0:
            stage("cpu baseline", deadline_s=args.cpu_seconds + 170.0)
            out["cpu_baseline"] = cpu_baseline(w, args.cpu_seconds)
            out["roofline"]["secondary"] = work_count(w, launch_s / frames_per_launch)
        print(json.dumps(out), flush=True)

    for g in gss:
        g.close()
    if dist_on and dist.is_initialized():
        dist.destroy_process_group()
    stage("done")
    _STAGE["name"] = "done"
    if check is False:
        sys.exit("bench.py: gathered frames differ from single-GPU frames")


if __name__ == "__main__":
    main()
