"""bench.py's contract with the driver, on the GPU box: the one JSON line of a default-shaped run
carries metric / value / roofline / cpu_baseline / the secondary measurements, and `--gpus 2` without a
launcher starts its own ranks and gathers frames equal to single-GPU frames (ranks share the one GPU
and talk over gloo: RCCL needs a GPU per rank)."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def _run(*args, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), *args], capture_output=True, text=True,
                       timeout=timeout, env=env, cwd=str(ROOT))
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_single_gpu_line_has_the_contract_fields():
    d = _run("--steps", "12", "--warmup", "4", "--cpu-seconds", "1")
    assert d["metric"].startswith("Mpixels/s at 1920x1080") and d["unit"] == "Mpixels/s"
    assert d["n_gpus"] == 1 and d["steps"] == 12 and d["warmup"] == 4 and d["higher_is_better"] is True
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and d["vs_baseline"] is None
    cfg = d["config"]
    assert cfg["workload"] == "cfg2_julia_1080p" and cfg["camera"].startswith("orbit")
    assert cfg["frames_per_launch"] == 48 and cfg["frames_per_step"] == 48
    assert d["value"] > 1000.0  # the north star's 1 Gpixel/s
    assert abs(d["value"] - 48 * 1920 * 1080 / (d["ms_per_step"] * 1e-3) / 1e6) < 0.01 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["unit"] == "GB/s"
    assert r["kernel"] == "render_wave_kernel" and r["launches_timed"] >= 2
    assert 0 < r["kernel_ms"] <= d["ms_per_step"] * 1.15  # (three sampled launches of twelve: launch times vary by +-10 %)
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-6
    assert r["algorithmic_bytes_per_launch"] == 48 * 1920 * 1080 * 4
    assert r["traffic_source"] == "profiles/pmc_traffic.json"
    # counters are reported only when the committed pass was measured on THESE kernels (kernel hash of the loaded
    # library = the hash recorded with the pass); otherwise traffic is null and the line says why
    import kifs_raymarching_amd as K
    recorded = json.loads((ROOT / "profiles" / "pmc_traffic.json").read_text()).get("cfg2_julia_1080p@48", {})
    if recorded.get("kernel_hash") == K._lib.kernel_hash_of_loaded_library():
        assert 0.9 < r["traffic"] / r["algorithmic_bytes_per_launch"] < 1.2 and "traffic_note" not in r
        v = r["valu_issue"]  # the bound that binds, from the same pass
        assert v["bound"] == "valu-issue" and 0.5 < v["frac_at_plain_rate"] < 1.0 and v["simds"] == 1024
    else:
        assert r["traffic"] is None and "valu_issue" not in r and "other kernels" in r["traffic_note"]
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "Mpixels/s" and c["cores"] >= 1 and c["value"] > 0
    assert d["settle_steps"] == cfg["settle_steps_before_warmup"] > 0
    sec = d["secondary"]
    assert set(sec) == {"lone_frame", "orbit_x8", "fixed_camera", "cfg4_julia_4096", "ref_constants_1080p",
                        "cfg3_sierpinski_1080p", "cfg5_whole_orbit", "cfg5_whole_orbit_reference_shading"}
    assert sec["lone_frame"]["frames_per_launch"] == 1 and sec["orbit_x8"]["frames_per_launch"] == 8
    # the reference's call pattern next to its floor (the critical ray's instructions x a lone wave's issue interval)
    lf = sec["lone_frame"]
    floor = json.loads((ROOT / "profiles" / "lone_frame_floor.json").read_text())["cfg2_julia_1080p"]
    if floor["kernel_hash"] == K._lib.kernel_hash_of_loaded_library():
        assert lf["floor_ms"] == floor["floor_ms_orbit_mean"] and 0.05 < lf["floor_ms"] < lf["kernel_ms"]
        assert lf["frac_of_floor"] == pytest.approx(lf["floor_ms"] / lf["kernel_ms"], abs=1e-3) and 0.4 < lf["frac_of_floor"] < 1.0
    else:
        assert lf["floor_ms"] is None and lf["frac_of_floor"] is None and "re-run" in lf["floor_note"]
    assert all(sec[k]["mpix_s"] > 0 and sec[k]["kernel_ms"] > 0 for k in ("lone_frame", "orbit_x8", "fixed_camera"))
    # the north star's 4096 x 4096 figure and the reference-constant run travel in the driver's own line
    c4 = sec["cfg4_julia_4096"]
    for form, frames in (("batched", 16), ("lone_frame", 1)):
        assert c4[form]["width"] == c4[form]["height"] == 4096 and c4[form]["frames_per_launch"] == frames
        assert c4[form]["mpix_s"] > 1000.0 and 0 < c4[form]["hbm_frac"] < 1 and c4[form]["kernel_ms"] > 0
    ref = sec["ref_constants_1080p"]
    assert ref["workload"] == "ref_julia_1080p" and ref["frames_per_launch"] == 48 and ref["mpix_s"] > 1000.0
    c3 = sec["cfg3_sierpinski_1080p"]
    assert c3["workload"] == "cfg3_sierpinski_1080p" and c3["frames_per_launch"] == 48 and c3["mpix_s"] > 1000.0
    # BASELINE config 5 as it is named: the whole 120-frame 8K orbit, timed in a child process
    for name in ("cfg5_whole_orbit", "cfg5_whole_orbit_reference_shading"):
        c5 = sec[name]
        assert "error" not in c5, c5
        assert c5["frames"] == 120 and c5["frames_per_launch"] == 24 and c5["kernel"] == "render_wave_kernel"
        assert 20.0 < c5["orbit_ms"] < 2000.0 and abs(c5["mpix_s"] - 120 * 7680 * 4320 / (c5["orbit_ms"] * 1e-3) / 1e6) < 0.01 * c5["mpix_s"]
    assert sec["cfg5_whole_orbit"]["orbit_ms"] > sec["cfg5_whole_orbit_reference_shading"]["orbit_ms"]  # the secondary rays cost something
    assert d["per_rank_kernel_ms"] == [pytest.approx(r["kernel_ms"], rel=1e-3)]


def test_two_ranks_are_started_by_bench_itself_and_gather_exact_frames():
    d = _run("--gpus", "2", "--backend", "gloo", "--share-device", "--check", "--steps", "4", "--warmup", "2",
             "--cpu-seconds", "0", "--frames-per-launch", "4")
    assert d["n_gpus"] == 2 and d["scaling"] == "weak"
    assert d["gathered_frame_equals_single_gpu_frame"] is True
    cfg = d["config"]
    assert cfg["frames_per_step"] == 8 and "8-row stripes" in cfg["parallelism"] and "REHEARSAL: gloo" in cfg["parallelism"]
    assert cfg["root_weight"] >= 1 and cfg["peer_weight"] >= 1 and sum(cfg["rows_per_rank"]) == 1080
    trial = cfg["root_weight_calibration"]["ms_per_step_by_share"]
    assert "1:1" in trial and len(trial) >= 3 and all(v > 0 for v in trial.values())
    assert f'{cfg["root_weight"]}:{cfg["peer_weight"]}' in trial
    assert cfg["gather"] == "sparse" and 0.0 < cfg["tiles_sent_fraction"] < 0.2
    assert len(d["per_rank_kernel_ms"]) == 2 and all(x > 0 for x in d["per_rank_kernel_ms"])
    assert "frame_parallel" in d["secondary"] and "NOT the north star" in d["secondary"]["frame_parallel"]["note"]
    comm = d["comm"]
    assert comm["backend"].startswith("gloo") and comm["world_size_seen"] == 2 and comm["gather"] == "sparse"
    assert comm["check"] is True
    # ... and, once the ranks' own collectives are over, the same two "devices" driven by ONE process through the C ABI
    # the north star's other size, through the same sharded pipeline
    c4 = d["secondary"]["cfg4_julia_4096"]
    assert c4["width"] == c4["height"] == 4096 and c4["frames_per_step"] == 32 and sum(c4["rows_per_rank"]) == 4096
    assert c4["check"] is True and c4["mpix_s"] > 1000.0 and 0 < c4["hbm_frac"] < 1
    one = d["secondary"]["one_process_c_abi"]
    assert "error" not in one, one
    assert one["value"] > 1000.0 and one["frames_per_step"] == 8 and one["comm"]["check"] is True
    assert one["comm"]["backend"] == "hip peer copies" and len(one["per_rank_kernel_ms"]) == 2


def test_two_ranks_check_their_frames_by_default():
    """Without --check: at N > 1 the last step's gathered frames are compared with single-GPU renders anyway."""
    d = _run("--gpus", "2", "--backend", "gloo", "--share-device", "--steps", "3", "--warmup", "1", "--cpu-seconds", "0",
             "--frames-per-launch", "3", "--root-weight", "1", "--no-secondary", "--gather", "dense")
    assert d["comm"]["check"] is True and d["gathered_frame_equals_single_gpu_frame"] is True
    assert d["comm"]["gather"] == "dense" and d["comm"]["count_channel"] is None


def test_one_process_drives_every_device_through_the_c_abi():
    """--host one-process: no ranks, no torch.distributed -- kifs_multi_render_batch_async.  Three 'devices' on the
    one GPU (peer-copy transport), frames checked against single-GPU renders; same JSON shape."""
    d = _run("--gpus", "3", "--host", "one-process", "--share-device", "--steps", "6", "--warmup", "2",
             "--frames-per-launch", "8")
    assert d["n_gpus"] == 3 and d["scaling"] == "weak" and d["unit"] == "Mpixels/s"
    cfg = d["config"]
    assert cfg["frames_per_step"] == 24 and sum(cfg["rows_per_rank"]) == 1080 and "ONE process" in cfg["parallelism"]
    comm = d["comm"]
    assert comm["backend"] == "hip peer copies" and comm["check"] is True and comm["gather"] == "sparse"
    assert 0.0 < comm["tiles_sent_fraction"] < 0.2 and comm["bytes_into_root_per_step"] > 0
    assert d["gathered_frame_equals_single_gpu_frame"] is True
    assert d["value"] > 1000.0 and len(d["per_rank_kernel_ms"]) == 3 and all(x > 0 for x in d["per_rank_kernel_ms"])
    assert abs(d["value"] - 24 * 1920 * 1080 / (d["ms_per_step"] * 1e-3) / 1e6) < 0.01 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["algorithmic_bytes_per_launch"] == 4 * 1920 * cfg["rows_per_rank"][0] * 24


def test_whole_orbit_step():
    """--whole-orbit: one step = all 120 poses, resident; reported with its launch count and total time."""
    d = _run("--workload", "cfg3_sierpinski_1080p", "--whole-orbit", "--frames-per-launch", "48", "--steps", "3",
             "--warmup", "1", "--cpu-seconds", "0", "--no-secondary")
    cfg = d["config"]
    assert cfg["frames_per_step"] == 120 and cfg["frames_per_launch"] == 48
    wo = cfg["whole_orbit"]
    assert wo["frames"] == 120 and wo["launches_per_step"] == 3 and wo["resident_bytes"] == 120 * 1920 * 1080 * 4
    assert abs(wo["total_ms"] - d["ms_per_step"]) < 1e-3
    assert abs(d["value"] - 120 * 1920 * 1080 / (d["ms_per_step"] * 1e-3) / 1e6) < 0.01 * d["value"]


def _run_raw(*args, env=None, timeout=600):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    e.update(env or {})
    return subprocess.run([sys.executable, str(ROOT / "bench.py"), *args], capture_output=True, text=True,
                          timeout=timeout, env=e, cwd=str(ROOT))


def test_the_rccl_backend_with_a_world_of_one_rank():
    """The environment half of a multi-GPU run, on the one GPU of this box: --dist-at-one runs
    bench.py's N > 1 code path with world size 1 over the REAL backend -- init_process_group("nccl", device_id=...),
    barrier(device_ids=...), all_reduce / all_gather on the device, the gloo side group for the message sizes, the
    sharded pipeline (no peers), the default frame check, frame_parallel and the 4096 x 4096 figure through the
    sharded pipeline, destroy_process_group -- so that hardware day only adds the second rank."""
    p = _run_raw("--dist-at-one", "--steps", "4", "--warmup", "2", "--cpu-seconds", "0", "--frames-per-launch", "4")
    assert p.returncode == 0, p.stderr[-3000:]
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    comm = d["comm"]
    assert comm["backend"].startswith("rccl") and comm["world_size_seen"] == 1 and comm["check"] is True
    assert comm["count_channel"] == "gloo side group" and comm["gather"] == "sparse"
    assert d["n_gpus"] == 1 and d["config"]["rows_per_rank"] == [1080] and "row shards" in d["config"]["parallelism"]
    assert d["secondary"]["cfg4_julia_4096"]["check"] is True and "frame_parallel" in d["secondary"]
    for name in ("import", "init process group (nccl)", "gloo side group", "build pipeline", "settle", "warmup", "timed",
                 "check", "secondary frame_parallel", "secondary cfg4_julia_4096", "reduce over ranks", "done"):
        assert f"bench.py[rank 0] stage: {name}" in p.stderr, name
    # and torch.distributed's own pieces the pipeline relies on at N > 1, with this rank as its own peer
    code = """
import os, torch, torch.distributed as dist
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29731", HSA_ENABLE_IPC_MODE_LEGACY="0")
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
dist.barrier(device_ids=[0])
t = torch.arange(4, dtype=torch.float64, device=dev); dist.all_reduce(t, op=dist.ReduceOp.MAX)
g = dist.new_group(backend="gloo"); c = torch.tensor([7]); out = [torch.zeros(1, dtype=torch.int64)]
dist.gather(c, out, dst=0, group=g)
src = torch.arange(1040 * 3, dtype=torch.int32, device=dev).to(torch.uint8); dst = torch.zeros_like(src)
for w in dist.batch_isend_irecv([dist.P2POp(dist.irecv, dst, 0), dist.P2POp(dist.isend, src, 0)]): w.wait()
torch.cuda.synchronize()
assert torch.equal(src, dst) and int(out[0]) == 7 and t.tolist() == [0.0, 1.0, 2.0, 3.0]
dist.destroy_process_group(); print("ok")
"""
    q = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert q.returncode == 0 and q.stdout.strip().endswith("ok"), q.stderr[-2000:]


def test_a_rank_asleep_in_the_timed_steps_ends_the_run_and_names_the_stage():
    """VERDICT r03 item 1 'done': the gloo --share-device rehearsal with one rank made to sleep in a stage ends within
    the deadline, rc != 0, naming the stage.  (120 + 0.5 x 4) s x 0.1: the timed steps get 12 s.)"""
    import time
    t0 = time.time()
    p = _run_raw("--gpus", "2", "--backend", "gloo", "--share-device", "--steps", "4", "--warmup", "2", "--cpu-seconds", "0",
                 "--frames-per-launch", "4", "--root-weight", "1", "--no-secondary",
                 env={"KIFS_BENCH_STALL": "timed:1", "KIFS_BENCH_DEADLINE_SCALE": "0.1"})
    assert p.returncode != 0 and time.time() - t0 < 150, p.stderr[-2000:]
    assert "STALLED in stage 'timed (4 steps)'" in p.stderr or "stalled in stage 'timed (4 steps)'" in p.stderr
    assert "last stage of every rank" in p.stderr and "'timed (4 steps)'" in p.stderr.split("last stage of every rank")[-1]
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]


def test_a_stalled_calibration_falls_back_to_even_shares_in_a_fresh_launch():
    """... and a calibration that stalls costs the calibration, not the run: self_launch() starts the ranks once more
    (fresh children; the parent never touches the GPU) with --root-weight 1:1 and the line says so."""
    p = _run_raw("--gpus", "2", "--backend", "gloo", "--share-device", "--steps", "4", "--warmup", "2", "--cpu-seconds", "0",
                 "--frames-per-launch", "4", "--no-secondary",
                 env={"KIFS_BENCH_STALL": "calibration 2:1", "KIFS_BENCH_DEADLINE_SCALE": "0.1"})
    assert p.returncode == 0, p.stderr[-3000:]
    assert "starting the ranks once more with --root-weight 1:1" in p.stderr
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    cal = d["config"]["root_weight_calibration"]
    assert cal["fallback"] == "1:1" and "calibration 2/7" in cal["reason"]
    assert d["config"]["root_weight"] == d["config"]["peer_weight"] == 1 and d["comm"]["check"] is True
