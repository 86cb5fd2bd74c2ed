#!/usr/bin/env python3
"""What to EXPECT at N GPUs per workload -- arithmetic from one-GPU timings, NOT a measurement of N GPUs.

A gpurun box has one GPU and the driver's 8-GPU run may not happen, so this prints, for every workload and
N = 2 / 4 / 8, the sides of a step of the default multi-GPU run (row shards of N x B frames, sparse gather to
device 0), each timed on ONE GPU exactly as that rank would run it:

  peer    : a peer's shard (its stripes of all the step's frames, one launch) + its pack into sparse records
  root    : device 0's shard in place + the erase under the previous records + the scatter of N - 1 peers' records
            (fill / scatter on a second stream beside the rendering, as the pipelines do)
  link    : one peer's payload over ONE xGMI link at 77 GB/s (7 links x 153 GB/s bidirectional per GPU;
            MI355X_MICROARCH.md), sparse and dense -- transfers overlap the next step's rendering, so they bound
            the step only when they take longer than it

  expected step = max(peer, root, link);   speed-up = N x (single-GPU step of B frames) / expected step

The single-GPU step is measured here too (B frames, one launch).  `tiles_sent` is the share of a peer's tiles that
hold something: close to 1 the sparse form degenerates to the dense one (+1.6 %) and the LINK sets the step.
    python tools/expected_scaling.py [workload ...]        (default: the BASELINE configs)"""
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

import kifs_raymarching_amd as K  # noqa: E402
from kifs_raymarching_amd.configs import WORKLOADS, orbit_camera  # noqa: E402

LINK_GBS = 77.0  # one direction of one xGMI link


def timed(stream, side, body, reps):
    for _ in range(3):
        body()
        stream.wait_stream(side)
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record(stream)
    for _ in range(reps):
        body()
        stream.wait_stream(side)
    t1.record(stream)
    torch.cuda.synchronize()
    return t0.elapsed_time(t1) / reps


def study(key, B, worlds=(2, 4, 8), quiet=False):
    w = WORKLOADS[key]
    W, H = w.screen.width, w.screen.height
    gs = K.GraphicState(0, screen_data=w.screen, camera_data=w.camera, gui_data=w.gui)
    gs.set_iters(*w.iters)
    if w.extensions:
        gs.set_extensions(**w.extensions)
    a, b = torch.cuda.Stream(), torch.cuda.Stream()
    reps = 6 if W * H > 4e6 else 12
    # the single-GPU step: B whole frames in one launch
    whole = torch.zeros((B, H, W, 4), dtype=torch.uint8, device="cuda:0")
    cams1 = K.camera_array([orbit_camera(w, k).into_buffer_data() for k in range(B)])
    outs1 = K.DevicePointers([whole[i] for i in range(B)])
    for _ in range(12):
        gs.render_batch_async(outs1, cams1, stream=a)
    single = timed(a, b, lambda: gs.render_batch_async(outs1, cams1, stream=a), reps)
    del whole, outs1
    rows = []
    for N in worlds:
        F = min(B * N, K.MAX_BATCH, max(1, int(20e9 // (4 * W * H))))
        frames = torch.zeros((F, H, W, 4), dtype=torch.uint8, device="cuda:0")
        cams = K.camera_array([orbit_camera(w, k).into_buffer_data() for k in range(F)])
        mine, _ = K.shard_stripes(H, 0, N)
        st1, rows1 = K.shard_stripes(H, 1, N)
        shard = torch.zeros((F, rows1, W, 4), dtype=torch.uint8, device="cuda:0")
        shard_outs = K.DevicePointers([shard[i] for i in range(F)])
        frame_outs = K.DevicePointers([frames[i] for i in range(F)])
        records = torch.zeros((gs.sparse_capacity(F, st1), 1040), dtype=torch.uint8, device="cuda:0")
        n_dev = torch.zeros(1, dtype=torch.int32, device="cuda:0")
        n_host = torch.zeros(1, dtype=torch.int32).pin_memory()
        peer_render = lambda: gs.render_shard_async(shard_outs, cams, st1, stream=a)
        pack = lambda: gs.pack_sparse_async(shard, st1, records, n_dev, n_host, stream=a)
        for _ in range(8):
            peer_render()
        pack()
        a.synchronize()
        n = int(n_host[0])
        t_peer = timed(a, b, lambda: (peer_render(), pack()), reps)
        root_render = lambda: gs.render_shard_async(frame_outs, cams, mine, in_place=True, stream=a)
        for _ in range(8):
            root_render()
        erase = lambda s: [gs.erase_sparse_async(frames, records, n, st1, stream=s) for _ in range(N - 1)]
        unpack = lambda s: [gs.unpack_sparse_async(frames, records, n, st1, stream=s) for _ in range(N - 1)]
        t_root = timed(a, b, lambda: (b.wait_stream(a), root_render(), erase(b), unpack(b)), reps)
        sparse_bytes, dense_bytes = n * 1040, shard.numel()
        t_link_sparse = sparse_bytes / (LINK_GBS * 1e6)  # ms
        t_link_dense = dense_bytes / (LINK_GBS * 1e6)
        step = max(t_peer, t_root, t_link_sparse)
        step_dense = max(t_peer, t_root, t_link_dense)
        scale = F / B  # frames per step relative to the single-GPU step
        rows.append({"workload": key, "n_gpus": N, "frames_per_step": F, "single_gpu_ms_per_B_frames": round(single, 4), "B": B,
                     "peer_ms": round(t_peer, 4), "root_ms": round(t_root, 4),
                     "tiles_sent": round(n / max(1, records.shape[0]), 4),
                     "payload_MB_sparse": round(sparse_bytes / 1e6, 2), "payload_MB_dense": round(dense_bytes / 1e6, 1),
                     "link_ms_sparse": round(t_link_sparse, 4), "link_ms_dense": round(t_link_dense, 4),
                     "expected_step_ms": round(step, 4), "bound": ("peer" if step == t_peer else "root" if step == t_root else "link"),
                     "expected_speedup": round(scale * single / step, 2),
                     "expected_gpixel_s": round(F * W * H / step / 1e6, 1),
                     "expected_speedup_dense_gather": round(scale * single / step_dense, 2)})
        if not quiet:
            print(json.dumps(rows[-1]), flush=True)
        del frames, shard, records, shard_outs, frame_outs
        # (no empty_cache() here: handing gigabytes back to the driver makes it scrub them in the background, and the
        # next size's timings then read up to twice too high -- seen on the first workload of a run, gone on a repeat)
    gs.close()
    torch.cuda.empty_cache()
    torch.cuda.synchronize()
    import time
    time.sleep(2.0)
    return rows


def main():
    keys = sys.argv[1:] or ["cfg2_julia_1080p", "cfg3_sierpinski_1080p", "cfg4_julia_4096", "cfg5_sierpinski_8k_orbit",
                            "cfg5_sierpinski_8k_orbit_shadows", "ref_julia_1080p", "n1_genjulia_1080p", "n2_bunny_1080p"]
    batch = lambda k: 48 if WORKLOADS[k].pixels <= 4e6 else 16 if WORKLOADS[k].pixels <= 2e7 else 4
    # a throw-away pass first: the process's first gigabytes of fresh allocations read up to twice too high for a
    # second or so (tools/fresh_memory_probe.py: the same launch into the same block settles after a few batches)
    study(keys[0], batch(keys[0]), quiet=True)
    for k in keys:
        study(k, batch(k))


if __name__ == "__main__":
    main()
