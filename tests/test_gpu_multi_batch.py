"""kifs_multi_render_batch_async: the batched, sparse, pipelined gather of one process driving N devices, through
the C ABI (ctypes).  A gpurun box has one GPU, so the device is listed several times (transport COPY); RCCL inside
the library is exercised with one rank (init, grouped self send/recv, a one-device step) and on distinct devices
when the box has them.  Every gathered frame must equal the frame one device renders alone, byte for byte -- and
that in turn is checked against the oracle on one frame of every scene."""
import ctypes as C

import numpy as np
import pytest

from helpers import oracle_frame

pytestmark = pytest.mark.gpu


def _scenes(kifs):
    julia = kifs.GuiData(max_iterations=96, fractal_group=kifs.FractalGroup.JuliaSet, constant=(-0.2, 0.6, 0.2, 0.2))
    # a close box: most of the frame is the primitive (few background tiles: the sparse form's worst case)
    box = kifs.GuiData(max_iterations=64, fractal_group=kifs.FractalGroup.KaleidoscopicIFS,
                       primitive_shape=kifs.PrimitiveShape.Box, background_color=(30, 60, 90))
    return {"julia": (julia, (12, 10, 10), 4.0), "box": (box, (100, 10, 10), 2.6)}


def _cameras(kifs, n, distance, first=0):
    return [kifs.CameraData(origin_distance=distance, phi=0.37 * (first + i), theta=0.25 * np.sin(first + i)) for i in range(n)]


def _single_frames(kifs, screen, gui, iters, cams, encode=1):
    import torch
    out = []
    with kifs.GraphicState(0, screen_data=screen, camera_data=cams[0], gui_data=gui) as gs:
        gs.set_iters(*iters)
        for cam in cams:
            gs.set_camera(cam)
            out.append(torch.from_numpy(gs.render(encode=encode)))
    return torch.stack(out)


@pytest.mark.parametrize("gather", ["sparse", "dense"])
@pytest.mark.parametrize("scene", ["julia", "box"])
def test_batch_on_four_listed_devices_equals_single_frames(kifs, oracle, gather, scene):
    """16 frames of a ragged frame size (neither dimension a multiple of the 32 x 8 tile) over [0, 0, 0, 0]."""
    import torch
    gui, iters, dist = _scenes(kifs)[scene]
    screen = kifs.ScreenData(333, 211)
    cams = _cameras(kifs, 16, dist)
    want = _single_frames(kifs, screen, gui, iters, cams)
    assert (want[3].numpy() == oracle_frame(oracle, kifs, screen, cams[3], gui, iters)).all()
    with kifs.MultiGraphicState([0, 0, 0, 0], screen, cams[0], gui, iters=iters) as mg:
        mg.set_gather(gather, "auto")
        frames = torch.full((16, 211, 333, 4), 99, dtype=torch.uint8, device="cuda:0")
        mg.render_batch(frames, cams)
        assert torch.equal(frames.cpu(), want), (gather, scene)
        st = mg.stats()
        assert st["transport"] == "copy" and st["gather"] == gather and st["steps"] == 1 and st["bytes_received"] > 0
        if gather == "sparse":
            # 27 stripes: three peers own 20 of them; 11 tile columns; 16 frames
            assert st["tiles_covered"] == 16 * 20 * 11 and 0 < st["records_received"] <= st["tiles_covered"]
            if scene == "julia":
                assert st["records_received"] < st["tiles_covered"] // 2  # mostly background: most tiles stay home
        shards = mg.shards()
        assert sum(s[2] for s in shards) == 211 and all(s[3] > 0 for s in shards)


def test_pipelined_steps_with_reused_buffers(kifs):
    """Seven steps through two frame buffers handed back as untouched (the erase-under-previous-records path),
    weights 3:1:1, a different pose set per step; each step is checked after its wait, before its buffer goes back."""
    import torch
    gui, iters, dist = _scenes(kifs)["julia"]
    screen = kifs.ScreenData(320, 200)
    B = 6
    with kifs.MultiGraphicState([0, 0, 0], screen, kifs.CameraData(), gui, iters=iters) as mg:
        mg.set_weights([3, 1, 1])
        bufs = [torch.zeros((B, 200, 320, 4), dtype=torch.uint8, device="cuda:0") for _ in range(2)]
        poses = [_cameras(kifs, B, dist, first=10 * k) for k in range(7)]
        wants = [_single_frames(kifs, screen, gui, iters, p) for p in poses]
        steps = []
        for k in range(7):
            if k >= 2:  # the consumer reads step k - 2 before its buffer is submitted again
                mg.wait(steps[k - 2])
                assert torch.equal(bufs[k % 2].cpu(), wants[k - 2]), k - 2
            steps.append(mg.render_batch_async(bufs[k % 2], poses[k], untouched=k >= 2))
            assert steps[-1] == k
        mg.wait(steps[5])
        assert torch.equal(bufs[1].cpu(), wants[5])
        mg.wait_all()
        assert torch.equal(bufs[0].cpu(), wants[6])
        mg.wait(steps[0])  # long complete: a no-op
        assert mg.stats()["steps"] == 7


def test_untouched_flag_is_ignored_when_anything_differs(kifs):
    """The flag is a promise about the SAME buffer and the same settings.  A changed background colour, a changed
    partition or another buffer must fall back to the full fill: frames stay exact."""
    import torch
    gui, iters, dist = _scenes(kifs)["julia"]
    screen = kifs.ScreenData(256, 128)
    cams = _cameras(kifs, 4, dist)
    with kifs.MultiGraphicState([0, 0], screen, cams[0], gui, iters=iters) as mg:
        a = torch.zeros((4, 128, 256, 4), dtype=torch.uint8, device="cuda:0")
        b = torch.zeros_like(a)
        for buf in (a, b, a):
            mg.wait(mg.render_batch_async(buf, cams, untouched=True))
        red = kifs.GuiData(max_iterations=96, fractal_group=kifs.FractalGroup.JuliaSet, constant=(-0.2, 0.6, 0.2, 0.2),
                           background_color=(200, 10, 10))
        mg.update_options(red)
        mg.render_batch_async(b, cams, untouched=True)
        mg.render_batch_async(a, cams, untouched=True)
        mg.wait_all()
        want = _single_frames(kifs, screen, red, iters, cams)
        assert torch.equal(a.cpu(), want) and torch.equal(b.cpu(), want)
        mg.set_weights([1, 2])
        mg.render_batch_async(b, cams, untouched=True)
        mg.render_batch_async(a, cams, untouched=True)
        other = torch.full_like(a, 5)
        mg.wait(mg.render_batch_async(other, cams, untouched=True))  # a buffer the library never saw
        mg.wait_all()
        assert torch.equal(a.cpu(), want) and torch.equal(b.cpu(), want) and torch.equal(other.cpu(), want)


def test_one_buffer_for_every_step_with_the_untouched_flag(kifs):
    """ONE buffer handed back on every step (waited for in between, as the contract asks) with the
    untouched flag set.  The slot of step k - 2 only knows the records IT scattered; step k - 1 (the other slot)
    scattered its own into the same buffer since, so an erase under step k - 2's records alone would leave step
    k - 1's tiles standing wherever step k is background.  Poses differ per step so that the tiles do."""
    import torch
    gui, iters, dist = _scenes(kifs)["julia"]
    screen = kifs.ScreenData(320, 200)
    B = 4
    with kifs.MultiGraphicState([0, 0, 0], screen, kifs.CameraData(), gui, iters=iters) as mg:
        a = torch.zeros((B, 200, 320, 4), dtype=torch.uint8, device="cuda:0")
        for k in range(5):
            poses = [kifs.CameraData(origin_distance=dist + 0.8 * (k % 3), phi=1.1 * k + 0.4 * i, theta=0.3 * (-1) ** k)
                     for i in range(B)]
            mg.wait(mg.render_batch_async(a, poses, untouched=True))
            assert torch.equal(a.cpu(), _single_frames(kifs, screen, gui, iters, poses)), k
        # ... and a lone kifs_multi_render into the first frame of the buffer between two steps
        mg.set_camera(kifs.CameraData(origin_distance=3.0, phi=2.0, theta=-0.5))
        mg.render(out=a[0])
        poses = _cameras(kifs, B, dist + 1.5, first=3)
        mg.wait(mg.render_batch_async(a, poses, untouched=True))
        assert torch.equal(a.cpu(), _single_frames(kifs, screen, gui, iters, poses))
        # overlapping buffers: frames 1..4 of a five-frame allocation after frames 0..3 of it
        big = torch.zeros((B + 1, 200, 320, 4), dtype=torch.uint8, device="cuda:0")
        for k, view in enumerate((big[:B], big[1:], big[:B], big[1:])):
            poses = _cameras(kifs, B, dist + 0.5 * k, first=7 * k)
            mg.wait(mg.render_batch_async(view, poses, untouched=True))
            assert torch.equal(view.cpu(), _single_frames(kifs, screen, gui, iters, poses)), k


def test_the_wrapper_orders_its_launches_after_torchs_current_stream(kifs):
    """VERDICT r03 item 5 / ADVICE r03: the library's streams are non-blocking, so nothing orders a render after the
    fill kernel torch.zeros / torch.full left on torch's current stream -- except the wrapper, which makes the launch
    stream wait for an event recorded there (kifs_order_after, kifs_multi_order_after).  A long fill is queued in
    front of the destination's own fill, on the default stream and on a side stream; a frame rendered 'at once' must
    survive both."""
    import torch
    gui, iters, dist = _scenes(kifs)["box"]
    screen = kifs.ScreenData(256, 128)
    cams = _cameras(kifs, 3, dist)
    want = _single_frames(kifs, screen, gui, iters, cams)
    side = torch.cuda.Stream()
    with kifs.GraphicState(0, screen_data=screen, camera_data=cams[0], gui_data=gui) as gs, \
            kifs.MultiGraphicState([0, 0], screen, cams[0], gui, iters=iters) as mg:
        gs.set_iters(*iters)
        for where in (None, side, None, side):
            with torch.cuda.stream(where) if where is not None else torch.cuda.stream(torch.cuda.default_stream()):
                ballast = torch.full((1 << 29,), 3, dtype=torch.uint8, device="cuda:0")  # 0.5 GB: tenths of a millisecond
                lone = torch.full((128, 256, 4), 77, dtype=torch.uint8, device="cuda:0")
                batch = [torch.full((128, 256, 4), 78, dtype=torch.uint8, device="cuda:0") for _ in range(3)]
                frames = torch.full((3, 128, 256, 4), 79, dtype=torch.uint8, device="cuda:0")
                gs.render(out=lone)                      # the context's stream, synchronous
                gs.render_batch_async(batch, cams)       # the context's stream
                mg.render_batch(frames, cams)            # the root's render and gather streams
                launch = torch.cuda.Stream()
                again = torch.full((128, 256, 4), 80, dtype=torch.uint8, device="cuda:0")
                gs.set_camera(cams[1])
                gs.render_async(again, stream=launch)    # a caller's stream
                gs.set_camera(cams[0])
            gs.synchronize()
            launch.synchronize()
            torch.cuda.synchronize()
            assert torch.equal(lone.cpu(), want[0])
            assert all(torch.equal(batch[i].cpu(), want[i]) for i in range(3))
            assert torch.equal(frames.cpu(), want)
            assert torch.equal(again.cpu(), want[1])
            del ballast


def test_stream_wait_orders_a_consumer_stream(kifs):
    import torch
    gui, iters, dist = _scenes(kifs)["julia"]
    screen = kifs.ScreenData(192, 96)
    cams = _cameras(kifs, 5, dist)
    want = _single_frames(kifs, screen, gui, iters, cams)
    with kifs.MultiGraphicState([0, 0, 0], screen, cams[0], gui, iters=iters) as mg:
        frames = torch.zeros((5, 96, 192, 4), dtype=torch.uint8, device="cuda:0")
        consumer = torch.cuda.Stream()
        step = mg.render_batch_async(frames, cams)
        mg.stream_wait(step, consumer)
        with torch.cuda.stream(consumer):
            copy = frames.clone()
        consumer.synchronize()
        assert torch.equal(copy.cpu(), want)
        mg.wait_all()


def test_rccl_inside_the_library_with_one_rank(kifs):
    """What a one-GPU box can verify of the RCCL transport: librccl opens, ncclCommInitAll over [0] succeeds,
    a grouped send-to-self / receive-from-self moves a known pattern, and a one-device step runs."""
    import torch
    gui, iters, dist = _scenes(kifs)["julia"]
    screen = kifs.ScreenData(160, 120)
    cams = _cameras(kifs, 3, dist)
    with kifs.MultiGraphicState([0], screen, cams[0], gui, iters=iters) as mg:
        mg.set_gather("sparse", "rccl")
        st = mg.stats()
        assert st["transport"] == "rccl" and st["comm_ranks"] == 1 and st["rccl_version"] > 20000
        mg.comm_selftest(1 << 20)
        mg.comm_selftest(1040 * 7)
        frames = torch.zeros((3, 120, 160, 4), dtype=torch.uint8, device="cuda:0")
        mg.render_batch(frames, cams)
        assert torch.equal(frames.cpu(), _single_frames(kifs, screen, gui, iters, cams))


def test_transport_errors_are_comm_status(kifs):
    """ncclCommInitAll refuses a device listed twice: asking for RCCL over [0, 0] is KIFS_ERR_COMM (6), not a crash,
    and the object stays usable with the copy transport."""
    import torch
    gui, iters, dist = _scenes(kifs)["julia"]
    screen = kifs.ScreenData(128, 64)
    cams = _cameras(kifs, 2, dist)
    with kifs.MultiGraphicState([0, 0], screen, cams[0], gui, iters=iters) as mg:
        with pytest.raises(kifs.KifsError) as e:
            mg.set_gather("sparse", "rccl")
        assert e.value.status == 6
        mg.set_gather("sparse", "copy")
        mg.comm_selftest(4096)
        frames = torch.zeros((2, 64, 128, 4), dtype=torch.uint8, device="cuda:0")
        mg.render_batch(frames, cams)
        assert torch.equal(frames.cpu(), _single_frames(kifs, screen, gui, iters, cams))


def test_argument_checks(kifs):
    import torch
    from kifs_raymarching_amd._lib import CameraUniform, lib
    gui, iters, dist = _scenes(kifs)["julia"]
    screen = kifs.ScreenData(128, 64)
    cams = kifs.camera_array(_cameras(kifs, 2, dist))
    frames = torch.zeros((2, 64, 128, 4), dtype=torch.uint8, device="cuda:0")
    host = np.zeros((2, 64, 128, 4), dtype=np.uint8)
    step = C.c_uint64()
    with kifs.MultiGraphicState([0, 0], screen, None, gui, iters=iters) as mg:
        m, ptr = mg._m, frames.data_ptr()
        call = lambda **kw: lib.kifs_multi_render_batch_async(
            kw.get("m", m), kw.get("count", 2), kw.get("cams", cams), kw.get("ptr", ptr), kw.get("pitch", 512),
            kw.get("stride", 64 * 512), kw.get("encode", 1), 0, C.byref(step))
        assert call(m=None) == 7 and call(count=0) == 7 and call(count=513) == 7 and call(cams=None) == 7
        assert call(ptr=None) == 7 and call(encode=2) == 7
        assert call(ptr=host.ctypes.data) == 7            # the frames must be device memory of the root
        assert call(pitch=508) == 3 and call(pitch=514) == 3 and call(stride=64 * 512 - 4) == 3
        assert lib.kifs_multi_wait(m, 0) == 7             # nothing submitted yet
        assert call() == 0 and step.value == 0
        assert lib.kifs_multi_wait(m, 1) == 7 and lib.kifs_multi_wait(m, 0) == 0
        assert lib.kifs_multi_set_gather(m, 2, 0) == 7 and lib.kifs_multi_set_gather(m, 0, 3) == 7
        assert lib.kifs_multi_stats(m, None, 0) == 7 and lib.kifs_multi_comm_selftest(m, 0) == 7
        assert lib.kifs_multi_wait_all(None) == 7 and lib.kifs_multi_stream_wait(m, 0, None) == 7
    unconfigured = C.c_int(0)
    m2 = lib.kifs_multi_create((C.c_int * 1)(0), 1, C.byref(unconfigured))
    try:
        assert lib.kifs_multi_render_batch(m2, 2, cams, frames.data_ptr(), 512, 64 * 512, 1) == 4
    finally:
        lib.kifs_multi_destroy(m2)


def test_batch_on_distinct_devices_over_rccl(kifs):
    """On a box with at least two GPUs: the default transport is RCCL, and the gathered frames still equal the
    single-device frames (skipped on a one-GPU box)."""
    import torch
    n = torch.cuda.device_count()
    if n < 2:
        pytest.skip("needs at least two GPUs")
    gui, iters, dist = _scenes(kifs)["julia"]
    screen = kifs.ScreenData(640, 360)
    cams = _cameras(kifs, 12, dist)
    want = _single_frames(kifs, screen, gui, iters, cams)
    with kifs.MultiGraphicState(list(range(n)), screen, cams[0], gui, iters=iters) as mg:
        mg.comm_selftest(1 << 20)
        bufs = [torch.zeros((12, 360, 640, 4), dtype=torch.uint8, device="cuda:0") for _ in range(2)]
        for k in range(4):
            mg.wait(mg.render_batch_async(bufs[k % 2], cams, untouched=k >= 2))
            assert torch.equal(bufs[k % 2].cpu(), want), k
        st = mg.stats()
        assert st["transport"] == "rccl" and st["comm_ranks"] == n
