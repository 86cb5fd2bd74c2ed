"""Multi-GPU sharding of the render path: load-balanced row shards of batches of frames
(ShardFrames, the default of bench.py at N > 1), contiguous row bands of one frame (BandFrame) or
whole frames of a sequence (FrameStream), all delivered to a root rank by grouped point-to-point
RCCL.

Which one to use.  A frame's run time is the critical path of its longest rays (hundreds of
dependent march steps), and every row band of the fractal still contains such rays, so bands
barely shorten a frame: they are the low-LATENCY option.  THROUGHPUT scales over frames --
the frames of an orbit are independent -- so FrameStream gives each rank whole frames and
streams the finished ones to the root; that is the default of bench.py at N > 1.

Row-band sharding of one frame over the ranks of a node, gathered to a root rank.

Pixels are independent (entry.wgsl:49-59 reads only uniforms and its own position), so
the frame shards by contiguous row bands: rank r renders rows kifs_band_range(H, r, N)
with GLOBAL pixel coordinates, which makes its band bit-identical to the same rows of a
single-GPU frame.  The only exchange step is the gather of the bands into the root's
frame: one grouped set of point-to-point transfers (root posts N-1 receives straight into
row views of its frame, every other rank posts one send).  On MI355X that is RCCL
(torch.distributed backend "nccl") and each sender->root transfer rides its own xGMI
link; a ring all-gather would be the wrong shape for point-to-point xGMI.

The reference has no multi-GPU code; this module is new.

Frames are double-buffered: the gather of frame k overlaps the render of frame k+1.
torch.distributed is plumbing here (process group, streams); with the "gloo" backend the
same code runs on CPU tensors, which is how the N>1 logic is tested without GPUs.
"""
from typing import Callable, List, Optional

import torch
import torch.distributed as dist

from .graphics import STRIPE_ROWS, band_range, shard_stripes


class BandFrame:
    def __init__(self, width: int, height: int, rank: int, world: int, device,
                 root: int = 0, buffers: int = 2, group=None):
        if width <= 0 or height <= 0 or world <= 0 or not (0 <= rank < world):
            raise ValueError("BandFrame: bad geometry")
        self.width, self.height = width, height
        self.rank, self.world, self.root = rank, world, root
        self.device = torch.device(device)
        self.group = group
        self.buffers = buffers
        self.ranges = [band_range(height, r, world) for r in range(world)]
        self.y0, self.y1 = self.ranges[rank]
        rows = self.y1 - self.y0
        if rank == root:
            # the root renders its own band in place and receives the others into row views
            self._frames = [torch.zeros((height, width, 4), dtype=torch.uint8, device=self.device)
                            for _ in range(buffers)]
            self._bands = [f[self.y0:self.y1] for f in self._frames]
        else:
            self._frames = [None] * buffers
            self._bands = [torch.zeros((rows, width, 4), dtype=torch.uint8, device=self.device)
                           for _ in range(buffers)]
        self._works: List[Optional[list]] = [None] * buffers
        # gloo cannot move device memory: rehearsals of the N > 1 path on a single GPU (two
        # ranks sharing cuda:0) stage the bands through host buffers.  RCCL never takes this path.
        self._staged = (world > 1 and self.device.type == "cuda"
                        and dist.get_backend(group) == "gloo")
        if self._staged:
            if rank == root:
                self._host = [[torch.empty((b - a, width, 4), dtype=torch.uint8) for a, b in self.ranges]
                              for _ in range(buffers)]
            else:
                self._host = [torch.empty((rows, width, 4), dtype=torch.uint8) for _ in range(buffers)]

    # -- buffers ----------------------------------------------------------------------
    def band(self, k: int) -> torch.Tensor:
        """Tensor (rows, W, 4) this rank renders into for frame k (contiguous rows)."""
        return self._bands[k % self.buffers]

    def frame(self, k: int) -> Optional[torch.Tensor]:
        """Full (H, W, 4) frame k on the root (valid after wait(k)); None elsewhere."""
        return self._frames[k % self.buffers]

    # -- exchange -----------------------------------------------------------------------
    def gather_async(self, k: int):
        """Post the band gather of frame k.  Ordered after whatever the current stream has
        enqueued (the render of this band).  Returns immediately."""
        if self.world == 1:
            return
        slot = k % self.buffers
        ops = []
        if self._staged:
            if self.rank == self.root:
                for r, (a, b) in enumerate(self.ranges):
                    if r != self.root and b > a:
                        ops.append(dist.P2POp(dist.irecv, self._host[slot][r], r, self.group))
            elif self.y1 > self.y0:
                self._host[slot].copy_(self._bands[slot])  # synchronises with the render stream
                ops.append(dist.P2POp(dist.isend, self._host[slot], self.root, self.group))
            self._works[slot] = dist.batch_isend_irecv(ops) if ops else None
            return
        if self.rank == self.root:
            frame = self._frames[slot]
            for r, (a, b) in enumerate(self.ranges):
                if r != self.root and b > a:
                    ops.append(dist.P2POp(dist.irecv, frame[a:b], r, self.group))
        elif self.y1 > self.y0:
            ops.append(dist.P2POp(dist.isend, self._bands[slot], self.root, self.group))
        self._works[slot] = dist.batch_isend_irecv(ops) if ops else None

    def wait(self, k: int):
        """Make the current stream (CPU thread for gloo) wait for the gather of frame k."""
        slot = k % self.buffers
        works = self._works[slot]
        if works:
            for w in works:
                w.wait()
            if self._staged and self.rank == self.root:
                for r, (a, b) in enumerate(self.ranges):
                    if r != self.root and b > a:
                        self._frames[slot][a:b].copy_(self._host[slot][r])
        self._works[slot] = None

    def wait_all(self):
        for slot in range(self.buffers):
            self.wait(slot)

    # -- one pipelined step ----------------------------------------------------------------
    def step(self, k: int, render_band: Callable[[torch.Tensor, int, int], None]):
        """Render this rank's band of frame k with `render_band(out, y0, y1)` and post its
        gather.  Before reusing a buffer, waits for the gather that last used it."""
        self.wait(k)  # gather of frame k - buffers
        if self.y1 > self.y0:
            render_band(self.band(k), self.y0, self.y1)
        self.gather_async(k)


class FrameStream:
    """Frame-parallel rendering: in step k every rank renders one whole frame (frame index
    k * world + rank of the sequence) into its own HBM and the root collects all `world`
    frames of the step: N-1 receives on the root, one send on every other rank, posted as one
    group so the transfers ride seven different xGMI links at once.  Double-buffered like
    BandFrame: the transfers of step k overlap the rendering of step k+1."""

    def __init__(self, width: int, height: int, rank: int, world: int, device,
                 root: int = 0, buffers: int = 2, group=None, deliver: bool = True):
        """deliver=False: frames stay in the HBM of the rank that rendered them (no exchange
        step at all: frames are independent units); frames(k) then holds only the root's own."""
        if width <= 0 or height <= 0 or world <= 0 or not (0 <= rank < world):
            raise ValueError("FrameStream: bad geometry")
        self.deliver = deliver and world > 1
        self.width, self.height = width, height
        self.rank, self.world, self.root = rank, world, root
        self.device = torch.device(device)
        self.group = group
        self.buffers = buffers
        shape = (height, width, 4)
        if rank == root:
            # slot[b][r] = frame rendered by rank r in a step using buffer b
            self._slots = [[torch.zeros(shape, dtype=torch.uint8, device=self.device)
                            for _ in range(world if self.deliver else 1)] for _ in range(buffers)]
            self._mine = [self._slots[b][root if self.deliver else 0] for b in range(buffers)]
        else:
            self._slots = None
            self._mine = [torch.zeros(shape, dtype=torch.uint8, device=self.device)
                          for _ in range(buffers)]
        self._works: List[Optional[list]] = [None] * buffers
        self._staged = (self.deliver and self.device.type == "cuda"
                        and dist.get_backend(group) == "gloo")  # rehearsal only, see BandFrame
        if self._staged:
            n = world if rank == root else 1
            self._host = [[torch.empty(shape, dtype=torch.uint8) for _ in range(n)]
                          for _ in range(buffers)]

    def frame_index(self, k: int) -> int:
        return k * self.world + self.rank

    def target(self, k: int) -> torch.Tensor:
        return self._mine[k % self.buffers]

    def frames(self, k: int):
        """On the root: the `world` frames of step k in sequence order (valid after wait(k))."""
        return self._slots[k % self.buffers] if self.rank == self.root else None

    def gather_async(self, k: int):
        if not self.deliver:
            return
        slot = k % self.buffers
        ops = []
        if self.rank == self.root:
            for r in range(self.world):
                if r != self.root:
                    dst = self._host[slot][r] if self._staged else self._slots[slot][r]
                    ops.append(dist.P2POp(dist.irecv, dst, r, self.group))
        else:
            src = self._mine[slot]
            if self._staged:
                self._host[slot][0].copy_(src)
                src = self._host[slot][0]
            ops.append(dist.P2POp(dist.isend, src, self.root, self.group))
        self._works[slot] = dist.batch_isend_irecv(ops)

    def wait(self, k: int):
        slot = k % self.buffers
        works = self._works[slot]
        if works:
            for w in works:
                w.wait()
            if self._staged and self.rank == self.root:
                for r in range(self.world):
                    if r != self.root:
                        self._slots[slot][r].copy_(self._host[slot][r])
        self._works[slot] = None

    def wait_all(self):
        for slot in range(self.buffers):
            self.wait(slot)

    def step(self, k: int, render_frame: Callable[[torch.Tensor, int], None]):
        """`render_frame(out, frame_index)` renders this rank's frame of step k."""
        self.wait(k)
        render_frame(self.target(k), self.frame_index(k))
        self.gather_async(k)


def unpack_shards_torch(frames: torch.Tensor, shards: torch.Tensor, stripes) -> None:
    """Packed shards (count, rows, W, 4) -> their rows of the frames (count, H, W, 4) with plain
    tensor indexing: what the CPU (gloo) tests use; on the GPU bench.py passes the library's unpack
    kernel (GraphicState.unpack_shard_async) instead."""
    height = frames.shape[1]
    rows = [y for s in stripes for y in range(s * STRIPE_ROWS, min(height, (s + 1) * STRIPE_ROWS))]
    frames.index_copy_(1, torch.tensor(rows, dtype=torch.long, device=frames.device), shards)


class ShardFrames:
    """Row shards of the `frames_per_step` frames of a step, gathered into the root's frames.

    The frame's 8-row stripes are dealt to the ranks in turn (kifs_shard_stripes: the expensive
    rows sit in the middle of the frame, so contiguous bands leave the outer ranks idle;
    `contiguous=True` gives every rank one run of consecutive stripes instead, for comparison).
    Rank r renders its stripes of every frame of the step into one packed buffer
    (frames, rows_r, W, 4) and sends it to the root in ONE message; the root renders its own stripes
    straight into the frames and posts one receive per peer -- a grouped set of point-to-point
    transfers, each sender -> root pair on its own xGMI link -- then copies the received stripes to
    their frame rows (`unpack`).  Pixel coordinates are global, so the gathered frames equal
    single-GPU frames byte for byte.  `weights` (one integer per rank) gives ranks unequal shares:
    the root's link-side ingest is the bottleneck of a gather, a root that renders more and
    receives less evens that out.

    Double-buffered: the transfers of step k overlap the rendering of step k + 1; the unpack of
    step k runs when its buffer slot is needed again (or at wait_all)."""

    def __init__(self, width: int, height: int, rank: int, world: int, device, frames_per_step: int = 1,
                 root: int = 0, buffers: int = 2, group=None, weights=None, contiguous: bool = False,
                 unpack: Callable = unpack_shards_torch):
        if width <= 0 or height <= 0 or world <= 0 or not (0 <= rank < world) or frames_per_step < 1:
            raise ValueError("ShardFrames: bad geometry")
        self.width, self.height = width, height
        self.rank, self.world, self.root = rank, world, root
        self.count = frames_per_step
        self.device = torch.device(device)
        self.group = group
        self.buffers = buffers
        self.unpack = unpack
        if contiguous:
            n_stripes = (height + STRIPE_ROWS - 1) // STRIPE_ROWS
            spans = [band_range(n_stripes, r, world) for r in range(world)]
            self.stripes = [list(range(a, b)) for a, b in spans]
            self.rows = [sum(min(STRIPE_ROWS, height - s * STRIPE_ROWS) for s in st) for st in self.stripes]
        else:
            dealt = [shard_stripes(height, r, world, weights) for r in range(world)]
            self.stripes = [d[0] for d in dealt]
            self.rows = [d[1] for d in dealt]
        assert sum(self.rows) == height
        shape = lambda r: (self.count, self.rows[r], width, 4)
        if rank == root:
            self._frames = [torch.zeros((self.count, height, width, 4), dtype=torch.uint8, device=self.device)
                            for _ in range(buffers)]
            self._recv = [[torch.zeros(shape(r), dtype=torch.uint8, device=self.device) if r != root else None
                           for r in range(world)] for _ in range(buffers)]
            self._mine = None
        else:
            self._frames = [None] * buffers
            self._mine = [torch.zeros(shape(rank), dtype=torch.uint8, device=self.device) for _ in range(buffers)]
        self._works: List[Optional[list]] = [None] * buffers
        self._staged = (world > 1 and self.device.type == "cuda"
                        and dist.get_backend(group) == "gloo")  # rehearsal on one GPU, see BandFrame
        if self._staged:
            if rank == root:
                self._host = [[torch.empty(shape(r), dtype=torch.uint8) if r != root else None
                               for r in range(world)] for _ in range(buffers)]
            else:
                self._host = [torch.empty(shape(rank), dtype=torch.uint8) for _ in range(buffers)]

    @property
    def my_stripes(self):
        return self.stripes[self.rank]

    def targets(self, k: int):
        """(tensors to render frame 0..count-1 of step k into, in_place): the root renders into its
        frames at the rows' frame positions, a peer into its packed shards."""
        slot = k % self.buffers
        if self.rank == self.root:
            return [self._frames[slot][i] for i in range(self.count)], True
        return [self._mine[slot][i] for i in range(self.count)], False

    def frames(self, k: int) -> Optional[torch.Tensor]:
        """(count, H, W, 4): the gathered frames of step k on the root (valid after wait(k))."""
        return self._frames[k % self.buffers]

    def gather_async(self, k: int):
        if self.world == 1:
            return
        slot = k % self.buffers
        ops = []
        if self.rank == self.root:
            for r in range(self.world):
                if r != self.root and self.rows[r] > 0:
                    dst = self._host[slot][r] if self._staged else self._recv[slot][r]
                    ops.append(dist.P2POp(dist.irecv, dst, r, self.group))
        elif self.rows[self.rank] > 0:
            src = self._mine[slot]
            if self._staged:
                self._host[slot].copy_(src)  # synchronises with the render stream
                src = self._host[slot]
            ops.append(dist.P2POp(dist.isend, src, self.root, self.group))
        self._works[slot] = dist.batch_isend_irecv(ops) if ops else None

    def wait(self, k: int):
        """Make the current stream (CPU thread for gloo) wait for the gather of step k; on the root,
        move the received stripes to their frame rows."""
        slot = k % self.buffers
        works = self._works[slot]
        if works:
            for w in works:
                w.wait()
            if self.rank == self.root:
                for r in range(self.world):
                    if r != self.root and self.rows[r] > 0:
                        if self._staged:
                            self._recv[slot][r].copy_(self._host[slot][r])
                        self.unpack(self._frames[slot], self._recv[slot][r], self.stripes[r])
        self._works[slot] = None

    def wait_all(self):
        for slot in range(self.buffers):
            self.wait(slot)

    def step(self, k: int, render_shard: Callable):
        """`render_shard(outs, first_frame, stripes, in_place)` renders this rank's stripes of the
        step's frames (sequence indices first_frame .. + count - 1)."""
        self.wait(k)
        if self.rows[self.rank] > 0:
            outs, in_place = self.targets(k)
            render_shard(outs, k * self.count, self.my_stripes, in_place)
        self.gather_async(k)
