"""Multi-GPU sharding of the render path, one process per GPU over torch.distributed: load-balanced row
shards of batches of frames with a sparse gather (SparseShardFrames: the default of bench.py at N > 1) or a
dense one (ShardFrames), contiguous row bands of one frame (BandFrame), or whole frames of a sequence
(FrameStream), all delivered to a root rank by grouped point-to-point RCCL.  (One process driving all the
devices uses the library's own kifs_multi_render_batch_async instead: graphics.MultiGraphicState.)

Which one to use.  THROUGHPUT comes from batches: every rank renders its 8-row stripes of all the step's
frames in ONE launch (the expensive rows sit in the middle of the frame, so stripes are dealt round-robin
rather than cut into contiguous bands) and rank 0 gathers them -- SparseShardFrames sends only the tiles
that hold something, which is what keeps the root's one xGMI link per peer from being the bound.  A lone
frame's run time is the critical path of its longest rays (hundreds of dependent march steps), and every
row band of the fractal still contains such rays, so BandFrame barely shortens a frame: it is the
low-LATENCY form.  FrameStream gives each rank whole frames and (optionally) streams the finished ones to
the root: frame-parallel, no gathered frame -- bench.py reports it as a secondary figure only.

Hand-off of gathered frames (ShardFrames / SparseShardFrames).  frames(k) is complete and stable from the
moment wait(k) returns until the slot is rendered into again, i.e. until step(k + buffers) starts; inside a
step() loop that window is empty (step(k + buffers) itself calls wait(k) and then overwrites).  A consumer
therefore registers `on_frames(k, frames)`: it is called from wait(k) right after the last stripe / record has
been enqueued on the current stream, BEFORE the slot is reused, and may return an event (recorded on whatever
stream the consumer reads the frames with); the slot's next fill and render are ordered after that event.

BandFrame: row-band sharding of one frame over the ranks of a node, gathered to a root rank.

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
            self._mine = None
        else:
            self._frames = [None] * buffers
            self._mine = [torch.zeros(shape(rank), dtype=torch.uint8, device=self.device) for _ in range(buffers)]
        self._works: List[Optional[list]] = [None] * buffers
        self._targets = [None] * buffers
        self._step_of_slot = [None] * buffers  # which step's frames a slot holds (for on_frames)
        self._consumed = [None] * buffers      # event returned by on_frames: the slot may be overwritten after it
        self.on_frames: Optional[Callable] = None  # on_frames(k, frames) -> optional event (see the module docstring)
        self.wrap_targets = lambda outs: outs  # e.g. graphics.DevicePointers: the pointers collected once per slot
        self._staged = (world > 1 and self.device.type == "cuda"
                        and dist.get_backend(group) == "gloo")  # rehearsal on one GPU, see BandFrame
        self._alloc_transfer_buffers()

    def _alloc_transfer_buffers(self):
        """What the root receives into (and, for a gloo rehearsal with device tensors, the host staging)."""
        rank, root, world, buffers = self.rank, self.root, self.world, self.buffers
        shape = lambda r: (self.count, self.rows[r], self.width, 4)
        if rank == root:
            self._recv = [[torch.zeros(shape(r), dtype=torch.uint8, device=self.device) if r != root else None
                           for r in range(world)] for _ in range(buffers)]
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
        if self._targets[slot] is None:  # the same tensors every time the slot comes round
            src = self._frames[slot] if self.rank == self.root else self._mine[slot]
            self._targets[slot] = self.wrap_targets([src[i] for i in range(self.count)])
        return self._targets[slot], self.rank == self.root

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
        self._hand_off(slot)

    def _hand_off(self, slot: int):
        """Root: the slot's frames are complete on the current stream -- give them to the consumer, once."""
        k = self._step_of_slot[slot]
        if k is None or self.rank != self.root:
            return
        self._step_of_slot[slot] = None
        if self.on_frames is not None:
            self._consumed[slot] = self.on_frames(k, self._frames[slot])

    def _before_overwrite(self, slot: int):
        """Order the current stream after the consumer of the slot's previous frames."""
        ev = self._consumed[slot]
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)
            self._consumed[slot] = None

    def wait_all(self):
        for slot in range(self.buffers):
            self.wait(slot)

    def step(self, k: int, render_shard: Callable):
        """`render_shard(outs, first_frame, stripes, in_place)` renders this rank's stripes of the
        step's frames (sequence indices first_frame .. + count - 1)."""
        self.wait(k)
        self._before_overwrite(k % self.buffers)
        if self.rows[self.rank] > 0:
            outs, in_place = self.targets(k)
            render_shard(outs, k * self.count, self.my_stripes, in_place)
        self._step_of_slot[k % self.buffers] = k
        self.gather_async(k)


# ---- sparse shards: only the tiles that hold something cross the links -------------------------------
SPARSE_RECORD_BYTES = 1040  # KIFS_SPARSE_RECORD_BYTES: uint32 id, three zero words, 8 rows of 32 RGBA8 pixels
_TILE_W = 32


def _tiles_x(width: int) -> int:
    return (width + _TILE_W - 1) // _TILE_W


def pack_sparse_torch(shards: torch.Tensor, stripes, height: int, background_rgba: int) -> torch.Tensor:
    """CPU form of kifs_pack_sparse_async (tests and the gloo path): (count, rows, W, 4) packed shards ->
    (n, 1040) uint8 records of the tiles that hold a pixel other than `background_rgba` (little-endian
    R | G << 8 | B << 16 | A << 24), in tile-id order."""
    count, rows, width, _ = shards.shape
    tx, ns = _tiles_x(width), len(stripes)
    bg = torch.tensor([(background_rgba >> s) & 255 for s in (0, 8, 16, 24)], dtype=torch.uint8)
    padded = bg.expand(count, ns * STRIPE_ROWS, tx * _TILE_W, 4).clone()
    padded[:, :rows, :width] = shards.cpu()
    tiles = padded.view(count, ns, STRIPE_ROWS, tx, _TILE_W, 4).permute(0, 1, 3, 2, 4, 5).reshape(-1, STRIPE_ROWS * _TILE_W * 4)
    keep = (tiles.view(-1, STRIPE_ROWS * _TILE_W, 4) != bg).any(-1).any(-1).nonzero().flatten()
    records = torch.zeros((keep.numel(), SPARSE_RECORD_BYTES), dtype=torch.uint8)
    records[:, :4] = keep.to(torch.int32).view(-1, 1).contiguous().view(torch.uint8).view(-1, 4)
    records[:, 16:] = tiles[keep]
    return records


def unpack_sparse_torch(frames: torch.Tensor, records: torch.Tensor, stripes) -> None:
    """CPU form of kifs_unpack_sparse_async: records -> their rows of frames (count, H, W, 4)."""
    count, height, width, _ = frames.shape
    tx, ns = _tiles_x(width), len(stripes)
    ids = records[:, :4].contiguous().view(torch.int32).flatten().tolist()
    for rec, tid in zip(records, ids):
        if not (0 <= tid < count * ns * tx):
            continue
        shard, rest = divmod(tid, ns * tx)
        k, col = divmod(rest, tx)
        y0, x0 = stripes[k] * STRIPE_ROWS, col * _TILE_W
        h, w = min(STRIPE_ROWS, height - y0), min(_TILE_W, width - x0)
        frames[shard, y0:y0 + h, x0:x0 + w] = rec[16:].view(STRIPE_ROWS, _TILE_W, 4)[:h, :w]


def erase_sparse_torch(frames: torch.Tensor, records: torch.Tensor, stripes, background_rgba: int) -> None:
    """CPU form of kifs_erase_sparse_async: the background over the records' tiles."""
    blank = records.clone()
    blank[:, 16:] = torch.tensor([(background_rgba >> s) & 255 for s in (0, 8, 16, 24)], dtype=torch.uint8).repeat(STRIPE_ROWS * _TILE_W)
    unpack_sparse_torch(frames, blank, stripes)


def fill_stripes_torch(frames: torch.Tensor, stripes, background_rgba: int) -> None:
    bg = torch.tensor([(background_rgba >> s) & 255 for s in (0, 8, 16, 24)], dtype=torch.uint8, device=frames.device)
    for s in stripes:
        frames[:, s * STRIPE_ROWS:(s + 1) * STRIPE_ROWS] = bg


class SparseShardFrames(ShardFrames):
    """ShardFrames whose peers send only the 32 x 8 tiles that hold something.

    The root of the gather takes each peer's rows over one xGMI link (~19 Gpixel/s inbound), a GPU renders
    several times faster, and most of a frame of these scenes is the background colour: dense shards make
    the links the bottleneck (two GPUs slower than one unless the root renders most of the frame itself).
    Here a peer packs its shards into records of their non-background tiles (kifs_pack_sparse_async), the
    root fills the peers' rows with the background (kifs_fill_shard_async, on `fill_stream` beside its own
    rendering) and scatters the received records over it (kifs_unpack_sparse_async).  Lossless: the
    gathered frames are the single-GPU frames byte for byte, whatever they show.

    Message sizes depend on the frames, and point-to-point transfers need them on the host of both sides:
    the number of records of step k reaches the peer's host asynchronously (pinned memory), and is
    exchanged -- one small gather over `count_group`, a CPU (gloo) group -- while step k + 1 is already
    rendering; then the payload of step k is posted.  So transfers lag the rendering by one step, exactly as
    the dense form's do, and nothing on the GPU waits for the host.

    A frame buffer that comes round again still holds the background everywhere but under its previous
    records: with `erase` given, only those tiles are reset (2 % of the headline's) instead of every row.

    pack(shards, stripes, records) -> callable returning the number of records (may block until known);
    unpack_sparse(frames, records, n, stripes); fill(frames, stripes); erase(frames, records, n, stripes)
    (optional): device forms from GraphicState
    (bench.py) or the *_torch functions above bound to a background pixel (CPU tests)."""

    def __init__(self, width: int, height: int, rank: int, world: int, device, frames_per_step: int = 1,
                 root: int = 0, buffers: int = 2, group=None, weights=None, contiguous: bool = False,
                 pack: Callable = None, unpack_sparse: Callable = None, fill: Callable = None, erase: Callable = None,
                 count_group=None, fill_stream=None):
        if pack is None or unpack_sparse is None or fill is None:
            raise ValueError("SparseShardFrames: pack, unpack_sparse and fill are required")
        if buffers < 2:
            raise ValueError("SparseShardFrames: transfers lag the rendering by a step: at least two buffers")
        super().__init__(width, height, rank, world, device, frames_per_step, root, buffers, group, weights, contiguous)
        self.pack, self.unpack_sparse, self.fill, self.erase = pack, unpack_sparse, fill, erase
        self._filled = [False] * buffers      # root: the slot's frames hold the background outside its last records
        self.count_group = count_group if count_group is not None else group
        self.fill_stream = fill_stream
        self._count_fn = [None] * buffers     # peer: the step's record count, once the host may know it
        self._counts = [None] * buffers       # per rank, after the exchange
        self._slot_free = [None] * buffers    # root: the slot's frames may be overwritten (event on the render stream)
        self._fill_done = [None] * buffers
        self._unflushed = None                # the step whose payload has not been posted yet
        self.records_sent = 0                 # statistics: records received and tiles they stand for
        self.tiles_seen = 0

    def _alloc_transfer_buffers(self):
        rank, root, world, buffers, dev = self.rank, self.root, self.world, self.buffers, self.device
        self.capacity = [self.count * len(self.stripes[r]) * _tiles_x(self.width) for r in range(world)]
        records = lambda r, **kw: torch.empty((max(1, self.capacity[r]), SPARSE_RECORD_BYTES), dtype=torch.uint8, **kw)
        if rank == root:
            self._recv = [[records(r, device=dev) if r != root and self.capacity[r] else None for r in range(world)]
                          for _ in range(buffers)]
            self.peer_stripes = sorted(s for r in range(world) if r != root for s in self.stripes[r])
        else:
            self._records = [records(rank, device=dev) for _ in range(buffers)]
        if self._staged:
            if rank == root:
                self._host = [[records(r) if r != root and self.capacity[r] else None for r in range(world)]
                              for _ in range(buffers)]
            else:
                self._host = [records(rank) for _ in range(buffers)]

    # -- the two halves of a step's gather
    def _start_fill(self, k: int):
        """Root, before step k's own rendering: the background under the peers' rows, on the fill stream
        (beside the rendering: disjoint rows) once the slot's previous frames have been consumed."""
        slot = k % self.buffers
        if self.rank != self.root or self.world == 1 or not self.peer_stripes:
            return
        def background():
            if self.erase is not None and self._filled[slot]:  # only where the slot's previous records went
                counts = self._counts[slot]
                for r in range(self.world):
                    if r != self.root and counts[r] > 0:
                        self.erase(self._frames[slot], self._recv[slot][r], counts[r], self.stripes[r])
            else:
                self.fill(self._frames[slot], self.peer_stripes)
                self._filled[slot] = True
        if self.fill_stream is not None:
            self.fill_stream.wait_event(self._slot_free[slot] or torch.cuda.current_stream().record_event())
            if self._consumed[slot] is not None:  # the consumer of the slot's previous frames reads them until here
                self.fill_stream.wait_event(self._consumed[slot])
            with torch.cuda.stream(self.fill_stream):
                background()
                self._fill_done[slot] = self.fill_stream.record_event()
        else:
            if self._consumed[slot] is not None:
                torch.cuda.current_stream().wait_event(self._consumed[slot])
            background()

    def _pack(self, k: int):
        """Peer, after step k's rendering is enqueued."""
        slot = k % self.buffers
        if self.rank != self.root and self.rows[self.rank] > 0:
            self._count_fn[slot] = self.pack(self._mine[slot], self.my_stripes, self._records[slot])

    def _flush(self, k: int):
        """Exchange step k's record counts (host) and post its transfers."""
        slot = k % self.buffers
        mine = 0
        if self.rank != self.root and self.rows[self.rank] > 0:
            mine = int(self._count_fn[slot]())
        t = torch.tensor([mine], dtype=torch.int64)
        if self.world > 1:
            backend = dist.get_backend(self.count_group)
            if backend == "nccl":  # no CPU group was given: the counts travel through device memory
                every = torch.empty(self.world, dtype=torch.int64, device=self.device)
                dist.all_gather_into_tensor(every, t.to(self.device), group=self.count_group)
                counts = every.tolist()
            else:
                gathered = [torch.zeros(1, dtype=torch.int64) for _ in range(self.world)] if self.rank == self.root else None
                dist.gather(t, gathered, dst=self.root, group=self.count_group)
                counts = [int(g.item()) for g in gathered] if gathered else None
        else:
            counts = [0]
        self._counts[slot] = counts
        ops = []
        if self.rank == self.root:
            if self._fill_done[slot] is not None:
                # the erase reads the slot's previous records: it precedes the receives that overwrite them
                torch.cuda.current_stream().wait_event(self._fill_done[slot])
            for r in range(self.world):
                if r != self.root and counts[r] > 0:
                    if counts[r] > self.capacity[r]:
                        raise RuntimeError(f"SparseShardFrames: rank {r} announced {counts[r]} records, capacity {self.capacity[r]}")
                    dst = self._host[slot][r] if self._staged else self._recv[slot][r]
                    ops.append(dist.P2POp(dist.irecv, dst[:counts[r]], r, self.group))
                    self.records_sent += counts[r]
                self.tiles_seen += self.capacity[r] if r != self.root else 0
        elif mine > 0:
            src = self._records[slot][:mine]
            if self._staged:
                self._host[slot][:mine].copy_(src)
                src = self._host[slot][:mine]
            ops.append(dist.P2POp(dist.isend, src, self.root, self.group))
        self._works[slot] = dist.batch_isend_irecv(ops) if ops else []

    def gather_async(self, k: int):
        """Both halves at once (tests, calibration): the host waits for step k's count."""
        if self._unflushed is not None:
            self._flush(self._unflushed)
            self._unflushed = None
        self._start_fill(k)
        self._pack(k)
        self._flush(k)

    def wait(self, k: int):
        slot = k % self.buffers
        if self._unflushed is not None and self._unflushed % self.buffers == slot:
            self._flush(self._unflushed)
            self._unflushed = None
        works = self._works[slot]
        if works is not None:
            for w in works:
                w.wait()
            if self.rank == self.root and self.world > 1:
                if self._fill_done[slot] is not None:
                    torch.cuda.current_stream().wait_event(self._fill_done[slot])
                    self._fill_done[slot] = None
                counts = self._counts[slot]
                for r in range(self.world):
                    if r != self.root and counts[r] > 0:
                        if self._staged:
                            self._recv[slot][r][:counts[r]].copy_(self._host[slot][r][:counts[r]])
                        self.unpack_sparse(self._frames[slot], self._recv[slot][r], counts[r], self.stripes[r])
                if self.fill_stream is not None:
                    self._slot_free[slot] = torch.cuda.current_stream().record_event()
        self._works[slot] = None
        if works is not None:
            self._hand_off(slot)

    def wait_all(self):
        if self._unflushed is not None:
            self._flush(self._unflushed)
            self._unflushed = None
        for slot in range(self.buffers):
            self.wait(slot)

    def step(self, k: int, render_shard: Callable):
        self.wait(k)
        self._start_fill(k)
        self._before_overwrite(k % self.buffers)  # (after _start_fill, which orders the fill stream after it too)
        if self.rows[self.rank] > 0:
            outs, in_place = self.targets(k)
            render_shard(outs, k * self.count, self.my_stripes, in_place)
        self._step_of_slot[k % self.buffers] = k
        self._pack(k)
        if self._unflushed is not None:  # step k - 1: its count has had a whole step to reach the host
            self._flush(self._unflushed)
        self._unflushed = k
