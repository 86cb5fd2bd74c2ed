"""Row-band sharding of one frame over the ranks of a node, gathered to a root rank.

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

from .graphics import band_range


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
