"""Multi-GPU partition of the render path: one process per GPU, contiguous
x-strips, gathered to rank 0 with torch.distributed (backend "nccl" = RCCL
over xGMI on the GPU node; "gloo" in the CPU tests).

This is the reference's static partitioning (PARTIONING_STRATEGY 1,
src/RayTracer.cpp:904-923: rank r renders one contiguous strip of the image)
with GPUs in place of tiles.  The reference splits z; the framebuffer is
x-major (pixels[x][z], src/RayTracer.h:44), so splitting x instead makes every
strip one contiguous block and rank order equal to memory order: the gather
needs no repacking.
"""
import torch
import torch.distributed as dist


def strip_bounds(W, world, rank):
    """(x0, x1, strip): rank renders columns [x0, x1); every rank's buffer holds
    `strip` = ceil(W / world) columns so the gather has equal counts (only
    trailing strips can be short or empty)."""
    strip = (W + world - 1) // world
    return min(rank * strip, W), min((rank + 1) * strip, W), strip


def alloc_full(W, H, world, device, dtype=torch.float32):
    """Rank 0's gathered framebuffer: world * strip columns (>= W), and the
    per-rank views into it that the gather writes."""
    _, _, strip = strip_bounds(W, world, 0)
    full = torch.empty((strip * world, H, 3), dtype=dtype, device=device)
    views = [full[r * strip:(r + 1) * strip] for r in range(world)]
    return full, views


def gather_strips(strip_buf, views, dst=0, async_op=False):
    """Gather every rank's strip buffer into rank `dst`'s views (None elsewhere).
    With async_op the collective runs on the backend's own stream and the
    returned work handle must be wait()ed before strip_buf is written again."""
    return dist.gather(strip_buf, views if dist.get_rank() == dst else None, dst=dst, async_op=async_op)


class StripPipeline:
    """Render/gather software pipeline over successive frames.

    Frames are independent, so while RCCL moves frame k's strips to rank 0 (on
    its own stream, over xGMI) the render kernel of frame k+1 already runs on
    the compute stream.  Two strip buffers alternate; a buffer is rendered into
    again only after the gather that read it has completed.  `render(buf)` is
    the caller's function that enqueues the kernel writing `buf`.
    """

    def __init__(self, W, H, world, rank, device, render, overlap=True, dtype=torch.float32, force_gather=False):
        self.world, self.rank, self.render, self.overlap = world, rank, render, overlap
        self.gather = world > 1 or force_gather          # force_gather: run the collective even with one rank
        self.x0, self.x1, self.strip = strip_bounds(W, world, rank)
        n_buf = 2 if (overlap and self.gather) else 1
        self.bufs = [torch.empty((self.strip, H, 3), dtype=dtype, device=device) for _ in range(n_buf)]
        self.pending = [None] * n_buf
        self.full, self.views = (alloc_full(W, H, world, device, dtype) if (self.gather and rank == 0)
                                 else (None, None))
        self.k = 0

    def step(self):
        b = self.k % len(self.bufs)
        self.k += 1
        if self.pending[b] is not None:
            self.pending[b].wait()
            self.pending[b] = None
        self.render(self.bufs[b])
        if self.gather:
            if self.overlap:
                self.pending[b] = gather_strips(self.bufs[b], self.views, dst=0, async_op=True)
            else:
                gather_strips(self.bufs[b], self.views, dst=0)

    def drain(self):
        for i, w in enumerate(self.pending):
            if w is not None:
                w.wait()
                self.pending[i] = None

    def image(self, W):
        """Rank 0: the gathered framebuffer (first W columns); single GPU: the strip."""
        self.drain()
        if not self.gather:
            return self.bufs[(self.k - 1) % len(self.bufs)]
        return self.full[:W] if self.full is not None else None
