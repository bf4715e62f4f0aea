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


def equal_bounds(W, world):
    """[(x0, x1)] per rank for the equal partition (trailing strips may be short or empty)."""
    return [strip_bounds(W, world, r)[:2] for r in range(world)]


def balanced_bounds(W, world, column_cost, send_cost_per_column, root=0, overlap=True):
    """Contiguous strips, in rank order, that minimise the frame time:

        rank `root`:  sum of column_cost over its strip          (it sends nothing)
        other ranks:  max(that sum, columns * send_cost_per_column)   overlap=True: a pipeline of
                      frames, the gather of frame k under the render of frame k+1
                      that sum + columns * send_cost_per_column       overlap=False: ONE frame,
                      a peer's columns leave after its kernel (max-rank
                      kernel + gather)

    column_cost[x] is the (measured) render time of image column x on one GPU,
    send_cost_per_column the (measured) time one peer needs to deliver one column
    to the root over its own link.  When a link is slower than a GPU renders --
    this kernel produces 200 GB/s of pixels, an xGMI link carries about 75 -- the
    root takes a larger strip and every peer just as many columns as its link can
    carry in that time.  With send_cost_per_column = 0 and a flat cost this is the
    equal partition.  Bisection on the frame time with a greedy fill; pure
    arithmetic on inputs every rank holds, so every rank computes the same bounds.
    Only root == 0 keeps rank order equal to memory order, which is what the
    callers use."""
    import numpy as np
    cost = np.maximum(np.asarray(column_cost, dtype=np.float64), 0.0)
    assert cost.shape == (W,) and world >= 1
    g = max(float(send_cost_per_column), 0.0)
    prefix = np.concatenate([[0.0], np.cumsum(cost)])

    def fill(limit):
        bounds, x = [], 0
        for r in range(world):
            # furthest x1 with prefix[x1] - prefix[x] <= limit
            if r != root and g > 0.0 and not overlap:
                # render + send <= limit: prefix[x1] + g * x1 <= prefix[x] + g * x + limit (monotone in x1)
                serial = prefix + g * np.arange(W + 1)
                x1 = int(np.searchsorted(serial, serial[x] + limit, side="right")) - 1
            else:
                x1 = int(np.searchsorted(prefix, prefix[x] + limit, side="right")) - 1
            x1 = max(x, min(W, x1))
            if r != root and g > 0.0 and overlap:
                x1 = min(x1, x + int(limit / g))
            bounds.append((x, x1))
            x = x1
        return bounds, x

    lo, hi = 0.0, float(prefix[-1]) + g * W + 1e-9
    for _ in range(60):
        mid = 0.5 * (lo + hi)
        if fill(mid)[1] >= W:
            hi = mid
        else:
            lo = mid
    bounds, reached = fill(hi)
    if reached < W:                                   # numerical corner: give the rest to the last rank
        bounds[-1] = (bounds[-1][0], W)
    return bounds


def gather_uneven(strip_buf, views, bounds, dst=0, async_op=False):
    """Strips of different widths to rank `dst`: every other rank sends its
    columns, `dst` receives each peer's directly into that peer's view of the full
    image (point-to-point over the same transport a gather uses).  `dst` renders
    straight into its own view, so there is nothing to move for it.  Returns the
    work handles (wait() them before strip_buf is written again)."""
    rank = dist.get_rank()
    ops = []
    if rank == dst:
        for r, (x0, x1) in enumerate(bounds):
            if r != dst and x1 > x0:
                ops.append(dist.P2POp(dist.irecv, views[r], r))
    else:
        x0, x1 = bounds[rank]
        if x1 > x0:
            ops.append(dist.P2POp(dist.isend, strip_buf[: x1 - x0], dst))
    works = dist.batch_isend_irecv(ops) if ops else []
    if not async_op:
        for w in works:
            w.wait()
        return []
    return works


def measure_and_balance(pipe, W, my_kernel_ms, sync, device, overlap=True):
    """Called by every rank between two warm-up frames on the EQUAL partition.
    Times one gather on its own (nothing else in flight; `sync()` must drain the
    device and end with a barrier), shares every rank's kernel time, and returns
    (bounds, note): the measured-cost partition of balanced_bounds() -- the same
    on every rank -- and a sentence for the bench record."""
    import time
    import numpy as np
    world = dist.get_world_size()
    sync()
    t0 = time.perf_counter()
    gather_strips(pipe.bufs[0], pipe.views, dst=0)
    sync()
    gather_ms = (time.perf_counter() - t0) * 1e3
    mine = torch.tensor([float(my_kernel_ms), gather_ms], dtype=torch.float64, device=device)
    everyone = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(everyone, mine)
    kernel_by_rank = [float(t[0]) for t in everyone]
    gather_ms = float(everyone[0][1])                    # rank 0's clock: it is the receiver
    eq = equal_bounds(W, world)
    cost = np.zeros(W, dtype=np.float64)
    for (a, b), k in zip(eq, kernel_by_rank):
        if b > a:
            cost[a:b] = max(k, 0.0) / (b - a)
    per_column_send = gather_ms / max(eq[0][1] - eq[0][0], 1)
    bounds = balanced_bounds(W, world, cost, per_column_send, overlap=overlap)
    note = (f"re-cut after warm-up frames on the equal partition: kernel ms per rank "
            f"{[round(k, 3) for k in kernel_by_rank]}, gather alone {gather_ms:.3f} ms")
    return bounds, note


class StripPipeline:
    """Render/gather software pipeline over successive frames.

    Frames are independent, so while RCCL moves frame k's strips to rank 0 (on
    its own stream, over xGMI) the render kernel of frame k+1 already runs on
    the compute stream.  Two strip buffers alternate; a buffer is rendered into
    again only after the gather that read it has completed.  `render(buf)` is
    the caller's function that enqueues the kernel writing `buf`.
    """

    def __init__(self, W, H, world, rank, device, render, overlap=True, dtype=torch.float32, force_gather=False,
                 bounds=None):
        self.world, self.rank, self.render, self.overlap = world, rank, render, overlap
        self.gather = world > 1 or force_gather          # force_gather: run the collective even with one rank
        self.x0, self.x1, self.strip = strip_bounds(W, world, rank)
        self.bounds = None                               # None: the equal partition and one gather collective
        if bounds is not None and list(bounds) != equal_bounds(W, world):
            assert len(bounds) == world and bounds[0][0] == 0 and bounds[-1][1] == W
            assert all(bounds[r][1] == bounds[r + 1][0] for r in range(world - 1))
            self.bounds = [tuple(b) for b in bounds]
            self.x0, self.x1 = self.bounds[rank]
            self.strip = max(self.x1 - self.x0, 1)
        n_buf = 2 if (overlap and self.gather) else 1
        self.full, self.views = None, None
        if self.gather and rank == 0:
            if self.bounds is None:
                self.full, self.views = alloc_full(W, H, world, device, dtype)
            else:
                self.full = torch.empty((W, H, 3), dtype=dtype, device=device)
                self.views = [self.full[a:b] for a, b in self.bounds]
        if self.bounds is not None and rank == 0 and self.gather:
            # rank 0 renders in place: its columns are not touched by the receives, so one buffer does
            self.bufs = [self.views[0] if self.x1 > self.x0 else torch.empty((1, H, 3), dtype=dtype, device=device)]
        else:
            self.bufs = [torch.empty((self.strip, H, 3), dtype=dtype, device=device) for _ in range(n_buf)]
        self.pending = [None] * n_buf                    # frame k waits for the gather of frame k - len(pending)
        self.k = 0

    def _wait(self, b):
        p = self.pending[b]
        if p is not None:
            for w in (p if isinstance(p, (list, tuple)) else [p]):
                w.wait()
            self.pending[b] = None

    def step(self):
        b, slot = self.k % len(self.bufs), self.k % len(self.pending)
        self.k += 1
        self._wait(slot)
        self.render(self.bufs[b])
        if self.gather:
            if self.bounds is None:
                work = gather_strips(self.bufs[b], self.views, dst=0, async_op=self.overlap)
            else:
                work = gather_uneven(self.bufs[b], self.views, self.bounds, dst=0, async_op=self.overlap)
            if self.overlap:
                self.pending[slot] = work

    def drain(self):
        for b in range(len(self.pending)):
            self._wait(b)

    def image(self, W):
        """Rank 0: the gathered framebuffer (first W columns); single GPU: the strip."""
        self.drain()
        if not self.gather:
            return self.bufs[(self.k - 1) % len(self.bufs)]
        return self.full[:W] if self.full is not None else None

    def describe(self):
        """The partition in words (bench.py's config.partition)."""
        if self.bounds is None:
            return f"{self.world} equal x-strip(s) of {self.strip} columns"
        return f"{self.world} x-strips of " + "/".join(str(b - a) for a, b in self.bounds) + " columns (measured-cost partition)"
