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


def chunk_bounds(x0, x1, chunks, align=1):
    """[(a, b)] * chunks: columns [x0, x1) cut into `chunks` contiguous column chunks of about equal
    width (trailing ones may be empty) whose inner boundaries lie a multiple of `align` columns from x0
    (the kernel's wavefront tiles are `align` columns wide: a chunk then never splits one).  The same
    arithmetic as rt_chunk_bounds() of the C ABI (csrc/rt_multi.hip)."""
    n, K, align = max(x1 - x0, 0), max(int(chunks), 1), max(int(align), 1)
    units = -(-n // align)
    return [(x0 + min(n, (units * k // K) * align), x0 + min(n, (units * (k + 1) // K) * align)) for k in range(K)]


def balanced_bounds(W, world, column_cost, send_cost_per_column, root=0, overlap=True, chunks=1):
    """Contiguous strips, in rank order, that minimise the frame time:

        rank `root`:  sum of column_cost over its strip          (it sends nothing)
        other ranks:  max(that sum, columns * send_cost_per_column)   overlap=True: a pipeline of
                      frames, the gather of frame k under the render of frame k+1
                      that sum + columns * send_cost_per_column       overlap=False: ONE frame,
                      a peer's columns leave after its kernel (max-rank
                      kernel + gather)
                      max(R, S) + min(R, S) / chunks                   overlap=False, chunks > 1: ONE
                      frame whose strip is rendered and sent in `chunks` column chunks, chunk k on its
                      way while chunk k + 1 is rendered (R = that sum, S = the send term): only the
                      first chunk of the slower activity is not covered by the other

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

    K = max(int(chunks), 1)

    def rank_time(r, x, x1):
        render = prefix[x1] - prefix[x]
        if r == root or g <= 0.0:
            return render
        send = g * (x1 - x)
        if overlap:
            return max(render, send)
        if K <= 1:
            return render + send
        return max(render, send) + min(render, send) / K

    def fill(limit):
        bounds, x = [], 0
        for r in range(world):
            # furthest x1 with rank_time(r, x, x1) <= limit (monotone in x1)
            lo_, hi_ = x, W
            while lo_ < hi_:
                mid_ = (lo_ + hi_ + 1) // 2
                if rank_time(r, x, mid_) <= limit:
                    lo_ = mid_
                else:
                    hi_ = mid_ - 1
            bounds.append((x, lo_))
            x = lo_
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


def suggest_chunks(kernel_ms, send_ms, most=8):
    """Column chunks per strip for the single-frame mode: 1 while a strip's send is short next to its
    kernel (every chunk is a launch of its own: fewer tiles per launch, a launch gap each), up to `most`
    where the send is as long as the kernel or longer (the built-in scene: 25 MB over a link against
    0.2 ms of rendering)."""
    if kernel_ms <= 0.0:
        return most if send_ms > 0.0 else 1
    return int(min(most, max(1, round(4.0 * send_ms / kernel_ms))))


def measure_and_balance(pipe, W, my_kernel_ms, sync, device, overlap=True, chunks=1):
    """Called by every rank between two warm-up frames on the EQUAL partition.
    Times one gather on its own (nothing else in flight; `sync()` must drain the
    device and end with a barrier), shares every rank's kernel time, and returns
    (bounds, note, chunks): the measured-cost partition of balanced_bounds() and
    the chunk count it was cut for (`chunks` as given, or -- 0 -- suggest_chunks()
    of what was measured), both the same on every rank, and a sentence for the
    bench record."""
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
    if chunks == 0:                                       # automatic: from what was just measured, the same on every rank
        chunks = suggest_chunks(float(np.mean(kernel_by_rank)), gather_ms) if not overlap else 1
    bounds = balanced_bounds(W, world, cost, per_column_send, overlap=overlap, chunks=chunks)
    note = (f"re-cut after warm-up frames on the equal partition: kernel ms per rank "
            f"{[round(k, 3) for k in kernel_by_rank]}, gather alone {gather_ms:.3f} ms"
            + (f"; every strip rendered and sent in {chunks} column chunks" if chunks > 1 else ""))
    return bounds, note, chunks


class StripPipeline:
    """Render/gather software pipeline over successive frames.

    Frames are independent, so while RCCL moves frame k's strips to rank 0 (on
    its own stream, over xGMI) the render kernel of frame k+1 already runs on
    the compute stream.  Two strip buffers alternate; a buffer is rendered into
    again only after the gather that read it has completed.  `render(buf)` is
    the caller's function that enqueues the kernel writing `buf`.
    """

    def __init__(self, W, H, world, rank, device, render, overlap=True, dtype=torch.float32, force_gather=False,
                 bounds=None, chunks=1, align=1):
        self.world, self.rank, self.render, self.overlap = world, rank, render, overlap
        self.gather = world > 1 or force_gather          # force_gather: run the collective even with one rank
        self.x0, self.x1, self.strip = strip_bounds(W, world, rank)
        # single-frame mode with chunks > 1: the strip is rendered and sent in column chunks (render(buf, a, b)
        # per chunk), chunk k on its way to rank 0 while chunk k + 1 is rendered -- within ONE frame, as the
        # reference's ranks write their pixels into the shared image while they render
        # (src/RayTracer.cpp:904-923, 1188-1193).  Chunks travel point-to-point, like uneven strips.
        self.chunks = max(int(chunks), 1) if not overlap else 1
        self.align = max(int(align), 1)
        if self.chunks > 1 and bounds is None:
            bounds = equal_bounds(W, world)
        self.bounds = None                               # None: the equal partition and one gather collective
        if bounds is not None and (list(bounds) != equal_bounds(W, world) or self.chunks > 1):
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

    def _step_chunked(self, buf):
        """One frame, chunk by chunk: render chunk k, then start its transfer and carry on with chunk k + 1."""
        per_rank = [chunk_bounds(a, b, self.chunks, self.align) for a, b in self.bounds]
        works = []
        for k in range(self.chunks):
            a, b = per_rank[self.rank][k]
            if b > a:
                self.render(buf[a - self.x0:b - self.x0], a, b)
            if not self.gather:
                continue
            ops = []
            if self.rank == 0:
                for r in range(1, self.world):
                    ra, rb = per_rank[r][k]
                    if rb > ra:
                        ops.append(dist.P2POp(dist.irecv, self.views[r][ra - self.bounds[r][0]:rb - self.bounds[r][0]], r))
            elif b > a:
                ops.append(dist.P2POp(dist.isend, buf[a - self.x0:b - self.x0], 0))
            if ops:
                works += dist.batch_isend_irecv(ops)
        for w in works:
            w.wait()

    def step(self):
        b, slot = self.k % len(self.bufs), self.k % len(self.pending)
        self.k += 1
        self._wait(slot)
        if self.chunks > 1:
            self._step_chunked(self.bufs[b])
            return
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
        return (f"{self.world} x-strips of " + "/".join(str(b - a) for a, b in self.bounds) + " columns"
                + (" (measured-cost partition)" if list(self.bounds) != equal_bounds(sum(b - a for a, b in self.bounds), self.world) else "")
                + (f", each rendered and sent in {self.chunks} column chunks" if self.chunks > 1 else ""))


# ---------------------------------------------------------------------------------------------------------------------
# The reference's ranks all write into ONE `pixels` array while they render (src/RayTracer.h:44; src/RayTracer.cpp:904-923,
# 1188-1196: the Tilera tiles share that memory) -- there is no gather.  The same for one process per GPU: rank 0 owns the
# image in its HBM, the other ranks map it (HIP IPC: include/rt_capi.h, rt_shared_image_*) and their kernels store their
# strips' pixels straight into it over xGMI.

def _collective_device(device):
    """where the small tensors of control collectives live: the GPU for RCCL, the host for gloo"""
    return device if dist.get_backend() == "nccl" else torch.device("cpu")


def _everyone(ok, device):
    """True if `ok` is true on every rank (one small all-reduce)."""
    t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=_collective_device(device))
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return int(t[0]) == 1


class _DeviceMemory:
    """memory this package allocated through the C ABI, for torch.as_tensor (zero copy)"""

    def __init__(self, ptr, shape):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": "<f4", "data": (int(ptr), False), "version": 2, "strides": None}


class SharedImage:
    """One W x H x 3 fp32 image in rank `owner`'s HBM that every rank of the node renders into.

    Collective: every rank constructs it at the same point.  `ptr` is the image's address in THIS process (the owner's
    allocation, or this rank's mapping of it); column x starts at ptr + x * H * 12.  Raises on every rank, or on none,
    when the image cannot be shared (no HIP IPC between these processes)."""

    def __init__(self, W, H, device, owner=0):
        import ctypes as C
        from . import capi
        self._capi, self._lib = capi, capi.load_library()
        self.W, self.H, self.owner, self.rank = int(W), int(H), owner, dist.get_rank()
        self.device = device
        self.device_index = device.index if device.index is not None else 0
        self.ptr, self._mapped = None, False
        handle = C.create_string_buffer(64)
        p = C.c_void_p()
        error = None
        if self.rank == owner:
            try:
                capi.check(self._lib.rt_shared_image_create(self.device_index, self.W * self.H * 12, C.byref(p), handle))
                self.ptr = p.value
            except capi.RtError as e:
                error = str(e)
        box = [(handle.raw, error)]
        dist.broadcast_object_list(box, src=owner)
        raw, error = box[0]
        if error is None and self.rank != owner:
            try:
                capi.check(self._lib.rt_shared_image_open(self.device_index, raw, C.byref(p)))
                self.ptr, self._mapped = p.value, True
            except capi.RtError as e:
                error = str(e)
        if not _everyone(error is None, device):
            self.close()
            raise RuntimeError(f"the image cannot be shared between the ranks: {error or 'another rank could not map it'}")

    def column_ptr(self, x):
        return self.ptr + int(x) * self.H * 12

    def tensor(self):
        """The image as a (W, H, 3) torch tensor on this rank's GPU (zero copy)."""
        return torch.as_tensor(_DeviceMemory(self.ptr, (self.W, self.H, 3)), device=self.device)

    def close(self):
        """Collective in effect: the owner frees the image after the others have unmapped it (call it on every rank,
        with a barrier in between if the owner could get here first -- DirectStrips.close() does)."""
        if self.ptr is None:
            return
        if self._mapped:
            self._lib.rt_shared_image_close(self.device_index, self.ptr)
        elif self.rank == self.owner:
            self._lib.rt_shared_image_destroy(self.device_index, self.ptr)
        self.ptr = None


class DirectStrips:
    """A frame without a gather: every rank's kernel stores its strip straight into the shared image.

    step(): render columns [x0, x1) to image.column_ptr(x0) (`render_ptr(address, x0, x1)` enqueues the kernel on the
    current stream), then a one-word all-reduce in stream order behind it -- when a rank's all-reduce has completed, every
    rank's kernel of that frame has, so the frame is whole in the owner's HBM, and the next frame's kernel starts behind it:
    one frame at a time, with no host synchronisation per frame (RCCL; gloo's collectives run on the host, so there the
    stream is drained first).

    overlap=True with TWO shared images: a stream of independent frames.  Frame k goes to image k % 2 and its all-reduce is
    only waited for before frame k + 2 is rendered into the same image, so a rank's next kernel starts as soon as its own
    last one is done, whatever the other ranks are doing (StripPipeline's pipelined mode without the transfers)."""

    def __init__(self, shared, world, rank, device, render_ptr, bounds=None, overlap=False):
        self.shareds = list(shared) if isinstance(shared, (list, tuple)) else [shared]
        self.shared = self.shareds[0]
        self.world, self.rank, self.device, self.render_ptr = world, rank, device, render_ptr
        self.overlap = bool(overlap) and len(self.shareds) > 1
        W = self.shared.W
        self.bounds = [tuple(b) for b in (bounds if bounds is not None else equal_bounds(W, world))]
        assert len(self.bounds) == world and self.bounds[0][0] == 0 and self.bounds[-1][1] == W
        assert all(self.bounds[r][1] == self.bounds[r + 1][0] for r in range(world - 1))
        self.x0, self.x1 = self.bounds[rank]
        self.strip = max(self.x1 - self.x0, 1)
        self._host_collectives = dist.get_backend() != "nccl"
        self._flags = [torch.zeros(1, dtype=torch.int32, device=_collective_device(device)) for _ in self.shareds]
        self.pending = [None] * len(self.shareds)
        self.k = 0

    def _wait(self, b):
        if self.pending[b] is not None:
            self.pending[b].wait()
            self.pending[b] = None

    def step(self):
        b = self.k % len(self.shareds) if self.overlap else 0
        self.k += 1
        self._wait(b)
        if self.x1 > self.x0:
            self.render_ptr(self.shareds[b].column_ptr(self.x0), self.x0, self.x1)
        if self._host_collectives and torch.device(self.device).type == "cuda":
            torch.cuda.synchronize(self.device)               # (the kernel must be done before a host-side collective says so)
        if self.overlap:
            self.pending[b] = dist.all_reduce(self._flags[b], async_op=True)
        else:
            dist.all_reduce(self._flags[b])

    def drain(self):
        for b in range(len(self.pending)):
            self._wait(b)

    def image(self, W=None):
        """Rank 0 (the owner): the last frame, as StripPipeline.image() gives it; None elsewhere."""
        self.drain()
        last = (self.k - 1) % len(self.shareds) if (self.overlap and self.k > 0) else 0
        return self.shareds[last].tensor() if self.rank == self.shared.owner else None

    def describe(self):
        eq = list(self.bounds) == equal_bounds(self.shared.W, self.world)
        return (f"{self.world} x-strips of " + "/".join(str(b - a) for a, b in self.bounds) + " columns"
                + ("" if eq else " (measured-cost partition)") + ", stored by the kernels straight into rank 0's image (shared over HIP IPC)")


def balance_direct(W, my_kernel_ms, device):
    """Every rank's kernel time on the EQUAL partition (its strip stored into the shared image: a peer's time includes what its
    link made of the stores) -> (bounds, note): strips of equal measured cost, the same on every rank."""
    import numpy as np
    world = dist.get_world_size()
    mine = torch.tensor([float(my_kernel_ms)], dtype=torch.float64, device=_collective_device(device))
    everyone = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(everyone, mine)
    kernel_by_rank = [float(t[0]) for t in everyone]
    cost = np.zeros(W, dtype=np.float64)
    for (a, b), k in zip(equal_bounds(W, world), kernel_by_rank):
        if b > a:
            cost[a:b] = max(k, 0.0) / (b - a)
    bounds = balanced_bounds(W, world, cost, 0.0)
    return bounds, f"re-cut after warm-up frames on the equal partition: kernel ms per rank {[round(k, 3) for k in kernel_by_rank]}"
