"""The N > 1 path on CPU: world_size 2 and 3 over gloo.  Every rank fills its
x-strip (here from the oracle, standing in for the GPU kernel), the strips are
gathered to rank 0 with the same code bench.py uses
(tilecoderaytracer_amd.distributed), and rank 0 must hold exactly the
single-process image -- including widths that do not divide evenly."""
import os
import sys
import tempfile

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _pipeline_worker(rank, world, W, H, depth, init_file, out_file, overlap):
    """bench.py's loop: StripPipeline.step() K times, frames differ so stale data would show."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    import oracle_lib
    from tilecoderaytracer_amd.distributed import StripPipeline
    dist.init_process_group("gloo", init_method=f"file://{init_file}", rank=rank, world_size=world)
    scene = oracle_lib.OracleScene.builtin()
    frame = {"d": 0}

    def render(buf):
        x0, x1 = pipe.x0, pipe.x1
        buf.zero_()
        if x1 > x0:                              # frame k is rendered at depth k % 3 (stands in for the kernel)
            buf[: x1 - x0] = torch.from_numpy(scene.render(W, H, frame["d"] % 3, x0, x1))
        frame["d"] += 1

    pipe = StripPipeline(W, H, world, rank, "cpu", render, overlap=overlap)
    for _ in range(5):
        pipe.step()
    img = pipe.image(W)
    dist.barrier()
    if rank == 0:
        np.save(out_file, img.numpy())
    dist.destroy_process_group()


@pytest.mark.parametrize("overlap", [True, False])
def test_pipelined_gather_delivers_the_last_frame(oracle, overlap):
    world, W, H = 2, 30, 16
    with tempfile.TemporaryDirectory() as d:
        init_file, out_file = os.path.join(d, "init"), os.path.join(d, "out.npy")
        mp.spawn(_pipeline_worker, args=(world, W, H, 0, init_file, out_file, overlap), nprocs=world, join=True)
        got = np.load(out_file)
    ref = oracle.OracleScene.builtin().render(W, H, 4 % 3)      # the 5th frame
    np.testing.assert_array_equal(got.view(np.uint32), ref.view(np.uint32))


def _worker(rank, world, W, H, depth, init_file, out_file):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    import oracle_lib
    from tilecoderaytracer_amd.distributed import alloc_full, gather_strips, strip_bounds
    dist.init_process_group("gloo", init_method=f"file://{init_file}", rank=rank, world_size=world)
    x0, x1, strip = strip_bounds(W, world, rank)
    buf = torch.zeros((strip, H, 3), dtype=torch.float32)
    if x1 > x0:
        buf[: x1 - x0] = torch.from_numpy(oracle_lib.OracleScene.builtin().render(W, H, depth, x0, x1))
    views = None
    full = None
    if rank == 0:
        full, views = alloc_full(W, H, world, "cpu")
    for _ in range(2):                       # twice: the buffers are reused step after step
        gather_strips(buf, views, dst=0)
    dist.barrier()
    if rank == 0:
        np.save(out_file, full[:W].numpy())
    dist.destroy_process_group()


@pytest.mark.parametrize("world,W", [(2, 64), (2, 37), (3, 8)])
def test_strips_gathered_to_rank0_equal_the_full_image(oracle, world, W):
    H, depth = 24, 3
    with tempfile.TemporaryDirectory() as d:
        init_file, out_file = os.path.join(d, "init"), os.path.join(d, "out.npy")
        mp.spawn(_worker, args=(world, W, H, depth, init_file, out_file), nprocs=world, join=True)
        got = np.load(out_file)
    ref = oracle.OracleScene.builtin().render(W, H, depth)
    np.testing.assert_array_equal(got.view(np.uint32), ref.view(np.uint32))


def test_strip_bounds_cover_the_image_exactly():
    from tilecoderaytracer_amd.distributed import strip_bounds
    for W in (1, 7, 8, 37, 4096, 8192):
        for world in (1, 2, 3, 4, 8):
            cols = []
            for r in range(world):
                x0, x1, strip = strip_bounds(W, world, r)
                assert 0 <= x0 <= x1 <= W and x1 - x0 <= strip
                cols += list(range(x0, x1))
            assert cols == list(range(W))


def test_balanced_bounds_properties():
    from tilecoderaytracer_amd.distributed import balanced_bounds, equal_bounds
    rng = np.random.RandomState(7)
    for W in (8, 37, 4096):
        for world in (1, 2, 3, 8):
            for cost in (np.full(W, 1.0 / W), rng.uniform(0.1, 2.0, W), np.r_[np.zeros(W // 2), np.ones(W - W // 2)]):
                for send in (0.0, 0.5 * cost.mean(), 3.0 * cost.mean()):
                    for overlap in (True, False):
                        b = balanced_bounds(W, world, cost, send, overlap=overlap)
                        assert len(b) == world and b[0][0] == 0 and b[-1][1] == W
                        assert all(0 <= x0 <= x1 <= W for x0, x1 in b)
                        assert all(b[r][1] == b[r + 1][0] for r in range(world - 1))
    # nothing to send and a flat cost: the equal partition
    assert balanced_bounds(4096, 8, np.full(4096, 1.0), 0.0) == equal_bounds(4096, 8)
    # a link slower than a GPU: the receiver renders more, the peers share the rest evenly
    b = balanced_bounds(4096, 8, np.full(4096, 1.08 / 4096), 2.68 / 4096)
    widths = [x1 - x0 for x0, x1 in b]
    assert widths[0] > 2 * widths[1] and max(widths[1:]) - min(widths[1:]) <= 8
    frame = max([widths[0] * 1.08 / 4096] + [w * 2.68 / 4096 for w in widths[1:]])
    assert frame < 0.30                                       # 0.335 with equal strips
    # one frame, nothing overlapped: a peer's time is kernel + send, the receiver takes even more
    b1 = balanced_bounds(4096, 8, np.full(4096, 1.08 / 4096), 2.68 / 4096, overlap=False)
    w1 = [x1 - x0 for x0, x1 in b1]
    single = max([w1[0] * 1.08 / 4096] + [w * (1.08 + 2.68) / 4096 for w in w1[1:]])
    equal_single = 512 * (1.08 + 2.68) / 4096
    assert w1[0] > widths[0] and single < 0.80 * equal_single
    # a costly middle: strips there get narrower
    cost = np.ones(4096)
    cost[1536:2560] = 4.0
    widths = [x1 - x0 for x0, x1 in balanced_bounds(4096, 8, cost, 0.0)]
    assert min(widths) < 300 and max(widths) > 600


def _uneven_worker(rank, world, W, H, bounds, init_file, out_file, overlap):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    import oracle_lib
    from tilecoderaytracer_amd.distributed import StripPipeline
    dist.init_process_group("gloo", init_method=f"file://{init_file}", rank=rank, world_size=world)
    scene = oracle_lib.OracleScene.builtin()
    frame = {"d": 0}

    def render(buf):
        x0, x1 = pipe.x0, pipe.x1
        if x1 > x0:
            buf[: x1 - x0] = torch.from_numpy(scene.render(W, H, frame["d"] % 3, x0, x1))
        frame["d"] += 1

    pipe = StripPipeline(W, H, world, rank, "cpu", render, overlap=overlap, bounds=bounds)
    assert (pipe.x0, pipe.x1) == tuple(bounds[rank])
    for _ in range(5):
        pipe.step()
    img = pipe.image(W)
    dist.barrier()
    if rank == 0:
        np.save(out_file, img.numpy())
    dist.destroy_process_group()


@pytest.mark.parametrize("overlap", [True, False])
@pytest.mark.parametrize("bounds", [[(0, 20), (20, 25), (25, 30)], [(0, 11), (11, 11), (11, 30)], [(0, 0), (0, 17), (17, 30)]])
def test_uneven_strips_reach_rank0_in_place(oracle, bounds, overlap):
    """The measured-cost partition: strips of different widths (one may be empty, rank 0's too),
    sent point-to-point into rank 0's image; five pipelined frames, the last one must be whole."""
    world, W, H = 3, 30, 12
    with tempfile.TemporaryDirectory() as d:
        init_file, out_file = os.path.join(d, "init"), os.path.join(d, "out.npy")
        mp.spawn(_uneven_worker, args=(world, W, H, bounds, init_file, out_file, overlap), nprocs=world, join=True)
        got = np.load(out_file)
    ref = oracle.OracleScene.builtin().render(W, H, 4 % 3)
    np.testing.assert_array_equal(got.view(np.uint32), ref.view(np.uint32))


def test_chunk_bounds_cover_a_strip_and_match_the_c_abi():
    """distributed.chunk_bounds and rt_chunk_bounds (csrc/rt_multi.hip) are the same arithmetic: chunks are
    contiguous, in order, cover [x0, x1) exactly, and inner boundaries lie a multiple of `align` from x0."""
    import ctypes as C
    from tilecoderaytracer_amd import capi
    from tilecoderaytracer_amd.distributed import chunk_bounds
    lib = capi.load_library()
    for x0, x1 in ((0, 4096), (512, 1024), (100, 612), (7, 7), (3, 8), (0, 37), (1000, 1001), (16, 48)):
        for chunks in (1, 2, 3, 4, 8, 64):
            for align in (1, 4, 16):
                got = chunk_bounds(x0, x1, chunks, align)
                assert len(got) == chunks and got[0][0] == x0 and got[-1][1] == x1
                for k, (a, b) in enumerate(got):
                    assert x0 <= a <= b <= x1
                    assert k == 0 or a == got[k - 1][1]
                    assert b == x1 or (b - x0) % align == 0
                    ca, cb = C.c_int(), C.c_int()
                    assert lib.rt_chunk_bounds(x0, x1, chunks, k, align, C.byref(ca), C.byref(cb)) == 0
                    assert (ca.value, cb.value) == (a, b)
    assert lib.rt_chunk_bounds(0, 8, 0, 0, 1, None, None) == 1 and lib.rt_chunk_bounds(0, 8, 2, 2, 1, None, None) == 1


def _chunked_worker(rank, world, W, H, bounds, chunks, align, init_file, out_file):
    """The single-frame mode with column chunks: render chunk k, start its transfer, render chunk k+1."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    import oracle_lib
    from tilecoderaytracer_amd.distributed import StripPipeline
    dist.init_process_group("gloo", init_method=f"file://{init_file}", rank=rank, world_size=world)
    scene = oracle_lib.OracleScene.builtin()
    frame = {"d": 0, "calls": []}

    def render(buf, a, b):
        assert buf.shape[0] == b - a and b > a
        buf.copy_(torch.from_numpy(scene.render(W, H, frame["d"] % 3, a, b)))
        frame["calls"].append((a, b))

    pipe = StripPipeline(W, H, world, rank, "cpu", render, overlap=False, bounds=bounds, chunks=chunks, align=align)
    for _ in range(3):
        frame["calls"] = []
        pipe.step()
        frame["d"] += 1
        # this rank's chunks: in order, contiguous, exactly its strip
        cols = [x for a, b in frame["calls"] for x in range(a, b)]
        assert cols == list(range(pipe.x0, pipe.x1))
        assert len(frame["calls"]) <= chunks
    img = pipe.image(W)
    dist.barrier()
    if rank == 0:
        np.save(out_file, img.numpy())
    dist.destroy_process_group()


@pytest.mark.parametrize("world,W,bounds,chunks,align", [
    (2, 64, None, 4, 4),                                     # equal strips, chunks of 8 columns
    (2, 37, None, 3, 4),                                     # widths that do not divide
    (3, 30, [(0, 20), (20, 25), (25, 30)], 4, 4),            # uneven strips: a peer with fewer tiles than chunks (empty chunks)
    (3, 30, [(0, 11), (11, 11), (11, 30)], 2, 16),           # an empty strip; a strip narrower than the alignment
    (3, 30, [(0, 0), (0, 17), (17, 30)], 8, 1),              # rank 0 renders nothing
])
def test_chunked_single_frame_delivers_the_image(oracle, world, W, bounds, chunks, align):
    H = 12
    with tempfile.TemporaryDirectory() as d:
        init_file, out_file = os.path.join(d, "init"), os.path.join(d, "out.npy")
        mp.spawn(_chunked_worker, args=(world, W, H, bounds, chunks, align, init_file, out_file), nprocs=world, join=True)
        got = np.load(out_file)
    ref = oracle.OracleScene.builtin().render(W, H, 2 % 3)      # the 3rd frame
    np.testing.assert_array_equal(got.view(np.uint32), ref.view(np.uint32))


def test_balanced_bounds_with_chunks():
    """Chunked single-frame cost: between the pipelined and the serial model; more chunks, wider peer strips."""
    from tilecoderaytracer_amd.distributed import balanced_bounds, suggest_chunks
    cost, g = np.full(4096, 1.08 / 4096), 2.68 / 4096

    def frame_time(b, K):
        t = []
        for r, (x0, x1) in enumerate(b):
            R, S = (x1 - x0) * 1.08 / 4096, (x1 - x0) * g
            t.append(R if r == 0 else (R + S if K == 1 else max(R, S) + min(R, S) / K))
        return max(t)
    times = []
    for K in (1, 2, 4, 8):
        b = balanced_bounds(4096, 8, cost, g, overlap=False, chunks=K)
        assert b[0][0] == 0 and b[-1][1] == 4096 and all(b[r][1] == b[r + 1][0] for r in range(7))
        times.append(frame_time(b, K))
    assert times[0] > times[1] > times[2] > times[3]
    pipelined = balanced_bounds(4096, 8, cost, g, overlap=True)
    assert times[3] > max([(pipelined[0][1] - pipelined[0][0]) * 1.08 / 4096] + [(x1 - x0) * g for x0, x1 in pipelined[1:]]) - 1e-9
    assert suggest_chunks(0.2, 0.34) >= 4 and suggest_chunks(1.2, 0.1) == 1 and suggest_chunks(0.0, 0.0) == 1


def _balance_worker(rank, world, W, H, init_file, out_file):
    """bench.py's N > 1 warm-up: equal partition, measure, re-cut, carry on."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    import json
    import oracle_lib
    from tilecoderaytracer_amd.distributed import StripPipeline, measure_and_balance
    dist.init_process_group("gloo", init_method=f"file://{init_file}", rank=rank, world_size=world)
    scene = oracle_lib.OracleScene.builtin()

    def make(bounds=None):
        pp = StripPipeline(W, H, world, rank, "cpu", None, overlap=True, bounds=bounds)

        def render(buf, pp=pp):
            if pp.x1 > pp.x0:
                buf[: pp.x1 - pp.x0] = torch.from_numpy(scene.render(W, H, 2, pp.x0, pp.x1))
        pp.render = render
        return pp

    pipe = make()

    def sync():
        pipe.drain()
        dist.barrier()

    pipe.step()
    pipe.step()
    fake_kernel_ms = [1.0, 3.0, 1.0][rank]               # the middle strip is the expensive one
    bounds, note, chunks = measure_and_balance(pipe, W, fake_kernel_ms, sync, "cpu")
    assert chunks == 1                                    # overlap mode never chunks
    pipe = make(bounds)
    for _ in range(3):
        pipe.step()
    img = pipe.image(W)
    dist.barrier()
    json.dump({"bounds": bounds, "note": note}, open(f"{out_file}.{rank}.json", "w"))
    if rank == 0:
        np.save(out_file, img.numpy())
    dist.destroy_process_group()


def test_measured_partition_is_agreed_on_and_delivers_the_image(oracle):
    import json
    world, W, H = 3, 48, 10
    with tempfile.TemporaryDirectory() as d:
        init_file, out_file = os.path.join(d, "init"), os.path.join(d, "out.npy")
        mp.spawn(_balance_worker, args=(world, W, H, init_file, out_file), nprocs=world, join=True)
        got = np.load(out_file)
        per_rank = [json.load(open(f"{out_file}.{r}.json")) for r in range(world)]
    assert per_rank[0]["bounds"] == per_rank[1]["bounds"] == per_rank[2]["bounds"]
    b = per_rank[0]["bounds"]
    assert b[0][0] == 0 and b[-1][1] == W and (b[1][1] - b[1][0]) < 16      # the costly strip got narrower
    ref = oracle.OracleScene.builtin().render(W, H, 2)
    np.testing.assert_array_equal(got.view(np.uint32), ref.view(np.uint32))


def _balance_direct_worker(rank, world, W, init_file, out_file):
    sys.path.insert(0, ROOT)
    import json
    from tilecoderaytracer_amd.distributed import balance_direct
    dist.init_process_group("gloo", init_method=f"file://{init_file}", rank=rank, world_size=world)
    bounds, note = balance_direct(W, [1.0, 3.0, 2.0][rank], torch.device("cpu"))
    json.dump({"bounds": bounds, "note": note}, open(f"{out_file}.{rank}.json", "w"))
    dist.barrier()
    dist.destroy_process_group()


def test_direct_transport_cuts_strips_of_equal_measured_kernel_time():
    """bench.py's re-cut for the direct-store transport: no transfer to weigh, every rank ends up with the same share of the
    measured kernel time, and every rank computes the same strips (tilecoderaytracer_amd.distributed.balance_direct)."""
    import json
    world, W = 3, 300
    with tempfile.TemporaryDirectory() as d:
        init_file, out_file = os.path.join(d, "init"), os.path.join(d, "out")
        mp.spawn(_balance_direct_worker, args=(world, W, init_file, out_file), nprocs=world, join=True)
        per_rank = [json.load(open(f"{out_file}.{r}.json")) for r in range(world)]
    assert per_rank[0]["bounds"] == per_rank[1]["bounds"] == per_rank[2]["bounds"]
    b = per_rank[0]["bounds"]
    assert b[0][0] == 0 and b[-1][1] == W and all(b[r][1] == b[r + 1][0] for r in range(world - 1))
    # columns 0-99 cost 0.01 each, 100-199 0.03, 200-299 0.02: 6 ms in all, 2 ms per rank
    cost = np.concatenate([np.full(100, 0.01), np.full(100, 0.03), np.full(100, 0.02)])
    shares = [cost[a:c].sum() for a, c in b]
    assert max(shares) - min(shares) <= 0.031 and abs(sum(shares) - 6.0) < 1e-9
    assert "kernel ms per rank [1.0, 3.0, 2.0]" in per_rank[0]["note"]


class _FileSharedImage:
    """Stands in for SharedImage on a machine without a GPU: the image is a file both rank processes map (shared memory between
    processes, as HIP IPC gives the GPUs); same attributes and methods as tilecoderaytracer_amd.distributed.SharedImage."""

    def __init__(self, path, W, H, owner=0):
        self.W, self.H, self.owner = W, H, owner
        self.map = np.memmap(path, dtype=np.float32, mode="r+", shape=(W, H, 3))
        self.ptr = self.map.ctypes.data

    def column_ptr(self, x):
        return self.ptr + int(x) * self.H * 12

    def tensor(self):
        return torch.from_numpy(np.asarray(self.map))


def _direct_cpu_worker(rank, world, W, H, bounds, overlap, frames, paths, init_file, out_file):
    """DirectStrips as bench.py drives it, the oracle standing in for the kernel: it `renders` its strip and the bytes go where
    the kernel would store them -- through the address DirectStrips hands to render_ptr."""
    import ctypes as C
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    import oracle_lib
    from tilecoderaytracer_amd.distributed import DirectStrips
    dist.init_process_group("gloo", init_method=f"file://{init_file}", rank=rank, world_size=world)
    scene = oracle_lib.OracleScene.builtin()
    images = [_FileSharedImage(p, W, H) for p in paths]
    frame = {"k": 0}

    def render_ptr(address, a, b):
        strip = np.ascontiguousarray(scene.render(W, H, frame["k"] % 3, a, b))      # frame k at depth k % 3: a stale frame would show
        C.memmove(address, strip.ctypes.data, strip.nbytes)
        frame["k"] += 1

    pipe = DirectStrips(images if overlap else images[0], world, rank, torch.device("cpu"), render_ptr, bounds=bounds, overlap=overlap)
    assert (pipe.x0, pipe.x1) == tuple((bounds or [(r * -(-W // world), min((r + 1) * -(-W // world), W)) for r in range(world)])[rank])
    for _ in range(frames):
        pipe.step()
        if pipe.x1 == pipe.x0:
            frame["k"] += 1                                  # (a rank with an empty strip renders nothing but counts the frame)
    img = pipe.image(W)
    assert (img is None) == (rank != 0)
    if rank == 0:
        np.save(out_file, img.numpy().copy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,W,bounds,overlap,frames", [
    (2, 40, None, False, 3),
    (2, 37, [(0, 9), (9, 37)], False, 4),
    (3, 30, [(0, 12), (12, 12), (12, 30)], False, 2),          # an empty strip
    (2, 40, None, True, 4),                                    # a stream of frames through two images: the last one is in image 1
    (3, 33, [(0, 5), (5, 20), (20, 33)], True, 5),             # ... in image 0
])
def test_direct_strips_deliver_the_frame_without_a_gather(oracle, world, W, bounds, overlap, frames):
    """The direct transport's host logic over gloo, world size 2 and 3 (tilecoderaytracer_amd.distributed.DirectStrips): every rank
    stores its strip into the one shared image, a one-word all-reduce per frame is the fence, a stream of frames alternates
    between two images -- rank 0 ends up holding exactly the last frame."""
    H = 12
    with tempfile.TemporaryDirectory() as d:
        paths = [os.path.join(d, f"image{i}.f32") for i in range(2 if overlap else 1)]
        for p in paths:
            np.full((W, H, 3), -7.0, np.float32).tofile(p)
        init_file, out_file = os.path.join(d, "init"), os.path.join(d, "out.npy")
        mp.spawn(_direct_cpu_worker, args=(world, W, H, bounds, overlap, frames, paths, init_file, out_file), nprocs=world, join=True)
        got = np.load(out_file)
    want = oracle.OracleScene.builtin().render(W, H, (frames - 1) % 3)
    np.testing.assert_array_equal(got.view(np.uint32), want.view(np.uint32))


def test_rt_render_multi_strip_arithmetic(oracle):
    """rt_render_multi (one process, N GPUs) on the CPU, with the oracle standing in for the
    kernel: every "GPU" renders the strip rt_strip_bounds gives it into its own strip-sized
    buffer, the gather places rank g at g * strip_floats of the full buffer (what ncclGather
    does at the root), and the first W * H * 3 floats of that buffer are the image
    (csrc/rt_multi.hip).  Widths that do not divide, short and empty trailing strips."""
    import ctypes as C
    from tilecoderaytracer_amd import capi
    lib = capi.load_library()
    H, depth = 12, 2
    scene = oracle.OracleScene.builtin()
    for W, n in ((64, 2), (64, 8), (37, 2), (37, 3), (5, 8), (9, 4), (1, 2), (100, 7)):
        want = scene.render(W, H, depth)
        strip = lib.rt_strip_bounds(W, n, 0, None, None)
        assert strip == -(-W // n)
        full = np.full((n * strip, H, 3), -7.0, np.float32)          # the root's receive buffer
        covered = 0
        for g in range(n):
            x0, x1 = C.c_int(), C.c_int()
            assert lib.rt_strip_bounds(W, n, g, C.byref(x0), C.byref(x1)) == strip
            x0, x1 = x0.value, x1.value
            assert 0 <= x0 <= x1 <= W and x0 == min(g * strip, W) and x1 - x0 <= strip
            assert x0 == covered                                      # contiguous, in rank order
            covered = x1
            mine = np.full((strip, H, 3), -9.0, np.float32)           # this rank's send buffer: strip_floats floats
            mine[: x1 - x0] = scene.render(W, H, depth, x0, x1)
            full[g * strip:(g + 1) * strip] = mine                    # ncclGather: rank g at g * sendcount
        assert covered == W
        image = full.reshape(-1)[: W * H * 3].reshape(W, H, 3)        # the image_bytes copy to the caller
        assert np.array_equal(image.view(np.uint32), want.view(np.uint32)), (W, n)
    assert lib.rt_strip_bounds(0, 2, 0, None, None) == 0 and lib.rt_strip_bounds(8, 2, 2, None, None) == 0


def test_c_side_partition_is_the_python_solver_bit_for_bit():
    """rt_balance_strips / rt_suggest_chunks (csrc/rt_multi.hip: what rt_multi_render(chunks = 0), rt_render_multi and
    bin/tcrt_raytracer --gpus G cut their strips with) against balanced_bounds / suggest_chunks (what bench.py's
    one-process-per-GPU path uses): the same bounds, column for column, on measured-looking and adversarial inputs."""
    import ctypes as C
    from tilecoderaytracer_amd import capi
    from tilecoderaytracer_amd.distributed import balanced_bounds, equal_bounds, suggest_chunks
    lib = capi.load_library()
    rng = np.random.RandomState(11)

    def c_bounds(W, n, measured, kernel_ms, send, chunks):
        mb = (C.c_int * (n + 1))(*([a for a, _ in measured] + [W]))
        km = (C.c_double * n)(*kernel_ms)
        out = (C.c_int * (n + 1))()
        assert lib.rt_balance_strips(W, n, mb, km, send, chunks, out) == 0
        return [(out[g], out[g + 1]) for g in range(n)]

    cases = 0
    for W in (8, 37, 500, 4096, 8192):
        for n in (1, 2, 3, 4, 8):
            measured = equal_bounds(W, n)
            for kind in range(4):
                if kind == 0:
                    kernel_ms = [1.0] * n
                elif kind == 1:
                    kernel_ms = list(rng.uniform(0.05, 3.0, n))
                elif kind == 2:
                    kernel_ms = [0.0 if g % 2 else 2.0 for g in range(n)]                 # strips that cost nothing
                else:
                    kernel_ms = [0.18, 0.2, 0.21, 0.25, 0.9, 0.88, 0.3, 0.2][:n]          # a costly middle, like the sphere grids
                cost = np.zeros(W)
                for (a, b), k in zip(measured, kernel_ms):
                    if b > a:
                        cost[a:b] = max(k, 0.0) / (b - a)
                for send in (0.0, 0.34 / max(W // n, 1), 3.0 * float(np.mean(kernel_ms)) / max(W // n, 1)):
                    for chunks in (1, 2, 4, 8):
                        want = [tuple(int(v) for v in b) for b in balanced_bounds(W, n, cost, send, overlap=False, chunks=chunks)]
                        got = c_bounds(W, n, measured, kernel_ms, send, chunks)
                        assert got == want, (W, n, kind, send, chunks, got, want)
                        assert got[0][0] == 0 and got[-1][1] == W and all(got[g][1] == got[g + 1][0] for g in range(n - 1))
                        cases += 1
    assert cases > 500
    # measured on a partition that is not the equal one (a re-cut of a re-cut)
    W, n = 4096, 4
    measured = [(0, 1500), (1500, 2100), (2100, 2500), (2500, 4096)]
    kernel_ms = [1.0, 1.1, 0.9, 1.05]
    cost = np.zeros(W)
    for (a, b), k in zip(measured, kernel_ms):
        cost[a:b] = k / (b - a)
    assert c_bounds(W, n, measured, kernel_ms, 1e-4, 2) == [tuple(int(v) for v in b) for b in balanced_bounds(W, n, cost, 1e-4, overlap=False, chunks=2)]
    # the chunk count, ties of round() included (half to even in both)
    for k in (0.0, 0.1, 0.2, 0.25, 0.5, 1.0, 1.2, 4.0):
        for s in (0.0, 0.0625, 0.1, 0.125, 0.3, 0.34, 0.375, 0.5, 0.625, 1.0, 2.5, 10.0):
            for most in (1, 4, 8):
                assert lib.rt_suggest_chunks(k, s, most) == suggest_chunks(k, s, most), (k, s, most)
    # bad arguments
    mb = (C.c_int * 3)(0, 4, 8)
    km = (C.c_double * 2)(1.0, 1.0)
    out = (C.c_int * 3)()
    assert lib.rt_balance_strips(8, 2, mb, km, 0.0, 1, out) == 0
    assert lib.rt_balance_strips(0, 2, mb, km, 0.0, 1, out) == 1 and lib.rt_balance_strips(8, 2, None, km, 0.0, 1, out) == 1
    assert lib.rt_balance_strips(9, 2, mb, km, 0.0, 1, out) == 1                      # the measured strips do not cover the image
    km[1] = float("nan")
    assert lib.rt_balance_strips(8, 2, mb, km, 0.0, 1, out) == 1
