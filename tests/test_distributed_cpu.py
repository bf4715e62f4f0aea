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
