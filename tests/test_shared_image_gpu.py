"""One image in one GPU's HBM that several PROCESSES render into (include/rt_capi.h, rt_shared_image_*; the reference's ranks
all write into the one `pixels` array, src/RayTracer.h:44, src/RayTracer.cpp:904-923, 1188-1196).  The GPU box has one GPU, so
the ranks of these tests share it: two or three processes, each with its own HIP context and its own scene handle, map rank 0's
image over HIP IPC and their kernels store their strips into it; the control collectives run over gloo (RCCL refuses two ranks
on one device -- the data path needs no RCCL).  Rank 0 must then hold the oracle's image, bit for bit."""
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


def _direct_worker(rank, world, scene_name, W, H, depth, bounds, frames, init_file, out_file):
    sys.path.insert(0, ROOT)
    from tilecoderaytracer_amd import HostScene, Renderer
    from tilecoderaytracer_amd.distributed import DirectStrips, SharedImage, balance_direct
    dist.init_process_group("gloo", init_method=f"file://{init_file}", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream(dev).cuda_stream
    renderer = Renderer(HostScene.named(scene_name), device=0)
    image = SharedImage(W, H, dev)
    if rank == 0:
        image.tensor().fill_(-7.0)                       # whatever is not rendered shows
        torch.cuda.synchronize(dev)
    dist.barrier()

    def render_ptr(address, a, b):
        renderer.render_device(W, H, depth, a, b, address, stream)

    if bounds == "measured":                             # bench.py's sequence: equal strips, every rank's kernel time, the re-cut
        pipe = DirectStrips(image, world, rank, dev, render_ptr)
        pipe.step()
        renderer.reset_timing()
        pipe.step()
        tm = renderer.timing()
        bounds, note = balance_direct(W, tm.sum_kernel_ms / max(tm.launches, 1), dev)
        assert "kernel ms per rank" in note
    pipe = DirectStrips(image, world, rank, dev, render_ptr, bounds=bounds)
    for _ in range(frames):
        pipe.step()
    # step() returns when every rank's kernel of the frame has completed: rank 0 may read the image now
    if rank == 0:
        np.save(out_file, pipe.image(W).cpu().numpy())
        assert "straight into rank 0's image" in pipe.describe()
    dist.barrier()
    if rank != 0:
        image.close()                                    # the mappings first ...
    dist.barrier()
    if rank == 0:
        image.close()                                    # ... then the owner's allocation
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("world,scene_name,W,H,depth,bounds", [
    (2, "builtin", 5, 5, 50, None),                                       # the reference's own rehearsal shape: IS_FOR_SIMULATION, 2 cores, 5 x 5 pixels (src/rt_project_parameters.h:45-52)
    (2, "builtin", 128, 96, 4, None),                                     # equal strips
    (2, "builtin", 150, 70, 4, [(0, 37), (37, 150)]),                     # uneven, not on tile boundaries
    (3, "grid16", 96, 64, 8, [(0, 40), (40, 40), (40, 96)]),              # an empty strip in the middle; the clustered-scene kernel
    (2, "grid32", 128, 64, 4, "measured"),                                # the cut from the ranks' own kernel times
])
def test_ranks_render_into_one_shared_image(oracle, world, scene_name, W, H, depth, bounds):
    with tempfile.TemporaryDirectory() as d:
        init_file, out_file = os.path.join(d, "init"), os.path.join(d, "out.npy")
        mp.spawn(_direct_worker, args=(world, scene_name, W, H, depth, bounds, 3, init_file, out_file), nprocs=world, join=True)
        got = np.load(out_file)
    want = oracle.OracleScene.named(scene_name).render(W, H, depth)
    assert got.shape == want.shape
    np.testing.assert_array_equal(got.view(np.uint32), want.view(np.uint32))


def _stream_worker(rank, world, W, H, frames, init_file, out_file):
    """A stream of frames through two shared images: frame k is rendered at depth k % 3 + 1 (a stale frame would show)."""
    sys.path.insert(0, ROOT)
    from tilecoderaytracer_amd import HostScene, Renderer
    from tilecoderaytracer_amd.distributed import DirectStrips, SharedImage
    dist.init_process_group("gloo", init_method=f"file://{init_file}", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream(dev).cuda_stream
    renderer = Renderer(HostScene.named("builtin"), device=0)
    images = [SharedImage(W, H, dev), SharedImage(W, H, dev)]
    frame = {"k": 0}

    def render_ptr(address, a, b):
        renderer.render_device(W, H, frame["k"] % 3 + 1, a, b, address, stream)
        frame["k"] += 1

    pipe = DirectStrips(images, world, rank, dev, render_ptr, bounds=[(0, 50), (50, W)], overlap=True)
    for _ in range(frames):
        pipe.step()
    img = pipe.image(W)                                  # drains: the last frame's all-reduce has completed
    if rank == 0:
        np.save(out_file, img.cpu().numpy())
    dist.barrier()
    if rank != 0:
        for image in images:
            image.close()
    dist.barrier()
    if rank == 0:
        for image in images:
            image.close()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("frames", [4, 5])
def test_a_stream_of_frames_alternates_between_two_shared_images(oracle, frames):
    W, H = 120, 64
    with tempfile.TemporaryDirectory() as d:
        init_file, out_file = os.path.join(d, "init"), os.path.join(d, "out.npy")
        mp.spawn(_stream_worker, args=(2, W, H, frames, init_file, out_file), nprocs=2, join=True)
        got = np.load(out_file)
    want = oracle.OracleScene.named("builtin").render(W, H, (frames - 1) % 3 + 1)
    np.testing.assert_array_equal(got.view(np.uint32), want.view(np.uint32))


@pytest.mark.gpu
def test_shared_image_handle_and_errors():
    """create / destroy in one process; NULL arguments and an empty image are refused."""
    import ctypes as C
    from tilecoderaytracer_amd import capi
    lib = capi.load_library()
    handle = C.create_string_buffer(64)
    p = C.c_void_p()
    capi.check(lib.rt_shared_image_create(0, 1 << 20, C.byref(p), handle))
    assert p.value and any(handle.raw)
    capi.check(lib.rt_shared_image_destroy(0, p))
    assert lib.rt_shared_image_create(0, 0, C.byref(p), handle) != 0 and b"bytes" in lib.rt_last_error()
    assert lib.rt_shared_image_create(0, 1 << 20, None, handle) != 0
    assert lib.rt_shared_image_open(0, None, C.byref(p)) != 0
    capi.check(lib.rt_shared_image_close(0, None))
    capi.check(lib.rt_shared_image_destroy(0, None))
