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


def gather_strips(strip_buf, views, dst=0):
    """Gather every rank's strip buffer into rank `dst`'s views (None elsewhere)."""
    dist.gather(strip_buf, views if dist.get_rank() == dst else None, dst=dst)
