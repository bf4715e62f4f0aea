import sys, time
sys.path.insert(0, '.')
from tilecoderaytracer_amd import HostScene, Renderer
for name, W, H, d in [("builtin", 4096, 4096, 4), ("grid32", 2048, 2048, 4), ("grid16", 2048, 2048, 8)]:
    r = Renderer(HostScene.named(name))
    r.render(256, 256, d)
    r.reset_timing()
    t = time.time(); a = r.render(W, H, d); wall = time.time() - t
    tm = r.timing(); li = r.launch_info()
    print(f"{name} {W}x{H} d{d}: kernel {tm.last_kernel_ms:.3f} ms = {W*H/tm.last_kernel_ms/1e3:.1f} Mrays/s; wall {wall:.3f}s; block {li.block_threads} lds {li.lds_bytes} grid {li.grid_blocks}; mean {a.mean():.6f}", flush=True)
