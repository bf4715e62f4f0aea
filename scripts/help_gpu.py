"""help=0 against help=1 (rt_set_option): the same pixels, and the kernel times, whole frames and strips."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tilecoderaytracer_amd import HostScene, Renderer
cases = (("grid16", 128, 128, 8, None), ("grid32", 256, 256, 4, None),
         ("grid32", 4096, 4096, 4, None), ("grid16", 4096, 4096, 8, None),
         ("grid32", 4096, 4096, 4, (1536, 2048)), ("grid32", 4096, 4096, 4, (1024, 2048)), ("grid32", 4096, 4096, 4, (0, 2048)))
for name, W, H, d, strip in cases:
    imgs, ms = {}, {}
    for h in (0, 1):
        r = Renderer(HostScene.named(name))
        r.set_option("help", h)
        best = 1e9
        for rep in range(4 if W > 1000 else 1):
            imgs[h] = r.render(W, H, d) if strip is None else r.render(W, H, d, strip[0], strip[1])
            best = min(best, r.timing().last_kernel_ms)
        ms[h] = best
        li = r.launch_info()
        print(name, W, strip, "help", h, "kernel ms", round(best, 3), "lds", li.lds_bytes, "grid", li.grid_blocks, flush=True)
    diff = (imgs[0].view(np.uint32) != imgs[1].view(np.uint32)).any(axis=-1)
    print(name, W, strip, "pixels differing:", int(diff.sum()), "speedup", round(ms[0] / ms[1], 3), flush=True)
