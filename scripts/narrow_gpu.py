"""Narrow strips through the horizon of the sphere grid: what HELP does to a heavy tile when helpers are free from the start."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tilecoderaytracer_amd import HostScene, Renderer
S = 4096
buf = torch.empty((S, S, 3), dtype=torch.float32, device="cuda:0")
st = torch.cuda.current_stream().cuda_stream
for name, d in (("grid32", 4),):
    for x0, x1 in ((2048, 2064), (2048, 2112), (2048, 2304)):
        for h in (0, 1, 2):
            r = Renderer(HostScene.named(name))
            r.set_option("help", h); r.set_option("defer", 0)
            r.render_device(S, S, d, x0, x1, buf.data_ptr(), st); torch.cuda.synchronize()
            r.reset_timing()
            for _ in range(3):
                r.render_device(S, S, d, x0, x1, buf.data_ptr(), st)
            torch.cuda.synchronize()
            tm = r.timing()
            print(f"{name} columns [{x0},{x1}) help {h}: {tm.sum_kernel_ms / tm.launches:.3f} ms, grid {r.launch_info().grid_blocks}", flush=True)
