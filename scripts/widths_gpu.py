import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tilecoderaytracer_amd import HostScene, Renderer
S = 4096
opts = dict(a.split("=") for a in sys.argv[1:] if "=" in a)
for name, d in (("grid32", 4), ("grid16", 8)):
    r = Renderer(HostScene.named(name))
    for k, v in opts.items():
        r.set_option(k, int(v))
    buf = torch.empty((S, S, 3), dtype=torch.float32, device="cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    for x0, x1 in ((0, 4096), (0, 3072), (0, 2048), (2048, 4096), (1024, 3072), (0, 1024), (1536, 2560), (0, 512)):
        r.render_device(S, S, d, x0, x1, buf.data_ptr(), st); torch.cuda.synchronize()
        r.reset_timing()
        for _ in range(3):
            r.render_device(S, S, d, x0, x1, buf.data_ptr(), st)
        torch.cuda.synchronize()
        tm = r.timing()
        ms = tm.sum_kernel_ms / tm.launches
        print(f"{name} columns [{x0},{x1}): {ms:.3f} ms = {ms / (x1 - x0) * 4096:.3f} ms per 4096 columns", flush=True)
