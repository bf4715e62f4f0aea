"""Kernel time of a whole 4096^2 frame under option settings (development aid).
usage: sweep_gpu.py scene depth key=v1,v2,... [key=v1,...]   (one option varied at a time, the others at their defaults)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tilecoderaytracer_amd import HostScene, Renderer
import torch
name, d = sys.argv[1], int(sys.argv[2])
S = 4096
buf = torch.empty((S, S, 3), dtype=torch.float32, device="cuda:0")
st = torch.cuda.current_stream().cuda_stream
def t(r, n=4):
    r.render_device(S, S, d, 0, S, buf.data_ptr(), st); torch.cuda.synchronize()
    r.reset_timing()
    for _ in range(n):
        r.render_device(S, S, d, 0, S, buf.data_ptr(), st)
    torch.cuda.synchronize()
    tm = r.timing()
    return tm.sum_kernel_ms / tm.launches
r = Renderer(HostScene.named(name))
print(f"{name} d{d} defaults: {t(r):.3f} ms  {r.launch_info().kernel if hasattr(r, 'launch_info') else ''}", flush=True)
for a in sys.argv[3:]:
    k, vs = a.split("=")
    for v in vs.split(","):
        r = Renderer(HostScene.named(name))
        try:
            r.set_option(k, int(v))
            ms = t(r)
            li = r.launch_info()
            print(f"   {k}={v}: {ms:.3f} ms   ({li.kernel.decode()}, {li.lds_bytes} B LDS of which tables {li.scene_lds_bytes}, {li.grid_blocks} workgroups of {li.block_threads})", flush=True)
        except Exception as e:
            print(f"   {k}={v}: {type(e).__name__} {str(e)[:80]}", flush=True)
