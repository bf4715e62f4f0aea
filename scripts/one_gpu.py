"""Render one workload a few times (for rocprofv3 runs).  usage: one_gpu.py scene depth [key=value ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tilecoderaytracer_amd import HostScene, Renderer
name, d = sys.argv[1], int(sys.argv[2])
S = 4096
r = Renderer(HostScene.named(name))
for a in sys.argv[3:]:
    k, v = a.split("=")
    if k == "size":
        S = int(v)
    else:
        r.set_option(k, int(v))
buf = torch.empty((S, S, 3), dtype=torch.float32, device="cuda:0")
st = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    r.render_device(S, S, d, 0, S, buf.data_ptr(), st)
torch.cuda.synchronize()
tm = r.timing()
print(f"{name} d{d} {sys.argv[3:]}: last kernel {tm.last_kernel_ms:.3f} ms (second pass {tm.last_second_pass_ms:.3f}), deferred {r.launch_info().deferred_tiles}")
