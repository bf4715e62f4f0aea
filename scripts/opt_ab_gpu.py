"""A/B of ONE run-time option inside one process, interleaved, with the frames compared bit for bit (development aid).

usage: opt_ab_gpu.py <option> <value> <value> ... [reps=3] [only=grid32,grid16] [stats=1] [strip=x0:x1]
Prints, per scene, the kernel ms of every value (minimum and mean over the repetitions) and whether all values gave the same
image; stats=1 adds the counting build's shadow-scan candidates per scan at 1024^2."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tilecoderaytracer_amd import HostScene, Renderer

args = [a for a in sys.argv[1:] if "=" not in a]
kv = dict(a.split("=") for a in sys.argv[1:] if "=" in a)
option, values = args[0], [int(v) for v in args[1:]]
reps = int(kv.get("reps", 3))
S = int(kv.get("size", 4096))
cases = [("builtin", 4), ("grid32", 4), ("grid16", 8), ("grid32-noshadow", 4)]
if "only" in kv:
    cases = [c for c in cases if c[0] in kv["only"].split(",")]
x0, x1 = (int(v) for v in kv["strip"].split(":")) if "strip" in kv else (0, S)
for name, d in cases:
    r = Renderer(HostScene.named(name))
    st = torch.cuda.current_stream().cuda_stream
    bufs = {v: torch.zeros((x1 - x0, S, 3), dtype=torch.float32, device="cuda:0") for v in values}
    times = {v: [] for v in values}
    for _ in range(reps):
        for v in values:
            r.set_option(option, v)
            for _ in range(3):
                r.render_device(S, S, d, x0, x1, bufs[v].data_ptr(), st)
            torch.cuda.synchronize()
            r.reset_timing()
            n = 20 if name == "builtin" else 6
            for _ in range(n):
                r.render_device(S, S, d, x0, x1, bufs[v].data_ptr(), st)
            torch.cuda.synchronize()
            tm = r.timing()
            times[v].append(tm.sum_kernel_ms / tm.launches)
    same = all(torch.equal(bufs[values[0]].view(torch.int32), bufs[v].view(torch.int32)) for v in values[1:])
    li = r.launch_info()
    print(f"{name:16s} {option}: " + "  ".join(f"{v}: min {min(times[v]):.3f} mean {np.mean(times[v]):.3f} ms" for v in values) +
          f"  images {'identical' if same else 'DIFFER'}  ({li.kernel.decode() if isinstance(li.kernel, bytes) else li.kernel}, block {li.block_threads})", flush=True)
    if kv.get("stats") == "1":
        for v in values:
            r.set_option(option, v)
            _, stt = r.render_stats(1024, 1024, d)
            scans = max(stt["wave_shadow_scans"], 1)
            print(f"    {option}={v}: shadow candidates per scan {stt['shadow_candidates'] / scans:.2f}, leaves needed {stt['shadow_leaves_union'] / scans:.2f}, "
                  f"box tests per 64 px {stt['wave_box_tests'] * 64 / 1024 / 1024:.1f}, sphere tests {stt['wave_sphere_tests'] * 64 / 1024 / 1024:.1f}", flush=True)
