"""Deferred-tile A/B (development aid): kernel time, N=8 strip times and the longest tile for defer / slices settings."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tilecoderaytracer_amd import HostScene, Renderer
from tilecoderaytracer_amd.distributed import strip_bounds
import torch
S = 4096
name, d = (sys.argv[1] if len(sys.argv) > 1 else "grid32"), int(sys.argv[2]) if len(sys.argv) > 2 else 4
buf = torch.empty((S, S, 3), dtype=torch.float32, device="cuda:0")
st = torch.cuda.current_stream().cuda_stream
ref = None
cases = [(0, 256), (-1, 256)]
if len(sys.argv) > 3:
    cases = [tuple(int(v) for v in a.split(",")) for a in sys.argv[3:]]
for band, slices in cases:
    r = Renderer(HostScene.named(name))
    r.set_option("defer", band); r.set_option("second_block", slices)
    def t(x0, x1, n=3):
        r.render_device(S, S, d, x0, x1, buf.data_ptr(), st); torch.cuda.synchronize()
        r.reset_timing()
        for _ in range(n):
            r.render_device(S, S, d, x0, x1, buf.data_ptr(), st)
        torch.cuda.synchronize()
        tm = r.timing()
        return tm.sum_kernel_ms / tm.launches
    full = t(0, S)
    tm, li = r.timing(), r.launch_info()
    second, ndef = tm.last_second_pass_ms, li.deferred_tiles
    img = buf.clone()
    if ref is None:
        ref = img
    same = bool(torch.equal(ref.view(torch.int32), img.view(torch.int32)))
    strips, seconds = [], []
    for k in range(8):
        strips.append(t(*strip_bounds(S, 8, k)[:2]))
        seconds.append((round(r.timing().last_second_pass_ms, 2), r.launch_info().deferred_tiles))
    _, stt, cyc = r.render_stats(S, S, d, wave_cycles=True)
    dur = (cyc[..., 5].astype(np.float64) - cyc[..., 4].astype(np.float64)) / 100.0
    dur = np.where(cyc[..., 4] > 0, dur, 0)
    print(f"{name} defer {band:4d} block {slices:3d}: full {full:7.3f} ms  N=8 strips max {max(strips):6.3f} ms -> {full / max(strips):5.2f}x (vs band0 full)  "
          f"longest tile (counting build) {dur.max():7.0f} us  same image {same}  second pass {second:6.3f} ms, {ndef} tiles deferred; strips (total ms, second pass ms, deferred): {[(round(a, 2),) + b for a, b in zip(strips, seconds)]}", flush=True)
