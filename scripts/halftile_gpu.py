"""How much shorter does a long tile get with half of its rays?  (development aid, r04_experiments 21)
One tile column of a frame is 1 024 tiles on a chip with 5 000+ wavefront slots: the launch lasts as long as its longest tile.
Rendered whole (16 columns: 64 rays per tile) and as its two halves (8 columns each: 32 active lanes per tile)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tilecoderaytracer_amd import HostScene, Renderer
import torch
S = 4096
st = torch.cuda.current_stream().cuda_stream
buf = torch.empty((64, S, 3), dtype=torch.float32, device="cuda:0")
def t(r, d, a, b, reps=4):
    best = 1e9
    for _ in range(reps):
        r.reset_timing()
        r.render_device(S, S, d, a, b, buf.data_ptr(), st)
        torch.cuda.synchronize()
        tm = r.timing()
        best = min(best, tm.sum_kernel_ms / tm.launches)
    return best
for name, d in (("grid32", 4), ("grid16", 8)):
    r = Renderer(HostScene.named(name))
    whole = torch.empty((S, S, 3), dtype=torch.float32, device="cuda:0")
    for _ in range(4):
        r.render_device(S, S, d, 0, S, whole.data_ptr(), st)
    torch.cuda.synchronize()
    del whole
    rows = []
    for a in range(0, S, 128):
        rows.append((t(r, d, a, a + 16), t(r, d, a, a + 8), t(r, d, a + 8, a + 16), t(r, d, a, a + 4), a))
    rows.sort(reverse=True)
    print(name, "depth", d, ": one tile column [a, a+16) / its halves / a quarter, ms (the eight longest of 32 columns)")
    for full, lo, hi, q, a in rows[:8]:
        print(f"   a={a:5d}  whole {full:7.3f}   halves {lo:7.3f} {hi:7.3f}  ({max(lo, hi) / full:5.2f} of the whole)   quarter {q:7.3f} ({q / full:5.2f})", flush=True)
