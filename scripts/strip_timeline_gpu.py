"""Timeline of one strip's tiles (counting build): when the passes end, which tiles finish last.  Development aid.
usage: strip_timeline_gpu.py scene depth x0 x1 [key=value ...]"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tilecoderaytracer_amd import HostScene, Renderer
name, d, x0, x1 = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
S = 4096
r = Renderer(HostScene.named(name))
for a in sys.argv[5:]:
    k, v = a.split("=")
    r.set_option(k, int(v))
r.render(64, 64, d)
_, st, cyc = r.render_stats(S, S, d, x0, x1, wave_cycles=True)
start = cyc[..., 4].astype(np.float64); end = cyc[..., 5].astype(np.float64)
ok = start > 0
t0 = start[ok].min()
start = (start - t0) / 100.0; end = (end - t0) / 100.0
dur = end - start
print(f"{name} d{d} columns [{x0},{x1}) {sys.argv[5:]}: span {end[ok].max():.0f} us (counting build); tiles {ok.sum()}")
order = np.argsort(end.ravel())[::-1][:15]
print("last finishers (row, col, start us, dur us, sphere tests):")
for i in order:
    row, col = np.unravel_index(i, end.shape)
    print(f"   {row:4d} {col:3d} {start[row, col]:8.0f} {dur[row, col]:7.0f} {cyc[row, col, 1]:6d}")
edges = np.linspace(0, end[ok].max(), 13)
for a, b in zip(edges[:-1], edges[1:]):
    mid = 0.5 * (a + b)
    print(f"   t={mid:7.0f} us: resident tiles {int(((start <= mid) & (end > mid) & ok).sum()):5d}, started in bin {int(((start >= a) & (start < b) & ok).sum()):6d}, mean dur of those {dur[(start >= a) & (start < b) & ok].mean() if ((start >= a) & (start < b) & ok).any() else 0:6.0f} us")
