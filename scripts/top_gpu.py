"""The longest tiles of a frame (counting build), development aid.  usage: top_gpu.py scene depth [key=value ...]"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tilecoderaytracer_amd import HostScene, Renderer
name, d = sys.argv[1], int(sys.argv[2])
S = 4096
r = Renderer(HostScene.named(name))
for a in sys.argv[3:]:
    k, v = a.split("=")
    r.set_option(k, int(v))
r.render(64, 64, d)
_, st, cyc = r.render_stats(S, S, d, wave_cycles=True)
dur = (cyc[..., 5].astype(np.float64) - cyc[..., 4].astype(np.float64)) / 100.0
t0 = cyc[..., 4][cyc[..., 4] > 0].min()
print(f"{name} d{d} {sys.argv[3:]}: frame span {(cyc[..., 5].max() - t0) / 100.0:.0f} us (counting build)")
order = np.argsort(dur.ravel())[::-1][:25]
for i in order:
    row, col = np.unravel_index(i, dur.shape)
    print(f"  tile row {row:4d} col {col:3d}: {dur[row, col]:7.0f} us, start {(cyc[row, col, 4] - t0) / 100.0:7.0f} us, sphere tests {cyc[row, col, 1]:6d}, box tests {cyc[row, col, 2]:5d}")
hist, edges = np.histogram(dur.ravel(), bins=[0, 100, 200, 400, 600, 800, 1000, 1500, 2000, 3000, 5000, 1e9])
print("  duration histogram (us):", [(int(edges[k]), int(hist[k])) for k in range(len(hist))])
