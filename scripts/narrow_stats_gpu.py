"""Counting build on a one-tile-column strip through the horizon of the sphere grid: what the heaviest tiles spend their time on."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tilecoderaytracer_amd import HostScene, Renderer
S = 4096
r = Renderer(HostScene.named("grid32"))
r.set_option("defer", 0)
r.render(64, 64, 4)
_, st, cyc = r.render_stats(S, S, 4, 2048, 2064, wave_cycles=True)
for k, v in st.items():
    print(f"   {k:28s} {v}")
dur = (cyc[..., 5].astype(np.float64) - cyc[..., 4].astype(np.float64)) / 100.0
order = np.argsort(dur.ravel())[::-1][:8]
print("longest tiles (row, col, dur us, cycles, sphere tests, box tests, scans):")
for i in order:
    row, col = np.unravel_index(i, dur.shape)
    print(f"   {row:4d} {col:3d} {dur[row, col]:8.0f} {cyc[row, col, 0]:10d} {cyc[row, col, 1]:6d} {cyc[row, col, 2]:6d} {cyc[row, col, 3]:4d}")
