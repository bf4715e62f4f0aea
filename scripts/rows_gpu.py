"""Per tile row near the horizon: longest and mean tile (counting build), development aid.
usage: rows_gpu.py scene depth band slices [row_lo row_hi]"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tilecoderaytracer_amd import HostScene, Renderer
name, d, band, slices = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
lo, hi = (int(sys.argv[5]), int(sys.argv[6])) if len(sys.argv) > 6 else (480, 545)
S = 4096
r = Renderer(HostScene.named(name))
r.set_option("block_threads", slices)
r.render(64, 64, d)
_, st, cyc = r.render_stats(S, S, d, wave_cycles=True)
dur = (cyc[..., 5].astype(np.float64) - cyc[..., 4].astype(np.float64)) / 100.0
print(f"{name} d{d} band {band} slices {slices}: tile rows {cyc.shape[0]}; whole-frame longest tile {dur.max():.0f} us at row {np.unravel_index(dur.argmax(), dur.shape)[0]}")
for row in range(lo, hi):
    print(f"row {row}: max {dur[row].max():7.0f} us mean {dur[row].mean():7.0f} us  cycles max {cyc[row, :, 0].max() / 1e6:6.2f} M  sphere tests max {cyc[row, :, 1].max():6d}  box tests max {cyc[row, :, 2].max():5d}")
