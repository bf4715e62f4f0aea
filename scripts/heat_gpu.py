"""Per-wavefront-tile work distribution from the counting build (development aid)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tilecoderaytracer_amd import HostScene, Renderer
name = sys.argv[1] if len(sys.argv) > 1 else "grid32"
S = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
d = int(sys.argv[3]) if len(sys.argv) > 3 else 4
r = Renderer(HostScene.named(name))
r.render(64, 64, d)     # sets launch info (tile shape)
img, st, t = r.render_stats(S, S, d, wave_cycles=True)
t = t.astype(np.float64)
for k, nm in enumerate(["cycles", "sphere tests", "box tests", "scans"]):
    c = t[..., k]
    print(f"{nm:13s} sum {c.sum():.3e} mean {c.mean():9.0f} median {np.median(c):9.0f} p99 {np.percentile(c, 99):9.0f} max {c.max():9.0f}")
    rows = c.mean(axis=1).reshape(32, -1).mean(axis=1)
    print("   per tile-row band (bottom->top):", " ".join(f"{v:.0f}" for v in rows))
# schedule: how many wavefronts are resident over time
t0 = t[..., 4].min()
start = (t[..., 4] - t0).ravel() / 100.0   # microseconds
end = (t[..., 5] - t0).ravel() / 100.0
total = end.max()
print(f"kernel span {total:.0f} us; wave duration mean {np.mean(end-start):.0f} us max {np.max(end-start):.0f} us")
edges = np.linspace(0, total, 21)
for a, b in zip(edges[:-1], edges[1:]):
    mid = 0.5 * (a + b)
    resident = np.sum((start <= mid) & (end > mid))
    started = np.sum((start >= a) & (start < b))
    print(f"   t={mid:8.0f} us resident waves {resident:6d}  started in bin {started:6d}")
late = np.argsort(end)[-8:]
tx = t.shape[1]
print("last finishers (tile_row, tile_col, start us, dur us):", [(int(i // tx), int(i % tx), int(start[i]), int(end[i]-start[i])) for i in late])
heavy = np.argsort(t[..., 0].ravel())[-12:]
print("heaviest tiles (tile_row, tile_col, Mcycles, sphere tests, box tests, start us, dur us):")
for i in heavy:
    r_, c_ = int(i // tx), int(i % tx)
    print("  ", r_, c_, f"{t[r_, c_, 0] / 1e6:.2f}", int(t[r_, c_, 1]), int(t[r_, c_, 2]), int(start[i]), int(end[i] - start[i]))
rows_ = t[..., 0].sum(axis=1)
order = np.argsort(rows_)[-8:]
print("heaviest tile rows (row, share of all cycles):", [(int(r_), round(float(rows_[r_] / rows_.sum()), 4)) for r_ in order])
