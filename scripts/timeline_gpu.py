"""Timeline of one launch's tiles from the production kernels' code (a diagnostic build: make -C tilecoderaytracer_amd/csrc
variant NAME=timeline DEFS=-DRT_TIMELINE=1, then TCRT_LIBRARY=.../lib/variants/libtcrt_timeline.so): when the queues run dry,
which tiles finish last, how busy the wavefront slots are.  Development aid.
usage: timeline_gpu.py scene depth x0 x1 [key=value ...]"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tilecoderaytracer_amd import HostScene, Renderer
import torch
name, d, x0, x1 = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
S = 4096
r = Renderer(HostScene.named(name))
for a in sys.argv[5:]:
    k, v = a.split("=")
    r.set_option(k, int(v))
buf = torch.empty((x1 - x0, S, 3), dtype=torch.float32, device="cuda:0")
st = torch.cuda.current_stream().cuda_stream
for _ in range(2):
    r.render_device(S, S, d, x0, x1, buf.data_ptr(), st)
torch.cuda.synchronize()
r.set_option("timeline", 1)
r.reset_timing()
r.render_device(S, S, d, x0, x1, buf.data_ptr(), st)
torch.cuda.synchronize()
tm = r.timing()
rec = r.timeline(x0, x1, S)
start, end, who, heavy = (rec[..., k].astype(np.float64) for k in range(4))
ok = start > 0
t0 = start[ok].min()
start = (start - t0) / 100.0; end = (end - t0) / 100.0
dur = end - start
span = end[ok].max()
slots = len(np.unique(who[ok]))
busy = dur[ok].sum() / (slots * span)
print(f"{name} d{d} columns [{x0},{x1}) {sys.argv[5:]}: kernel {tm.last_kernel_ms * 1e3:.0f} us, tiles' span {span:.0f} us, tiles {int(ok.sum())}, heavy {int(heavy[ok].sum())}, "
      f"wavefront slots used {slots}, mean tile {dur[ok].mean():.1f} us, max {dur[ok].max():.0f} us, slots busy {busy:.2f} of the span")
order = np.argsort(end.ravel())[::-1][:12]
print("last finishers (row, col, start us, dur us, heavy):")
for i in order:
    row, col = np.unravel_index(i, end.shape)
    print(f"   {row:4d} {col:3d} {start[row, col]:8.0f} {dur[row, col]:7.0f} {int(heavy[row, col])}")
edges = np.linspace(0, span, 11)
for a, b in zip(edges[:-1], edges[1:]):
    mid = 0.5 * (a + b)
    res = (start <= mid) & (end > mid) & ok
    sel = (start >= a) & (start < b) & ok
    print(f"   t={mid:7.0f} us: tiles in flight {int(res.sum()):5d}, started in bin {int(sel.sum()):6d}, mean dur of those {dur[sel].mean() if sel.any() else 0:6.0f} us, max {dur[sel].max() if sel.any() else 0:6.0f}")
h = np.histogram(dur[ok], bins=[0, 25, 50, 100, 200, 400, 800, 1600, 1e9])
print("tile duration histogram (us):", dict(zip(["<25", "<50", "<100", "<200", "<400", "<800", "<1600", ">=1600"], h[0].tolist())))
# per wavefront slot: when it got its first tile, when it ran out, how long it sat between tiles
ids = who[ok].astype(np.int64); s_ = start[ok]; e_ = end[ok]
order = np.lexsort((s_, ids))
ids, s_, e_ = ids[order], s_[order], e_[order]
first = np.r_[True, ids[1:] != ids[:-1]]; last = np.r_[ids[1:] != ids[:-1], True]
gaps = (s_[1:] - e_[:-1])[~first[1:]]
print(f"slots: first tile starts at {np.percentile(s_[first], [0, 10, 50, 90, 100]).round(1).tolist()} us (min/p10/median/p90/max); "
      f"last tile ends at {np.percentile(e_[last], [0, 10, 50, 90, 100]).round(1).tolist()} us; "
      f"gap between consecutive tiles of a slot: mean {gaps.mean():.2f} us, p90 {np.percentile(gaps, 90):.2f}, max {gaps.max():.1f}; tiles per slot mean {len(ids) / first.sum():.2f}")
if os.environ.get("TIMELINE_DUMP"):
    np.savez_compressed(os.environ["TIMELINE_DUMP"], start=start, end=end, who=who, heavy=heavy, ok=ok)
