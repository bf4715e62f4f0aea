"""Kernel time of each rank's x-strip on ONE GPU: what the static partition's load balance
would be on N GPUs (development aid; the N-GPU runs themselves are the driver's)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tilecoderaytracer_amd import HostScene, Renderer
from tilecoderaytracer_amd.distributed import strip_bounds
import torch
S = 4096
NS = tuple(int(a[2:]) for a in sys.argv[1:] if a.startswith("N=")) or (2, 4, 8)      # N=8 cut=0: equal strips of 8 only
CUT = "cut=0" not in sys.argv[1:]
ONLY = [a[5:].split(",") for a in sys.argv[1:] if a.startswith("only=")]
for name, d in [("builtin", 4), ("grid32", 4), ("grid16", 8), ("grid32-noshadow", 4)]:
    if ONLY and name not in ONLY[0]:
        continue
    r = Renderer(HostScene.named(name))
    for a in sys.argv[1:]:
        if "=" in a and not a.startswith(("N=", "cut=", "only=", "learn=")):
            r.set_option(a.split("=")[0], int(a.split("=")[1]))
    buf = torch.empty((S, S, 3), dtype=torch.float32, device="cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    LEARN = "learn=1" in sys.argv[1:]                  # learn=1: rt_learn_tile_order for every strip before it is timed
    def t(x0, x1, n=6):
        if LEARN and (x0, x1) != (0, S):
            r.learn_tile_order(S, S, d, x0, x1)
        for _ in range(3):                      # (a process's first launches run on clocks that are still rising)
            r.render_device(S, S, d, x0, x1, buf.data_ptr(), st)
        torch.cuda.synchronize()
        r.reset_timing()
        for _ in range(n):
            r.render_device(S, S, d, x0, x1, buf.data_ptr(), st)
        torch.cuda.synchronize()
        tm = r.timing()
        return tm.sum_kernel_ms / tm.launches
    t(0, S)
    full = t(0, S)
    for N in NS:
        ts = [t(*strip_bounds(S, N, k)[:2]) for k in range(N)]
        print(f"{name:8s} N={N}: full {full:.3f} ms; strips " + " ".join(f"{x:.3f}" for x in ts) +
              f"; max {max(ts):.3f} sum {sum(ts):.3f} -> render-bound speedup {full / max(ts):.2f}x", flush=True)
        if not CUT:
            continue
        # what bench.py does at N > 1: re-cut the strips by the measured cost (balanced_bounds, nothing to send in this
        # render-only model), twice; boundaries on multiples of 16 columns
        import numpy as np
        from tilecoderaytracer_amd.distributed import balanced_bounds
        bounds = [strip_bounds(S, N, k)[:2] for k in range(N)]
        for _ in range(2):
            cost = np.zeros(S)
            for (a, b), tk in zip(bounds, ts):
                cost[a:b] = tk / max(b - a, 1)
            bounds = balanced_bounds(S, N, cost, 0.0)
            bounds = [(min(S, (a + 8) // 16 * 16) if a else 0, S if b == S else min(S, (b + 8) // 16 * 16)) for a, b in bounds]
            ts = [t(a, b) if b > a else 0.0 for a, b in bounds]
        print(f"{name:8s} N={N}: strips cut by measured cost " + "/".join(str(b - a) for a, b in bounds) + " columns: " +
              " ".join(f"{x:.3f}" for x in ts) + f"; max {max(ts):.3f} -> render-bound speedup {full / max(ts):.2f}x", flush=True)
