"""Kernel-time sweep over the bench workloads (development aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tilecoderaytracer_amd import HostScene, Renderer
import torch
opts = dict(a.split("=") for a in sys.argv[1:] if "=" in a)
cases = [("builtin", 4096, 4), ("grid32", 4096, 4), ("grid32-noshadow", 4096, 4), ("grid16", 4096, 8)]
if "only" in opts:
    cases = [c for c in cases if c[0] in opts["only"].split(",")]
for name, S, d in cases:
    r = Renderer(HostScene.named(name))
    for k in ("tile_z", "block_threads", "cluster_leaf", "grid_mult", "aa_planes", "stack", "first_row", "pairs", "heavy", "help", "fast", "tight_planes", "tables", "wide", "help", "tile_prio"):
        if k in opts:
            r.set_option(k, int(opts[k]))
    buf = torch.empty((S, S, 3), dtype=torch.float32, device="cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(20 if name == "builtin" else 6):          # (the first frames of a process run on clocks that are still rising)
        r.render_device(S, S, d, 0, S, buf.data_ptr(), st)
    torch.cuda.synchronize()
    r.reset_timing()
    n = 20 if name == "builtin" else 8
    for _ in range(n):
        r.render_device(S, S, d, 0, S, buf.data_ptr(), st)
    torch.cuda.synchronize()
    tm = r.timing(); li = r.launch_info()
    ms = tm.sum_kernel_ms / tm.launches
    print(f"{name:16s} {S}x{S} d{d}: {ms:9.3f} ms  {S*S/ms/1e3:9.1f} Mrays/s  block {li.block_threads} lds {li.lds_bytes} tile {li.tile_x}x{li.tile_z} grid {li.grid_blocks}", flush=True)
# the reference's SCENE 2 (two mirrors, 3 920 objects) at 1024^2, depth 50: tables in LDS vs global memory
if "only" not in opts or "twomirrors" in opts["only"]:
    for tables in (1, 2):
        r = Renderer(HostScene.named("twomirrors"))
        r.set_option("tables", tables)
        S, d = 1024, 50
        buf = torch.empty((S, S, 3), dtype=torch.float32, device="cuda:0")
        st = torch.cuda.current_stream().cuda_stream
        r.render_device(S, S, d, 0, S, buf.data_ptr(), st); torch.cuda.synchronize()
        r.reset_timing()
        for _ in range(2):
            r.render_device(S, S, d, 0, S, buf.data_ptr(), st)
        torch.cuda.synchronize()
        tm = r.timing(); li = r.launch_info()
        ms = tm.sum_kernel_ms / tm.launches
        print(f"twomirrors tables={tables} {S}x{S} d{d}: {ms:9.3f} ms  {S*S/ms/1e3:9.1f} Mrays/s  block {li.block_threads} lds {li.lds_bytes} scene lds {li.scene_lds_bytes} grid {li.grid_blocks}", flush=True)
