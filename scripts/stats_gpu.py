"""Work counters per pixel from the counting build (development aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tilecoderaytracer_amd import HostScene, Renderer
S = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
for name, d in [("builtin", 4), ("grid32", 4), ("grid32-noshadow", 4), ("grid16", 8)]:
    if len(sys.argv) > 1 and sys.argv[1] != "all" and name not in sys.argv[1].split(","):
        continue
    r = Renderer(HostScene.named(name))
    img, st = r.render_stats(S, S, d)
    px = S * S
    print(name, f"{S}x{S} d{d}")
    for k, v in st.items():
        if k.startswith("cycles"):
            print(f"   {k:20s} {v:14d}   share of tile cycles: {v / max(st['cycles_tile'], 1):.3f}")
            continue
        if k.startswith("shadow_") and k != "shadow_rays":
            print(f"   {k:20s} {v:14d}   per shadow scan: {v / max(st['wave_shadow_scans'], 1):.2f}")
            continue
        per = v / px if not k.startswith("wave") else v * 64 / px
        print(f"   {k:20s} {v:14d}   per pixel{' (x64 lanes)' if k.startswith('wave') else ''}: {per:10.2f}")
    ws, ls = st["wave_sphere_tests"] * 64, st["lane_sphere_tests"]
    if ws:
        print(f"   sphere-test lane efficiency (needed / issued): {ls / ws:.3f}")
