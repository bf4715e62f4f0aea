"""How full are the wavefronts' nearest-hit scans (what bounce compaction could gain), counting build."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tilecoderaytracer_amd import HostScene, Renderer
S = 2048
for name, d in [("builtin", 4), ("grid32", 4), ("grid16", 8), ("grid16", 13), ("twomirrors", 50)]:
    r = Renderer(HostScene.named(name))
    if name == "twomirrors":
        r.set_option("tables", 1)
    _, st = r.render_stats(S, S, d)
    b = [st["nearest_scans_1_16"], st["nearest_scans_17_32"], st["nearest_scans_33_48"], st["nearest_scans_49_64"]]
    tot = sum(b)
    lanes = st["nearest_rays"] / (64.0 * tot)
    # merging partially filled scans four at a time (a workgroup's wavefronts): scans that would remain
    merged = b[3] + b[2] + b[1] / 2.0 + b[0] / 4.0
    print(f"{name:10s} d{d:2d}: scans by live lanes 1-16/17-32/33-48/49-64 = {[round(x / tot, 3) for x in b]}; "
          f"lane utilisation {lanes:.3f}; ideal 4-way merge would leave {merged / tot:.3f} of the scans", flush=True)
