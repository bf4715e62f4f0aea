import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tilecoderaytracer_amd import HostScene, Renderer
for name, d in [("grid32", 4), ("grid16", 8), ("builtin", 4)]:
    r = Renderer(HostScene.named(name))
    _, st = r.render_stats(1024, 1024, d)
    print(name, d, "nearest scans", st["wave_nearest_scans"], "unculled", st["nearest_scans_unculled"], "box tests all", st["wave_box_tests"],
          "box tests in unculled nearest scans", st["nearest_unculled_box_tests"], "sphere tests all", st["wave_sphere_tests"], "in nearest scans", st["nearest_sphere_tests"], "in unculled nearest scans", st["nearest_unculled_sphere_tests"])
