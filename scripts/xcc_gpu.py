import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tilecoderaytracer_amd import HostScene, Renderer
r = Renderer(HostScene.named("builtin"))
r.render(64, 64, 4)
img, st, t = r.render_stats(1024, 1024, 4, wave_cycles=True)
x = t[..., 3].astype(np.int64)
xcc, steal = x // 1000, x % 1000
print("tiles", x.shape, "xcc histogram", np.bincount(xcc.ravel(), minlength=8), "steal histogram", np.bincount(steal.ravel(), minlength=8))
print("xcc of first 4 tile rows, first 16 columns:\n", xcc[:4, :16])
