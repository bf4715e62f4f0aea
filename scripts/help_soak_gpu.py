"""Soak of HELP (development aid): strips and frames of the sphere-grid scenes with help from 2 leaves on and with the
default threshold, several times each (who helps whom depends on timing), compared bit for bit with help off."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tilecoderaytracer_amd import HostScene, Renderer
S = 2048
bad = 0
t0 = time.time()
for name, d in (("grid32", 4), ("grid16", 8), ("grid9", 6), ("grid32-noshadow", 4)):
    rs = {}
    for h in (0, 1, 2):
        rs[h] = Renderer(HostScene.named(name))
        rs[h].set_option("help", h)
    bufs = {h: torch.empty((S, S, 3), dtype=torch.float32, device="cuda:0") for h in rs}
    st = torch.cuda.current_stream().cuda_stream
    for x0, x1 in [(0, S)] + [(k * S // 8, (k + 1) * S // 8) for k in range(8)] + [(3, 70), (S - 130, S - 1)]:
        for rep in range(3):
            for h, r in rs.items():
                bufs[h].zero_()
                r.render_device(S, S, d, x0, x1, bufs[h].data_ptr(), st)
            torch.cuda.synchronize()
            n = (x1 - x0) * S * 3
            ref = bufs[0].view(-1)[:n].view(torch.int32)
            for h in (1, 2):
                if not torch.equal(ref, bufs[h].view(-1)[:n].view(torch.int32)):
                    bad += 1
                    print("MISMATCH", name, x0, x1, "help", h, "rep", rep, flush=True)
    print(name, "done", round(time.time() - t0, 1), "s", flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
