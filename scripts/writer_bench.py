"""Throughput of the byte-exact raytracer_screen.txt writer (csrc/host/screen_txt.hpp: integer %f, one thread per slice of the
columns; the reference's printPixelsToLog, src/RayTracer.cpp:1574-1626, is fprintf("(%f, %f, %f)\\n") per pixel -- about 10 s for
4096^2 by SURVEY.md section 6).  Host only, no GPU.  usage: writer_bench.py [sizes ...]  (default 4096 8192); writes to /dev/shm
(memory) and to the working directory's file system, and formats the same pixels with C printf through Python for a sample."""
import os, sys, time, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tilecoderaytracer_amd import host
sizes = [int(a) for a in sys.argv[1:]] or [4096, 8192]
rng = np.random.RandomState(1)
print(f"host: {os.cpu_count()} logical CPUs, affinity {len(os.sched_getaffinity(0))}")
for S in sizes:
    # pixel values like a frame's: most in [0, 1], some above (no final clamp in the reference), exact zeros
    img = rng.uniform(0.0, 1.2, (S, S, 3)).astype(np.float32)
    img[rng.rand(S, S) < 0.1] = 0.0
    for where in ("/dev/shm", tempfile.gettempdir()):
        path = os.path.join(where, f"tcrt_writer_bench_{os.getpid()}.txt")
        times = []
        for rep in range(3):
            t0 = time.perf_counter()
            host.write_screen_txt(path, img)
            times.append(time.perf_counter() - t0)
        size = os.path.getsize(path)
        os.unlink(path)
        t = min(times)
        print(f"{S}x{S}: {size / 1e6:8.1f} MB to {where:10s} in {t:6.2f} s = {size / t / 1e6:7.1f} MB/s = {S * S / t / 1e6:6.2f} Mpixel/s "
              f"(runs {[round(x, 2) for x in times]})", flush=True)
    # the reference's way on a sample of 2^20 pixels: C's printf("%f") three times per pixel, one thread
    n = 1 << 20
    sample = img.reshape(-1, 3)[:n].astype(np.float64)
    t0 = time.perf_counter()
    text = "".join("(%f, %f, %f)\n" % (r, g, b) for r, g, b in sample)
    dt = time.perf_counter() - t0
    print(f"   (Python '%f' formatting of {n} pixels, one thread: {dt:.2f} s -> {S * S / n * dt:.1f} s for the frame; the reference's fprintf "
          f"took about 10 s at 4096^2, SURVEY.md section 6)")
