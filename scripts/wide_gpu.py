"""Strip model (scripts/strips_gpu.py) for several (library variant, options) configurations, interleaved on one box
(development aid): usage wide_gpu.py reps=2 only=grid32,grid16 N=8 -- main: b512:block_threads=512 b1024:block_threads=1024"""
import os, subprocess, sys, collections, re
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
common = [a for a in args if "=" in a and ":" not in a and not a.startswith("reps=")]
reps = int(dict(a.split("=") for a in args if a.startswith("reps=")).get("reps", 2))
configs = [a for a in args if ":" in a]
best = collections.defaultdict(dict)
for _ in range(reps):
    for c in configs:
        lib, _, opts = c.partition(":")
        env = dict(os.environ)
        if lib != "main":
            env["TCRT_LIBRARY"] = os.path.join(R, "tilecoderaytracer_amd", "lib", "variants", f"libtcrt_{lib}.so")
        out = subprocess.run([sys.executable, os.path.join(R, "scripts", "strips_gpu.py")] + common + [o for o in opts.split(",") if o],
                             env=env, capture_output=True, text=True)
        if out.returncode:
            print(c, "FAILED", out.stderr[-400:], flush=True)
            continue
        for line in out.stdout.splitlines():
            m = re.match(r"(\S+)\s+N=(\d+): (full|strips cut)", line)
            if not m:
                continue
            mx = float(re.search(r"max ([0-9.]+)", line).group(1))
            key = (m.group(1), int(m.group(2)), "cut" if m.group(3) != "full" else "equal")
            full = re.search(r"full ([0-9.]+)", line)
            if full:
                best[(m.group(1), 0, "frame")][c] = min(best[(m.group(1), 0, "frame")].get(c, 1e9), float(full.group(1)))
            best[key][c] = min(best[key].get(c, 1e9), mx)
for key in sorted(best):
    print(f"{key[0]:16s} N={key[1]} {key[2]:6s} " + "  ".join(f"{c}: {ms:7.3f}" for c, ms in best[key].items()), flush=True)
