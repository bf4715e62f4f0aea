"""A/B kernel timing of library variants on one box (development aid).

usage: ab_gpu.py [reps=3] [only=builtin,grid32] name1 name2 ...   (names of lib/variants/libtcrt_<name>.so; `main` = lib/libtcrt.so)
Runs scripts/quick_gpu.py once per variant per repetition, interleaved, and prints the minimum per case."""
import os, subprocess, sys, collections
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
kv = dict(a.split("=") for a in sys.argv[1:] if "=" in a)
names = [a for a in sys.argv[1:] if "=" not in a]
reps = int(kv.pop("reps", 3))
best = collections.defaultdict(dict)
for _ in range(reps):
    for n in names:
        env = dict(os.environ)
        if n != "main":
            env["TCRT_LIBRARY"] = os.path.join(R, "tilecoderaytracer_amd", "lib", "variants", f"libtcrt_{n}.so")
        out = subprocess.run([sys.executable, os.path.join(R, "scripts", "quick_gpu.py")] + [f"{k}={v}" for k, v in kv.items()],
                             env=env, capture_output=True, text=True, check=True).stdout
        for line in out.splitlines():
            f = line.split()
            try:
                k = f.index("ms")
                case, ms = " ".join(f[:k - 2]) if f[0] == "twomirrors" else f[0], float(f[k - 1])
            except (ValueError, IndexError):
                continue
            best[case][n] = min(best[case].get(n, 1e9), ms)
for case, d in best.items():
    print(f"{case:16s} " + "  ".join(f"{n}: {ms:8.3f} ms" for n, ms in d.items()), flush=True)
