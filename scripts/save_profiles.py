#!/usr/bin/env python3
"""Copy the summaries of a gpurun_out/<dir> profile run into profiles/."""
import csv, glob, json, os, shutil, sys
src = sys.argv[1]
tag = sys.argv[2] if len(sys.argv) > 2 else "r02"
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(R, "profiles")

def one(pattern):
    g = sorted(glob.glob(os.path.join(src, pattern)), key=os.path.getmtime)     # newest run of that pass
    return g[-1] if g else None

def rows(path, want="rt_render_kernel"):
    with open(path) as f:
        r = list(csv.reader(f))
    return r[0], [x for x in r[1:] if any(want in c for c in x)]

def counter(path):
    h, rs = rows(path)
    i, j = h.index("Counter_Name"), h.index("Counter_Value")
    out = {}
    for r in rs:
        out[r[i]] = float(r[j])          # last dispatch wins
    return out

f0 = one("trace/*/*_kernel_stats.csv")
if f0: shutil.copy(f0, os.path.join(P, f"{tag}_builtin4096d4_kernel_stats.csv"))
for w in ("grid32", "grid16d8"):
    f = one(f"trace_{w}/*/*_kernel_stats.csv")
    if f: shutil.copy(f, os.path.join(P, f"{tag}_{w}_kernel_stats.csv"))
h, rs = rows(one("trace/*/*_kernel_trace.csv"))
with open(os.path.join(P, f"{tag}_builtin4096d4_kernel_trace_head.csv"), "w") as f:
    w = csv.writer(f); w.writerow(h); w.writerows(rs[:6])
traffic = {"_how": "rocprofv3 --pmc WRITE_SIZE and --pmc FETCH_SIZE in separate passes of `python3 bench.py [--workload W] --no-cpu-baseline --steps 5 --warmup 1` (profiles/*_pmc_hbm.csv). Both counters are KiB per dispatch. WRITE_SIZE needs no correction: with this kernel's 12-B-per-lane stores it equalled the algorithmic 201 326 592 B to 5 digits in the first build of the round, which calibrates it. FETCH_SIZE is doubled (gfx950 reports half of a wide coalesced read, MI355X_MICROARCH.md HBM section)."}
for w, suffix in (("builtin", ""), ("grid32", "_grid32"), ("grid16d8", "_grid16d8")):
    fw, ff = one(f"pmc_write{suffix}/*/*_counter_collection.csv"), one(f"pmc_fetch{suffix}/*/*_counter_collection.csv")
    if not fw or not ff: continue
    cw, cf = counter(fw)["WRITE_SIZE"], counter(ff)["FETCH_SIZE"]
    traffic[w] = {"write_size_kib": cw, "fetch_size_kib": cf, "hbm_bytes_per_launch": int(cw * 1024 + 2 * cf * 1024)}
    name = "builtin4096d4" if w == "builtin" else w
    with open(os.path.join(P, f"{tag}_{name}_pmc_hbm.csv"), "w") as f:
        wr = csv.writer(f)
        for path in (fw, ff):
            h, rs = rows(path)
            if path == fw: wr.writerow(h)
            wr.writerows(rs[-3:])
sys.path.insert(0, R)
import bench  # noqa: E402  (kernel_source_digest only)
traffic["kernel_source_sha256"] = bench.kernel_source_digest()
for w, dirs in (("builtin", ("sq1", "sq2")), ("grid32", ("sq1_grid32", "sq2_grid32")), ("grid16d8", ("sq1_grid16d8", "sq2_grid16d8"))):
    c = {}
    for d in dirs:
        path = one(f"{d}/*/*_counter_collection.csv")
        if path: c.update(counter(path))
    if w in traffic and "SQ_INSTS_VALU" in c:
        traffic[w]["sq_insts_valu"] = c["SQ_INSTS_VALU"]
        traffic[w]["sq_insts_salu"] = c.get("SQ_INSTS_SALU")
        traffic[w]["sq_insts_lds"] = c.get("SQ_INSTS_LDS")
        traffic[w]["sq_waves"] = c.get("SQ_WAVES")
        if c.get("SQ_BUSY_CYCLES"): traffic[w]["sq_busy_cycles_per_engine"] = c["SQ_BUSY_CYCLES"] / 32.0
json.dump(traffic, open(os.path.join(P, "pmc_traffic.json"), "w"), indent=1)
with open(os.path.join(P, f"{tag}_builtin4096d4_pmc_sq.csv"), "w") as f:
    wr = csv.writer(f)
    first = True
    for d in ("sq1", "sq2"):
        path = one(f"{d}/*/*_counter_collection.csv")
        if not path: continue
        h, rs = rows(path)
        if first: wr.writerow(h); first = False
        wr.writerows(rs[-8:])
with open(os.path.join(P, f"{tag}_grid32_pmc_sq.csv"), "w") as f:
    wr = csv.writer(f)
    first = True
    for d in ("sq1_grid32", "sq2_grid32"):
        path = one(f"{d}/*/*_counter_collection.csv")
        if not path: continue
        h, rs = rows(path)
        if first: wr.writerow(h); first = False
        wr.writerows(rs[-8:])
for w in ("builtin", "grid32", "grid16d8", "grid32-noshadow", "builtin8k", "twomirrors"):
    f = os.path.join(src, f"bench_{w}.json")
    if os.path.exists(f): shutil.copy(f, os.path.join(P, f"{tag}_bench_{w}.json"))
print(json.dumps(traffic, indent=1)[:600])
