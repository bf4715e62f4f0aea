#!/usr/bin/env python3
"""Copy the summaries of a gpurun_out/<dir> profile run (scripts/profile_all.sh) into profiles/.

usage: save_profiles.py gpurun_out/<dir> <tag>"""
import csv, glob, json, os, shutil, sys
src = sys.argv[1]
tag = sys.argv[2] if len(sys.argv) > 2 else "r03"
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(R, "profiles")
WORKLOADS = ("builtin", "grid32", "grid16d8", "builtin8k", "twomirrors")
NAME = {"builtin": "builtin4096d4"}


def one(pattern):
    g = sorted(glob.glob(os.path.join(src, pattern)), key=os.path.getmtime)     # newest run of that pass
    return g[-1] if g else None


def rows(path, want="rt_render_kernel"):
    with open(path) as f:
        r = list(csv.reader(f))
    return r[0], [x for x in r[1:] if any(want in c for c in x)]


def counter(path):
    """({counter: value of the last dispatch}, kernel name)"""
    h, rs = rows(path)
    i, j, k = h.index("Counter_Name"), h.index("Counter_Value"), h.index("Kernel_Name")
    out, name = {}, None
    for r in rs:
        out[r[i]] = float(r[j])          # last dispatch wins
        name = r[k]
    return out, name


traffic = {"_how": "rocprofv3 --pmc WRITE_SIZE and --pmc FETCH_SIZE in separate passes of `python3 bench.py --workload W --no-cpu-baseline "
                   "--no-extra --steps 5 --warmup 2` (profiles/*_pmc_hbm.csv), last dispatch of the render kernel. Both counters are KiB per "
                   "dispatch. WRITE_SIZE needs no correction: with this kernel's 12-B-per-lane stores it equalled the algorithmic "
                   "201 326 592 B to 5 digits in round 2's first build, which calibrates it. FETCH_SIZE is doubled (gfx950 reports half of "
                   "a wide coalesced read, MI355X_MICROARCH.md HBM section)."}
for w in WORKLOADS:
    name = NAME.get(w, w)
    f = one(f"trace_{w}/*/*_kernel_stats.csv")
    if f:
        shutil.copy(f, os.path.join(P, f"{tag}_{name}_kernel_stats.csv"))
    if w == "builtin":
        t = one(f"trace_{w}/*/*_kernel_trace.csv")
        if t:
            h, rs = rows(t)
            with open(os.path.join(P, f"{tag}_{name}_kernel_trace_head.csv"), "w") as fo:
                wr = csv.writer(fo); wr.writerow(h); wr.writerows(rs[:6])
    fw, ff = one(f"pmc_write_{w}/*/*_counter_collection.csv"), one(f"pmc_fetch_{w}/*/*_counter_collection.csv")
    if fw and ff:
        (cw, kname), (cf, _) = counter(fw), counter(ff)
        traffic[w] = {"kernel": kname, "write_size_kib": cw["WRITE_SIZE"], "fetch_size_kib": cf["FETCH_SIZE"],
                      "hbm_bytes_per_launch": int(cw["WRITE_SIZE"] * 1024 + 2 * cf["FETCH_SIZE"] * 1024)}
        with open(os.path.join(P, f"{tag}_{name}_pmc_hbm.csv"), "w") as fo:
            wr = csv.writer(fo)
            for path in (fw, ff):
                h, rs = rows(path)
                if path == fw:
                    wr.writerow(h)
                wr.writerows(rs[-3:])
    c = {}
    sq_rows, sq_head = [], None
    for d in (f"sq1_{w}", f"sq2_{w}"):
        path = one(f"{d}/*/*_counter_collection.csv")
        if path:
            c.update(counter(path)[0])
            h, rs = rows(path)
            sq_head = sq_head or h
            sq_rows += rs[-8:]
    if sq_rows:
        with open(os.path.join(P, f"{tag}_{name}_pmc_sq.csv"), "w") as fo:
            wr = csv.writer(fo); wr.writerow(sq_head); wr.writerows(sq_rows)
    if w in traffic and "SQ_INSTS_VALU" in c:
        traffic[w]["sq_insts_valu"] = c["SQ_INSTS_VALU"]
        traffic[w]["sq_insts_salu"] = c.get("SQ_INSTS_SALU")
        traffic[w]["sq_insts_lds"] = c.get("SQ_INSTS_LDS")
        traffic[w]["sq_insts_branch"] = c.get("SQ_INSTS_BRANCH")
        traffic[w]["sq_waves"] = c.get("SQ_WAVES")
        if c.get("SQ_BUSY_CYCLES"):
            traffic[w]["sq_busy_cycles_per_engine"] = c["SQ_BUSY_CYCLES"] / 32.0
sys.path.insert(0, R)
import bench  # noqa: E402  (kernel_source_digest only)
traffic["kernel_source_sha256"] = bench.kernel_source_digest()
json.dump(traffic, open(os.path.join(P, "pmc_traffic.json"), "w"), indent=1)
for w in ("builtin", "grid32", "grid16d8", "grid32-noshadow", "builtin8k", "twomirrors", "shipped", "shipped512"):
    f = os.path.join(src, f"bench_{w}.json")
    if os.path.exists(f) and os.path.getsize(f) > 0:
        shutil.copy(f, os.path.join(P, f"{tag}_bench_{w}.json"))
print(json.dumps(traffic, indent=1)[:1200])
