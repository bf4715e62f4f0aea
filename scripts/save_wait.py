#!/usr/bin/env python3
"""Summaries of scripts/profile_wait.sh's counter passes -> profiles/<tag>_<workload>_pmc_wait.csv and a reading.

usage: save_wait.py gpurun_out/<dir> <tag> workload...
Per workload: the last dispatch of the render kernel in each of the three passes, one row per counter, plus the
derived shares the stall attribution in DESIGN.md quotes (all of them ratios of counters of ONE pass, or of
per-launch totals of the same kernel)."""
import csv, glob, json, os, sys
src, tag = sys.argv[1], sys.argv[2]
workloads = sys.argv[3:] or ["builtin", "grid32", "grid16d8"]
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(R, "profiles")


def last_dispatch(path, want="rt_render_kernel"):
    with open(path) as f:
        r = list(csv.reader(f))
    h = r[0]
    kn, dn, cn, cv = h.index("Kernel_Name"), h.index("Dispatch_Id"), h.index("Counter_Name"), h.index("Counter_Value")
    rows = [x for x in r[1:] if want in x[kn]]
    if not rows:
        return None, {}
    last = max(int(x[dn]) for x in rows)
    out = {}
    for x in rows:
        if int(x[dn]) == last:
            out[x[cn]] = out.get(x[cn], 0.0) + float(x[cv])
    return rows[-1][kn], out


summary = {}
for w in workloads:
    c, kernel = {}, None
    passes = {}
    for p in "ABCD":
        g = sorted(glob.glob(os.path.join(src, f"wait_{w}_{p}", "*", "*_counter_collection.csv")), key=os.path.getmtime)
        if not g:
            continue
        k, d = last_dispatch(g[-1])
        kernel = kernel or k
        passes[p] = d
        c.update(d)
    if not c:
        continue
    name = "builtin4096d4" if w == "builtin" else w
    with open(os.path.join(P, f"{tag}_{name}_pmc_wait.csv"), "w") as f:
        wr = csv.writer(f)
        wr.writerow(["kernel", "counter", "value_last_dispatch"])
        for k in sorted(c):
            wr.writerow([kernel, k, f"{c[k]:.0f}"])
    g = lambda k: c.get(k, float("nan"))
    wc = g("SQ_WAVE_CYCLES")
    s = {
        "kernel": kernel,
        "wave_cycles": wc,
        "share_active_inst_any": g("SQ_ACTIVE_INST_ANY") / wc,
        "share_wait_inst_any": g("SQ_WAIT_INST_ANY") / wc,
        "share_wait_any": g("SQ_WAIT_ANY") / wc,
        "active_inst_valu_per_wave_cycle": g("SQ_ACTIVE_INST_VALU") / wc,
        "active_inst_sca_per_wave_cycle": g("SQ_ACTIVE_INST_SCA") / wc,
        "active_inst_lds_per_wave_cycle": g("SQ_ACTIVE_INST_LDS") / wc,
        "active_inst_misc_per_wave_cycle": g("SQ_ACTIVE_INST_MISC") / wc,
        "wait_inst_lds_per_wave_cycle": g("SQ_WAIT_INST_LDS") / wc,
        "inst_cycles_salu": g("SQ_INST_CYCLES_SALU"),
        "busy_cycles": g("SQ_BUSY_CYCLES"),
        "insts_valu": g("SQ_INSTS_VALU"), "insts_salu": g("SQ_INSTS_SALU"), "insts_lds": g("SQ_INSTS_LDS"),
        "insts_branch": g("SQ_INSTS_BRANCH"), "insts_smem": g("SQ_INSTS_SMEM"), "ifetch": g("SQ_IFETCH"),
        "ifetch_level_per_ifetch": g("SQ_IFETCH_LEVEL") / g("SQ_IFETCH") if g("SQ_IFETCH") else None,
        "lds_bank_conflict_share_of_idx_active": g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE") if g("SQ_LDS_IDX_ACTIVE") else None,
        "inst_level_lds_per_lds_inst": g("SQ_INST_LEVEL_LDS") / g("SQ_INSTS_LDS") if g("SQ_INSTS_LDS") else None,
        # pass D (both counters from the same pass): lanes active per vector instruction / 64 = rocprof's VALUUtilization
        "valu_lane_utilisation": (passes["D"]["SQ_THREAD_CYCLES_VALU"] / (passes["D"]["SQ_ACTIVE_INST_VALU"] * 64.0)
                                  if "D" in passes and passes["D"].get("SQ_ACTIVE_INST_VALU") else None),
    }
    summary[w] = s
json.dump(summary, open(os.path.join(P, f"{tag}_pmc_wait_summary.json"), "w"), indent=1)
print(json.dumps(summary, indent=1))
