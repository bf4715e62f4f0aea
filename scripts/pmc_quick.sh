#!/bin/bash
# Instruction counters of the render kernel for a few workloads, run ON the GPU box (development aid):
#   scripts/gpu.sh 'bash scripts/pmc_quick.sh <dir> "<bench options>" workload...'
# one --pmc pass per workload (SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES); prints the last dispatch
set -e
D=${1:-pmcq}; OPTS=$2; shift 2 || true
W=${@:-grid32 grid16d8}
R=$PWD; O=$R/gpurun_out/$D; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for w in $W; do
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $O/q_$w --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-extra --steps 3 --warmup 1 --workload $w $OPTS > $O/q_$w.log 2>&1
  python3 - $O $w <<'PY'
import csv, glob, sys
o, w = sys.argv[1], sys.argv[2]
f = sorted(glob.glob(f"{o}/q_{w}/*/*_counter_collection.csv"))[-1]
r = [x for x in csv.DictReader(open(f)) if "rt_render_kernel" in x["Kernel_Name"]]
last = max(int(x["Dispatch_Id"]) for x in r)
c = {}
for x in r:
    if int(x["Dispatch_Id"]) == last: c[x["Counter_Name"]] = c.get(x["Counter_Name"], 0.0) + float(x["Counter_Value"])
print(w, r[-1]["Kernel_Name"][:34], " ".join(f"{k[3:]} {v/1e9:.4f}G" for k, v in sorted(c.items())), flush=True)
PY
done
