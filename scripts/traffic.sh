#!/bin/bash
# HBM traffic of the render kernel, run ON the GPU box: scripts/gpu.sh 'bash scripts/traffic.sh <dir> [workloads...]'
# (two counter passes per workload, as in profile_all.sh; prints KiB written / fetched by the last dispatch)
set -e
D=${1:-traffic}
shift || true
W=${@:-grid32 grid16d8}
R=$PWD
O=$R/gpurun_out/$D
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for w in $W; do
  B="python3 $R/bench.py --no-cpu-baseline --no-extra --steps 5 --warmup 2 --workload $w $OPTS"
  rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write_$w --output-format csv -- $B > $O/pmc_write_$w.log 2>&1
  rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch_$w --output-format csv -- $B > $O/pmc_fetch_$w.log 2>&1
  python3 - $O $w <<'PY'
import csv, glob, sys
o, w = sys.argv[1], sys.argv[2]
for c in ("write", "fetch"):
    f = sorted(glob.glob(f"{o}/pmc_{c}_{w}/*/*_counter_collection.csv"))[-1]
    r = list(csv.DictReader(open(f)))
    r = [x for x in r if "rt_render_kernel" in x["Kernel_Name"]]
    last = max(int(x["Dispatch_Id"]) for x in r)
    print(w, c, "KiB", sum(float(x["Counter_Value"]) for x in r if int(x["Dispatch_Id"]) == last), r[-1]["Kernel_Name"][:40], flush=True)
PY
done
