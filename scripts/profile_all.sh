#!/bin/bash
# The round's profile set, run ON the GPU box (scripts/gpu.sh 'bash scripts/profile_all.sh <dir>').
# Results land in gpurun_out/<dir>; scripts/save_profiles.py copies the summaries into profiles/.
# Counters are collected in their own passes (no trace domains next to --pmc).
set -e
D=${1:-r02}
WHAT=${2:-all}
R=$PWD
O=$R/gpurun_out/$D
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --no-extra --steps 5 --warmup 1"
if [ $WHAT = all ]; then
  rocprofv3 --kernel-trace --stats -d $O/trace --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-extra --steps 20 --warmup 5 > $O/trace.log 2>&1
  for w in grid32 grid16d8; do
    rocprofv3 --kernel-trace --stats -d $O/trace_$w --output-format csv -- $B --workload $w > $O/trace_$w.log 2>&1
  done
  rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write --output-format csv -- $B > $O/pmc_write.log 2>&1
  rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch --output-format csv -- $B > $O/pmc_fetch.log 2>&1
  for w in grid32 grid16d8; do
    rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write_$w --output-format csv -- $B --workload $w > $O/pmc_write_$w.log 2>&1
    rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch_$w --output-format csv -- $B --workload $w > $O/pmc_fetch_$w.log 2>&1
  done
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY -d $O/sq1 --output-format csv -- $B > $O/sq1.log 2>&1
  rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU -d $O/sq2 --output-format csv -- $B > $O/sq2.log 2>&1
fi
for w in grid32 grid16d8; do
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY -d $O/sq1_$w --output-format csv -- $B --workload $w > $O/sq1_$w.log 2>&1
  rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU -d $O/sq2_$w --output-format csv -- $B --workload $w > $O/sq2_$w.log 2>&1
done
cd $R
if [ $WHAT = all ]; then
  # counters first into profiles/pmc_traffic.json (with the kernel-source digest), so that the bench lines
  # below can report roofline.traffic / roofline.compute from the same build
  python3 scripts/save_profiles.py $O r02 > $O/save_profiles.log 2>&1 || true
  for w in builtin grid32 grid16d8 grid32-noshadow builtin8k twomirrors; do
    python3 bench.py --workload $w > $O/bench_$w.json 2> $O/bench_$w.log
  done
fi
echo done
