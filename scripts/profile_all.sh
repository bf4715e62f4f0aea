#!/bin/bash
# The round's profile set, run ON the GPU box (scripts/gpu.sh 'bash scripts/profile_all.sh <dir> <tag>').
# Results land in gpurun_out/<dir>; scripts/save_profiles.py copies the summaries into profiles/<tag>_*.
# Counters are collected in their own passes (no trace domains next to --pmc).  Five workloads: the three BASELINE
# configs a single GPU runs (built-in, grid-32, grid-16 depth 8), the 8192^2 frame, and the reference's SCENE 2
# (two mirrors: the large-scene kernel).
set -e
D=${1:-r04}
TAG=${2:-r04}
R=$PWD
O=$R/gpurun_out/$D
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
WL="builtin grid32 grid16d8 builtin8k twomirrors"
for w in $WL; do
  B="python3 $R/bench.py --no-cpu-baseline --no-extra --steps 5 --warmup 2 --workload $w"
  st=5; [ $w = builtin ] && st=20
  rocprofv3 --kernel-trace --stats -d $O/trace_$w --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-extra --steps $st --warmup 5 --workload $w > $O/trace_$w.log 2>&1
  rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write_$w --output-format csv -- $B > $O/pmc_write_$w.log 2>&1
  rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch_$w --output-format csv -- $B > $O/pmc_fetch_$w.log 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY -d $O/sq1_$w --output-format csv -- $B > $O/sq1_$w.log 2>&1
  rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_BRANCH -d $O/sq2_$w --output-format csv -- $B > $O/sq2_$w.log 2>&1
  echo "$w counters done"
done
cd $R
# stall attribution (three more passes per workload)
TAG=$TAG bash scripts/profile_wait.sh $D $WL > $O/wait.log 2>&1
# counters first into profiles/pmc_traffic.json (with the kernel-source digest), so that the bench lines
# below can report roofline.traffic / roofline.compute from the same build
python3 scripts/save_profiles.py $O $TAG > $O/save_profiles.log 2>&1 || true
for w in builtin grid32 grid16d8 grid32-noshadow builtin8k twomirrors shipped shipped512; do
  python3 bench.py --workload $w > $O/bench_$w.json 2> $O/bench_$w.log || true
done
python3 scripts/save_profiles.py $O $TAG > $O/save_profiles.log 2>&1 || true
echo done
