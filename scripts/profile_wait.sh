#!/bin/bash
# Stall attribution for the render kernels, run ON the GPU box:
#   scripts/gpu.sh 'bash scripts/profile_wait.sh <dir> [workloads...]'
# Four --pmc passes per workload (counters only, no trace domains), results in
# gpurun_out/<dir>/wait_<workload>_<pass>; scripts/save_wait.py writes the per-kernel sums to profiles/.
set -e
D=${1:-r03wait}
shift || true
W=${@:-builtin grid32 grid16d8}
R=$PWD
O=$R/gpurun_out/$D
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
PA="SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES"
PB="SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_IFETCH SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC"
PC="SQ_LDS_BANK_CONFLICT SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_IFETCH_LEVEL SQ_INSTS_VALU SQ_INSTS_SALU"
PD="SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VALU"      # lanes active per vector instruction: what divergence costs
for w in $W; do
  B="python3 $R/bench.py --no-cpu-baseline --no-extra --steps 3 --warmup 1 --workload $w"
  rocprofv3 --pmc $PA -d $O/wait_${w}_A --output-format csv -- $B > $O/wait_${w}_A.log 2>&1
  rocprofv3 --pmc $PB -d $O/wait_${w}_B --output-format csv -- $B > $O/wait_${w}_B.log 2>&1
  rocprofv3 --pmc $PC -d $O/wait_${w}_C --output-format csv -- $B > $O/wait_${w}_C.log 2>&1
  rocprofv3 --pmc $PD -d $O/wait_${w}_D --output-format csv -- $B > $O/wait_${w}_D.log 2>&1
  echo "$w done"
done
cd $R
python3 scripts/save_wait.py $O ${TAG:-r03} $W > $O/save_wait.log 2>&1 || true
cat $O/save_wait.log
