#!/bin/bash
# one development iteration ON the GPU box: the GPU suite, then kernel times A/B against library variants, then the counting build
# usage: scripts/gpu.sh 'bash scripts/iter.sh <tag> [variant names for ab_gpu.py, default: r2 main]'
T=${1:-it}
shift || true
V=${@:-main}
mkdir -p gpurun_out
python3 -m pytest tests -m gpu -x -q > gpurun_out/${T}_tests.log 2>&1
rc=$?
tail -3 gpurun_out/${T}_tests.log
[ $rc -ne 0 ] && { grep -E "^(FAILED|ERROR)|Error|assert" gpurun_out/${T}_tests.log | head -20; exit $rc; }
python3 scripts/ab_gpu.py reps=3 $V > gpurun_out/${T}_ab.log 2>&1 || { tail -5 gpurun_out/${T}_ab.log; exit 1; }
cat gpurun_out/${T}_ab.log
python3 scripts/stats_gpu.py builtin 2048 > gpurun_out/${T}_stats.log 2>&1
grep -E "cycles|candidates|plane_tests|sphere_tests" gpurun_out/${T}_stats.log
