mkdir -p gpurun_out/s2k
python3 -m pytest tests -m gpu -x -q > gpurun_out/s2k/tests.log 2>&1; echo tests rc=$?; tail -5 gpurun_out/s2k/tests.log
python3 scripts/ab_gpu.py reps=3 base main > gpurun_out/s2k/ab.log 2>&1; cat gpurun_out/s2k/ab.log
