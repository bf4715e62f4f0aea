for hv in 0 2 3 5 -1; do echo "== heavy=$hv"; python3 scripts/strips_gpu.py heavy=$hv 2>&1 | grep -E "grid32 |grid16 "; done
