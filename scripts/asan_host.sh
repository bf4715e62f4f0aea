#!/bin/bash
# The HOST side of libtcrt.so (table packer, SHADOW VOXELS, strip solver) under AddressSanitizer, on a machine without a GPU:
# rt_scene_create packs everything before it asks for a device, so the CPU tests below run the whole packer.
# (The device code is the product build's rt_kernel.o; GPU sanitizers are not available on this pool.)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
O=${1:-/tmp/tcrt_asan}
mkdir -p $O
cd $R/tilecoderaytracer_amd/csrc
make rt_kernel.o >/dev/null
HF="--offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero -fno-fast-math -fno-slp-vectorize"
for f in rt_capi rt_multi; do /opt/rocm/bin/hipcc $HF -Xarch_host -fsanitize=address -Xarch_host -fno-omit-frame-pointer -c $f.hip -o $O/$f.o; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=address -o $O/libtcrt_asan.so rt_kernel.o $O/rt_capi.o $O/rt_multi.o -ldl
cd $R
ASAN_RT=$(/opt/rocm/lib/llvm/bin/clang --print-file-name=libclang_rt.asan-x86_64.so)
TCRT_LIBRARY=$O/libtcrt_asan.so LD_PRELOAD=$ASAN_RT ASAN_OPTIONS=detect_leaks=0 python3 -m pytest tests/test_capi_library.py tests/test_host_model.py -q -m "not gpu" \
    --deselect tests/test_capi_library.py::test_library_path_can_be_overridden
