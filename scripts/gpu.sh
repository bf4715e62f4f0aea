#!/bin/bash
# build, then (only if the build succeeded) run a command on the GPU box
set -e
cd /root/repo
make -C tilecoderaytracer_amd/csrc 2>&1 | grep -E "error:|Error [0-9]" -A6 && exit 1
make -C tilecoderaytracer_amd/csrc >/dev/null
exec /usr/local/graft/bin/gpurun --timeout "${GPU_TIMEOUT:-900}" -- "$@"
