#!/bin/bash
# build lib/variants/libtcrt_<name>.so from the kernel sources of a git revision (A/B baseline for scripts/ab_gpu.py)
# usage: scripts/mkvariant_from.sh <rev> <name>
set -e
REV=$1; NAME=$2
R=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
mkdir -p $T/tilecoderaytracer_amd $T/include
git -C $R archive $REV tilecoderaytracer_amd/csrc include | tar -x -C $T
make -C $T/tilecoderaytracer_amd/csrc ../lib/libtcrt.so >/dev/null 2>&1
mkdir -p $R/tilecoderaytracer_amd/lib/variants
cp $T/tilecoderaytracer_amd/lib/libtcrt.so $R/tilecoderaytracer_amd/lib/variants/libtcrt_$NAME.so
rm -rf $T
ls -la $R/tilecoderaytracer_amd/lib/variants/libtcrt_$NAME.so
