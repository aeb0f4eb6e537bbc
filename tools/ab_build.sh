#!/bin/bash
# Build the library of a given git revision (default HEAD) into tools/probe/lib_base.so for A/B timing inside one gpurun
# call:  EGS_LIB_PATH=$PWD/tools/probe/lib_base.so python tools/check_fused.py
set -e
REV=${1:-HEAD}
R=$(cd "$(dirname "$0")/.." && pwd)
P=task-level-insights-from-eigenvalues-across-sequence-models_amd
rm -rf /tmp/abpkg && mkdir -p /tmp/abpkg/$P /tmp/abpkg/include
git -C $R archive $REV $P/csrc $P/Makefile include | tar -x -C /tmp/abpkg
rm -rf /tmp/abpkg/$P/csrc/build
make -C /tmp/abpkg/$P > /tmp/abpkg/build.log 2>&1 || { tail -20 /tmp/abpkg/build.log; exit 1; }
mkdir -p $R/tools/probe && cp /tmp/abpkg/$P/libeigspec_hip.so $R/tools/probe/lib_base.so
echo "built $REV -> tools/probe/lib_base.so"
