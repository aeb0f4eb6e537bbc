#!/bin/bash
# Build the working tree's library with extra -D flags into tools/probe/lib_<name>.so (A/B timing inside one gpurun call):
#   tools/build_variant.sh name -DEGS_FOO -DEGS_BAR=2
set -e
NAME=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd)
P=task-level-insights-from-eigenvalues-across-sequence-models_amd
rm -rf /tmp/var_$NAME && mkdir -p /tmp/var_$NAME/$P /tmp/var_$NAME/include
cp -r $R/$P/csrc $R/$P/Makefile /tmp/var_$NAME/$P/ && cp $R/include/*.h /tmp/var_$NAME/include/
rm -rf /tmp/var_$NAME/$P/csrc/build
make -j8 -C /tmp/var_$NAME/$P CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -I/tmp/var_$NAME/include -I/tmp/var_$NAME/$P/csrc -ffp-contract=on -Wall -Wno-unused-function $*" > /tmp/var_$NAME/build.log 2>&1 || { tail -20 /tmp/var_$NAME/build.log; exit 1; }
cp /tmp/var_$NAME/$P/libeigspec_hip.so $R/tools/probe/lib_$NAME.so
echo "built $NAME ($*) -> tools/probe/lib_$NAME.so"
