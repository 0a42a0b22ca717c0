#!/bin/bash
# build_variant.sh NAME [-DFLAG ...]: libsquidstitch_NAME.so with extra compile flags (experiments; use with SQ_LIB_PATH)
set -e
cd "$(dirname "$0")/../image-stitcher_amd/csrc"
name=$1; shift
mkdir -p build_$name
FLAGS="-O3 -std=c++17 -fPIC -Wall -Wno-unused-function --offload-arch=gfx950 -ffp-contract=off $*"
for f in plan_expand.hip arena.hip fuse.hip register.hip pyramid.hip synth.hip basic.hip blosc.hip; do /opt/rocm/bin/hipcc $FLAGS -c $f -o build_$name/$f.o & done
/opt/rocm/bin/hipcc $FLAGS -x hip -c plan.cpp -o build_$name/plan.cpp.o &
/opt/rocm/bin/hipcc $FLAGS -x hip -c chunkio.cpp -o build_$name/chunkio.cpp.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libsquidstitch_$name.so build_$name/*.o
echo built libsquidstitch_$name.so
