#!/bin/bash
# Developer helper: build an experimental variant of libpcr_hip.so with extra -D flags into
# build/exp/libpcr_<name>.so (git-ignored, travels to the GPU box).  Select it with PCR_LIB=<path>.
#   tools/build_variant.sh <name> "<-DFLAG=1 ...>"
set -e
name=$1; defs=$2
root=$(cd "$(dirname "$0")/.." && pwd)
src=$root/point_cloud_registration_amd/csrc
out=$root/build/exp/$name
mkdir -p "$out"
flags="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -I$root/include -I$src -DPCR_DEV=1 $defs"
pids=()
for f in api kernels kernels_dev index_build comm voxel_build knn_normals host_hash; do
    # only kernels.hip sees the experiment macros; the other objects are reused from the main build when present
    # (variant libraries are developer builds: api / kernels / kernels_dev are compiled with -DPCR_DEV)
    if [ "$f" != "kernels" ] && [ -f "$src/$f.dev.o" ] && [ -z "$ALL" ]; then cp "$src/$f.dev.o" "$out/$f.o"; continue; fi
    if [ "$f" != "kernels" ] && [ "$f" != "api" ] && [ "$f" != "kernels_dev" ] && [ -f "$src/$f.o" ] && [ -z "$ALL" ]; then cp "$src/$f.o" "$out/$f.o"; continue; fi
    /opt/rocm/bin/hipcc $flags -c "$src/$f.hip" -o "$out/$f.o" &
    pids+=($!)
done
for p in "${pids[@]}"; do wait "$p"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$root/build/exp/libpcr_$name.so" "$out"/*.o -ldl -lpthread -Wl,-rpath,/opt/rocm/lib
echo "$root/build/exp/libpcr_$name.so"
