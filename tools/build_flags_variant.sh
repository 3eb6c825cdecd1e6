#!/bin/bash
# developer: build a variant of the library with extra COMPILER flags for every TU that holds hot kernels
#   tools/build_flags_variant.sh <name> "<flags>"
set -e
name=$1; extra=$2
root=$(cd "$(dirname "$0")/.." && pwd)
src=$root/point_cloud_registration_amd/csrc
out=$root/build/exp/$name; mkdir -p "$out"
flags="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -I$root/include -I$src"
for f in api kernels kernels_dev index_build comm voxel_build knn_normals host_hash; do
    if [ "$f" = "kernels" ]; then /opt/rocm/bin/hipcc $flags $extra -c "$src/$f.hip" -o "$out/$f.o"; else cp "$src/$f.o" "$out/$f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$root/build/exp/libpcr_$name.so" "$out"/*.o -ldl -lpthread -Wl,-rpath,/opt/rocm/lib
echo "$root/build/exp/libpcr_$name.so"
