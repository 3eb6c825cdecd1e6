#!/bin/bash
# developer: a variant of the library with ONE translation unit rebuilt with extra compiler flags (the others are the built .o files)
#   tools/build_tu_variant.sh <name> <tu without .hip> "<flags>"      -> build/exp/libpcr_<name>.so   (load it with PCR_LIB=...)
set -e
name=$1; tu=$2; extra=$3
root=$(cd "$(dirname "$0")/.." && pwd)
src=$root/point_cloud_registration_amd/csrc
out=$root/build/exp/$name; mkdir -p "$out"
flags="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -I$root/include -I$src"
for f in api kernels index_build comm voxel_build knn_normals host_hash group; do
    if [ "$f" = "$tu" ]; then /opt/rocm/bin/hipcc $flags $extra -c "$src/$f.hip" -o "$out/$f.o"; else cp "$src/$f.o" "$out/$f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$root/build/exp/libpcr_$name.so" "$out"/*.o -ldl -lpthread -Wl,-rpath,/opt/rocm/lib
rm -rf "$out"
echo "$root/build/exp/libpcr_$name.so"
