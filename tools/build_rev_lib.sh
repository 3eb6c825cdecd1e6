#!/bin/bash
# Developer helper: build libpcr_hip.so of a COMMITTED revision into build/exp/libpcr_<name>.so (A/B baseline for the
# working tree's library; select with PCR_LIB=<path>).   tools/build_rev_lib.sh <git-rev> <name>
set -e
rev=$1; name=$2
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d /tmp/pcr_rev_XXXX)
git -C "$root" archive "$rev" point_cloud_registration_amd/csrc include | tar -x -C "$tmp"
make -s -C "$tmp/point_cloud_registration_amd/csrc" -j8 >/dev/null
mkdir -p "$root/build/exp"
cp "$tmp/point_cloud_registration_amd/libpcr_hip.so" "$root/build/exp/libpcr_$name.so"
rm -rf "$tmp"
echo "$root/build/exp/libpcr_$name.so"
