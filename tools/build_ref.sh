#!/bin/bash
# Build the library of another revision side by side (A/B on the GPU box):
#   tools/build_ref.sh <git rev> <name>   ->  groundgrid_amd/variants/lib_<name>.so   (load with GROUNDGRID_HIP_LIB=<path>)
set -e
rev=$1; name=$2
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=/tmp/gg_ref_$name
rm -rf "$tmp"; mkdir -p "$tmp" "$root/groundgrid_amd/variants"
git -C "$root" archive "$rev" groundgrid_amd/csrc include | tar -x -C "$tmp"
make -C "$tmp/groundgrid_amd/csrc" -j8 > "$tmp/build.log" 2>&1 || { tail -5 "$tmp/build.log"; exit 1; }
cp "$tmp/groundgrid_amd/libgroundgrid_hip.so" "$root/groundgrid_amd/variants/lib_$name.so"
echo "built groundgrid_amd/variants/lib_$name.so from $rev"
