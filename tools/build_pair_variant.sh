#!/bin/bash
# Side-by-side build of the library with extra flags on k4p_sweep_pair.hip only (timing experiments of the pair sweep on the GPU box):
#   tools/build_pair_variant.sh <name> "<extra flags>"  ->  groundgrid_amd/variants/lib_<name>.so   (GROUNDGRID_HIP_LIB=<path>)
# The other objects come from the main in-tree build (make -C groundgrid_amd/csrc first).
set -e
name=$1; shift
extra="$*"
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p /tmp/gg_pvar_$name "$root/groundgrid_amd/variants"
cd "$root/groundgrid_amd/csrc"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero -fno-slp-vectorize -I../../include -I. -Wno-unused-result -Wno-unused-value $extra"
/opt/rocm/bin/hipcc $FLAGS -c k4p_sweep_pair.hip -o /tmp/gg_pvar_$name/k4p_sweep_pair.o
objs=$(ls *.o | grep -v k4p_sweep_pair.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$root/groundgrid_amd/variants/lib_$name.so" $objs /tmp/gg_pvar_$name/k4p_sweep_pair.o
echo "built groundgrid_amd/variants/lib_$name.so with: $extra"
