#!/bin/bash
# Side-by-side build of the library with extra compiler flags (A/B experiments on the GPU box):
#   tools/build_variant.sh <name> "<extra flags>"   ->  groundgrid_amd/variants/lib_<name>.so   (load with GROUNDGRID_HIP_LIB=<path>)
set -e
name=$1; shift
extra="$*"
root=$(cd "$(dirname "$0")/.." && pwd)
obj=/tmp/gg_var_$name
mkdir -p "$obj" "$root/groundgrid_amd/variants"
cd "$root/groundgrid_amd/csrc"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero -fno-slp-vectorize -I../../include -I. -Wno-unused-result -Wno-unused-value $extra"
pids=()
for f in gg_context sweep_emul k0_scroll k1_classify k_sort k2_reduce k3_patch k4_sweep k4p_sweep_pair k4b_sweep_pair_batch k5_label k6_wire k7_stage; do
  /opt/rocm/bin/hipcc $FLAGS -c $f.hip -o "$obj/$f.o" 2> "$obj/$f.log" & pids+=($!)
done
fail=0; for p in "${pids[@]}"; do wait $p || fail=1; done
if [ $fail = 1 ]; then grep -h "error" "$obj"/*.log | head; exit 1; fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$root/groundgrid_amd/variants/lib_$name.so" "$obj"/*.o
echo "built groundgrid_amd/variants/lib_$name.so with: $extra"
