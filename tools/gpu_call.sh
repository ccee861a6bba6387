# scratch: what the next gpurun call runs (edited per call)
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
V=$GRAFT_REPO_ROOT/groundgrid_amd/variants
for rep in 1 2 3; do
  GROUNDGRID_HIP_LIB=$V/lib_base.so timeout 200 python tools/ab_kernels.py 1024 8 base 2>&1 | tail -1
  GROUNDGRID_HIP_LIB=$V/lib_occ5.so timeout 200 python tools/ab_kernels.py 1024 8 old_code_5_waves 2>&1 | tail -1
  timeout 200 python tools/ab_kernels.py 1024 8 k2_light_by_record 2>&1 | tail -1
done | tee gpurun_out/t18_ab.log
