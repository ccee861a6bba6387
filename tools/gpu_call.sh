# scratch: what the next gpurun call runs (edited per call)
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/t.log 2>&1; grep -E "passed|failed" gpurun_out/t.log | tail -2
for lib in rng1 new rng1 new; do
  L=$GRAFT_REPO_ROOT/groundgrid_amd/variants/lib_$lib.so
  [ $lib = new ] && L=$GRAFT_REPO_ROOT/groundgrid_amd/libgroundgrid_hip.so
  GROUNDGRID_HIP_LIB=$L MODES=cold,warm timeout 300 python tools/ab_kernels.py 1024 8 $lib 2>&1 | tail -1 | tee -a gpurun_out/ab_ranges.log
done
