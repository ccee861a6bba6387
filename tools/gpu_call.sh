# scratch: what the next gpurun call runs (edited per call)
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
for lib in base r3 new base r3 new; do
  L=$GRAFT_REPO_ROOT/groundgrid_amd/variants/lib_$lib.so
  [ $lib = new ] && L=$GRAFT_REPO_ROOT/groundgrid_amd/libgroundgrid_hip.so
  GROUNDGRID_HIP_LIB=$L MODES=cold timeout 300 python tools/ab_kernels.py 1024 6 $lib 2>&1 | tail -1 | tee -a gpurun_out/ab_peel12.log
  GROUNDGRID_HIP_LIB=$L timeout 300 python tools/host_call_probe.py 2>&1 | tail -1 | tee -a gpurun_out/ab_peel12.log
done
