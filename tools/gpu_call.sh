# scratch: what the next gpurun call runs (edited per call)
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for w in 0 1 2 3 0 1 2 3; do
  GG_SWEEP_WAVES=$w MODES=cold timeout 300 python tools/ab_kernels.py 1024 6 waves$w 2>&1 | tail -1 | tee -a gpurun_out/ab_waves.log
done
