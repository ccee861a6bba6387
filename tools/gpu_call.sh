# scratch: what the next gpurun call runs (edited per call)
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for sp in 1500 0 4000 1500 0 4000; do
GG_HOST_HELPER_SPINS=$sp timeout 200 python tools/direct_probe.py 2>&1 | tail -1 | tee -a gpurun_out/helper_spin.log
done
nproc
