# scratch: what the next gpurun call runs (edited per call)
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python tools/fuzz_knobs.py 0 120 2>&1 | grep -v amdgpu.ids | tail -12 | tee gpurun_out/t28_fuzz_knobs.log
