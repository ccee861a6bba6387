# scratch: what the next gpurun call runs (edited per call)
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python tools/front_probe.py 2>&1 | tail -1 | tee gpurun_out/front_probe.json
