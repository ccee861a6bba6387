# scratch: what the next gpurun call runs (edited per call)
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for rep in 1 2; do timeout 300 python tools/fill_overlap_probe.py 1024 2>&1 | tail -1; done | tee gpurun_out/t25_fill.log
