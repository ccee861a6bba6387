# scratch: what the next gpurun call runs (edited per call)
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 120 tools/ubench/ta_lines 2>&1 | tee gpurun_out/t35_ta_lines.log
