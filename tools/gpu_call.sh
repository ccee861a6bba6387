set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python tools/fill_in_batch_probe.py 2>&1 | tail -1 | tee gpurun_out/r05_fill_in_batch_probe.json
