set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python tools/fuzz_walk.py 0 60 2>&1 | tail -8 | tee gpurun_out/r05_fuzz_walk.log
timeout 600 python tools/fuzz_more.py 6 60 2>&1 | tail -4 | tee gpurun_out/r05_fuzz_more.log
timeout 600 python tools/fuzz_knobs.py 0 30 2>&1 | tail -4 | tee gpurun_out/r05_fuzz_knobs.log
