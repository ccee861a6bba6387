# scratch: what the next gpurun call runs (edited per call)
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_stages_wire.py -x -q 2>&1 | tail -25 | tee gpurun_out/r05_new_tests.log
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -12 | tee gpurun_out/r05_all_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
