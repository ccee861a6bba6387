set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
LIBS="default scnt k1nts k4nt" REPS=3 bash tools/run_ab.sh 2>&1 | grep -v "^+" | tee gpurun_out/r05_nt_ab3.log
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "hdl64_full or golden or minimal or dense" 2>&1 | tail -3
