set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
LIBS="default k1ntl k5nts k5ntl allnt" REPS=2 bash tools/run_ab.sh 2>&1 | grep -v "^+" | tee gpurun_out/r05_nt_ab.log
