set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
LIBS="default k4ntl scntl k3ntl" REPS=3 bash tools/run_ab.sh 2>&1 | grep -v "^+" | tee gpurun_out/r05_nt_ab4.log
