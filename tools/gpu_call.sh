# scratch: what the next gpurun call runs (edited per call)
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/profile_round.sh r05c > gpurun_out/r05c_profile_round.log 2>&1
tail -5 gpurun_out/r05c_profile_round.log | cut -c1-600
ls gpurun_out/r05c
