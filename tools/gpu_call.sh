# scratch: what the next gpurun call runs (edited per call)
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/profile_round.sh r04a > gpurun_out/r04a_profile.log 2>&1
tail -5 gpurun_out/r04a_profile.log
bash tools/sq_pmc.sh > gpurun_out/r04a/sq_counters.txt 2>&1
tail -c 600 gpurun_out/r04a/sq_counters.txt
rm -rf gpurun_out/sq_1 gpurun_out/sq_2 gpurun_out/sq_3
cp profiles/pmc_summary.json gpurun_out/r04a/pmc_summary_final.json
ls -la gpurun_out/r04a
