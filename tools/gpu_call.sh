set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for r in 1 2 3; do
  for g in 16 1 8 32; do
    GG_FILL_GROUP=$g timeout 150 python tools/ab_kernels.py 1024 8 fillgroup$g 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); c=j['cold']; print(j['tag'], c['ms_per_step'], 'classify', c['k_classify'], 'reduce', c['k_reduce'])" | tee -a gpurun_out/r05_fill_group_ab.log
  done
done
LIBS="default k5nts" REPS=4 bash tools/run_ab.sh 2>&1 | grep -v "^+" | tee gpurun_out/r05_k5_nt_ab.log
