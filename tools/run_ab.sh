# scratch: ABAB of library variants on the headline workload; usage: LIBS="base default" REPS=3 bash tools/run_ab.sh
V=$PWD/groundgrid_amd/variants
for r in $(seq 1 ${REPS:-3}); do
  for lib in ${LIBS:-base default}; do
    if [ $lib = default ]; then unset GROUNDGRID_HIP_LIB; else export GROUNDGRID_HIP_LIB=$V/lib_$lib.so; fi
    timeout 150 python tools/ab_kernels.py 1024 8 $lib 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); c=j['cold']; print(j['tag'], c['ms_per_step'], 'reduce', c['k_reduce'], 'classify', c['k_classify'], 'patch', c['k_patch'], 'sweep', c['k_sweep'], 'label', c['k_label'], 'scatter', c['k_scatter'])"
  done
done
