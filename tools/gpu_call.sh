set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
V=$PWD/groundgrid_amd/variants
for r in 1 2; do
  for lib in default prio3; do
    if [ $lib = default ]; then unset GROUNDGRID_HIP_LIB; else export GROUNDGRID_HIP_LIB=$V/lib_$lib.so; fi
    SKIP_BIG=1 BATCHES_SMALL=1,8 timeout 200 python tools/latency_probe.py 2>/dev/null | tail -1 | tee -a gpurun_out/r05_prio_ab.log
  done
done
