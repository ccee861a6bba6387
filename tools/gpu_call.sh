# scratch: what the next gpurun call runs (edited per call)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for v in base nodiv nolds nostore noload noperm none4; do
  if [ $v = base ]; then unset GROUNDGRID_HIP_LIB; else export GROUNDGRID_HIP_LIB=$GRAFT_REPO_ROOT/groundgrid_amd/variants/lib_$v.so; fi
  echo "== $v"
  timeout 300 python tools/pair_timing.py 2>&1 | grep "wg 1" | grep "wave  [0567]"
done 2>&1 | tee gpurun_out/pair_variants.log
