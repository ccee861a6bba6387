mkdir -p gpurun_out/r03t; O=gpurun_out/r03t
V=$PWD/groundgrid_amd/variants
for lib in default tight2 tight4; do
  if [ $lib = default ]; then unset GROUNDGRID_HIP_LIB; else export GROUNDGRID_HIP_LIB=$V/lib_$lib.so; fi
  BATCHES_SMALL=1,64 BATCHES_BIG=1 timeout 300 python tools/latency_probe.py > $O/lat_$lib.json 2>>$O/err.log; cat $O/lat_$lib.json
done
