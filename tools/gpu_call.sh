# scratch: what the next gpurun call runs (edited per call)
set -x
cd $GRAFT_REPO_ROOT
REPO=$GRAFT_REPO_ROOT; OUT=$REPO/gpurun_out/r04c_c4; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
C4="python $REPO/bench.py --only-config4 --cpu-seconds 0"
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/fetch4" -o f -- $C4 > /dev/null 2> "$OUT/fetch4.err"
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/write4" -o w -- $C4 > /dev/null 2> "$OUT/write4.err"
F4=$(find "$OUT/fetch4" -name '*counter_collection.csv' | head -1)
W4=$(find "$OUT/write4" -name '*counter_collection.csv' | head -1)
python "$REPO/tools/pmc_summary.py" "$F4" "$W4" "$REPO/profiles/r04c" 128 config4_kernels > "$OUT/pmc_summary_config4.log" 2>&1
cp "$REPO/profiles/r04c/pmc_raw_per_launch_config4_kernels.json" "$REPO/profiles/pmc_summary.json" "$OUT/"
rm -rf "$OUT/fetch4" "$OUT/write4"
cat $OUT/pmc_summary_config4.log | tail -12
cd $REPO && timeout 600 python bench.py --only-config4 > $OUT/config4_line.json 2> $OUT/config4_line.err; tail -c 600 $OUT/config4_line.json
