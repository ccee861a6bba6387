#!/bin/bash
# Run on the MI355X box (through gpurun):  bash tools/profile_round.sh r02a
# Produces, under gpurun_out/<rev>/: the rocprofv3 --kernel-trace --stats summary of the default bench (no extras), two separate
# --pmc passes (FETCH_SIZE, WRITE_SIZE; never combined with other trace domains), the derived pmc_raw_per_launch.json /
# pmc_summary.json, and the bench JSON line of the same revision.  Copy the directory into profiles/<rev>/ afterwards.
set -u
REV=${1:-r02a}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$REV
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --steps 10 --warmup 3 --no-extras"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -o kt -- $BENCH > "$OUT/bench_under_rocprof.json" 2> "$OUT/kt.err"
cp "$(find "$OUT/kt" -name '*kernel_stats.csv' | head -1)" "$OUT/kernel_stats.csv" 2>/dev/null
PMC="python $REPO/bench.py --steps 3 --warmup 1 --no-extras"
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/fetch" -o f -- $PMC > /dev/null 2> "$OUT/fetch.err"
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/write" -o w -- $PMC > /dev/null 2> "$OUT/write.err"
F=$(find "$OUT/fetch" -name '*counter_collection.csv' | head -1)
W=$(find "$OUT/write" -name '*counter_collection.csv' | head -1)
mkdir -p "$OUT/profiles/$REV"
python "$REPO/tools/pmc_summary.py" "$F" "$W" "$OUT/profiles/$REV" 1024 > "$OUT/pmc_summary.log" 2>&1
cp "$OUT/profiles/$REV/pmc_raw_per_launch.json" "$OUT/profiles/pmc_summary.json" "$OUT/" 2>/dev/null
# the bench line proper (with the fresh pmc_summary in place so that roofline.traffic is filled in)
cp "$OUT/pmc_summary.json" "$REPO/profiles/pmc_summary.json" 2>/dev/null
cd "$REPO" && timeout 900 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
rm -rf "$OUT/kt" "$OUT/fetch" "$OUT/write" "$OUT/profiles"
ls -la "$OUT"
tail -c 1500 "$OUT/bench_default.json"
