#!/bin/bash
# Run on the MI355X box (through gpurun):  bash tools/profile_round.sh r02a
# (SKIP_CONFIG4=1 leaves the configs[3] passes out.)
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
cp "$OUT/pmc_summary.json" "$REPO/profiles/pmc_summary.json" 2>/dev/null
# configs[3] (2.1 M points, 1000 x 1000, 128 clouds per launch): kernel stats and the same two counter passes, merged into pmc_summary.json
C4="python $REPO/bench.py --only-config4 --cpu-seconds 0"
if [ -z "${SKIP_CONFIG4:-}" ]; then
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt4" -o kt -- $C4 > "$OUT/config4_under_rocprof.json" 2> "$OUT/kt4.err"
cp "$(find "$OUT/kt4" -name '*kernel_stats.csv' | head -1)" "$OUT/config4_kernel_stats.csv" 2>/dev/null
# ... and of the 128-cloud launches alone (the run also launches 8 clouds and one cloud at a time: rocprofv3's averages mix them)
python "$REPO/tools/kernel_stats_of_batch.py" "$(find "$OUT/kt4" -name '*kernel_trace.csv' | head -1)" 128 > "$OUT/config4_kernel_stats_128_clouds.csv" 2> "$OUT/kt4_128.err"
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/fetch4" -o f -- $C4 > /dev/null 2> "$OUT/fetch4.err"
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/write4" -o w -- $C4 > /dev/null 2> "$OUT/write4.err"
F4=$(find "$OUT/fetch4" -name '*counter_collection.csv' | head -1)
W4=$(find "$OUT/write4" -name '*counter_collection.csv' | head -1)
mkdir -p "$REPO/profiles/$REV"
python "$REPO/tools/pmc_summary.py" "$F4" "$W4" "$REPO/profiles/$REV" 128 config4_kernels > "$OUT/pmc_summary_config4.log" 2>&1
cp "$REPO/profiles/$REV/pmc_raw_per_launch_config4_kernels.json" "$OUT/" 2>/dev/null
cp "$REPO/profiles/pmc_summary.json" "$OUT/pmc_summary.json" 2>/dev/null
rm -rf "$OUT/kt4" "$OUT/fetch4" "$OUT/write4"
fi
# one cloud per launch (the latency launches: k_sweep_records + k_sweep_pair + k_sweep_finish), device-resident input
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt1" -o kt -- python $REPO/bench.py --batch 1 --steps 200 --warmup 20 --no-extras --no-live-pmc > "$OUT/bench_single_cloud_under_rocprof.json" 2> "$OUT/kt1.err"
cp "$(find "$OUT/kt1" -name '*kernel_stats.csv' | head -1)" "$OUT/kernel_stats_single_cloud.csv" 2>/dev/null
rm -rf "$OUT/kt1"
# the N > 1 code path with one rank (RCCL process group, all-gather per step), with and without the collective in the headline steps
cd "$REPO"
timeout 600 python bench.py --force-dist --no-extras --no-live-pmc --steps 10 --warmup 3 > "$OUT/bench_force_dist.json" 2> "$OUT/bench_force_dist.err"
timeout 600 python bench.py --force-dist --gather config3-only --no-extras --no-live-pmc --steps 10 --warmup 3 > "$OUT/bench_force_dist_no_gather.json" 2>> "$OUT/bench_force_dist.err"
# the bench line proper (with the fresh pmc_summary in place so that roofline.traffic is filled in)
cd "$REPO" && timeout 900 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
rm -rf "$OUT/kt" "$OUT/fetch" "$OUT/write" "$OUT/profiles"
ls -la "$OUT"
tail -c 1500 "$OUT/bench_default.json"
