# scratch: what the next gpurun call runs (edited per call)
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/final_tests_full.log 2>&1; grep -E "passed|failed|error" gpurun_out/final_tests_full.log | tail -2 | tee gpurun_out/final_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err; head -c 600 gpurun_out/final_bench.json
( timeout 200 python tools/fuzz_more.py 1500 2500; timeout 200 python tools/fuzz_knobs.py 900 1500; timeout 200 python tools/fuzz_walk.py 400 520 ) 2>&1 | grep -E "done|FAILED" | tee gpurun_out/final_fuzz.log
