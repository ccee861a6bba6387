set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_cpp_adapter.py tests/test_gpu_stages_wire.py -x -q 2>&1 | tail -5
timeout 900 python bench.py --steps 12 --warmup 3 > gpurun_out/r05_bench2.json 2> gpurun_out/r05_bench2.err; python - <<'PY'
import json
j=json.loads(open('gpurun_out/r05_bench2.json').read().strip().split('\n')[-1])
print(j['value'], j['ms_per_step'], {k:v for k,v in j.get('concurrent_halves',{}).items() if k!='note'}, j['summary']['parity_checked_in_run'])
PY
tail -3 gpurun_out/r05_bench2.err
