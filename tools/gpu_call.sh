set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_stages_wire.py -x -q -k "empty_and_all_outside" 2>&1 | tail -3
( time timeout 900 python bench.py > gpurun_out/r05_bench3.json 2> gpurun_out/r05_bench3.err ) 2>&1 | tail -4
python - <<'PY'
import json
j=json.loads(open('gpurun_out/r05_bench3.json').read().strip().split('\n')[-1])
print(j['value'], j['ms_per_step'], j['roofline'])
print({k:(v.get('pmc_MB_per_launch'), v.get('real_frac_hbm')) for k,v in j['kernels'].items()})
PY
tail -3 gpurun_out/r05_bench3.err
