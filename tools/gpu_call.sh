cd $GRAFT_REPO_ROOT
F='RCCL\|HIP version\|ROCm\|Hostname\|Librccl\|amdgpu'
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stages_wire.py -m gpu -x -q -k "fresh or halves or divided or batch" 2>&1 | grep -v "$F" | tail -5
python bench.py --steps 6 --warmup 2 --no-live-pmc --drive-frames 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['config3']['clouds_per_s'], d['config3']['ms_per_step'], d['summary']['config4'])"
