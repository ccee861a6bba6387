set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
SKIP_BIG=1 BATCHES_SMALL=1,8 timeout 200 python tools/latency_probe.py 2>/dev/null | tail -1 | tee gpurun_out/r05_pf_latency.log
BATCHES_BIG=1 SKIP_SMALL=1 timeout 200 python tools/latency_probe.py 2>/dev/null | tail -1 | tee -a gpurun_out/r05_pf_latency.log
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "sweep or geometr or hdl64 or config4 or golden" 2>&1 | tail -3
