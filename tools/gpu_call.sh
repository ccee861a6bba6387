# scratch: what the next gpurun call runs (edited per call)
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
timeout 300 python tools/ab_config4.py 128 parts_auto 2>&1 | tail -1
SCAN_PARTS=8 timeout 300 python tools/ab_config4.py 128 parts8 2>&1 | tail -1
SCAN_PARTS=16 timeout 300 python tools/ab_config4.py 128 parts16 2>&1 | tail -1
GG_K2_PER_CLOUD=128 timeout 300 python tools/ab_config4.py 128 k2_per_cloud128 2>&1 | tail -1
GG_K2_DENSE_SHARE=14 timeout 300 python tools/ab_config4.py 128 k2_dense14 2>&1 | tail -1
GG_K2_DENSE_SHARE=8 timeout 300 python tools/ab_config4.py 128 k2_dense8 2>&1 | tail -1
GG_K2_PER_CLOUD=128 GG_K2_DENSE_SHARE=14 timeout 300 python tools/ab_config4.py 128 k2_128_dense14 2>&1 | tail -1
} | tee gpurun_out/t21_c4.log
{
timeout 200 python tools/ab_kernels.py 1 40 one_cloud_auto 2>&1 | tail -1
SCAN_PARTS=2 timeout 200 python tools/ab_kernels.py 1 40 one_cloud_parts2 2>&1 | tail -1
SCAN_PARTS=4 timeout 200 python tools/ab_kernels.py 1 40 one_cloud_parts4 2>&1 | tail -1
timeout 200 python tools/ab_kernels.py 8 40 eight_auto 2>&1 | tail -1
SCAN_PARTS=2 timeout 200 python tools/ab_kernels.py 8 40 eight_parts2 2>&1 | tail -1
} | tee gpurun_out/t21_one.log
