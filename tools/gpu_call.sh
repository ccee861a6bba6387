cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
GEOM=big timeout 900 python tools/pair_ab.py 1,2,4,8,16 2>&1 | grep -v amdgpu | tee gpurun_out/pair_ab.log
