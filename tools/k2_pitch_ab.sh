#!/bin/bash
# EXPERIMENT: k_reduce with layer rows padded to 16 / 32 cells (GG_K2_DEBUG=7 / 8; results are wrong, only k_reduce's time and write requests count)
R=$GRAFT_REPO_ROOT
for i in 1 2 3; do for m in 0 7 8; do echo -n "mode $m: "; GG_K2_DEBUG=$m timeout 200 python $R/tools/k4_ab.py 2>&1 | tail -1 | sed 's/.*scatter/scatter/; s/patch.*//'; done; done
cd /tmp; export TMPDIR=/tmp
for m in 7 8; do
GG_K2_DEBUG=$m timeout 200 rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --kernel-trace --output-format csv -d $R/gpurun_out/wr$m -o p -- python $R/bench.py --steps 3 --warmup 1 --cpu-seconds 0 --no-extras > /dev/null 2>&1
python $R/tools/pmc_report.py $R/gpurun_out/wr$m/p_counter_collection.csv | grep reduce
done
