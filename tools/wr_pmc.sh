#!/bin/bash
# which kernels send partly written lines to memory?  TCC -> EA write requests, all sizes vs 64-byte ones (separate pass, TCC only)
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -E "Counter_Name\s*:\s*TCC_EA0_WR" | sort -u
timeout 200 rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --kernel-trace --output-format csv -d $R/gpurun_out/wr -o p -- python $R/bench.py --steps 3 --warmup 1 --cpu-seconds 0 --no-extras > /dev/null 2>&1
python $R/tools/pmc_report.py $R/gpurun_out/wr/p_counter_collection.csv
