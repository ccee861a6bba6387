#!/bin/bash
# Texture-addresser (TA) load of every kernel at the default batch: a wave instruction whose 64 lanes name 64 different cache lines
# occupies the CU's TA for ~2 cycles per line.  Separate passes, no trace domains beside the kernel trace.
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
i=0
for set in "TA_BUSY_avr TA_BUSY_max GRBM_GUI_ACTIVE" \
           "TCP_PENDING_STALL_CYCLES_sum TD_TD_BUSY_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout 150 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/ta_$i -o p -- python $R/bench.py --steps 3 --warmup 1 --no-extras --no-profile > /dev/null 2>$R/gpurun_out/ta_$i.err
  python $R/tools/pmc_report.py $R/gpurun_out/ta_$i/p_counter_collection.csv
done
