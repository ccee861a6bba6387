#!/bin/bash
# SQ counter breakdown of the kernels for one cloud per launch (tools/kprof.py hdl); separate passes, SQ counters only
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" \
           "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS"; do
  i=$((i+1))
  timeout 150 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/spmc_$i -o p -- python $R/tools/kprof.py hdl > /dev/null 2>&1
  python $R/tools/pmc_report.py $R/gpurun_out/spmc_$i/p_counter_collection.csv | grep -E "spiral|reduce"
done
