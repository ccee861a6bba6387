#!/bin/bash
# VALU / scalar / LDS instruction counts of k_reduce by phase: separate --pmc passes with the measurement switches
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
export GG_K2_PERSISTENT=2
for cfg in ${CFGS:-0:0 0:1 0:2 2:1 3:1 10:2 11:2}; do
  IFS=: read d k <<< "$cfg"
  set -- $d $k
  export GG_K2_DEBUG=$1 GG_K2_SKIP=$2
  timeout 150 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $R/gpurun_out/k2v_$1_$2 -o p -- python $R/bench.py --steps 3 --warmup 1 --no-extras --no-profile > /dev/null 2>&1
  echo "debug=$1 skip=$2: $(python $R/tools/pmc_report.py $R/gpurun_out/k2v_$1_$2/p_counter_collection.csv | grep k_reduce)"
done
