#!/bin/bash
# L2 (TCC) write-side counters of the 960x544 bench step, one workgroup per picture and cluster of 4: which requests leave the L2 and why.
# usage (GPU box): bash tools/r02_tcc_run.sh   -> gpurun_out/tcc/<config>_<n>/  (read with tools/collect_profiles_r02.py --tcc)
cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out/tcc
B="python3 bench.py --steps 1 --warmup 0 --no-extras --no-cpu-baseline --height 544 --width 960"
n=0
for set in "TCC_NORMAL_WRITEBACK_sum TCC_ALL_TC_OP_WB_WRITEBACK_sum" "TCC_NORMAL_EVICT_sum TCC_WRITE_REQ_sum" "TCC_UC_WRITE_REQ_sum TCC_RW_WRITE_REQ_sum" "TCC_NC_WRITE_REQ_sum TCC_CC_WRITE_REQ_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_HIT_sum TCC_MISS_sum"; do
  n=$((n+1))
  HEVCE_CLUSTER=1 timeout -k 10 90 rocprofv3 --pmc $set -d gpurun_out/tcc/one_$n -- $B > gpurun_out/tcc/one_$n.log 2>&1 || { echo "set $n failed"; tail -3 gpurun_out/tcc/one_$n.log; exit 1; }
  echo "one $n ok: $set"
done
