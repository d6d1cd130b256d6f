#!/bin/bash
# SQ counter passes over the single-stream bench (each --pmc group is its own run)
set -o pipefail
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc
export TMPDIR=/tmp
cd /tmp
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" \
           "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $grp -d $R/gpurun_out/pmc/g$i -o p --output-format csv -- python3 $R/bench.py --streams 1 --images-per-launch 1 --steps 10 --warmup 3 --no-cpu-baseline > $R/gpurun_out/pmc/g$i.log 2>&1 || { tail -5 $R/gpurun_out/pmc/g$i.log; }
  python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc/g$i/p_counter_collection.csv
done
