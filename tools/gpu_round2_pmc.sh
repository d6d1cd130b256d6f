#!/bin/bash
# Round-2 counter evidence over the single-stream bench, every --pmc group in its own run (with --kernel-trace only):
# four SQ groups, then FETCH_SIZE and WRITE_SIZE (MI355X_MICROARCH.md, HBM section: separate passes).
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02pmc; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" \
           "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT" \
           "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $grp -d $O/g$i -o p --output-format csv -- python3 $R/bench.py --streams 1 --images-per-launch 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/g$i.log 2>&1 || { echo "group $i failed"; tail -5 $O/g$i.log; continue; }
  python3 $R/tools/pmc_summary.py $O/g$i/p_counter_collection.csv > $O/g$i.txt; tail -30 $O/g$i.txt | grep -E "k_tile|FETCH|WRITE" | head -12
done
