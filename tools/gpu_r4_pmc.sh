#!/bin/bash
# Round-4 counter evidence over the single-stream bench (one image per launch unless IPL is set), every --pmc group in its own
# run with --kernel-trace only (MI355X_MICROARCH.md: separate passes).  PASSES="1 2 3 4 5" selects the passes.
set -o pipefail
R=$GRAFT_REPO_ROOT
TAG=${TAG:-r4pmc}
O=$R/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
IPL=${IPL:-1}
declare -a G
G[1]="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES"
G[2]="SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT"
G[3]="SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS"
G[4]="FETCH_SIZE"
G[5]="WRITE_SIZE"
for i in ${PASSES:-1 2}; do
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc ${G[$i]} -d $O/g$i -o p --output-format csv -- python3 $R/bench.py --streams 1 --images-per-launch $IPL --steps 20 --warmup 5 --no-cpu-baseline --no-one-image-pass "$@" > $O/g$i.log 2>&1 || { echo "group $i failed"; tail -5 $O/g$i.log; continue; }
  python3 $R/tools/pmc_summary.py $O/g$i/p_counter_collection.csv > $O/g$i.txt; cat $O/g$i.txt
done
