#!/bin/bash
# further SQ counter passes over the single-stream bench: where the issue time of each instruction class goes
set -o pipefail
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc
export TMPDIR=/tmp
cd /tmp
i=2
for grp in "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS" \
           "SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_SMEM SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_VALU_MFMA_COEXEC_CYCLES SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU2 SQ_WAVE_CYCLES"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $grp -d $R/gpurun_out/pmc/g$i -o p --output-format csv -- python3 $R/bench.py --streams 1 --images-per-launch 1 --steps 10 --warmup 3 --no-cpu-baseline > $R/gpurun_out/pmc/g$i.log 2>&1 || { tail -5 $R/gpurun_out/pmc/g$i.log; }
  python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc/g$i/p_counter_collection.csv
done
