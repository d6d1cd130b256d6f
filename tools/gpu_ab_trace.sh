#!/bin/bash
# same-box A/B by rocprofv3 kernel trace: default + variants, two rounds interleaved
set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python tests/manual/gpu_quick.py > gpurun_out/quick.log 2>&1 || { tail -30 gpurun_out/quick.log; exit 1; }
tail -1 gpurun_out/quick.log
for round in 1 2; do BENCH_EXTRA="--images-per-launch 1" bash tools/gpu_trace.sh "$@" 2>&1 | grep -E "^==|k_tile|k_entropy|k_finalize|sum of"; done
