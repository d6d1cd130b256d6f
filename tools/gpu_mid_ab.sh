#!/bin/bash
# quick probe + selected GPU tests on the default build, then the same-box kernel-trace A/B (default + variants, two rounds)
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/mid; mkdir -p $O
timeout -k 10 300 python tests/manual/gpu_quick.py > $O/quick.log 2>&1 || { tail -30 $O/quick.log; exit 1; }
tail -1 $O/quick.log
timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "${K:-goldens or random or large or sharded or segment or concurrent or quality}" > $O/pytest_gpu.log 2>&1 || { tail -40 $O/pytest_gpu.log; exit 1; }
tail -2 $O/pytest_gpu.log
for round in 1 2; do BENCH_EXTRA="--images-per-launch 1" bash tools/gpu_trace.sh "$@" 2>&1 | grep -E "^==|k_tile|k_entropy|k_finalize|sum of"; done
