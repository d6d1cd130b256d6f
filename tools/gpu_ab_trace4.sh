#!/bin/bash
# same-box A/B by kernel trace at 4 images per launch (steady state dominates): default + variants, two rounds
set -o pipefail
cd $GRAFT_REPO_ROOT
for round in 1 2; do BENCH_EXTRA="--images-per-launch 4" bash tools/gpu_trace.sh "$@" 2>&1 | grep -E "^==|k_tile|MISMATCH"; done
