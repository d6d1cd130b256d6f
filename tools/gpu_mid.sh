#!/bin/bash
# quick probe + selected GPU tests + bench + trace  (args: pytest -k expression)
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/mid; mkdir -p $O
timeout -k 10 300 python tests/manual/gpu_quick.py > $O/quick.log 2>&1 || { tail -30 $O/quick.log; exit 1; }
tail -1 $O/quick.log
timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "${1:-goldens or random or large or sharded or segment or concurrent}" > $O/pytest_gpu.log 2>&1 || { tail -40 $O/pytest_gpu.log; exit 1; }
tail -2 $O/pytest_gpu.log
bash tools/gpu_trace.sh > $O/trace.txt 2>&1; cat $O/trace.txt | grep -v rocclr
