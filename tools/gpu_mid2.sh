#!/bin/bash
# quick probe + selected GPU tests + trace + stamps
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/mid; mkdir -p $O
timeout -k 10 300 python tests/manual/gpu_quick.py > $O/quick.log 2>&1 || { tail -30 $O/quick.log; exit 1; }
tail -1 $O/quick.log
timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "${1:-goldens or random or large or sharded or taps or ties or exact}" > $O/pytest_gpu.log 2>&1 || { tail -40 $O/pytest_gpu.log; exit 1; }
tail -2 $O/pytest_gpu.log
bash tools/gpu_trace.sh > $O/trace.txt 2>&1; cat $O/trace.txt | grep -v "rocclr\|elementwise\|sum_stats"
if [ -f build_variants/lib_stamps.so ]; then
JPEGAMD_LIB=$PWD/build_variants/lib_stamps.so timeout -k 10 200 python tools/stamp_profile_tile.py > $O/stamps.txt 2>&1 || { tail -20 $O/stamps.txt; exit 1; }
grep -v "std\|corr\|xcd-slot" $O/stamps.txt
fi
