#!/bin/bash
# how memory-bound is k_tile_transform?  the same bench with 1 input (Infinity-Cache resident) and with 6 (603 -> 1206 MB in rotation)
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/rot; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for n in 3 1 6 3; do
JPEGAMD_BENCH_ROTATE=$n timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/r$n -o t --output-format csv -- python3 $R/bench.py --streams 1 --images-per-launch 1 --steps 100 --warmup 10 --no-cpu-baseline > $O/r$n.log 2>&1 || { tail -5 $O/r$n.log; }
echo "== rotate $n"; python3 $R/tools/trace_gaps.py $O/r$n/t_kernel_trace.csv | grep -E "k_tile|k_entropy|k_finalize"
done
