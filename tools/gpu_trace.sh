#!/bin/bash
# kernel trace of the single-stream bench: per-kernel durations and the gaps between consecutive kernels (tools/trace_gaps.py)
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/trace; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for v in default "$@"; do
L=$R/jpeg-image-compression_amd/libjpegamd.so; [ $v != default ] && L=$R/build_variants/lib_$v.so
JPEGAMD_LIB=$L timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/$v -o t --output-format csv -- python3 $R/bench.py --streams 1 --steps 100 --warmup 10 --no-cpu-baseline --no-one-image-pass $BENCH_EXTRA > $O/$v.log 2>&1 || [ $? -eq 3 ] || { tail -20 $O/$v.log; exit 1; }
echo "== $v"; python3 $R/tools/trace_gaps.py $O/$v/t_kernel_trace.csv
done
