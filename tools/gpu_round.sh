#!/bin/bash
# One GPU-box call: parity suite, bench (2-stream default and single-stream), rocprof summaries.
set -o pipefail
mkdir -p gpurun_out/r01
cd $GRAFT_REPO_ROOT
timeout -k 10 500 python -m pytest tests -x -q -m gpu > gpurun_out/r01/pytest_gpu.log 2>&1 || { tail -20 gpurun_out/r01/pytest_gpu.log; exit 1; }
tail -2 gpurun_out/r01/pytest_gpu.log
timeout -k 10 300 python bench.py > gpurun_out/r01/bench_default.json 2> gpurun_out/r01/bench_default.err || { tail -20 gpurun_out/r01/bench_default.err; exit 1; }
cat gpurun_out/r01/bench_default.json
timeout -k 10 300 python bench.py --streams 1 > gpurun_out/r01/bench_s1.json 2> gpurun_out/r01/bench_s1.err || { tail -20 gpurun_out/r01/bench_s1.err; exit 1; }
cat gpurun_out/r01/bench_s1.json
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r01/prof_stats -o s1 --output-format csv -- python3 $R/bench.py --streams 1 --steps 100 --warmup 10 --no-cpu-baseline > $R/gpurun_out/r01/prof_stats.log 2>&1 || { tail -20 $R/gpurun_out/r01/prof_stats.log; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/r01/prof_fetch -o f --output-format csv -- python3 $R/bench.py --streams 1 --steps 20 --warmup 5 --no-cpu-baseline > $R/gpurun_out/r01/prof_fetch.log 2>&1 || { tail -20 $R/gpurun_out/r01/prof_fetch.log; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/r01/prof_write -o w --output-format csv -- python3 $R/bench.py --streams 1 --steps 20 --warmup 5 --no-cpu-baseline > $R/gpurun_out/r01/prof_write.log 2>&1 || { tail -20 $R/gpurun_out/r01/prof_write.log; exit 1; }
ls -R $R/gpurun_out/r01 | head -40
