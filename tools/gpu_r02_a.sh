#!/bin/bash
# Round-2 call A: baseline of the round-1 kernels (parity suite, single-stream bench, phase stamps) + the issue-model micro-benchmark
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02a; mkdir -p $O
timeout -k 10 400 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1 || { tail -20 $O/pytest_gpu.log; exit 1; }
tail -2 $O/pytest_gpu.log
timeout -k 10 200 python bench.py --streams 1 --steps 100 --warmup 20 --no-cpu-baseline > $O/bench_s1.json 2> $O/bench_s1.err || { tail -20 $O/bench_s1.err; exit 1; }
cat $O/bench_s1.json
JPEGAMD_LIB=$PWD/build_variants/lib_stamps.so timeout -k 10 200 python tools/stamp_profile_tile.py > $O/stamps.txt 2>&1 || { tail -20 $O/stamps.txt; exit 1; }
cat $O/stamps.txt
timeout -k 10 300 tools/ubench/issue_model > $O/issue_model.txt 2>&1 || { tail -20 $O/issue_model.txt; exit 1; }
cat $O/issue_model.txt
