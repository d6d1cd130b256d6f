#!/bin/bash
# Round-2 evidence run: whole GPU suite, driver-style bench lines (default, single stream, Q=10/90, noise, batch4096 rehearsal),
# rocprofv3 kernel stats of the single-stream bench.
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02; mkdir -p $O
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1 || { tail -40 $O/pytest_gpu.log; exit 1; }
tail -2 $O/pytest_gpu.log
run() { name=$1; shift; timeout -k 10 400 python bench.py "$@" > $O/bench_$name.json 2> $O/bench_$name.err || { echo "bench $name failed"; tail -5 $O/bench_$name.err; return 1; }
  python - <<PY
import json
d=json.load(open("$O/bench_$name.json")); r=d["roofline"]
print("%-10s value %.0f ms/step %.4f  frac %.3f dominant %.3f  transform %.2f entropy %.2f pack %.2f sum %.2f  %s" % ("$name", d["value"], d["ms_per_step"], r["frac"], r["dominant_frac"], r["kernel_us"], r["entropy_us"], r["pack_us"], r["sum_kernels_us"], d["parity"]))
PY
}
run default || exit 1
run driver20 --steps 20 --warmup 5 --no-cpu-baseline
run s1 --streams 1 --steps 100 --warmup 10 --no-cpu-baseline
run launch1 --images-per-launch 1 --no-cpu-baseline
run launch1_s1 --images-per-launch 1 --streams 1 --steps 200 --warmup 20 --no-cpu-baseline
run launch4 --images-per-launch 4 --no-cpu-baseline
run q10 --quality 10 --steps 100 --warmup 10 --no-cpu-baseline
run q90 --quality 90 --steps 100 --warmup 10 --no-cpu-baseline
run kind1 --kind 1 --steps 50 --warmup 5 --no-cpu-baseline
run batch4096 --workload batch4096 --force-gather --steps 50 --warmup 10 --no-cpu-baseline
run batch4096_launch1 --workload batch4096 --images-per-launch 1 --force-gather --steps 50 --warmup 10 --no-cpu-baseline
# kernel traces of the single-stream bench: the default (8 images per launch) and one image per launch
bash tools/gpu_trace.sh > $O/trace.txt 2>&1; grep -v "rocclr\|elementwise" $O/trace.txt
cp gpurun_out/trace/default/t_kernel_stats.csv $O/kernel_stats.csv
BENCH_EXTRA="--images-per-launch 1" bash tools/gpu_trace.sh > $O/trace_launch1.txt 2>&1; grep -v "rocclr\|elementwise" $O/trace_launch1.txt
cp gpurun_out/trace/default/t_kernel_stats.csv $O/kernel_stats_launch1.csv
