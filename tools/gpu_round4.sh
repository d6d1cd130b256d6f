#!/bin/bash
# Round-4 evidence run, part 1: the whole GPU suite, driver-style bench lines (default, driver's 20 steps, single stream, one image per
# launch, Q=10/90, noise, batch4096 through the gather path, the N > 1 rehearsal with its configs3 leg).
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1 || { tail -40 $O/pytest_gpu.log; exit 1; }
tail -2 $O/pytest_gpu.log
run() { name=$1; shift; timeout -k 10 400 python bench.py "$@" > $O/bench_$name.json 2> $O/bench_$name.err || { echo "bench $name failed"; tail -5 $O/bench_$name.err; return 1; }
  python - <<PY
import json
d=json.load(open("$O/bench_$name.json")); r=d["roofline"]; o=r.get("one_image_per_launch")
print("%-10s value %.0f ms/step %.4f  frac %.3f dominant %.3f  encode %.2f merge %.2f finalize %.2f sum %.2f  one-image %s  %s" % ("$name", d["value"], d["ms_per_step"], r["frac"], r["dominant_frac"], r["kernel_us"], r["merge_us"], r["finalize_us"], r["sum_kernels_us"], (o["sum_kernels_us"], o["frac"]) if o else None, d["parity"][:60]))
PY
}
run default || exit 1
run driver20 --steps 20 --warmup 5
run s1 --streams 1 --steps 100 --warmup 10 --no-cpu-baseline
run launch1 --images-per-launch 1 --no-cpu-baseline
run launch1_s1 --images-per-launch 1 --streams 1 --steps 200 --warmup 20 --no-cpu-baseline
run q10 --quality 10 --steps 100 --warmup 10 --no-cpu-baseline
run q90 --quality 90 --steps 100 --warmup 10 --no-cpu-baseline
run kind1 --kind 1 --steps 50 --warmup 5 --no-cpu-baseline
run batch4096 --workload batch4096 --force-gather --steps 30 --warmup 5 --no-cpu-baseline
run batch4096_launch8 --workload batch4096 --images-per-launch 8 --force-gather --steps 30 --warmup 5 --no-cpu-baseline
run gather_rehearsal --force-gather --steps 300 --warmup 20 --no-cpu-baseline
run stitch --pipeline stitch --steps 100 --warmup 10 --no-cpu-baseline
run stitch_q90 --pipeline stitch --quality 90 --steps 100 --warmup 10 --no-cpu-baseline
run image16384 --width 16384 --height 16384 --images-per-launch 1 --steps 40 --warmup 5 --no-cpu-baseline
run image16384_pair --width 16384 --height 16384 --images-per-launch 1 --pipeline pair --steps 40 --warmup 5 --no-cpu-baseline
