#!/bin/bash
# quick parity probe + single-stream and default bench, compact output
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 120 python tests/manual/gpu_quick.py > gpurun_out/quick.log 2>&1; tail -1 gpurun_out/quick.log
for s in 1 2; do
  timeout -k 10 200 python bench.py --streams $s --no-cpu-baseline > gpurun_out/bench_s$s.json 2> gpurun_out/bench_s$s.err || { tail -5 gpurun_out/bench_s$s.err; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/bench_s$s.json")); r=d["roofline"]
print("streams $s: value %.0f  ms/step %.4f  transform %.2f  entropy %.2f  pack %.2f  total %.2f  frac %.3f  exact %d  %s" % (d["value"], d["ms_per_step"], r["kernel_us"], r["entropy_us"], r["pack_us"], r["sum_kernels_us"], r["frac"], d["exact_fallbacks_per_image"], d["parity"]))
PY
done
