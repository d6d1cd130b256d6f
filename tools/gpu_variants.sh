#!/bin/bash
# A/B of prebuilt library variants (build_variants/lib_<name>.so): single-stream kernel times and multi-stream throughput
set -o pipefail
L=jpeg-image-compression_amd/libjpegamd.so
for f in build_variants/lib_*.so; do
  n=$(basename $f .so); n=${n#lib_}
  cp $f $L
  for s in 1 4; do
  timeout -k 10 200 python bench.py --streams $s --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/v_$n.json 2> gpurun_out/v_$n.err || { echo "$n FAILED"; tail -3 gpurun_out/v_$n.err; continue; }
  python - <<PY
import json
d=json.load(open("gpurun_out/v_$n.json")); r=d["roofline"]
print("%-10s streams $s value %.0f  transform %.2f  entropy %.2f  pack %.2f  total %.2f  %s" % ("$n", d["value"], r["kernel_us"], r["entropy_us"], r["pack_us"], r["sum_kernels_us"], d["parity"]))
PY
  done
done
