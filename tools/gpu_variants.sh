#!/bin/bash
# A/B of prebuilt library variants (build_variants/lib_<name>.so): single-stream bench, kernel times
set -o pipefail
L=jpeg-image-compression_amd/libjpegamd.so
for f in build_variants/lib_*.so; do
  n=$(basename $f .so); n=${n#lib_}
  cp $f $L
  timeout -k 10 200 python bench.py --streams 1 --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/v_$n.json 2> gpurun_out/v_$n.err || { echo "$n FAILED"; tail -3 gpurun_out/v_$n.err; continue; }
  python - <<PY
import json
d=json.load(open("gpurun_out/v_$n.json")); r=d["roofline"]
print("%-12s transform %.2f  entropy %.2f  pack %.2f  total %.2f  %s" % ("$n", r["kernel_us"], r["entropy_us"], r["pack_us"], r["all_kernels_us"], d["parity"]))
PY
done
