#!/bin/bash
# Same-box A/B over workloads: the shipped library against build_variants/lib_<name>.so at Q=50 / Q=90 / noise, eight images per launch
# (bench.py's own events, no profiler).  usage: gpu_r3_ab_q.sh <name>
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${TAG:-r3abq}; mkdir -p $O
cd /tmp
timeout -k 10 300 python3 $R/tests/manual/gpu_quick.py > $O/quick.log 2>&1 || { tail -30 $O/quick.log; exit 1; }
for round in $(seq 1 ${ROUNDS:-2}); do
for w in "--quality 50" "--quality 90" "--kind 1" ${MORE_W:+"$MORE_W"}; do
for v in default "$@"; do
  L=$R/jpeg-image-compression_amd/libjpegamd.so; [ $v != default ] && L=$R/build_variants/lib_$v.so
  tag=$(echo $w | tr -d ' -')
  JPEGAMD_LIB=$L timeout -k 10 300 python3 $R/bench.py --images-per-launch 8 --steps 60 --warmup 10 --no-cpu-baseline --no-one-image-pass $w > $O/$v.$tag.$round.json 2> $O/$v.$tag.$round.err || [ $? -eq 3 ] || { tail -20 $O/$v.$tag.$round.err; exit 1; }
  python3 - <<PY
import json
d=json.load(open("$O/$v.$tag.$round.json")); r=d["roofline"]
print("%-8s %-10s round $round: value %.0f frac %.3f encode %.1f merge %.1f finalize %.1f sum %.1f exact/img %s  %s" % ("$v", "$tag", d["value"], r["frac"], r["kernel_us"], r["merge_us"], r["finalize_us"], r["sum_kernels_us"], d.get("exact_fallbacks_per_image"), d["parity"][:40]))
PY
done
done
done
