#!/bin/bash
# Same-box A/B (round 4): the shipped library against build_variants/lib_<name>.so -- bench.py's own per-kernel events at eight images
# per launch and at one (the literal configs[2] shape).  usage: gpu_r4_ab.sh <name>...   env: TAG, ROUNDS, WORKLOADS ("--quality 50" ...)
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${TAG:-r4ab}; mkdir -p $O
cd /tmp
timeout -k 10 300 python3 $R/tests/manual/gpu_quick.py > $O/quick.log 2>&1 || { tail -30 $O/quick.log; exit 1; }
IFS=';' read -ra WL <<< "${WORKLOADS:---quality 50}"
for round in $(seq 1 ${ROUNDS:-2}); do
for w in "${WL[@]}"; do
for v in default "$@"; do
  L=$R/jpeg-image-compression_amd/libjpegamd.so; [ $v != default ] && L=$R/build_variants/lib_$v.so
  tag=$(echo $w | tr -d ' -')
  JPEGAMD_LIB=$L timeout -k 10 300 python3 $R/bench.py --images-per-launch 8 --steps 60 --warmup 10 --no-cpu-baseline ${BENCH_ARGS:-} $w > $O/$v.$tag.$round.json 2> $O/$v.$tag.$round.err || [ $? -eq 3 ] || { tail -20 $O/$v.$tag.$round.err; exit 1; }
  python3 - <<PY
import json
d=json.load(open("$O/$v.$tag.$round.json")); r=d["roofline"]; o=r.get("one_image_per_launch") or {}
print("%-10s %-10s r$round: value %.0f frac %.3f | 8/launch encode %.1f tail %.1f sum %.1f | 1/launch encode %s tail %s sum %s frac %s | %s" % ("$v", "$tag", d["value"], r["frac"], r["kernel_us"], r["merge_us"] + r["finalize_us"], r["sum_kernels_us"], o.get("transform_us"), (o.get("merge_us") or 0) + (o.get("finalize_us") or 0), o.get("sum_kernels_us"), o.get("frac"), d["parity"][:24]))
PY
done
done
done
