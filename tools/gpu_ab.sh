#!/bin/bash
# gpu_ab.sh [variant ...]: quick parity probe on the default build, then single-stream bench of default + each build_variants/lib_<variant>.so
# (two rounds, interleaved), then the stamp profile if build_variants/lib_stamps.so exists.
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/ab; mkdir -p $O
timeout -k 10 300 python tests/manual/gpu_quick.py > $O/quick.log 2>&1 || { tail -20 $O/quick.log; exit 1; }
tail -1 $O/quick.log
for round in 1 2; do
for v in default "$@"; do
  L=$PWD/jpeg-image-compression_amd/libjpegamd.so; [ $v != default ] && L=$PWD/build_variants/lib_$v.so
  JPEGAMD_LIB=$L timeout -k 10 200 python bench.py --streams 1 --images-per-launch 1 --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_$v.json 2> $O/bench_$v.err || [ $? -eq 3 ] || { tail -20 $O/bench_$v.err; exit 1; }
  python - <<PY
import json
d=json.load(open("$O/bench_$v.json")); r=d["roofline"]
print("%-10s value %.0f  transform %.2f  entropy %.2f  pack %.2f  total %.2f  %s" % ("$v", d["value"], r["kernel_us"], r["entropy_us"], r["pack_us"], r["sum_kernels_us"], d["parity"]))
PY
done
done
if [ -f build_variants/lib_stamps.so ]; then
JPEGAMD_LIB=$PWD/build_variants/lib_stamps.so timeout -k 10 200 python tools/stamp_profile_tile.py > $O/stamps.txt 2>&1 || { tail -20 $O/stamps.txt; exit 1; }
grep -v "std\|corr\|xcd-slot" $O/stamps.txt
fi
