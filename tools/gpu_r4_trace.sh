#!/bin/bash
# Round-4 quick measurement: kernel traces (rocprofv3 --kernel-trace --stats) of the single-stream bench at eight images per launch
# and at one; prints the per-kernel mean durations.  Optional args: bench.py flags for both runs.
set -o pipefail
R=$GRAFT_REPO_ROOT
TAG=${TAG:-r4}
O=$R/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for ipl in 8 1; do
  steps=60; [ $ipl = 1 ] && steps=200
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/trace$ipl -o t --output-format csv -- python3 $R/bench.py --streams 1 --images-per-launch $ipl --steps $steps --warmup 10 --no-cpu-baseline --no-one-image-pass "$@" > $O/trace$ipl.json 2> $O/trace$ipl.err || [ $? -eq 3 ] || { tail -20 $O/trace$ipl.err; exit 1; }
  echo "== $ipl image(s) per launch"; python3 $R/tools/trace_gaps.py $O/trace$ipl/t_kernel_trace.csv | grep -v "rocclr\|elementwise"
  cp $O/trace$ipl/t_kernel_stats.csv $O/kernel_stats_ipl$ipl.csv
  python3 - <<PY
import json
d=json.load(open("$O/trace$ipl.json")); r=d["roofline"]
print("bench: value %.0f  frac %.3f  transform %.2f merge %.2f finalize %.2f sum %.2f per-image %.2f  %s" % (d["value"], r["frac"], r["kernel_us"], r["merge_us"], r["finalize_us"], r["sum_kernels_us"], r["per_image_us"], d["parity"]))
PY
done
