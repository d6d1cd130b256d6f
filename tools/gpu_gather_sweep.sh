#!/bin/bash
# one-rank rehearsal of bench.py's gather path for several batch sizes
mkdir -p gpurun_out
for g in 8 32 64; do
  timeout -k 10 200 python bench.py --force-gather --gather-every $g --steps 256 --warmup 64 --no-cpu-baseline > gpurun_out/fg_$g.json 2> gpurun_out/fg_$g.err
  python - <<PY
import json
d = json.load(open("gpurun_out/fg_$g.json"))
print("gather-every $g:", d["value"], d["ms_per_step"], d["host_issue_us_per_step"])
PY
done
