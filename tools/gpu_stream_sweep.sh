#!/bin/bash
mkdir -p gpurun_out
for s in 3 4 5 6 8; do
  timeout -k 10 200 python bench.py --streams $s --no-cpu-baseline > gpurun_out/ss_$s.json 2>/dev/null
  python - <<PY
import json
d = json.load(open("gpurun_out/ss_$s.json"))
print("streams $s:", d["value"], d["ms_per_step"], d["host_issue_us_per_step"])
PY
done
