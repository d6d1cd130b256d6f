#!/bin/bash
# tools/ubench/launch_cost under rocprofv3: per-kernel median durations, by kernel AND grid argument order of appearance
set -o pipefail
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 120 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/launch_cost -o t --output-format csv -- $R/tools/ubench/${1:-launch_cost} > /dev/null 2>&1 || exit 1
python3 - <<PY
import csv,collections
rows=list(csv.DictReader(open("$R/gpurun_out/launch_cost/t_kernel_trace.csv")))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))

per=max(1,len(rows)//int("${2:-20}"))
d=collections.defaultdict(list)
for i,r in enumerate(rows): d[(i%per, r["Kernel_Name"])].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1000)
for (i,k),v in sorted(d.items()):
    v=sorted(v[2:]); print(f"{i:2d} {k[:60]:60s} median {v[len(v)//2]:7.2f} us  min {v[0]:7.2f}")
PY
