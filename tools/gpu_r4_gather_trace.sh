#!/bin/bash
# Kernel trace of the N > 1 path rehearsed on one rank (bench.py --force-gather) and of the same run without the gather: where do the
# extra microseconds per step go?  Output: gpurun_out/$TAG/{gather,plain}_kernel_trace.csv + a timeline of a few steps.
set -o pipefail
R=$GRAFT_REPO_ROOT
TAG=${TAG:-r04/gtrace}
O=$R/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for mode in gather plain; do
  extra=""; [ $mode = gather ] && extra="--force-gather --configs3-steps 0"
  timeout -k 10 400 rocprofv3 --kernel-trace -d $O/$mode -o t --output-format csv -- python3 $R/bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-one-image-pass $extra "$@" > $O/$mode.json 2> $O/$mode.err || [ $? -eq 3 ] || { tail -20 $O/$mode.err; exit 1; }
  f=$(find $O/$mode -name "*kernel_trace.csv" | head -1)
  cp $f $O/${mode}_kernel_trace.csv
  python3 - <<PY
import csv, json
d = json.load(open("$O/$mode.json"))
print("== $mode: value %.0f Mpx/s, %.4f ms/step" % (d["value"], d["ms_per_step"]), d.get("gather"))
rows = list(csv.DictReader(open("$O/${mode}_kernel_trace.csv")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the timed region: the last 40 * 1 launches of k_tile_encode before the event passes; print a window of it
enc = [i for i, r in enumerate(rows) if "k_tile_encode" in r["Kernel_Name"]]
i0 = enc[len(enc) // 3]
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0:i0 + 40]:
    n = r["Kernel_Name"].split("(")[0].replace("jpegamd::", "").replace("void ", "")[:46]
    print("%9.1f us  +%8.1f us  %-46s q%s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, n, r.get("Queue_Id", "?")))
PY
done
