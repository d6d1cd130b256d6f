#!/usr/bin/env python3
"""Per-kernel mean duration and the idle gap in front of each kernel from a rocprofv3 --kernel-trace CSV (single stream)."""
import csv, sys
from collections import defaultdict
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
dur, gap, prev_end = defaultdict(list), defaultdict(list), None
for r in rows:
    n = r["Kernel_Name"].split("(")[0].replace("jpegamd::", "").replace("void ", "")
    b, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    dur[n].append(e - b)
    if prev_end is not None:
        gap[n].append(b - prev_end)
    prev_end = e
tot = 0.0
for n in dur:
    d = sorted(dur[n]); g = sorted(gap[n]) or [0]
    med = d[len(d) // 2] / 1e3
    print(f"{n[:40]:40s} calls {len(d):5d}  dur mean {sum(d) / len(d) / 1e3:8.2f} us  median {med:8.2f}  min {d[0] / 1e3:8.2f}   gap before: median {g[len(g) // 2] / 1e3:6.2f} us")
    if len(d) > 50: tot += sum(d) / len(d) / 1e3
print(f"sum of per-image kernel means: {tot:.2f} us")
