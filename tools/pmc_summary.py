#!/usr/bin/env python3
"""Mean counter value per kernel launch from rocprofv3 --pmc counter_collection.csv files."""
import collections
import csv
import sys

for f in sys.argv[1:]:
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("jpegamd::", "")
        d[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in sorted(d.items()):
        if k.startswith("k_") and not k.startswith("k_sum"):
            print(f"{k:28s} {c:28s} {sum(v) / len(v):16.1f}  launches {len(v)}")
