#!/usr/bin/env python3
"""Static instruction census of k_tile_encode<false> between the TSTAMP markers.
  hipcc ... -DJPEGAMD_MARKS --save-temps=obj -c csrc/jpegamd_tile_pipeline.hip ; isa_census.py file.s"""
import re
import sys
from collections import Counter

src = open(sys.argv[1]).read()
m = re.search(r"^_ZN7jpegamd13k_tile_encodeILb0EEE.*?:\n(.*?)\n\s*s_endpgm", src, re.S | re.M)
body = m.group(1).splitlines()
phase = "pre"
cnt = {}
for line in body:
    t = line.strip()
    if t.startswith("; MARK"):
        phase = "after MARK " + t.split()[2]
        continue
    if not t or t.startswith((";", ".")) or t.endswith(":"):
        continue
    op = t.split()[0]
    cls = ("mfma" if op.startswith("v_mfma") else "valu" if op.startswith("v_") else "branch" if op.startswith("s_cbranch") or op == "s_branch"
           else "waitcnt" if op == "s_waitcnt" else "salu" if op.startswith("s_") else "lds" if op.startswith("ds_")
           else "vmem" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else "other")
    cnt.setdefault(phase, Counter())[cls] += 1
for ph, c in cnt.items():
    print(f"{ph:16s} " + "  ".join(f"{k}={v}" for k, v in sorted(c.items())))
