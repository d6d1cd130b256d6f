#!/usr/bin/env python3
"""Static VALU-cycle estimate of k_tile_encode<false> between the TSTAMP markers, with the measured issue costs
(profiles/r02_issue_model_forms.txt): VOP2/VOP1 add/sub/xor/and/or/mov/mul_f32/add_f32/fmac/fma with VGPR operands 2 cycles,
every other VALU form 4, MFMA 8 (issue), DPP / SDWA 4.  Straight-line count: every branch of the marked region is summed."""
import re, sys
from collections import Counter, defaultdict
src = open(sys.argv[1]).read()
m = re.search(r"^_ZN7jpegamd13k_tile_encodeILb0EEE.*?:\n(.*?)\n\s*s_endpgm", src, re.S | re.M)
FAST = {"v_add_u32","v_sub_u32","v_subrev_u32","v_xor_b32","v_and_b32","v_or_b32","v_mov_b32","v_add_f32","v_sub_f32","v_mul_f32","v_fmac_f32","v_fma_f32","v_mov_b64"}
phase = "pre"; cyc = defaultdict(float); cnt = defaultdict(Counter)
for line in m.group(1).splitlines():
    t = line.strip()
    if t.startswith("; MARK"): phase = "after MARK " + t.split()[2]; continue
    if not t or t.startswith((";", ".")) or t.endswith(":"): continue
    op = t.split()[0]; base = re.sub(r"_(e32|e64|dpp|sdwa)$", "", op)
    if op.startswith("v_mfma"): c, k = 8, "mfma"
    elif op.startswith("v_"):
        ops = t[len(op):]
        fast = base in FAST and not op.endswith(("_dpp", "_sdwa")) and not re.search(r"\bs\d|\bs\[|0x|vcc|exec|\b-?\d+(\.\d+)?\b(?!\])", ops.split(";")[0].replace("v[","").replace("a[",""))
        c, k = (2, "valu2") if fast else (4, "valu4")
    elif op.startswith("s_"): c, k = 0, "salu"
    elif op.startswith("ds_"): c, k = 0, "lds"
    else: c, k = 0, "vmem"
    cyc[phase] += c; cnt[phase][k] += 1
tot = 0
for ph in cnt:
    print(f"{ph:14s} valu-cycles {cyc[ph]:7.0f}   " + "  ".join(f"{k}={v}" for k, v in sorted(cnt[ph].items())))
