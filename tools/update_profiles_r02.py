#!/usr/bin/env python3
"""Copy the judged summaries of tools/gpu_round2.sh / gpu_round2_pmc.sh runs from gpurun_out/ into profiles/ (tracked):
bench lines, rocprofv3 kernel stats + trace summary, SQ counter summary, HBM traffic (profiles/hbm_traffic.json)."""
import json
import re
import shutil
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
r02, pmc, dst = ROOT / "gpurun_out" / "r02", ROOT / "gpurun_out" / "r02pmc", ROOT / "profiles"
for name in ("default", "driver20", "s1", "launch1", "launch1_s1", "launch4", "q10", "q90", "kind1", "batch4096", "batch4096_launch1"):
    f = r02 / f"bench_{name}.json"
    if f.exists():
        shutil.copy(f, dst / f"r02_bench_{name}.json")
for tag, flags in (("", ""), ("_launch1", " --images-per-launch 1")):
    if (r02 / f"kernel_stats{tag}.csv").exists():
        shutil.copy(r02 / f"kernel_stats{tag}.csv", dst / f"r02_kernel_stats{tag}.csv")
    if (r02 / f"trace{tag}.txt").exists():
        (dst / f"r02_kernel_trace_summary{tag}.txt").write_text(
            f"# rocprofv3 --kernel-trace --stats -- python3 bench.py --streams 1 --steps 100 --warmup 10 --no-cpu-baseline --no-one-image-pass{flags}  (tools/gpu_trace.sh, tools/trace_gaps.py)\n"
            + ("# default: 8 images of 8192^2 per launch -- divide a duration by 8 for the per-image figure\n" if not tag else "")
            + "".join(l for l in (r02 / f"trace{tag}.txt").read_text().splitlines(True) if "rocclr" not in l and "elementwise" not in l))

vals = {}
sq = ["# rocprofv3 --kernel-trace --pmc <group> -- python3 bench.py --streams 1 --steps 20 --warmup 5 --no-cpu-baseline, one run per group (tools/gpu_round2_pmc.sh)",
      "# mean counter value per kernel launch; SQ_*_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES cycles"]
for i in range(1, 6):
    f = pmc / f"g{i}.txt"
    if not f.exists():
        continue
    for line in f.read_text().splitlines():
        m = re.match(r"(\S+)\s+(\S+)\s+([0-9.]+)\s+launches", line)
        if not m:
            continue
        vals.setdefault(m.group(1), {})[m.group(2)] = float(m.group(3))
        if i <= 3:
            sq.append(line)
(dst / "r02_pmc_sq.txt").write_text("\n".join(sq) + "\n")

out = ["# rocprofv3 --kernel-trace --pmc FETCH_SIZE  /  --pmc WRITE_SIZE (separate passes), python3 bench.py --streams 1 --steps 20 --warmup 5 --no-cpu-baseline",
       "# 8192x8192 kind 0 Q=50; means over the launches, raw counter units (KiB)",
       "# HBM bytes = FETCH_SIZE KiB x 1024 x 2 (gfx950 tallies 128-B requests at 64 B, MI355X_MICROARCH.md HBM section) + WRITE_SIZE KiB x 1024",
       "# the x2 is calibrated for wide streaming reads: k_tile_transform reads 24 B per lane (16 + 8), measured ratio to its 201.3 MB of pixels in the last column;",
       "# k_entropy reads 8 B per lane, k_finalize 4 B: their corrected figures are upper bounds",
       "kernel,fetch_kib_raw,write_kib_raw,hbm_bytes_corrected,fetch_raw_over_algorithmic_read"]
tj = {}
for k, v in vals.items():
    if not k.startswith("k_") or "FETCH_SIZE" not in v or "WRITE_SIZE" not in v:
        continue
    b = int(v["FETCH_SIZE"] * 1024 * 2 + v["WRITE_SIZE"] * 1024)
    ratio = f"{v['FETCH_SIZE'] * 1024 / 201326592:.3f}" if k.startswith("k_tile") else ""
    out.append(f"{k},{v['FETCH_SIZE']:.1f},{v['WRITE_SIZE']:.1f},{b},{ratio}")
    tj[k] = b
if tj:
    (dst / "r02_hbm_pmc.txt").write_text("\n".join(out) + "\n")
    pipe = sum(b for k, b in tj.items() if k.split("<")[0] in ("k_tile_transform", "k_entropy", "k_finalize"))
    j = {"8192x8192_kind0": {"pipeline_bytes_per_image": pipe, "per_kernel_bytes_per_launch": tj,
                             "dominant_kernel": "k_tile_transform", "dominant_kernel_bytes_per_launch": tj.get("k_tile_transform<false>"),
                             "algorithmic_read_bytes": 201326592,
                             "how": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (profiles/r02_hbm_pmc.txt); FETCH_SIZE (KiB) x2 per "
                                    "MI355X_MICROARCH.md HBM section (gfx950 tallies 128-B requests at 64 B), WRITE_SIZE x1; sum over k_tile_transform, k_entropy, k_finalize"}}
    json.dump(j, open(dst / "hbm_traffic.json", "w"), indent=1)
    print("\n".join(out)); print("pipeline bytes per image", pipe)
