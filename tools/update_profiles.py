#!/usr/bin/env python3
"""Copy the judged summaries of the last tools/gpu_round.sh run from gpurun_out/r01 into profiles/ (tracked)."""
import collections
import csv
import json
import shutil
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
src, dst = ROOT / "gpurun_out" / "r01", ROOT / "profiles"
shutil.copy(src / "prof_stats" / "s1_kernel_stats.csv", dst / "r01_kernel_stats.csv")
shutil.copy(src / "bench_default.json", dst / "r01_bench.json")
shutil.copy(src / "bench_s1.json", dst / "r01_bench_streams1.json")
out = ["# rocprofv3 --kernel-trace --pmc FETCH_SIZE  /  --pmc WRITE_SIZE (separate passes), python3 bench.py --streams 1 --steps 20 --warmup 5 --no-cpu-baseline",
       "# 8192x8192 kind 0 Q=50, split pipeline (k_tile_transform + k_entropy + k_fin_count + k_fin_write); means over the launches, raw counter units (KiB)",
       "# HBM bytes = FETCH_SIZE KiB x 1024 x 2 (gfx950 tallies 128-B requests at 64 B, MI355X_MICROARCH.md HBM section) + WRITE_SIZE KiB x 1024",
       "kernel,fetch_kib_raw,write_kib_raw,hbm_bytes_corrected"]
vals = {}
for tag, f in (("f", src / "prof_fetch" / "f_counter_collection.csv"), ("w", src / "prof_write" / "w_counter_collection.csv")):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("jpegamd::", "")
        d[n].append(float(r["Counter_Value"]))
    for k, v in d.items():
        vals.setdefault(k, {})[tag] = sum(v) / len(v)
tj = {}
for k, v in vals.items():
    if not k.startswith("k_"):
        continue
    b = int(v["f"] * 1024 * 2 + v["w"] * 1024)
    out.append(f"{k},{v['f']:.1f},{v['w']:.1f},{b}")
    tj[k] = b
(dst / "r01_hbm_pmc.txt").write_text("\n".join(out) + "\n")
j = {"8192x8192_kind0": {"dominant_kernel": "k_tile_transform", "dominant_kernel_bytes_per_launch": tj["k_tile_transform<false>"],
                         "per_kernel_bytes_per_launch": tj, "all_kernels_bytes_per_image": sum(v for k, v in tj.items() if k != "k_sum_stats"),
                         "how": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (profiles/r01_hbm_pmc.txt); FETCH_SIZE (KiB) x2 per "
                                "MI355X_MICROARCH.md HBM section (gfx950 tallies 128-B requests at 64 B; measured ratio to the algorithmic read 0.505), WRITE_SIZE x1",
                         "algorithmic_read_bytes": 201326592}}
json.dump(j, open(dst / "hbm_traffic.json", "w"), indent=1)
print("\n".join(out))
