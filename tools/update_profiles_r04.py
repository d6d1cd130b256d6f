#!/usr/bin/env python3
"""Copy the judged summaries of tools/gpu_round4.sh / gpu_round4_prof.sh runs from gpurun_out/ into profiles/ (tracked):
bench lines, rocprofv3 kernel stats + trace summaries, SQ counter summary, HBM traffic (profiles/hbm_traffic.json), stamps."""
import json
import re
import shutil
import subprocess
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
go, dst = ROOT / "gpurun_out", ROOT / "profiles"
r04 = go / "r04"
for name in ("default", "driver20", "s1", "launch1", "launch1_s1", "q10", "q90", "kind1", "batch4096", "batch4096_launch8", "gather_rehearsal", "stitch", "stitch_q90", "image16384", "image16384_pair"):
    f = r04 / f"bench_{name}.json"
    if f.exists():
        shutil.copy(f, dst / f"r04_bench_{name}.json")
for ipl, tag in ((8, ""), (1, "_launch1")):
    f = r04 / f"kernel_stats_ipl{ipl}.csv"
    if f.exists():
        shutil.copy(f, dst / f"r04_kernel_stats{tag}.csv")
    tr = r04 / f"trace{ipl}" / "t_kernel_trace.csv"
    if tr.exists():
        out = subprocess.run(["python3", str(ROOT / "tools" / "trace_gaps.py"), str(tr)], capture_output=True, text=True).stdout
        steps = 60 if ipl == 8 else 200
        (dst / f"r04_kernel_trace_summary{tag}.txt").write_text(
            f"# rocprofv3 --kernel-trace --stats -- python3 bench.py --streams 1 --images-per-launch {ipl} --steps {steps} --warmup 10 --no-cpu-baseline --no-one-image-pass  (tools/gpu_r4_trace.sh, tools/trace_gaps.py)\n"
            + (f"# {ipl} images of 8192^2 per launch -- divide a duration by {ipl} for the per-image figure\n" if ipl > 1 else "")
            + "# (the run's burn-in, warm-up, timed and per-kernel-event passes are all in the trace: the means are over every launch)\n"
            + "".join(l for l in out.splitlines(True) if "rocclr" not in l and "elementwise" not in l))


def read_pmc(d):
    vals, lines = {}, {}
    for i in range(1, 6):
        f = d / f"g{i}.txt"
        if not f.exists():
            continue
        for line in f.read_text().splitlines():
            m = re.match(r"(\S+)\s+(\S+)\s+([0-9.]+)\s+launches", line)
            if m:
                vals.setdefault(m.group(1), {})[m.group(2)] = float(m.group(3))
                lines.setdefault(i, []).append(line)
    return vals, lines


v1, l1 = read_pmc(go / "r04_pmc1")
v8, l8 = read_pmc(go / "r04_pmc8")
sq = ["# rocprofv3 --kernel-trace --pmc <group> -- python3 bench.py --streams 1 --images-per-launch N --steps 20 --warmup 5 --no-cpu-baseline --no-one-image-pass,",
      "# one run per group (tools/gpu_r4_pmc.sh); mean counter value per kernel launch;",
      "# SQ_*_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES cycles",
      "# ---- one image per launch (comparable with r02_pmc_sq.txt) ----"]
for i in (1, 2, 3):
    sq += l1.get(i, [])
sq.append("# ---- eight images per launch (k_segment_merge with 16-tile segments) ----")
sq += l8.get(1, [])
(dst / "r04_pmc_sq.txt").write_text("\n".join(sq) + "\n")

out = ["# rocprofv3 --kernel-trace --pmc FETCH_SIZE  /  --pmc WRITE_SIZE (separate passes), python3 bench.py --streams 1 --images-per-launch N --steps 20 --warmup 5 --no-cpu-baseline --no-one-image-pass",
       "# 8192x8192 kind 0 Q=50; means over the launches, raw counter units (KiB)",
       "# HBM bytes = FETCH_SIZE KiB x 1024 x 2 (gfx950 tallies 128-B requests at 64 B, MI355X_MICROARCH.md HBM section) + WRITE_SIZE KiB x 1024",
       "# the x2 is calibrated for wide streaming reads: k_tile_encode reads 24 B per lane (16 + 8), measured ratio to its 201.3 MB of pixels per image in the last column;",
       "# k_segment_merge reads 16 B per lane, k_finalize 16 + 4 B: their corrected figures are upper bounds",
       "images_per_launch,kernel,fetch_kib_raw,write_kib_raw,hbm_bytes_corrected,fetch_raw_over_algorithmic_read"]
tj = {}
for ipl, vals in ((1, v1), (8, v8)):
    for k, v in sorted(vals.items()):
        if not k.startswith("k_") or "FETCH_SIZE" not in v or "WRITE_SIZE" not in v:
            continue
        b = int(v["FETCH_SIZE"] * 1024 * 2 + v["WRITE_SIZE"] * 1024)
        ratio = f"{v['FETCH_SIZE'] * 1024 / (201326592 * ipl):.3f}" if k.startswith("k_tile") else ""
        out.append(f"{ipl},{k},{v['FETCH_SIZE']:.1f},{v['WRITE_SIZE']:.1f},{b},{ratio}")
        tj.setdefault(ipl, {})[k] = b
if tj:
    (dst / "r04_hbm_pmc.txt").write_text("\n".join(out) + "\n")
    pipe1 = sum(tj.get(1, {}).values())
    pipe8 = sum(tj.get(8, {}).values())
    j = {"8192x8192_kind0": {"pipeline_bytes_per_image": (pipe8 // 8) if pipe8 else pipe1,
                             "pipeline_bytes_per_image_one_image_per_launch": pipe1,
                             "pipeline_bytes_per_launch_of_eight": pipe8,
                             "per_kernel_bytes_per_launch": {str(k): v for k, v in tj.items()},
                             "dominant_kernel": "k_tile_encode",
                             "algorithmic_read_bytes": 201326592,
                             "how": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes at eight images per launch (profiles/r04_hbm_pmc.txt, "
                                    "tools/gpu_round4_prof.sh); FETCH_SIZE (KiB) x2 per MI355X_MICROARCH.md HBM section (gfx950 tallies 128-B requests at 64 B), "
                                    "WRITE_SIZE x1; sum over k_tile_encode, k_segment_merge, k_finalize, divided by 8"}}
    json.dump(j, open(dst / "hbm_traffic.json", "w"), indent=1)
    print("\n".join(out)); print("pipeline bytes per image: one per launch", pipe1, " eight per launch", pipe8 // 8 if pipe8 else None)
st = go / "stamps" / "stamps.txt"
if st.exists():
    (dst / "r04_stamps.txt").write_text("# tools/gpu_stamps.sh: build_variants/lib_stamps.so (-DJPEGAMD_STAMPS), tools/stamp_profile_tile.py, 8192^2 seed 1000, one image per launch\n"
                                        + "".join(l for l in st.read_text().splitlines(True) if "amdgpu.ids" not in l))
