#!/usr/bin/env python3
"""Generate the committed golden vectors from the COMPILED REFERENCE (oracle/_ref).

Run in the build container only (it needs /root/reference for `make -C oracle ref`):

    python tests/golden/make_goldens.py [--large]

Outputs (all data, no reference source):
  tests/golden/manifest.json   fixture list: synthetic generator parameters or asset-crop file,
                               expected JFIF size + sha256
  tests/golden/<name>.jpg      expected JFIF bytes written by the reference's own
                               jpeg_compression_app (natural_c built with its Makefile flags)
  tests/golden/<name>.bmp      input, only for crops of the reference's sample images
                               (assets/input/*.bmp); synthetic inputs are regenerated from the
                               integer-only generator (jpegamd_synth_bmp) and are not stored
  tests/golden/large.json      size + sha256 of the reference's output for the big synthetic
                               configs (1920x1080, 4096^2, 8192^2; BASELINE.json configs)
  tests/golden/quality.json    size + sha256 for quality 10 / 90: the reference rebuilt in a
                               temp dir with ONLY the 64 table bytes of jpeg_tables.c:3-12
                               replaced by the libjpeg-scaled table
"""
from __future__ import annotations

import argparse
import hashlib
import json
import re
import struct
import subprocess
import sys
import tempfile
from pathlib import Path

HERE = Path(__file__).resolve().parent
ROOT = HERE.parents[1]
sys.path.insert(0, str(ROOT / "jpeg-image-compression_amd" / "python"))
sys.path.insert(0, str(ROOT))
REF_ROOT = Path("/root/reference")

import jpegamd                      # noqa: E402  (generator only; no device needed)
from oracle import oracle           # noqa: E402

# name, width, height, seed, kind, flags
SYNTH = [
    ("flat_odd_64", 64, 64, 101, 2, 0),            # every DC on a rounding tie
    ("flat_even_40x24", 40, 24, 100, 2, 0),
    ("gradient_200x120", 200, 120, 0, 3, 0),
    ("noise_203x117", 203, 117, 5, 1, 0),          # W%8 != 0, row padding, no-EOB blocks, stuffing
    ("photo_333x250_topdown", 333, 250, 4, 0, 1),  # negative biHeight
    ("photo_320x180_off138", 320, 180, 2, 0, 2),   # bfOffBits = 138
    ("one_pixel", 1, 1, 7, 1, 0),
    ("one_block", 8, 8, 1, 0, 0),
    ("noise_7x9", 7, 9, 7, 1, 0),
    ("noise_520x16", 520, 16, 9, 1, 0),            # 65 blocks per row: a 1-block second segment
    ("photo_1024x64", 1024, 64, 3, 0, 0),          # two full segments per row
    ("photo_512x512", 512, 512, 11, 0, 0),
]
CROPS = [  # name, asset, x0, y0, w, h
    ("lena_crop_128", "lena.bmp", 200, 200, 128, 128),
    ("greenland_crop_130x75", "greenland.bmp", 300, 500, 130, 75),
    ("offset_sample_crop_160x120", "offset_sample.bmp", 600, 300, 160, 120),
    ("blackbuck_crop_96x200", "blackbuck.bmp", 100, 60, 96, 200),
]
LARGE = [  # width, height, seed, kind
    (1920, 1080, 1, 0), (2048, 2048, 7, 1), (4096, 4096, 2, 0),
    (8192, 8192, 1000, 0), (8192, 8192, 1001, 0), (8192, 8192, 1002, 0),
]
# BASELINE.json configs[4] at its stated shape (quality sweep on 8192^2) + the symbol-dense stress (noise) the bench can be
# pointed at: the three rotating bench seeds each, so that bench.py's parity key resolves whichever step comes last.
LARGE_Q = [(8192, 8192, seed, 0, q) for q in (10, 90) for seed in (1000, 1001, 1002)]
LARGE_NOISE = [(8192, 8192, seed, 1) for seed in (1000, 1001, 1002)]
# The 16 rotating 8192^2 inputs of bench.py's default step (seeds 1000..1015) and the first eight of them at Q=10 / Q=90: what
# the -m gpu tests of the batched launch (8 images through one launch of each kernel) are hashed against.
LARGE_BENCH = [(8192, 8192, seed, 0) for seed in range(1003, 1016)]
LARGE_BENCH_Q = [(8192, 8192, seed, 0, q) for q in (10, 90) for seed in range(1003, 1008)]
# BASELINE.json configs[3]: 64 independent 4096x4096 images, distinct seeds
BATCH4096 = [(4096, 4096, 2000 + i, 0) for i in range(64)]


def ref_encode(bmp: bytes, app: Path) -> bytes:
    with tempfile.TemporaryDirectory(dir="/dev/shm") as td:
        src, dst = Path(td) / "in.bmp", Path(td) / "out.jpg"
        src.write_bytes(bmp)
        subprocess.run([str(app), str(src), str(dst)], check=True, stdout=subprocess.DEVNULL)
        return dst.read_bytes()


def crop_asset(asset: str, x0: int, y0: int, w: int, h: int) -> bytes:
    """Crop a reference sample image (top-down coordinates) into a plain bottom-up 24-bit BMP."""
    data = (REF_ROOT / "assets" / "input" / asset).read_bytes()
    off = struct.unpack_from("<I", data, 10)[0]
    W, H = struct.unpack_from("<ii", data, 18)
    top_down = H < 0
    H = abs(H)
    stride = (3 * W + 3) & ~3
    out_stride = (3 * w + 3) & ~3
    rows = []
    for y in range(y0 + h - 1, y0 - 1, -1):                  # bottom-up output
        fr = y if top_down else H - 1 - y
        row = data[off + fr * stride + 3 * x0: off + fr * stride + 3 * (x0 + w)]
        rows.append(row + b"\0" * (out_stride - len(row)))
    hdr = b"BM" + struct.pack("<IHHI", 54 + out_stride * h, 0, 0, 54)
    hdr += struct.pack("<IiiHHIIiiII", 40, w, h, 1, 24, 0, out_stride * h, 2835, 2835, 0, 0)
    return hdr + b"".join(rows)


def build_quality_ref(quality: int, workdir: Path) -> Path:
    """The reference with only its quantisation-table bytes replaced (temp dir; nothing kept)."""
    src_root = REF_ROOT / "natural_c"
    text = (src_root / "src/core/jpeg_tables.c").read_text()
    table = oracle.quant_table(quality)
    body = ",\n    ".join(", ".join(str(int(v)) for v in table[r * 8:(r + 1) * 8]) for r in range(8))
    new, n = re.subn(r"(std_luminance_quant_tbl\[64\]\s*=\s*\{)[^}]*(\})", r"\1\n    " + body + r"\n\2", text, count=1)
    assert n == 1
    patched = workdir / "jpeg_tables_q.c"
    patched.write_text(new)
    srcs = [str(p) for p in (src_root / "src/core").glob("*.c") if p.name != "jpeg_tables.c"]
    srcs += [str(p) for p in (src_root / "src/io").glob("*.c")] + [str(src_root / "src/main.c"), str(patched)]
    app = workdir / f"app_q{quality}"
    subprocess.run(["gcc", f"-I{src_root}/include", "-g", "-w", *srcs, "-o", str(app), "-lm"], check=True)
    return app


def _one_large(job):
    """(w, h, seed, kind, quality, app path) -> (key, entry); run in a worker process (the reference is single-threaded)."""
    w, h, seed, kind, q, app = job
    bmp = jpegamd.synth_bmp(w, h, seed, kind, 0)
    jpg = ref_encode(bmp, Path(app))
    return (f"{w}x{h}_seed{seed}_kind{kind}_q{q}",
            dict(size=len(jpg), sha256=hashlib.sha256(jpg).hexdigest(), bmp_sha256=hashlib.sha256(bmp).hexdigest()))


def extend_large(which: str):
    """--configs: add the BASELINE configs[3] / configs[4] answers (and the 8192^2 noise images) without touching the rest."""
    from concurrent.futures import ProcessPoolExecutor
    oracle.build(ref=True)
    with tempfile.TemporaryDirectory() as td:
        jobs = []
        if which in ("all", "quality"):
            apps = {q: str(build_quality_ref(q, Path(td))) for q in (10, 90)}
            jobs += [(w, h, seed, kind, q, apps[q]) for (w, h, seed, kind, q) in LARGE_Q]
        if which in ("all", "noise"):
            jobs += [(w, h, seed, kind, 50, str(oracle.REF_APP)) for (w, h, seed, kind) in LARGE_NOISE]
        if which in ("all", "bench"):
            apps = {q: str(build_quality_ref(q, Path(td))) for q in (10, 90)}
            jobs += [(w, h, seed, kind, 50, str(oracle.REF_APP)) for (w, h, seed, kind) in LARGE_BENCH]
            jobs += [(w, h, seed, kind, q, apps[q]) for (w, h, seed, kind, q) in LARGE_BENCH_Q]
        big = {}
        with ProcessPoolExecutor(max_workers=3) as ex:           # ~1.3 GB per 8192^2 reference run
            for key, ent in ex.map(_one_large, jobs):
                big[key] = ent
                print("large", key, ent["size"], flush=True)
        path = HERE / "large.json"
        large = json.loads(path.read_text())
        large.update(big)
        path.write_text(json.dumps(large, indent=1) + "\n")
        if which in ("all", "batch"):
            batch = {}
            with ProcessPoolExecutor(max_workers=6) as ex:
                for key, ent in ex.map(_one_large, [(w, h, seed, kind, 50, str(oracle.REF_APP)) for (w, h, seed, kind) in BATCH4096]):
                    batch[key] = ent
                    print("batch", key, ent["size"], flush=True)
            (HERE / "batch4096.json").write_text(json.dumps(batch, indent=1) + "\n")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--large", action="store_true", help="also (re)generate large.json (minutes of CPU)")
    ap.add_argument("--configs", choices=["all", "quality", "noise", "batch", "bench"],
                    help="only ADD the BASELINE configs[3]/[4] answers: 8192^2 at Q=10/90 and noise into large.json, "
                         "64 x 4096^2 into batch4096.json (the other files stay as they are)")
    args = ap.parse_args()
    if args.configs:
        extend_large(args.configs)
        return
    oracle.build(ref=True)
    app = oracle.REF_APP
    manifest = []
    for name, w, h, seed, kind, flags in SYNTH:
        bmp = jpegamd.synth_bmp(w, h, seed, kind, flags)
        jpg = ref_encode(bmp, app)
        (HERE / f"{name}.jpg").write_bytes(jpg)
        manifest.append(dict(name=name, source="synth", width=w, height=h, seed=seed, kind=kind, flags=flags,
                             bmp_sha256=hashlib.sha256(bmp).hexdigest(), jpg_size=len(jpg),
                             jpg_sha256=hashlib.sha256(jpg).hexdigest()))
    for name, asset, x0, y0, w, h in CROPS:
        bmp = crop_asset(asset, x0, y0, w, h)
        (HERE / f"{name}.bmp").write_bytes(bmp)
        jpg = ref_encode(bmp, app)
        (HERE / f"{name}.jpg").write_bytes(jpg)
        manifest.append(dict(name=name, source="asset-crop", asset=asset, x0=x0, y0=y0, width=w, height=h,
                             bmp_sha256=hashlib.sha256(bmp).hexdigest(), jpg_size=len(jpg),
                             jpg_sha256=hashlib.sha256(jpg).hexdigest()))
    (HERE / "manifest.json").write_text(json.dumps(manifest, indent=1) + "\n")

    # whole reference assets: known answers (inputs are not redistributed; checked only where present)
    assets = {}
    for p in sorted((REF_ROOT / "assets" / "input").glob("*.bmp")):
        bmp = p.read_bytes()
        jpg = ref_encode(bmp, app)
        assets[p.name] = dict(bmp_sha256=hashlib.sha256(bmp).hexdigest(), jpg_size=len(jpg),
                              jpg_sha256=hashlib.sha256(jpg).hexdigest())
    (HERE / "assets.json").write_text(json.dumps(assets, indent=1) + "\n")

    # quality extension
    quality = {}
    with tempfile.TemporaryDirectory() as td:
        for q in (10, 90):
            qapp = build_quality_ref(q, Path(td))
            for name, w, h, seed, kind, flags in SYNTH[:6] + [SYNTH[-1]]:
                bmp = jpegamd.synth_bmp(w, h, seed, kind, flags)
                jpg = ref_encode(bmp, qapp)
                quality[f"{name}_q{q}"] = dict(width=w, height=h, seed=seed, kind=kind, flags=flags, quality=q,
                                               jpg_size=len(jpg), jpg_sha256=hashlib.sha256(jpg).hexdigest())
    (HERE / "quality.json").write_text(json.dumps(quality, indent=1) + "\n")

    if args.large:
        large = {}
        for w, h, seed, kind in LARGE:
            bmp = jpegamd.synth_bmp(w, h, seed, kind, 0)
            jpg = ref_encode(bmp, app)
            large[f"{w}x{h}_seed{seed}_kind{kind}_q50"] = dict(size=len(jpg), sha256=hashlib.sha256(jpg).hexdigest(),
                                                               bmp_sha256=hashlib.sha256(bmp).hexdigest())
            print("large", w, h, seed, kind, len(jpg), flush=True)
        (HERE / "large.json").write_text(json.dumps(large, indent=1) + "\n")
    print(f"wrote {len(manifest)} fixtures, {len(assets)} asset answers, {len(quality)} quality answers")


if __name__ == "__main__":
    main()
