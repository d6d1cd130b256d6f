#!/usr/bin/env python3
"""Randomised GPU-vs-oracle sweep (development aid; the committed suite is tests/ -m gpu): many small and
odd-shaped images, every synthetic kind and flag combination, three qualities."""
import random
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "jpeg-image-compression_amd" / "python"))
sys.path.insert(0, str(ROOT))
import jpegamd                      # noqa: E402
from oracle import oracle           # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
t0 = time.time()
for i in range(n):
    cls = rng.random()
    if cls < 0.5:
        w, h = rng.randint(1, 300), rng.randint(1, 200)
    elif cls < 0.8:
        w, h = rng.randint(250, 2100), rng.randint(8, 64)            # wide: many tiles per row, partial last tiles
    elif cls < 0.97:
        w, h = rng.randint(300, 1400), rng.randint(200, 900)
    else:
        w, h = rng.randint(1500, 6000), rng.randint(100, 2500)         # occasional large shapes: the shared context grows in steps
    seed, kind, flags = rng.randint(1, 10 ** 6), rng.randint(0, 3), rng.randint(0, 3)
    q = rng.choice([50, 50, 50, 10, 90, rng.randint(1, 100)])
    bmp = jpegamd.synth_bmp(w, h, seed, kind, flags)
    got = jpegamd.encode_bmp_bytes(bmp, q if q != 50 else 0)
    exp = oracle.encode_bmp(bmp, q)
    if got != exp:
        bad += 1
        print(f"DIFF {w}x{h} seed={seed} kind={kind} flags={flags} q={q}: gpu {len(got)} B, oracle {len(exp)} B", flush=True)
    if i % 50 == 49:
        print(f"{i + 1} cases, {bad} bad, {time.time() - t0:.0f} s", flush=True)
print("FAILED" if bad else f"ALL {n} MATCH")
sys.exit(1 if bad else 0)
