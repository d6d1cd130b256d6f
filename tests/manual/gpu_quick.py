#!/usr/bin/env python3
"""Quick end-to-end GPU parity probe (development aid; the real suite is tests/ -m gpu)."""
import hashlib
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "jpeg-image-compression_amd" / "python"))
sys.path.insert(0, str(ROOT))

import jpegamd                      # noqa: E402
from oracle import oracle           # noqa: E402


def first_diff(a: bytes, b: bytes) -> int:
    n = min(len(a), len(b))
    for i in range(n):
        if a[i] != b[i]:
            return i
    return n if len(a) != len(b) else -1


def main():
    cases = [(8, 8, 1, 2, 0), (16, 8, 3, 0, 0), (64, 64, 1, 0, 0), (200, 120, 2, 0, 0), (203, 117, 5, 1, 0),
             (333, 250, 4, 0, 1), (512, 512, 101, 2, 0), (520, 16, 9, 1, 0), (1, 1, 7, 1, 0), (7, 9, 7, 1, 0),
             (640, 480, 6, 3, 2), (1920, 1080, 1, 0, 0), (1024, 1024, 7, 1, 0)]
    if len(sys.argv) > 1:
        cases = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]]
    bad = 0
    for (w, h, seed, kind, flags) in cases:
        bmp = jpegamd.synth_bmp(w, h, seed, kind, flags)
        t0 = time.time()
        got = jpegamd.encode_bmp_bytes(bmp)
        t1 = time.time()
        exp = oracle.encode_bmp(bmp)
        ok = got == exp
        bad += not ok
        print(f"{w}x{h} seed={seed} kind={kind} flags={flags}: gpu {len(got)} B, oracle {len(exp)} B, "
              f"{'MATCH' if ok else 'DIFF at byte %d' % first_diff(got, exp)}  ({(t1 - t0) * 1e3:.1f} ms) "
              f"sha {hashlib.sha256(got).hexdigest()[:12]}", flush=True)
    print("FAILED" if bad else "ALL MATCH")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
