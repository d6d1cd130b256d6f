#!/usr/bin/env python3
"""PCIe- and file-inclusive rate of jpegamd_encode_files (never bench.py's `value`): N synthetic BMPs in /dev/shm ->
N .jpg files, pipelined, against the same files through the one-at-a-time path (jpegamd_encode_bmp_memory)."""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "jpeg-image-compression_amd" / "python"))
import jpegamd  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
w = h = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
d = Path("/dev/shm/jpegamd_bench_files")
d.mkdir(exist_ok=True)
ins, outs = [], []
for i in range(n):
    p = d / f"in_{i}.bmp"
    p.write_bytes(jpegamd.synth_bmp(w, h, 1000 + i, 0, 0))
    ins.append(p)
    outs.append(d / f"out_{i}.jpg")
try:
    jpegamd.encode_files(ins, outs)                               # warm up: contexts, page-in
    t0 = time.perf_counter()
    rc, status, st = jpegamd.encode_files(ins, outs)
    t1 = time.perf_counter()
    assert rc == 0, (rc, status)
    print(f"pipelined : {n} x {w}x{h}: {t1 - t0:.3f} s  {n * w * h / (t1 - t0) / 1e6:.0f} Mpixels/s  "
          f"({st.bytes_in / (t1 - t0) / 1e9:.1f} GB/s of BMP; fread {st.seconds_read:.3f} s, fwrite {st.seconds_write:.3f} s)")
    t0 = time.perf_counter()
    for i, o in zip(ins, outs):
        o.write_bytes(jpegamd.encode_bmp_bytes(i.read_bytes()))
    t1 = time.perf_counter()
    print(f"one by one: {n} x {w}x{h}: {t1 - t0:.3f} s  {n * w * h / (t1 - t0) / 1e6:.0f} Mpixels/s")
finally:
    for p in ins + outs:
        p.unlink(missing_ok=True)
    d.rmdir()
