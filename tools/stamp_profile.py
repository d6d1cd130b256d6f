#!/usr/bin/env python3
"""Per-phase cycle shares of k_transform_mfma from in-kernel s_memtime stamps.
Needs a diagnostic build:  make -C jpeg-image-compression_amd EXTRA_HIPFLAGS=-DJPEGAMD_STAMPS
(read the SHARES, not the absolute time: the stamps' fences forbid overlap the real kernel has)."""
import ctypes as C
import os
import sys
from pathlib import Path

os.environ["JPEGAMD_STAMPS"] = "1"
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "jpeg-image-compression_amd" / "python"))
import numpy as np
import torch
import jpegamd

w = h = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
bmp = jpegamd.synth_bmp(w, h, 1000, 0, 0)
img, off = jpegamd.parse_bmp(bmp)
px = torch.frombuffer(bytearray(bmp[off:off + img.row_stride * h]), dtype=torch.uint8).cuda()
enc = jpegamd.Encoder(w, h)
cap = 4096 + w * h
out = torch.empty(cap, dtype=torch.uint8, device="cuda")
size = torch.zeros(1, dtype=torch.int64, device="cuda")
d = jpegamd.Encoder.image(px.data_ptr(), w, h, img.row_stride, True)
for _ in range(3):
    enc.encode_async(d, out.data_ptr(), cap, size.data_ptr(), True, 0)
    enc.finish()
nseg = ((h + 7) // 8) * (((w + 7) // 8 + 127) // 128)
buf = np.zeros((nseg, 16), np.uint64)
fn = jpegamd.lib.jpegamd_debug_read_stamps
fn.restype = C.c_int32
fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
rc = fn(enc._h, buf.ctypes.data, nseg)
assert rc == 0, rc
names = ["tile prologue", "loads+luma", "mfma", "quantise", "exact fallback", "counts+scans", "scatter", "symbol batches", "epilogue"]
tot = buf[:, :9].sum()
print(f"segments {nseg}, mean cycles per segment {buf[:, :9].sum(axis=1).mean():.0f}")
for i, n in enumerate(names):
    print(f"  {n:16s} {buf[:, i].mean():10.0f} cycles/segment  {100.0 * buf[:, i].sum() / tot:5.1f} %")
