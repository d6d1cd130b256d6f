#!/usr/bin/env python3
"""Where do the exact-order events sit?  Histogram of the exact-path mask by zigzag position (2048^2 bench-like image)."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "jpeg-image-compression_amd" / "python"))
import numpy as np, torch, jpegamd
zz = [0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
      35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63]
w = h = 2048
bmp = jpegamd.synth_bmp(w, h, 1000, 0, 0)
img, off = jpegamd.parse_bmp(bmp)
px = torch.frombuffer(bytearray(bmp[off:off + img.row_stride * h]), dtype=torch.uint8).cuda()
enc = jpegamd.Encoder(w, h)
nb = (w // 8) * (h // 8)
y = torch.zeros(nb * 64, dtype=torch.int8, device="cuda"); z = torch.zeros(nb * 64, dtype=torch.int16, device="cuda")
mask = torch.zeros(nb, dtype=torch.int64, device="cuda")
d = jpegamd.Encoder.image(px.data_ptr(), w, h, img.row_stride, True)
enc.debug_stages(d, y.data_ptr(), z.data_ptr(), mask.data_ptr())
m = mask.cpu().numpy().astype(np.uint64)
cnt = np.array([int(((m >> np.uint64(zz[p])) & np.uint64(1)).sum()) for p in range(64)])
print("events", cnt.sum(), "per block", cnt.sum() / nb)
print("by zigzag position:", cnt.tolist())
print("by group of 16:", [int(cnt[16 * g:16 * g + 16].sum()) for g in range(4)])
c = jpegamd.mfma_consts(50)
print("delta by zigzag (x1e-4):", [round(float(c["delta"][zz[p]]) * 1e4, 2) for p in range(64)])
nz = (z.cpu().numpy().reshape(nb, 64) != 0).sum(axis=0)
print("non-zero coefficients by zigzag position:", nz.tolist())
