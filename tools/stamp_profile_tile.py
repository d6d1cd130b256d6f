#!/usr/bin/env python3
"""Per-phase cycle shares of k_tile_encode from in-kernel s_memtime stamps (per wave, summed over its tiles).
Needs a diagnostic build:  make -C jpeg-image-compression_amd EXTRA_HIPFLAGS=-DJPEGAMD_STAMPS
The stamps do not drain vmcnt, so memory overlap is as shipped; s_memtime ticks at 100 MHz on gfx950."""
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
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
quality = int(sys.argv[3]) if len(sys.argv) > 3 else 0                 # 0: the default (50)
kind = int(sys.argv[4]) if len(sys.argv) > 4 else 0
bmp = jpegamd.synth_bmp(w, h, seed, kind, 0)
img, off = jpegamd.parse_bmp(bmp)
px = torch.frombuffer(bytearray(bmp[off:off + img.row_stride * h]), dtype=torch.uint8).cuda()
enc = jpegamd.Encoder(w, h)
cap = 4096 + w * h
out = torch.empty(cap, dtype=torch.uint8, device="cuda")
size = torch.zeros(1, dtype=torch.int64, device="cuda")
d = jpegamd.Encoder.image(px.data_ptr(), w, h, img.row_stride, True, jpegamd.ORDER_BGR, quality)
for _ in range(3):
    enc.encode_async(d, out.data_ptr(), cap, size.data_ptr(), True, 0)
    enc.finish()
nwaves = min(512, (((h + 7) // 8) * (((w + 7) // 8 + 31) // 32) + 7) // 8) * 8
buf = np.zeros((nwaves, 16), np.uint64)
fn = jpegamd.lib.jpegamd_debug_read_stamps
fn.restype = C.c_int32
fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
rc = fn(enc._h, buf.ctypes.data, nwaves)
assert rc == 0, rc
names = ["loop/geometry", "wait rows + luma", "luma -> LDS", "mfma", "quantise", "exact fallback", "counts+scans", "ticket, geometry, row requests",
         "appends", "coding", "record, copy-out"]
NP = len(names)
ph = buf[:, :NP]                                                   # phase sums: slots 0..10 (11..14 hold the real-time stamps and the cycle count)
tot = ph.sum()
print(f"waves {nwaves}, mean ticks per wave {ph.sum(axis=1).mean():.0f} (max {ph.sum(axis=1).max()})")
for i, n in enumerate(names):
    print(f"  {n:26s} {ph[:, i].mean():10.0f} ticks/wave  {100.0 * ph[:, i].sum() / tot:5.1f} %")
tw = ph.sum(axis=1).astype(np.float64).reshape(-1, 8)          # [workgroup][wave]
print(f"per-wave total: mean {tw.mean():.0f}  std {tw.std():.0f}  p50 {np.percentile(tw, 50):.0f}  p90 {np.percentile(tw, 90):.0f}  p99 {np.percentile(tw, 99):.0f}  max {tw.max():.0f}")
print(f"workgroup means: std {tw.mean(axis=1).std():.0f}  min {tw.mean(axis=1).min():.0f}  max {tw.mean(axis=1).max():.0f};  within-workgroup std (mean) {tw.std(axis=1).mean():.0f}")
for i, n in enumerate(names):
    col = ph[:, i].astype(np.float64)
    print(f"  {n:26s} std {col.std():8.0f}  p99 {np.percentile(col, 99):8.0f}  max {col.max():8.0f}")
ex = buf[:, 5].astype(np.float64)
print("corr(total, exact) =", np.corrcoef(tw.reshape(-1), ex)[0, 1], " corr(total, coding) =", np.corrcoef(tw.reshape(-1), buf[:, 9].astype(np.float64))[0, 1])

rt = buf[:, 11:14].astype(np.float64)
span = (rt[:, 2].max() - rt[:, 0].min()) / 100.0
print(f"kernel span seen by the waves: {span:.2f} us; entry spread {(rt[:, 0].max() - rt[:, 0].min()) / 100.0:.2f} us; "
      f"prologue (entry -> loop) mean {(rt[:, 1] - rt[:, 0]).mean() / 100.0:.2f} us max {(rt[:, 1] - rt[:, 0]).max() / 100.0:.2f} us; "
      f"loop mean {(rt[:, 2] - rt[:, 1]).mean() / 100.0:.2f} us max {(rt[:, 2] - rt[:, 1]).max() / 100.0:.2f} us; "
      f"first wave done at {(rt[:, 2].min() - rt[:, 0].min()) / 100.0:.2f} us")
clk = buf[:, 14].astype(np.float64) / np.maximum(rt[:, 2] - rt[:, 1], 1.0) * 0.1
print(f"shader clock inside the loop: mean {clk.mean():.3f} GHz (min {clk.min():.3f}, max {clk.max():.3f})")
# end-of-kernel balance: when do the ticket groups (blockIdx & 63) and the XCDs (blockIdx % 8) run dry?
blk = np.arange(nwaves) // 8
t0 = rt[:, 0].min()
end = (rt[:, 2] - t0) / 100.0
grp_end = np.array([end[(blk & 63) == g].max() for g in range(64)])
grp_first = np.array([end[(blk & 63) == g].min() for g in range(64)])
print(f"group end times (us): min {grp_end.min():.2f} mean {grp_end.mean():.2f} max {grp_end.max():.2f}; first wave of a group done: min {grp_first.min():.2f} mean {grp_first.mean():.2f}")
for x in range(8):
    m = (blk % 8) == x
    print(f"  xcd-slot {x}: clock {clk[m].mean():.3f} GHz  waves end mean {end[m].mean():.2f} max {end[m].max():.2f} us")
print(f"idle SIMD-time at the end: {(end.max() - end).mean():.2f} us per wave of {end.max():.2f} ({100 * (end.max() - end).mean() / end.max():.1f} %)")

hist, edges = np.histogram(end, bins=np.arange(np.floor(end.min()), np.ceil(end.max()) + 1.0, 1.0))
print("waves finishing per microsecond:", " ".join(f"{int(e)}:{c}" for e, c in zip(edges[:-1], hist)))
busy = np.array([(end > t).sum() for t in np.arange(0.0, end.max(), 1.0)])
print("waves still running at t (us):", " ".join(f"{int(t)}:{b}" for t, b in zip(np.arange(0.0, end.max(), 1.0), busy) if t >= end.min() - 2))
os.makedirs(str(ROOT / "gpurun_out" / "stamps"), exist_ok=True)
np.save(str(ROOT / "gpurun_out" / "stamps" / (os.path.basename(os.environ.get("JPEGAMD_LIB", "default")) + ".npy")), buf)
