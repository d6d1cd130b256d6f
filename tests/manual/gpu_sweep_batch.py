#!/usr/bin/env python3
"""Randomised sweep of the batched launch (development aid): 2..8 images of one random geometry per launch, every
output against the oracle's file for that image alone."""
import random
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "jpeg-image-compression_amd" / "python"))
sys.path.insert(0, str(ROOT))
import torch                        # noqa: E402
import jpegamd                      # noqa: E402
from oracle import oracle           # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
bad = 0
t0 = time.time()
for i in range(n):
    cls = rng.random()
    if cls < 0.5:
        w, h = rng.randint(1, 300), rng.randint(1, 200)
    elif cls < 0.8:
        w, h = rng.randint(250, 2600), rng.randint(8, 64)
    else:
        w, h = rng.randint(300, 1400), rng.randint(100, 700)
    nb = rng.randint(2, 8)
    kind, flags = rng.randint(0, 3), rng.randint(0, 3)
    q = rng.choice([50, 50, 10, 90, rng.randint(1, 100)])
    bmps = [jpegamd.synth_bmp(w, h, rng.randint(1, 10 ** 6), kind, flags) for _ in range(nb)]
    ups = []
    for b in bmps:
        img, off = jpegamd.parse_bmp(b)
        ups.append((img, torch.frombuffer(bytearray(b[off:off + img.row_stride * img.height]), dtype=torch.uint8).cuda()))
    enc = jpegamd.Encoder(w, nb * ((h + 7) // 8 * 8))
    cap = 4096 + 2 * w * h
    outs = [torch.empty(cap, dtype=torch.uint8, device="cuda") for _ in range(nb)]
    sizes = [torch.zeros(1, dtype=torch.int64, device="cuda") for _ in range(nb)]
    imgs = [jpegamd.Encoder.image(px.data_ptr(), im.width, im.height, im.row_stride, bool(im.bottom_up), jpegamd.ORDER_BGR, q if q != 50 else 0)
            for im, px in ups]
    enc.encode_batch_async(imgs, [o.data_ptr() for o in outs], cap, [s.data_ptr() for s in sizes], True, torch.cuda.current_stream().cuda_stream)
    enc.finish()
    for j in range(nb):
        got = bytes(outs[j][:int(sizes[j].item())].cpu().numpy())
        if got != oracle.encode_bmp(bmps[j], q):
            bad += 1
            print(f"DIFF {w}x{h} batch {nb} image {j} kind={kind} flags={flags} q={q}", flush=True)
    del enc
    if i % 25 == 24:
        print(f"{i + 1} launches, {bad} bad, {time.time() - t0:.0f} s", flush=True)
print("FAILED" if bad else f"ALL {n} LAUNCHES MATCH")
sys.exit(1 if bad else 0)
