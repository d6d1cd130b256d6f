"""Edges of the declared operating range, and a fixed-seed slice of the randomised sweeps (tests/manual/), in the driver-run suite.

jpegamd_encoder_create accepts images up to 65535 pixels wide / high; the tile geometry of k_tile_encode is division-free
(multiply-high by a rounded reciprocal + one correction, jpegamd_tile_pipeline.hip: geo()), row offsets are 32-bit inside a tile
and 64-bit across rows, and k_stitch's look-back runs over as many workgroups as the picture has segments / 8.  Everything
against the oracle, byte for byte.  Nothing here reads /root/reference."""
from __future__ import annotations

import random

import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test started without a GPU: the product path has no CPU fallback")
    return torch.device("cuda:0")


def _encode_batch(jpegamd, bmps, quality, dev):
    """bmps: BMP files of ONE geometry -> their JFIF files through ONE launch of each kernel (one image: the plain entry)."""
    ups = []
    for b in bmps:
        img, off = jpegamd.parse_bmp(b)
        ups.append((img, torch.frombuffer(bytearray(b[off:off + img.row_stride * img.height]), dtype=torch.uint8).to(dev)))
    w, h = ups[0][0].width, ups[0][0].height
    enc = jpegamd.Encoder(w, len(bmps) * ((h + 7) // 8 * 8))
    cap = 4096 + 2 * w * h
    outs = [torch.empty(cap, dtype=torch.uint8, device=dev) for _ in bmps]
    sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in bmps]
    imgs = [jpegamd.Encoder.image(px.data_ptr(), im.width, im.height, im.row_stride, bool(im.bottom_up), jpegamd.ORDER_BGR, quality)
            for im, px in ups]
    stream = torch.cuda.current_stream().cuda_stream
    if len(bmps) == 1:
        enc.encode_async(imgs[0], outs[0].data_ptr(), cap, sizes[0].data_ptr(), True, stream)
    else:
        enc.encode_batch_async(imgs, [o.data_ptr() for o in outs], cap, [s.data_ptr() for s in sizes], True, stream)
    enc.finish()
    return [bytes(o[:int(n.item())].cpu().numpy()) for o, n in zip(outs, sizes)]


@pytest.mark.parametrize("w,h", [(65535, 8), (8, 65535), (65528, 24), (4104, 8), (4096, 8), (65535, 1), (1, 65535)])
def test_dimension_limits(jpegamd, oracle, dev, w, h):
    """The widest and the tallest picture the header admits, 256 tiles per row with and without a ragged last tile, and one /
    two tiles per row next to a row boundary (4104 = 16 tiles + 1 block, 4096 = 16 tiles): alone and as a batch of two."""
    for kind, flags, q in ((0, 0, 0), (1, 1, 0), (0, 2, 90)):
        bmps = [jpegamd.synth_bmp(w, h, 70 + i + kind, kind, flags) for i in range(2)]
        want = [oracle.encode_bmp(b, q if q else 50) for b in bmps]
        assert _encode_batch(jpegamd, bmps[:1], q, dev)[0] == want[0], (w, h, kind, flags, q, "single")
        assert _encode_batch(jpegamd, bmps, q, dev) == want, (w, h, kind, flags, q, "batch of two")


def test_sweep_slice_through_the_shared_context(jpegamd, oracle, dev):
    """300 cases of tests/manual/gpu_sweep.py (seed 4): 1x1 .. 6000x2500, every synthetic kind and flag combination, qualities
    1..100, through jpegamd_encode_bmp_memory (the shared context grows and is reused)."""
    rng = random.Random(4)
    for i in range(300):
        cls = rng.random()
        if cls < 0.5:
            w, h = rng.randint(1, 300), rng.randint(1, 200)
        elif cls < 0.8:
            w, h = rng.randint(250, 2100), rng.randint(8, 64)
        elif cls < 0.97:
            w, h = rng.randint(300, 1400), rng.randint(200, 900)
        else:
            w, h = rng.randint(1500, 6000), rng.randint(100, 2500)
        seed, kind, flags = rng.randint(1, 10 ** 6), rng.randint(0, 3), rng.randint(0, 3)
        q = rng.choice([50, 50, 50, 10, 90, rng.randint(1, 100)])
        bmp = jpegamd.synth_bmp(w, h, seed, kind, flags)
        assert jpegamd.encode_bmp_bytes(bmp, q if q != 50 else 0) == oracle.encode_bmp(bmp, q), (i, w, h, seed, kind, flags, q)


def test_sweep_slice_of_batched_launches(jpegamd, oracle, dev):
    """12 launches of tests/manual/gpu_sweep_batch.py (seed 9) with 2 .. 32 images of one random geometry each."""
    rng = random.Random(9)
    for i in range(12):
        cls = rng.random()
        if cls < 0.5:
            w, h = rng.randint(1, 300), rng.randint(1, 200)
        elif cls < 0.8:
            w, h = rng.randint(250, 2600), rng.randint(8, 64)
        else:
            w, h = rng.randint(300, 1400), rng.randint(100, 700)
        nb = rng.choice([2, 3, 5, 8, 13, 32]) if w * h < 200_000 else rng.randint(2, 8)
        kind, flags = rng.randint(0, 3), rng.randint(0, 3)
        q = rng.choice([50, 50, 10, 90, rng.randint(1, 100)])
        bmps = [jpegamd.synth_bmp(w, h, rng.randint(1, 10 ** 6), kind, flags) for _ in range(nb)]
        got = _encode_batch(jpegamd, bmps, q if q != 50 else 0, dev)
        for j in range(nb):
            assert got[j] == oracle.encode_bmp(bmps[j], q), (i, w, h, nb, j, kind, flags, q)
