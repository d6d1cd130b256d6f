"""Edges of the declared operating range, and a fixed-seed slice of the randomised sweeps (tests/manual/), in the driver-run suite.

jpegamd_encoder_create accepts images up to 65535 pixels wide / high; the tile geometry of k_tile_encode is division-free
(multiply-high by a rounded reciprocal + one correction, jpegamd_tile_pipeline.hip: geo()), row offsets are 32-bit inside a tile
and 64-bit across rows, and k_stitch's look-back runs over as many workgroups as the picture has segments / 8.  Everything
against the oracle, byte for byte.  Nothing here reads /root/reference."""
from __future__ import annotations

import random

import numpy as np

import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test started without a GPU: the product path has no CPU fallback")
    return torch.device("cuda:0")


def _encode_batch(jpegamd, bmps, quality, dev, pipeline=None):
    """bmps: BMP files of ONE geometry -> their JFIF files through ONE launch of each kernel (one image: the plain entry)."""
    ups = []
    for b in bmps:
        img, off = jpegamd.parse_bmp(b)
        ups.append((img, torch.frombuffer(bytearray(b[off:off + img.row_stride * img.height]), dtype=torch.uint8).to(dev)))
    w, h = ups[0][0].width, ups[0][0].height
    enc = jpegamd.Encoder(w, len(bmps) * ((h + 7) // 8 * 8))
    if pipeline is not None:
        enc.set_pipeline(pipeline)
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
        for pipeline in (jpegamd.PIPELINE_PAIR, jpegamd.PIPELINE_STITCH):
            assert _encode_batch(jpegamd, bmps[:1], q, dev, pipeline)[0] == want[0], (w, h, kind, flags, q, pipeline, "single")
            assert _encode_batch(jpegamd, bmps, q, dev, pipeline) == want, (w, h, kind, flags, q, pipeline, "batch of two")


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
        got = _encode_batch(jpegamd, bmps, q if q != 50 else 0, dev, jpegamd.PIPELINE_STITCH if i % 2 else jpegamd.PIPELINE_PAIR)
        for j in range(nb):
            assert got[j] == oracle.encode_bmp(bmps[j], q), (i, w, h, nb, j, kind, flags, q)


def test_single_pass_stitch_on_every_kind_of_content(jpegamd, oracle, dev):
    """k_stitch (JPEGAMD_PIPELINE_STITCH) where AUTO would take the pair: photo-like, noise (segments longer than the kernel's bit
    window: several parts, counted then written), flat (6-bit segments: bytes straddling several segments), gradients; Q = 10, 50, 90,
    100 (0xFF counts that saturate the hand-off granule); one to many workgroups; the segment only (no container)."""
    cases = [(8, 8, 2, 0), (16, 8, 2, 0), (1, 1, 1, 0), (264, 1200, 2, 0), (4104, 8, 1, 0), (2048, 2048, 1, 100), (2048, 1024, 1, 90), (3000, 2000, 0, 50),
             (3000, 2000, 0, 10), (1920, 1080, 3, 50), (520, 16, 1, 100), (8200, 520, 1, 50), (8200, 520, 0, 95), (6000, 4000, 0, 50)]
    for (w, h, kind, q) in cases:
        for flags in (0, 3):
            bmp = jpegamd.synth_bmp(w, h, 300 + w + kind, kind, flags)
            got = _encode_batch(jpegamd, [bmp], q if q != 50 else 0, dev, jpegamd.PIPELINE_STITCH)[0]
            assert got == oracle.encode_bmp(bmp, q if q else 50), (w, h, kind, q, flags)
    # the segment without container, and an output buffer that is too small: the would-be size and status -8
    bmp = jpegamd.synth_bmp(1500, 700, 5, 1, 0)
    want = oracle.encode_bmp(bmp)
    img, off = jpegamd.parse_bmp(bmp)
    px = torch.frombuffer(bytearray(bmp[off:off + img.row_stride * img.height]), dtype=torch.uint8).to(dev)
    enc = jpegamd.Encoder(1500, 700)
    enc.set_pipeline(jpegamd.PIPELINE_STITCH)
    d = jpegamd.Encoder.image(px.data_ptr(), 1500, 700, img.row_stride, True, jpegamd.ORDER_BGR, 0)
    for cap, container in ((len(want), False), (len(want), True), (len(want) // 2, True)):
        out = torch.zeros(len(want) + 64, dtype=torch.uint8, device=dev)
        size = torch.zeros(1, dtype=torch.int64, device=dev)
        enc.encode_async(d, out.data_ptr(), cap, size.data_ptr(), container, torch.cuda.current_stream().cuda_stream)
        if cap < len(want):
            with pytest.raises(jpegamd.JpegAmdError) as err:
                enc.finish()
            assert err.value.code == -8 and int(size.item()) == len(want)
            assert bytes(out[cap:].cpu().numpy()) == bytes(len(want) + 64 - cap)          # nothing written beyond the capacity
        else:
            enc.finish()
            n = int(size.item())
            assert bytes(out[:n].cpu().numpy()) == (want if container else want[328:-2])


def test_auto_takes_the_single_pass_for_very_large_pictures(jpegamd, oracle, dev):
    """16 384 segments and more (2056 x 65535: two segments per block row, 8 192 block rows): JPEGAMD_PIPELINE_AUTO codes the picture
    with k_stitch -- many look-back rounds -- and the pair gives the same bytes (its scan then reads 2 048 predecessors per workgroup)."""
    w, h = 2056, 65535
    bmp = jpegamd.synth_bmp(w, h, 11, 0, 0)
    want = oracle.encode_bmp(bmp)
    assert _encode_batch(jpegamd, [bmp], 0, dev) == [want]
    assert _encode_batch(jpegamd, [bmp], 0, dev, jpegamd.PIPELINE_PAIR) == [want]


def test_c_level_gather_of_streams_over_rccl(jpegamd, oracle, dev):
    """jpegamd_gather_streams (the multi-GPU exchange as a C entry) with a one-rank RCCL communicator made through ctypes: five
    images encoded straight into staging records, one of them too small for its stream; the size table comes back in full (the
    cut stream's would-be size included), the streams that fit land densely, 8-byte aligned, in record order."""
    import ctypes as C
    try:
        rccl = C.CDLL("librccl.so")
    except OSError:
        rccl = C.CDLL("/opt/rocm/lib/librccl.so")

    class UniqueId(C.Structure):
        _fields_ = [("internal", C.c_char * 128)]

    uid, comm = UniqueId(), C.c_void_p()
    rccl.ncclGetUniqueId.argtypes = [C.POINTER(UniqueId)]
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    rccl.ncclCommDestroy.argtypes = [C.c_void_p]
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
    try:
        w, h, slots = 640, 360, 5
        bmps = [jpegamd.synth_bmp(w, h, 60 + k, 1 if k == 3 else 0, 0) for k in range(slots)]       # record 3: noise, several times the others
        want = [oracle.encode_bmp(b) for b in bmps]
        slot_bytes = (max(len(x) for k, x in enumerate(want) if k != 3) + 64 + 7) // 8 * 8 + 8
        assert len(want[3]) > slot_bytes - 8
        records = torch.zeros(slots * slot_bytes, dtype=torch.uint8, device=dev)
        enc = jpegamd.Encoder(w, h)
        keep = []
        for k, b in enumerate(bmps):
            img, off = jpegamd.parse_bmp(b)
            px = torch.frombuffer(bytearray(b[off:off + img.row_stride * img.height]), dtype=torch.uint8).to(dev)
            keep.append(px)
            d = jpegamd.Encoder.image(px.data_ptr(), w, h, img.row_stride, True, jpegamd.ORDER_BGR, 0)
            base = records.data_ptr() + k * slot_bytes
            enc.encode_async(d, base, slot_bytes - 8, base + slot_bytes - 8, True, torch.cuda.current_stream().cuda_stream)
        with pytest.raises(jpegamd.JpegAmdError) as err:
            enc.finish()
        assert err.value.code == -8                                        # record 3 did not fit
        sizes = (C.c_uint64 * slots)()
        stride = slots * slot_bytes
        recv = torch.zeros(stride, dtype=torch.uint8, device=dev)
        rc = jpegamd.lib.jpegamd_gather_streams(comm, 0, 1, 0, records.data_ptr(), slot_bytes, slots, C.addressof(sizes), recv.data_ptr(), stride,
                                                torch.cuda.current_stream().cuda_stream)
        assert rc == 0
        assert list(sizes) == [len(x) for x in want]
        got, off = recv.cpu().numpy().tobytes(), 0
        for k in range(slots):
            if k == 3:
                continue                                                   # cut by the encoder: its owner sends it by itself, at sizes[3] bytes
            assert got[off:off + len(want[k])] == want[k], k
            off += (len(want[k]) + 7) // 8 * 8
        assert jpegamd.lib.jpegamd_gather_streams(comm, 0, 1, 0, records.data_ptr(), slot_bytes, slots, C.addressof(sizes), recv.data_ptr(), 64,
                                                  torch.cuda.current_stream().cuda_stream) == -8      # a receive area that is too small
        assert jpegamd.lib.jpegamd_gather_streams(None, 0, 1, 0, records.data_ptr(), slot_bytes, slots, C.addressof(sizes), recv.data_ptr(), stride, None) == -1
    finally:
        rccl.ncclCommDestroy(comm)


def _gray_bmp(y: np.ndarray) -> bytes:
    """24-bit bottom-up BMP whose three channels all equal `y` (uint8 [H, W]): the luma weights sum to 256, so Y == y exactly."""
    h, w = y.shape
    stride = (3 * w + 3) & ~3
    rows = np.zeros((h, stride), np.uint8)
    rows[:, :3 * w] = np.repeat(y[::-1], 3, axis=1)
    head = b"BM" + (54 + stride * h).to_bytes(4, "little") + bytes(4) + (54).to_bytes(4, "little")
    info = (40).to_bytes(4, "little") + w.to_bytes(4, "little") + h.to_bytes(4, "little") + (1).to_bytes(2, "little") + (24).to_bytes(2, "little") + bytes(24)
    return head + info + rows.tobytes()


@pytest.mark.gpu
def test_extreme_blocks_through_the_subnormal_matrix_operand(jpegamd, oracle, dev):
    """The transform's B operand is the uncentred luma 0 .. 255 as binary16 subnormals: its exactness rests on every
    partial sum of a chain staying below 2^24 units whatever order the matrix pipe adds in.  The pictures that push the sums furthest --
    for each of the 64 basis functions the block that is 255 where the function is positive and 0 elsewhere, its complement, all-white,
    all-black, checkerboards, single bright / dark pixels -- go through the kernel at the finest, the default and a coarse quantiser
    (Q=100: every step is 1, the largest coefficient magnitudes) and must come out byte-identical to the oracle; the stage taps pin
    the quantised coefficients themselves."""
    lut = jpegamd.cos_lut().astype(np.float64)
    blocks = []
    for k in range(64):
        u, v = divmod(k, 8)
        pos = np.outer(lut[:, u], lut[:, v]) > 0
        blocks += [np.where(pos, 255, 0), np.where(pos, 0, 255), np.where(pos, 255, 1), np.where(pos, 128, 127)]
    blocks += [np.full((8, 8), 255), np.zeros((8, 8), int), (np.indices((8, 8)).sum(0) % 2) * 255, (np.indices((8, 8))[0] % 2) * 255]
    for p in ((0, 0), (7, 7), (3, 4)):
        b = np.zeros((8, 8), int); b[p] = 255; blocks.append(b)
        b = np.full((8, 8), 255); b[p] = 0; blocks.append(b)
    rng = np.random.default_rng(4)
    blocks += [rng.choice([0, 255], size=(8, 8)) for _ in range(58)]
    nb = len(blocks)
    per_row = 40                                                   # 320 pixels: a full tile and a ragged one per block row
    rows = (nb + per_row - 1) // per_row
    y = np.zeros((rows * 8, per_row * 8), np.uint8)
    for i, b in enumerate(blocks):
        y[(i // per_row) * 8:(i // per_row) * 8 + 8, (i % per_row) * 8:(i % per_row) * 8 + 8] = b
    bmp = _gray_bmp(y)
    enc = jpegamd.Encoder(y.shape[1], y.shape[0])
    for q in (100, 50, 10):
        got = _encode_batch(jpegamd, [bmp], q, dev)[0]
        assert got == oracle.encode_bmp(bmp, q), q
    st = oracle.stages(bmp)
    img, off = jpegamd.parse_bmp(bmp)
    px = torch.frombuffer(bytearray(bmp[off:off + img.row_stride * img.height]), dtype=torch.uint8).to(dev)
    n = st["zigzag"].shape[0]
    yt = torch.zeros(n * 64, dtype=torch.int8, device=dev)
    zz = torch.zeros(n * 64, dtype=torch.int16, device=dev)
    mask = torch.zeros(n, dtype=torch.int64, device=dev)
    d = jpegamd.Encoder.image(px.data_ptr(), img.width, img.height, img.row_stride, bool(img.bottom_up))
    enc.debug_stages(d, yt.data_ptr(), zz.data_ptr(), mask.data_ptr())
    assert np.array_equal(zz.cpu().numpy().reshape(n, 64), st["zigzag"])
