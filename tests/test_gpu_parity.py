"""Parity tests proper: the HIP path, called through the C-ABI, against the oracle and the committed goldens.

Bar: bit-exact (integer / byte work; the one float stage, the DCT, is bit-exact as well because the
reference's float32 evaluation order is reproduced wherever a rounding decision depends on it).
Nothing here reads /root/reference."""
from __future__ import annotations

import ctypes
import hashlib
import json
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, PKG, ROOT, fixture_bmp, golden_jpg

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test started without a GPU: the product path has no CPU fallback")
    return torch.device("cuda:0")


def upload_pixels(bmp: bytes, jpegamd, dev):
    img, off = jpegamd.parse_bmp(bmp)
    n = img.row_stride * img.height
    t = torch.frombuffer(bytearray(bmp[off:off + n]), dtype=torch.uint8).to(dev)
    return img, t


def device_encode(jpegamd, enc, bmp, dev, quality=0, container=True, cap=None):
    img, px = upload_pixels(bmp, jpegamd, dev)
    cap = cap or (4096 + 2 * img.width * img.height)
    out = torch.empty(cap, dtype=torch.uint8, device=dev)
    size = torch.zeros(1, dtype=torch.int64, device=dev)
    d = jpegamd.Encoder.image(px.data_ptr(), img.width, img.height, img.row_stride, bool(img.bottom_up), jpegamd.ORDER_BGR, quality)
    enc.encode_async(d, out.data_ptr(), cap, size.data_ptr(), container, torch.cuda.current_stream().cuda_stream)
    st = enc.finish()
    n = int(size.item())
    assert n == st.jfif_bytes
    return bytes(out[:n].cpu().numpy()), st


def test_goldens_through_c_abi(jpegamd, manifest, dev):
    """Every committed fixture (flat/tie, gradient, noise, W%8 != 0, top-down, bfOffBits=138, 1x1, crops of
    the reference's sample images): device JFIF == bytes written by the compiled reference."""
    for e in manifest:
        got = jpegamd.encode_bmp_bytes(fixture_bmp(e, jpegamd))
        assert got == golden_jpg(e), e["name"]


def test_random_shapes_against_oracle(jpegamd, oracle, dev):
    rng = np.random.default_rng(2026)
    enc = jpegamd.Encoder(1400, 900)
    for i in range(40):
        w, h = int(rng.integers(1, 1400)), int(rng.integers(1, 900))
        kind, flags = int(rng.choice([0, 0, 1, 2, 3])), int(rng.integers(0, 4))
        bmp = jpegamd.synth_bmp(w, h, 500 + i, kind, flags)
        got, _ = device_encode(jpegamd, enc, bmp, dev)
        assert got == oracle.encode_bmp(bmp), (w, h, kind, flags)


def test_stage_taps_against_oracle_stages(jpegamd, oracle, dev):
    """Per-stage parity: centred luma (converter.c), quantised zigzag coefficients (dct.c + quantization.c +
    zigzag.c).  Also: the exact-path mask only ever marks coefficients whose value was recomputed."""
    enc = jpegamd.Encoder(640, 480)
    for (w, h, seed, kind, flags) in [(203, 117, 5, 1, 0), (512, 256, 11, 0, 0), (333, 250, 4, 0, 1), (64, 64, 101, 2, 0)]:
        bmp = jpegamd.synth_bmp(w, h, seed, kind, flags)
        st = oracle.stages(bmp)
        img, px = upload_pixels(bmp, jpegamd, dev)
        nb = st["zigzag"].shape[0]
        y = torch.zeros(nb * 64, dtype=torch.int8, device=dev)
        zz = torch.zeros(nb * 64, dtype=torch.int16, device=dev)
        mask = torch.zeros(nb, dtype=torch.int64, device=dev)
        d = jpegamd.Encoder.image(px.data_ptr(), img.width, img.height, img.row_stride, bool(img.bottom_up))
        enc.debug_stages(d, y.data_ptr(), zz.data_ptr(), mask.data_ptr())
        ph, pw = st["y"].shape
        y_blocks = st["y"].reshape(ph // 8, 8, pw // 8, 8).transpose(0, 2, 1, 3).reshape(nb, 64)
        assert np.array_equal(y.cpu().numpy().reshape(nb, 64), y_blocks), "luma / level shift"
        assert np.array_equal(zz.cpu().numpy().reshape(nb, 64), st["zigzag"]), "quantised zigzag coefficients"
        assert int((mask.cpu().numpy() & 1).sum()) == 0       # DC never needs the fallback (bit 0 = raster k = 0)


def test_exact_order_dct_is_bit_identical(jpegamd, oracle, dev):
    """dct.c:63-96 in float32: device == oracle bit for bit, on random, extreme and tie-prone blocks."""
    rng = np.random.default_rng(5)
    blocks = np.concatenate([
        rng.integers(-128, 128, size=(2000, 8, 8)),
        np.full((1, 8, 8), -128), np.full((1, 8, 8), 127), np.full((1, 8, 8), -27),
        ((np.indices((8, 8)).sum(0) % 2) * 255 - 128)[None],
        rng.integers(-3, 4, size=(500, 8, 8)),
    ]).astype(np.int8)
    enc = jpegamd.Encoder(64, 64)
    b = torch.from_numpy(blocks.reshape(-1)).to(dev)
    c = torch.zeros(blocks.shape[0] * 64, dtype=torch.float32, device=dev)
    enc.debug_dct_exact(b.data_ptr(), c.data_ptr(), blocks.shape[0])
    got = c.cpu().numpy().reshape(-1, 8, 8)
    exp = oracle.dct_blocks(blocks)
    assert np.array_equal(got.view(np.uint32), exp.view(np.uint32))


def test_forced_exact_path_ties(jpegamd, oracle, dev):
    """Inputs built so that many coefficients sit on or next to a rounding tie: flat odd greys (DC ties,
    SURVEY.md H1c) and low-amplitude patterns; the fallback must fire and the bytes must still match."""
    enc = jpegamd.Encoder(512, 512)
    total_exact = 0
    for level in (1, 27, 101, 129, 255):
        bmp = jpegamd.synth_bmp(256, 64, level, 2, 0)
        got, _ = device_encode(jpegamd, enc, bmp, dev)
        assert got == oracle.encode_bmp(bmp)
    for seed in range(6):
        bmp = jpegamd.synth_bmp(512, 512, 900 + seed, 1 if seed % 2 else 0, 0)
        got, st = device_encode(jpegamd, enc, bmp, dev)
        total_exact += st.exact_fallbacks
        assert got == oracle.encode_bmp(bmp)
    assert total_exact > 0, "the exact-order fallback never ran; this test must exercise it"


def test_quality_extension(jpegamd, oracle, dev):
    q = json.loads((GOLDEN / "quality.json").read_text())
    enc = jpegamd.Encoder(512, 512)
    for name, e in q.items():
        bmp = jpegamd.synth_bmp(e["width"], e["height"], e["seed"], e["kind"], e["flags"])
        got, _ = device_encode(jpegamd, enc, bmp, dev, quality=e["quality"])
        assert (len(got), hashlib.sha256(got).hexdigest()) == (e["jpg_size"], e["jpg_sha256"]), name
    for quality in (1, 25, 75, 100):                       # table-patched goldens exist only for 10/90: oracle here
        bmp = jpegamd.synth_bmp(203, 117, 5, 1, 0)
        got, _ = device_encode(jpegamd, enc, bmp, dev, quality=quality)
        assert got == oracle.encode_bmp(bmp, quality), quality


def test_segment_only_and_capacity(jpegamd, oracle, dev):
    enc = jpegamd.Encoder(512, 512)
    bmp = jpegamd.synth_bmp(203, 117, 5, 1, 0)
    full = oracle.encode_bmp(bmp)
    seg, st = device_encode(jpegamd, enc, bmp, dev, container=False)
    assert seg == full[328:-2]                             # what the reference's accelerator hands back
    assert st.stuffed_bytes == seg.count(b"\xff\x00")
    with pytest.raises(jpegamd.JpegAmdError) as ei:        # reference: -8 when the Huffman buffer is too small
        device_encode(jpegamd, enc, bmp, dev, cap=1000)
    assert ei.value.code == -8


def test_dto_boundary(jpegamd, oracle, dev):
    """JpegCompression_Init / convertToJpeg(DTO): caller-owned buffers, callee fills sizes, counters and the
    first-block taps (dsp_port/jpeg_compression/src/jpeg_compression.c:35-216)."""
    lib = jpegamd.lib
    assert lib.JpegCompression_Init() == 0 and lib.JpegCompression_Init() == 0      # idempotent
    bmp = jpegamd.synth_bmp(333, 250, 4, 0, 0)
    st = oracle.stages(bmp)
    img, px = upload_pixels(bmp, jpegamd, dev)
    cap = 333 * 250
    huff = torch.zeros(cap, dtype=torch.uint8, device=dev)
    y = (ctypes.c_int8 * 64)()
    dct = (ctypes.c_float * 64)()
    quant = (ctypes.c_int16 * 64)()
    zz = (ctypes.c_int16 * 64)()
    dto = jpegamd.DTO(width=333, height=250, r_phy_ptr=px.data_ptr(), huff_phy_ptr=huff.data_ptr(), huff_size=cap,
                      y_phy_ptr=ctypes.addressof(y), dct_phy_ptr=ctypes.addressof(dct), quant_phy_ptr=ctypes.addressof(quant),
                      zigzag_phy_ptr=ctypes.addressof(zz), row_stride=img.row_stride, bottom_up=img.bottom_up,
                      channel_order=jpegamd.ORDER_BGR, quality=0)
    assert lib.convertToJpeg(ctypes.byref(dto)) == 0
    full = oracle.encode_bmp(bmp)
    assert bytes(huff[:dto.huff_size].cpu().numpy()) == full[328:-2]
    assert dto.rle_count == len(oracle.rle_symbols(st["zigzag"]))
    # the six stage counters of the reference's DTO (jpeg_compression.c:188-210), from the stamped variant of the fused kernel: every
    # stage that exists as instructions has time (zigzag is the row order of the matrix operand), and they add up to the kernels' time
    stages = [dto.cycles_color_conversion, dto.cycles_dct, dto.cycles_quantization, dto.cycles_rle, dto.cycles_huffman]
    assert all(c > 0 for c in stages) and dto.cycles_zigzag == 0 and dto.cycles_total > 0
    assert sum(stages) <= dto.cycles_total                          # (first kernel's begin .. last kernel's end: the launch gaps are in the total only)
    assert list(y) == list(st["y"][:8, :8].reshape(-1))
    assert list(quant) == list(st["quant"][:8, :8].reshape(-1)) and list(zz) == list(st["zigzag"][0])
    assert np.array_equal(np.array(dct, np.float32).view(np.uint32), st["dct"][:8, :8].reshape(-1).view(np.uint32))
    dto.huff_size = 100
    assert lib.convertToJpeg(ctypes.byref(dto)) == -8
    assert lib.JpegCompression_RemoteServiceHandler(b"com.etfbl.sdos.jpeg_compression", 0, None, 0, 0) == -1
    assert lib.JpegCompression_DeInit() == 0
    assert lib.convertToJpeg(ctypes.byref(dto)) == -4      # not initialised any more


def test_natural_c_surface_and_cli(jpegamd, oracle, dev, tmp_path):
    """loadBMPImage + saveJPEGGrayscale (library) and jpeg_compression_app (CLI) write the reference's bytes."""
    bmp = jpegamd.synth_bmp(333, 250, 4, 0, 1)
    src, out1, out2 = tmp_path / "in.bmp", tmp_path / "lib.jpg", tmp_path / "cli.jpg"
    src.write_bytes(bmp)
    img = jpegamd.lib.loadBMPImage(str(src).encode())
    assert img
    assert jpegamd.lib.saveJPEGGrayscale(str(out1).encode(), img) is True
    jpegamd.lib.freeBMPImage(img)
    exp = oracle.encode_bmp(bmp)
    assert out1.read_bytes() == exp
    r = subprocess.run([str(PKG / "jpeg_compression_app"), str(src), str(out2)], capture_output=True, text=True)
    assert r.returncode == 0 and out2.read_bytes() == exp
    assert "Starting processing..." in r.stdout and "Natural C quant (First Block):" in r.stdout
    assert "Compression successful. File saved:" in r.stdout and r.stdout.rstrip().endswith("Save is sucesfull")
    assert jpegamd.lib.saveJPEGGrayscale(str(tmp_path / "no_such_dir" / "x.jpg").encode(), None) is False


def test_large_configs_against_reference_hashes(jpegamd, dev):
    """BASELINE.json configs at full size: sha256 of the device output == sha256 of what the compiled
    reference wrote for the same deterministic input (tests/golden/large.json)."""
    large = json.loads((GOLDEN / "large.json").read_text())
    enc = jpegamd.Encoder(8192, 8192)
    assert {"8192x8192_seed1000_kind0_q10", "8192x8192_seed1000_kind0_q90", "8192x8192_seed1000_kind1_q50"} <= set(large)
    for key, e in large.items():                       # configs[1], [2], [4] (Q = 10 / 50 / 90 on 8192^2) and the noise stress
        dims, seed, kind, q = key.split("_")
        w, h = (int(x) for x in dims.split("x"))
        if w == 8192 and int(seed[4:]) > 1002:         # (the other bench seeds go through the batched launch, below)
            continue
        bmp = jpegamd.synth_bmp(w, h, int(seed[4:]), int(kind[4:]), 0)
        assert hashlib.sha256(bmp).hexdigest() == e["bmp_sha256"]
        got, st = device_encode(jpegamd, enc, bmp, dev, quality=int(q[1:]), cap=4096 + 2 * w * h)
        assert (len(got), hashlib.sha256(got).hexdigest()) == (e["size"], e["sha256"]), key
        # size-independent properties of the stream
        assert got[:2] == b"\xff\xd8" and got[-2:] == b"\xff\xd9"
        body = got[328:-2]
        assert b"\xff" not in body.replace(b"\xff\x00", b""), "an unstuffed 0xFF inside the entropy-coded segment"
        assert st.entropy_bits > 0 and (st.entropy_bits + 7) // 8 + st.stuffed_bytes == len(body)


def test_batched_launch_at_full_size_against_reference_hashes(jpegamd, dev):
    """The entry point bench.py times, at the sizes it times: jpegamd_encode_batch_async with EIGHT images through ONE launch of
    each kernel.  The 16 rotating 8192^2 bench inputs (seeds 1000..1015, two launches at Q=50), the first eight of them at Q=10 and
    at Q=90 (one launch each; BASELINE configs[4]), and eight 4096^2 images of configs[3] (seeds 2000..2007) in one launch: every
    output hashed against what the compiled reference (natural_c's own flags; table-patched for Q != 50) wrote for that image
    alone -- the DC chain (rle.c:59-70), the bit offsets and the stuffing (huffman.c:26-81) restart with every image."""
    large = json.loads((GOLDEN / "large.json").read_text())
    batch = json.loads((GOLDEN / "batch4096.json").read_text())
    stream = torch.cuda.current_stream().cuda_stream

    def run(enc, w, h, seeds, quality, answers, px):
        n = len(seeds)
        cap = 4096 + w * h // 2 if quality <= 50 else 4096 + 2 * w * h
        outs = [torch.empty(cap, dtype=torch.uint8, device=dev) for _ in range(n)]
        sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(n)]
        stride = (3 * w + 3) & ~3
        imgs = [jpegamd.Encoder.image(px[s].data_ptr(), w, h, stride, True, jpegamd.ORDER_BGR, quality) for s in seeds]
        enc.encode_batch_async(imgs, [o.data_ptr() for o in outs], cap, [s.data_ptr() for s in sizes], True, stream)
        enc.finish()
        for i, seed in enumerate(seeds):
            e = answers[f"{w}x{h}_seed{seed}_kind0_q{quality}"]
            got = bytes(outs[i][:int(sizes[i].item())].cpu().numpy())
            assert (len(got), hashlib.sha256(got).hexdigest()) == (e["size"], e["sha256"]), (w, h, seed, quality, i)

    def pixels(w, h, seeds, answers):
        px = {}
        for seed in seeds:
            bmp = jpegamd.synth_bmp(w, h, seed, 0, 0)
            assert hashlib.sha256(bmp).hexdigest() == answers[f"{w}x{h}_seed{seed}_kind0_q50"]["bmp_sha256"]
            px[seed] = upload_pixels(bmp, jpegamd, dev)[1]
        return px

    enc = jpegamd.Encoder(8192, 8 * 8192)
    px = pixels(8192, 8192, range(1000, 1016), large)
    run(enc, 8192, 8192, list(range(1000, 1008)), 50, large, px)
    run(enc, 8192, 8192, list(range(1008, 1016)), 50, large, px)
    run(enc, 8192, 8192, list(range(1000, 1008)), 10, large, px)
    run(enc, 8192, 8192, list(range(1000, 1008)), 90, large, px)
    del px
    # configs[3]'s images: eight per launch, and JPEGAMD_MAX_BATCH (32) per launch -- as much work as a launch of eight 8192^2
    enc = jpegamd.Encoder(4096, 32 * 4096)
    px = pixels(4096, 4096, range(2000, 2032), batch)
    run(enc, 4096, 4096, list(range(2000, 2008)), 50, batch, px)
    run(enc, 4096, 4096, list(range(2000, 2032)), 50, batch, px)


def test_row_stride_of_16_mib_and_more(jpegamd, oracle, dev):
    """A region of interest inside a very wide buffer: row_stride >= 2^24 does not fit the dword loader's 24-bit multiply and
    must take the byte loader (64-bit addresses) instead of silently reading wrong rows."""
    w, h, stride = 200, 24, (1 << 24) + 64
    bmp = jpegamd.synth_bmp(w, h, 31, 0, 1)                       # top-down, so that stored row r is image row r
    img, off = jpegamd.parse_bmp(bmp)
    want = oracle.encode_bmp(bmp)
    wide = torch.zeros(stride * h, dtype=torch.uint8, device=dev)
    rows = torch.frombuffer(bytearray(bmp[off:off + img.row_stride * h]), dtype=torch.uint8).to(dev).view(h, img.row_stride)
    wide.view(h, stride)[:, :img.row_stride] = rows
    enc = jpegamd.Encoder(w, h)
    cap = 1 << 20
    out = torch.empty(cap, dtype=torch.uint8, device=dev); size = torch.zeros(1, dtype=torch.int64, device=dev)
    d = jpegamd.Encoder.image(wide.data_ptr(), w, h, stride, False, jpegamd.ORDER_BGR, 0)
    enc.encode_async(d, out.data_ptr(), cap, size.data_ptr(), True, torch.cuda.current_stream().cuda_stream)
    enc.finish()
    assert bytes(out[:int(size.item())].cpu().numpy()) == want


def test_corrupt_tile_record_ends_in_a_status_code(jpegamd, oracle, dev):
    """A tile record that claims more bits than 32 blocks can hold (stale or corrupt scratch) must not become a device fault:
    k_segment_merge trusts a record only up to the worst case, ORs a status bit, and jpegamd_encoder_finish returns
    JPEGAMD_ERR_RLE_CAPACITY (-6, dsp_port/jpeg_compression/src/jpeg_compression.c:180-181).  The next encode is clean again."""
    import ctypes as C
    bmp = jpegamd.synth_bmp(640, 360, 21, 0, 0)
    want = oracle.encode_bmp(bmp)
    enc = jpegamd.Encoder(640, 360)
    got, _ = device_encode(jpegamd, enc, bmp, dev)
    assert got == want
    fn = jpegamd.lib.jpegamd_debug_poison_tile_record
    fn.restype = C.c_int32
    fn.argtypes = [C.c_void_p, C.c_int32, C.c_uint32]
    for tile, value in ((5, 0xFFFFFFFF), (0, 1 << 20), (134, 32 * 1721 + 1)):      # 640x360: 3 tiles per row, 45 rows
        assert fn(enc._h, tile, value) == 0
        with pytest.raises(jpegamd.JpegAmdError) as err:
            device_encode(jpegamd, enc, bmp, dev, cap=1 << 22)
        assert err.value.code == -6
        got, _ = device_encode(jpegamd, enc, bmp, dev)           # the status was cleared; the scratch is simply rewritten
        assert got == want
    assert fn(enc._h, 10 ** 6, 0) != 0


def test_reference_sample_images(jpegamd, dev):
    """The reference's own assets/input/*.bmp (512x512 lena / blackbuck = BASELINE configs[0]'s input, greenland 762x1309,
    offset_sample with bfOffBits = 138) through the HIP path: sha256 == what the compiled reference wrote."""
    answers = json.loads((GOLDEN / "assets.json").read_text())
    assert len(answers) == 4
    for name, e in answers.items():
        bmp = (GOLDEN / "assets" / name).read_bytes()
        assert hashlib.sha256(bmp).hexdigest() == e["bmp_sha256"], name
        got = jpegamd.encode_bmp_bytes(bmp)
        assert (len(got), hashlib.sha256(got).hexdigest()) == (e["jpg_size"], e["jpg_sha256"]), name
    assert answers["lena.bmp"]["jpg_sha256"].startswith("95cf58fe")


@pytest.mark.gpu
def test_profiling_ring_reports_the_kernels_own_durations(jpegamd, dev):
    """jpegamd_encoder_set_profiling: every kernel carries its own begin / end events, so the durations are positive, their sum is
    below the first-begin-to-last-end span (launch gaps), and a 4x larger image takes longer.  With the single-pass pipeline
    (k_tile_encode, k_stitch) the merge figure stays 0 and k_stitch's duration is reported as ns_pack."""
    res = {}
    for (w, h, pipeline) in ((1024, 1024, jpegamd.PIPELINE_PAIR), (2048, 2048, jpegamd.PIPELINE_PAIR), (2048, 2048, jpegamd.PIPELINE_STITCH)):
        enc = jpegamd.Encoder(w, h)
        enc.set_pipeline(pipeline)
        bmp = jpegamd.synth_bmp(w, h, 3, 0, 0)
        img, px = upload_pixels(bmp, jpegamd, dev)
        cap = 4096 + w * h
        out = torch.empty(cap, dtype=torch.uint8, device=dev)
        size = torch.zeros(1, dtype=torch.int64, device=dev)
        d = jpegamd.Encoder.image(px.data_ptr(), w, h, img.row_stride, True)
        enc.set_profiling(8)
        for _ in range(8):
            enc.encode_async(d, out.data_ptr(), cap, size.data_ptr(), True, torch.cuda.current_stream().cuda_stream)
        enc.finish()
        prof = [enc.profile(s) for s in range(2, 8)]
        for p in prof:
            assert p.ns_transform > 0 and p.ns_pack > 0 and (p.ns_entropy > 0) == (pipeline == jpegamd.PIPELINE_PAIR)
            assert p.ns_transform + p.ns_entropy + p.ns_pack <= p.ns_total
            assert p.ns_total < 5_000_000
        res[(w, pipeline)] = sum(p.ns_transform for p in prof) / len(prof)
    assert res[(2048, jpegamd.PIPELINE_PAIR)] > res[(1024, jpegamd.PIPELINE_PAIR)]


@pytest.mark.gpu
def test_batched_launch_matches_the_oracle_image_by_image(jpegamd, oracle, dev):
    """jpegamd_encode_batch_async: several images of one geometry through ONE launch of each kernel.  DC prediction
    (rle.c:59-70), bit offsets and 0xFF stuffing (huffman.c:26-62) must restart with every image: each output equals the
    oracle's file for that image alone.  Shapes: ragged widths (edge tiles, a short last segment), one tile per image,
    several segments per row, noise (stuffing, dense lists on the direct path), Q != 50, segment only."""
    rng = np.random.default_rng(77)
    cases = [(8, 8, 3, 0, 0), (203, 117, 4, 1, 0), (512, 256, 8, 0, 1), (333, 250, 5, 2, 3), (2056, 72, 6, 0, 0), (1024, 1024, 8, 0, 0),
             (640, 480, 2, 3, 2)]
    for (w, h, n, kind, flags) in cases:
        for quality, container in ((0, True), (90, True), (10, False)) if w in (203, 512) else ((0, True),):
            enc = jpegamd.Encoder(w, n * ((h + 7) // 8) * 8 + 8)
            bmps = [jpegamd.synth_bmp(w, h, 900 + 17 * i + w, kind, flags) for i in range(n)]
            ups = [upload_pixels(b, jpegamd, dev) for b in bmps]
            cap = 4096 + 2 * w * h
            outs = [torch.empty(cap, dtype=torch.uint8, device=dev) for _ in range(n)]
            sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(n)]
            imgs = [jpegamd.Encoder.image(px.data_ptr(), im.width, im.height, im.row_stride, bool(im.bottom_up), jpegamd.ORDER_BGR, quality)
                    for im, px in ups]
            enc.encode_batch_async(imgs, [o.data_ptr() for o in outs], cap, [s.data_ptr() for s in sizes], container,
                                   torch.cuda.current_stream().cuda_stream)
            st = enc.finish()
            for i in range(n):
                got = bytes(outs[i][:int(sizes[i].item())].cpu().numpy())
                want = oracle.encode_bmp(bmps[i], quality=quality) if quality else oracle.encode_bmp(bmps[i])
                if not container:
                    want = want[328:-2]
                assert got == want, (w, h, n, kind, flags, quality, container, i)
            assert st.jfif_bytes == int(sizes[-1].item())
    # a context that is too small for the batch refuses it
    enc = jpegamd.Encoder(512, 512)
    bmp = jpegamd.synth_bmp(512, 512, 1, 0, 0)
    im, px = upload_pixels(bmp, jpegamd, dev)
    d = jpegamd.Encoder.image(px.data_ptr(), 512, 512, im.row_stride, True)
    out = torch.empty(1 << 20, dtype=torch.uint8, device=dev); size = torch.zeros(1, dtype=torch.int64, device=dev)
    with pytest.raises(jpegamd.JpegAmdError):
        enc.encode_batch_async([d, d], [out.data_ptr()] * 2, 1 << 20, [size.data_ptr()] * 2)



def test_batch_of_64_4096_through_the_gather_path(jpegamd, dev):
    """BASELINE configs[3] at its stated shape on one GPU: 64 distinct 4096x4096 images encoded straight into the staging
    records of jpegamd.sharding.ExactStreamGather (a one-rank RCCL group: a rank's own streams never move, the code path is the
    N > 1 one of bench.py); one record is deliberately too small for its stream, which its owner then encodes again at the exact
    size; every collected stream is hashed against the compiled reference's answer."""
    import os
    import torch.distributed as dist
    from jpegamd.sharding import ExactStreamGather
    batch = json.loads((GOLDEN / "batch4096.json").read_text())
    seeds = sorted(int(k.split("_")[1][4:]) for k in batch)
    assert len(seeds) == 64
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    own_group = not dist.is_initialized()
    if own_group:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        w = h = 4096
        G, nstreams = 32, 2
        sizes = sorted(e["size"] for e in batch.values())
        slot_bytes = ((sizes[-2] + 255) // 256) * 256 + 8                       # the second largest stream fits its record, the largest does not
        assert sizes[-1] > slot_bytes - 8
        encs = [jpegamd.Encoder(w, h) for _ in range(nstreams)]
        streams = [torch.cuda.Stream() for _ in range(nstreams)]
        stride = 3 * w
        keep, got, again = [], {}, []

        def reencode(step, payload, size):
            again.append(step)
            d = jpegamd.Encoder.image(keep[step % G].data_ptr(), w, h, stride, True, jpegamd.ORDER_BGR, 0)
            encs[0].encode_async(d, payload.data_ptr(), payload.numel(), size.data_ptr(), True, torch.cuda.current_stream().cuda_stream)
            try:
                encs[0].finish()
            except jpegamd.JpegAmdError as err:                                # (the capacity status is sticky: the FIRST encode of this image set it)
                assert err.code == -8
            assert 0 < int(size.item()) <= payload.numel()

        gather = ExactStreamGather(slot_bytes, G, dev, dst=0, depth=3, reencode=reencode)
        for i, seed in enumerate(seeds):
            bmp = jpegamd.synth_bmp(w, h, seed, 0, 0)
            assert hashlib.sha256(bmp).hexdigest() == batch[f"4096x4096_seed{seed}_kind0_q50"]["bmp_sha256"]
            px = torch.frombuffer(bytearray(bmp[54:54 + stride * h]), dtype=torch.uint8).to(dev)
            keep.append(px)
            si = i % nstreams
            with torch.cuda.stream(streams[si]):
                streams[si].wait_stream(torch.cuda.current_stream())          # the upload ran on the default stream
                if i % G < nstreams:
                    gather.reserve(i)
                pl, sz = gather.record(i)
                d = jpegamd.Encoder.image(px.data_ptr(), w, h, stride, True, jpegamd.ORDER_BGR, 0)
                encs[si].encode_async(d, pl.data_ptr(), pl.numel(), sz.data_ptr(), True, streams[si].cuda_stream)
                if i % G == G - 1:
                    for sj in streams:
                        if sj != streams[si]:
                            streams[si].wait_event(sj.record_event())
                    gather.commit(i)
            if i % G == G - 1:
                gather.drain()
                per_rank = gather.result(i)
                assert len(per_rank) == 1 and len(per_rank[0]) == G
                for k, sbytes in enumerate(per_rank[0]):
                    got[seeds[i - G + 1 + k]] = (len(sbytes), hashlib.sha256(sbytes).hexdigest())
                keep.clear()
        for e in encs:
            try:
                e.finish()
            except jpegamd.JpegAmdError as err:                                # the stream that outgrew its record: -8 from the first encode
                assert err.code in (-1, -8)
        assert len(again) == 1 and gather.reencoded == 1 and gather.exchanges == 0      # (one rank: nothing crosses a link)
        for seed in seeds:
            e = batch[f"4096x4096_seed{seed}_kind0_q50"]
            assert got[seed] == (e["size"], e["sha256"]), seed
    finally:
        if own_group:
            dist.destroy_process_group()


def test_repeatable_and_order_independent(jpegamd, dev):
    """Idempotence: encoding the same pixels twice, and interleaved with other images on one context, gives
    identical bytes (no state leaks through the scratch buffers)."""
    enc = jpegamd.Encoder(1024, 1024)
    a = jpegamd.synth_bmp(1024, 1024, 1, 0, 0)
    b = jpegamd.synth_bmp(640, 333, 2, 1, 0)
    a1, _ = device_encode(jpegamd, enc, a, dev)
    b1, _ = device_encode(jpegamd, enc, b, dev)
    a2, _ = device_encode(jpegamd, enc, a, dev)
    b2, _ = device_encode(jpegamd, enc, b, dev)
    assert a1 == a2 and b1 == b2


def test_decodes_with_an_independent_decoder(jpegamd, dev):
    """Container sanity: PIL decodes the stream to the right size and a plausible image."""
    PIL = pytest.importorskip("PIL.Image")
    import io
    bmp = jpegamd.synth_bmp(640, 360, 2, 0, 2)
    jpg = jpegamd.encode_bmp_bytes(bmp)
    im = PIL.open(io.BytesIO(jpg))
    im.load()
    assert im.size == (640, 360) and im.mode == "L"
    src = PIL.open(io.BytesIO(bmp)).convert("L")
    mse = float(np.mean((np.asarray(im, np.float64) - np.asarray(src, np.float64)) ** 2))
    assert mse < 60.0, mse


@pytest.mark.gpu
def test_file_batch_pipeline_matches_oracle(jpegamd, oracle, dev, tmp_path):
    """jpegamd_encode_files: several files of different geometry in flight at once, one bad input in the middle;
    every written stream equals the oracle's, the bad file reports its own error and does not stop the batch."""
    cases = [(640, 480, 11, 0, 0), (203, 117, 5, 1, 0), (1024, 768, 3, 0, 0), (64, 64, 1, 2, 0), (333, 250, 4, 0, 1), (1920, 1080, 9, 0, 0), (8, 8, 2, 1, 0)]
    ins, outs, bmps = [], [], []
    for i, (w, h, seed, kind, flags) in enumerate(cases):
        bmp = jpegamd.synth_bmp(w, h, seed, kind, flags)
        p = tmp_path / f"in_{i}.bmp"
        p.write_bytes(bmp)
        ins.append(p); outs.append(tmp_path / f"out_{i}.jpg"); bmps.append(bmp)
    bad = tmp_path / "bad.bmp"
    bad.write_bytes(b"BM" + bytes(40))
    ins.insert(3, bad); outs.insert(3, tmp_path / "bad.jpg"); bmps.insert(3, None)
    rc, status, st = jpegamd.encode_files(ins, outs)
    assert rc == -7
    assert status[3] == -7 and not (tmp_path / "bad.jpg").exists()
    assert st.files_ok == len(cases) and st.files_failed == 1
    total = 0
    for i, bmp in enumerate(bmps):
        if bmp is None:
            continue
        assert status[i] == 0
        got = outs[i].read_bytes()
        assert got == oracle.encode_bmp(bmp), f"file {i}"
        total += len(got)
    assert st.bytes_out == total and st.seconds_total > 0
    # all files readable -> return code 0
    rc2, status2, st2 = jpegamd.encode_files(ins[:3], outs[:3])
    assert rc2 == 0 and status2 == [0, 0, 0] and st2.files_failed == 0
    # noise at quality 100 needs more than the 1 byte per pixel the batch path reserves: the file is encoded a second time
    # into a worst-case buffer instead of failing with -8 (saveJPEGGrayscale / encode_bmp_memory always could)
    noisy = [jpegamd.synth_bmp(256, 192, 21, 1, 0), jpegamd.synth_bmp(640, 480, 11, 0, 0), jpegamd.synth_bmp(200, 200, 22, 1, 0)]
    nin, nout = [], []
    for i, bmp in enumerate(noisy):
        (tmp_path / f"n_{i}.bmp").write_bytes(bmp)
        nin.append(tmp_path / f"n_{i}.bmp"); nout.append(tmp_path / f"n_{i}.jpg")
    rc3, status3, _ = jpegamd.encode_files(nin, nout, quality=100)
    assert rc3 == 0 and status3 == [0, 0, 0]
    for i, bmp in enumerate(noisy):
        exp = oracle.encode_bmp(bmp, 100)
        assert nout[i].read_bytes() == exp, f"noisy file {i}"
    assert len(oracle.encode_bmp(noisy[0], 100)) > 256 * 192 + 4096          # the case really exceeds the first buffer


@pytest.mark.gpu
def test_context_capacity_is_checked_per_derived_count(jpegamd, oracle, dev):
    """A context sized for 256x256 must REFUSE a 320x160 image: it has fewer 64-block segments (20 vs 32) but more
    32-block tiles (40 vs 32).  Sizing by one count alone once let a wider-but-shorter image write past the scratch
    (5000x3000 in a context created for 4096x4096).  The shared context of the host functions grows instead."""
    enc = jpegamd.Encoder(256, 256)
    ok, _ = device_encode(jpegamd, enc, jpegamd.synth_bmp(256, 256, 3, 0, 0), dev)
    assert ok == oracle.encode_bmp(jpegamd.synth_bmp(256, 256, 3, 0, 0))
    with pytest.raises(jpegamd.JpegAmdError) as ei:
        device_encode(jpegamd, enc, jpegamd.synth_bmp(320, 160, 4, 0, 0), dev)
    assert ei.value.code == -5                                        # JPEGAMD_ERR_TOO_LARGE
    for (w, h) in [(256, 256), (320, 160), (2100, 40), (96, 900)]:    # the shared context: every shape after a different one
        bmp = jpegamd.synth_bmp(w, h, 5, 0, 0)
        assert jpegamd.encode_bmp_bytes(bmp) == oracle.encode_bmp(bmp), (w, h)


@pytest.mark.gpu
def test_one_image_sharded_by_block_rows(jpegamd, oracle, dev):
    """SURVEY.md 8e, single large image: every 'rank' (a separate context here) codes its block rows into unstuffed
    segments, the root imports all packs and finalizes once.  Bytes equal the oracle's for even and ragged splits,
    one- and several-segment rows, narrow images (one tile per row), and more ranks than block rows."""
    from jpegamd.sharding import encode_image_virtual_ranks
    for (w, h, kind, ranks) in [(640, 480, 0, 2), (2100, 333, 1, 3), (200, 120, 0, 4), (4200, 160, 0, 8), (64, 40, 1, 7), (1024, 1024, 0, 5)]:
        bmp = jpegamd.synth_bmp(w, h, 40 + ranks, kind, 0)
        img, px = upload_pixels(bmp, jpegamd, dev)
        d = jpegamd.Encoder.image(px.data_ptr(), img.width, img.height, img.row_stride, bool(img.bottom_up), jpegamd.ORDER_BGR, 0)
        got = encode_image_virtual_ranks(jpegamd, d, w, h, ranks, dev)
        assert got == oracle.encode_bmp(bmp), (w, h, kind, ranks)


def _segment_meta(bits: np.ndarray, word_off: int):
    """Metadata of one unstuffed segment string as k_segment_merge leaves it (jpegamd_export_segments' 12 words): bit count, word
    offset, (first 8 bits << 8) | last 7 bits, and for each byte phase p the 0xFF bytes lying WHOLLY inside the string when its
    first bit sits at bit p of a byte (16 bits each)."""
    n = len(bits)
    first8 = int("".join(map(str, np.concatenate([bits[:8], np.zeros(max(0, 8 - n), np.uint8)]))), 2)
    last7 = int("".join(map(str, bits[-7:])), 2) if n else 0
    ones8 = np.zeros(max(n - 7, 0), bool)
    if n >= 8:
        c = np.concatenate([[0], np.cumsum(bits, dtype=np.int64)])
        ones8 = (c[8:] - c[:-8]) == 8                                  # ones8[o]: the 8 bits from offset o are all ones
    ff = [int(ones8[(8 - ph) % 8::8].sum()) for ph in range(8)]
    m = np.zeros(12, np.uint32)
    m[0], m[1], m[2] = n, word_off, (first8 << 8) | last7
    for ph in range(8):
        m[8 + ph // 2] |= np.uint32(ff[ph] << (16 * (ph & 1)))
    return m


def test_finalize_over_synthetic_segments(jpegamd, dev):
    """k_finalize alone, fed through the segment exchange (jpegamd_import_segments) with bit strings no picture produces on
    demand: dense in ones (every byte 0xFF, runs of 0xFF across segment borders), lengths around the kernel's pass sizes
    (1024 owned bytes per pass, 16-byte output pieces), segments that end on and off byte boundaries, long segments, a last
    segment that leaves 1 .. 7 bits for the zero-padded flush.  Expected bytes: the concatenated bits, padded with zeros
    to a byte, 0x00 behind every 0xFF (huffman.c:26-81) -- computed here in numpy."""
    rng = np.random.default_rng(77)
    W = 2048                                                           # 256 blocks per row: one 8-tile segment per block row
    lens = [16, 24, 23, 8191, 8192, 8193, 8200, 8184 + 5, 16384, 16390, 3 * 8192 + 1, 40001, 127, 128, 129, 64, 9000, 15, 8, 100000, 8 * 1023, 8 * 1024,
            8 * 1025, 8 * 2047 + 3, 33, 7777, 8 * 16, 8 * 16 + 1, 8 * 15, 250000, 17, 4095 * 8, 4097 * 8, 61, 12345, 8 * 3000 + 4]
    for case, dens in enumerate((0.5, 0.9, 0.995, 1.0, 0.0)):
        lens_c = lens if case % 2 == 0 else lens[::-1]
        H = 8 * len(lens_c)
        strings = []
        for n in lens_c:
            b = (rng.random(n) < dens).astype(np.uint8)
            if n < 8:
                b[:] = 0
            strings.append(b)
        words, metas, off = [], [], 0
        for b in strings:
            nw = (len(b) + 31) // 32
            padded = np.concatenate([b, np.zeros(nw * 32 - len(b), np.uint8)])
            w = np.packbits(padded).view(">u4").astype(np.uint32)
            metas.append(_segment_meta(b, off))
            words.append(w)
            off += nw
        dense = torch.from_numpy(np.concatenate(words).view(np.int32).copy()).to(dev)
        meta = torch.from_numpy(np.concatenate(metas).view(np.int32).copy()).to(dev)
        allbits = np.concatenate(strings)
        padded = np.concatenate([allbits, np.zeros((-len(allbits)) % 8, np.uint8)])
        raw = np.packbits(padded)
        want = bytearray()
        for x in raw.tobytes():
            want.append(x)
            if x == 0xFF:
                want.append(0)
        px = torch.zeros(64, dtype=torch.uint8, device=dev)           # never read: k_finalize works on the imported segments alone
        img = jpegamd.Encoder.image(px.data_ptr(), W, H, 3 * W, False, jpegamd.ORDER_BGR, 0)
        enc = jpegamd.Encoder(W, H)
        for shift in (0, 5):                                           # output buffers at two alignments
            cap = len(want) + 64
            out = torch.full((cap + 16,), 0xAA, dtype=torch.uint8, device=dev)
            size = torch.zeros(1, dtype=torch.int64, device=dev)
            enc.import_segments(img, 0, len(strings), dense.data_ptr(), meta.data_ptr(), 0)
            enc.finalize_async(img, out.data_ptr() + shift, cap, size.data_ptr(), False, 0)
            enc.finish()
            n = int(size.item())
            got = bytes(out[shift:shift + n].cpu().numpy())
            assert n == len(want) and got == bytes(want), (case, dens, shift, n, len(want),
                                                           next((i for i, (x, y) in enumerate(zip(got, want)) if x != y), None))
            assert bytes(out[shift + n:shift + n + 8].cpu().numpy()) == b"\xaa" * 8 and (shift == 0 or bytes(out[:shift].cpu().numpy()) == b"\xaa" * shift)


@pytest.mark.gpu
def test_sharded_image_encoder_single_rank(jpegamd, oracle, dev):
    """jpegamd.sharding.ShardedImageEncoder without a process group: rows -> export -> finalize on one context."""
    from jpegamd.sharding import ShardedImageEncoder
    w, h = 777, 333
    bmp = jpegamd.synth_bmp(w, h, 12, 0, 0)
    img, px = upload_pixels(bmp, jpegamd, dev)
    d = jpegamd.Encoder.image(px.data_ptr(), img.width, img.height, img.row_stride, bool(img.bottom_up), jpegamd.ORDER_BGR, 0)
    enc = jpegamd.Encoder(w, h)
    she = ShardedImageEncoder(enc, w, h, dev)
    cap = 4096 + 2 * w * h
    out = torch.empty(cap, dtype=torch.uint8, device=dev)
    size = torch.zeros(1, dtype=torch.int64, device=dev)
    she.encode(d, out, size)
    enc.finish()
    assert bytes(out[:int(size.item())].cpu().numpy()) == oracle.encode_bmp(bmp)
    assert int(she.total.item()) > 0 and she.rows_of(0) == (0, (h + 7) // 8)


@pytest.mark.gpu
def test_sharded_image_encoder_two_processes(jpegamd, oracle, dev, tmp_path):
    """SURVEY.md 8e, one image over several ranks, with world_size 2 for real: two fresh processes (a gloo group; both on
    GPU 0, packs staged through host memory) each transform and entropy-code their block rows, the root imports the other
    rank's segments at their global indices and finalizes.  Even split, ragged split (odd number of block rows), several
    segments per row, noise (stuffing across the seam): bytes equal the oracle's (rle.c:59-70 DC chain, huffman.c:35-81)."""
    import torch.multiprocessing as mp
    import sharded_worker
    cases = [(640, 480, 31, 0), (2100, 333, 32, 1), (4200, 168, 33, 0), (300, 72, 34, 1)]
    mp.spawn(sharded_worker.run, args=(2, 29533, cases, str(tmp_path)), nprocs=2, join=True)
    for ci, (w, h, seed, kind) in enumerate(cases):
        got = (tmp_path / f"case{ci}.jpg").read_bytes()
        assert got == oracle.encode_bmp(jpegamd.synth_bmp(w, h, seed, kind, 0)), (w, h, kind)


@pytest.mark.gpu
def test_stage_functions_match_the_oracle_stage_by_stage(jpegamd, oracle, dev):
    """The reference's own verification method: every stage function (natural_c_stages.h, GPU kernels behind them)
    against the same stage of the oracle -- luma, centring, float32 DCT bit for bit, quantiser, zigzag, RLE symbols,
    entropy-coded bytes -- and the chained result against the whole-pipeline encoder."""
    from jpegamd import stages
    for (w, h, seed, kind) in [(96, 64, 1, 0), (203, 117, 5, 1), (64, 64, 101, 2), (333, 250, 4, 0), (8, 8, 2, 1)]:
        bmp = jpegamd.synth_bmp(w, h, seed, kind, 1)                   # top-down: rows already in BMPImage order
        img, off = jpegamd.parse_bmp(bmp)
        assert not img.bottom_up and img.channel_order == jpegamd.ORDER_BGR
        rows = np.frombuffer(bmp, np.uint8, offset=off).reshape(h, img.row_stride)[:, :3 * w].reshape(h, w, 3)
        rgb = rows[:, :, ::-1]                                          # loadBMPImage swaps BGR -> RGB (bmp_handler.c:112-120)
        got = stages.run_stages(rgb)
        exp = oracle.stages(bmp)
        assert np.array_equal(got["centered"], exp["y"]) and np.array_equal(got["y"].astype(np.int16) - 128, exp["y"].astype(np.int16))
        assert np.array_equal(got["dct"].view(np.uint32), exp["dct"].view(np.uint32)), (w, h, "dct bits")
        assert np.array_equal(got["quant"], exp["quant"]) and np.array_equal(got["zigzag"], exp["zigzag"])
        assert got["blocks"] == ((w + 7) // 8, (h + 7) // 8, ((w + 7) // 8) * ((h + 7) // 8))
        assert got["rle"] == oracle.rle_symbols(exp["zigzag"])
        assert got["entropy"] == oracle.entropy(exp["zigzag"])
        whole = jpegamd.encode_bmp_bytes(bmp)
        assert whole == oracle.jfif_prefix(w, h) + got["entropy"] + b"\xff\xd9"
    blk = (np.arange(64).reshape(8, 8) * 3 - 90).astype(np.int8)
    out = np.zeros((8, 8), np.float32)
    jpegamd.lib.computeDCTBlock(blk.ctypes.data, out.ctypes.data)
    assert np.array_equal(out.view(np.uint32), oracle.dct_blocks(blk[None])[0].view(np.uint32))


@pytest.mark.gpu
def test_concurrent_streams_and_contexts(jpegamd, oracle, dev):
    """bench.py's default mode: several encoder contexts on several HIP streams with many encodes in flight and no host
    synchronisation in between.  Every output (not only the last) must equal the oracle's."""
    shapes = [(640, 480, 0), (1000, 700, 1), (333, 250, 0), (1280, 64, 3), (96, 1100, 0), (1024, 1024, 2)]
    bmps = [jpegamd.synth_bmp(w, h, 900 + i, kind, i & 1) for i, (w, h, kind) in enumerate(shapes)]
    ups = [upload_pixels(b, jpegamd, dev) for b in bmps]
    nstreams, rounds = 3, 4
    encs = [jpegamd.Encoder(1280, 1100) for _ in range(nstreams)]
    streams = [torch.cuda.Stream() for _ in range(nstreams)]
    jobs = []
    for r in range(rounds):
        for i, (img, px) in enumerate(ups):
            k = len(jobs) % nstreams
            cap = 4096 + 2 * img.width * img.height
            out = torch.empty(cap, dtype=torch.uint8, device=dev)
            size = torch.zeros(1, dtype=torch.int64, device=dev)
            d = jpegamd.Encoder.image(px.data_ptr(), img.width, img.height, img.row_stride, bool(img.bottom_up), jpegamd.ORDER_BGR, 0)
            with torch.cuda.stream(streams[k]):
                encs[k].encode_async(d, out.data_ptr(), cap, size.data_ptr(), True, streams[k].cuda_stream)
            jobs.append((i, out, size))
    torch.cuda.synchronize()
    for e in encs:
        e.finish()
    expect = [oracle.encode_bmp(b) for b in bmps]
    for n, (i, out, size) in enumerate(jobs):
        assert bytes(out[:int(size.item())].cpu().numpy()) == expect[i], (n, shapes[i])


@pytest.mark.gpu
def test_reference_main_linked_against_the_library(jpegamd, oracle, dev, tmp_path):
    """INTEGRATION.md section 1 as an executable: the reference's UNMODIFIED natural_c/src/main.c, compiled against the
    reference's own headers and linked with libjpegamd.so instead of its src/core + src/io (oracle/Makefile `ref`),
    writes the same bytes as the reference and keeps main.c's exit codes."""
    app = ROOT / "oracle" / "_ref" / "main_on_jpegamd"
    if not app.exists():
        pytest.skip("oracle/_ref/main_on_jpegamd not built (needs the reference sources at build time)")
    for i, (w, h, kind, flags) in enumerate([(320, 200, 0, 0), (203, 117, 1, 1), (333, 250, 0, 2)]):
        bmp = jpegamd.synth_bmp(w, h, 70 + i, kind, flags)
        src, dst = tmp_path / f"in{i}.bmp", tmp_path / f"out{i}.jpg"
        src.write_bytes(bmp)
        r = subprocess.run([str(app), str(src), str(dst)], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr
        assert dst.read_bytes() == oracle.encode_bmp(bmp), (w, h, kind, flags)
    assert subprocess.run([str(app)], capture_output=True, timeout=60).returncode == 1                       # main.c:9-12
    assert subprocess.run([str(app), str(tmp_path / "missing.bmp"), str(tmp_path / "x.jpg")], capture_output=True, timeout=60).returncode == 1   # main.c:29-31


@pytest.mark.gpu
def test_reference_driver_over_the_gpu_stage_functions(jpegamd, oracle, dev, tmp_path):
    """The reference's own driver (main.c + io/jpeg_handler.c + io/bmp_handler.c, unmodified) with every stage function
    resolved from libjpegamd.so (natural_c_stages.h): the bytes it writes equal the reference's."""
    app = ROOT / "oracle" / "_ref" / "driver_on_jpegamd_stages"
    if not app.exists():
        pytest.skip("oracle/_ref/driver_on_jpegamd_stages not built (needs the reference sources at build time)")
    for i, (w, h, kind, flags) in enumerate([(200, 120, 0, 0), (203, 117, 1, 1)]):
        bmp = jpegamd.synth_bmp(w, h, 80 + i, kind, flags)
        src, dst = tmp_path / f"in{i}.bmp", tmp_path / f"out{i}.jpg"
        src.write_bytes(bmp)
        r = subprocess.run([str(app), str(src), str(dst)], capture_output=True, text=True, timeout=180)
        assert r.returncode == 0, r.stderr
        assert dst.read_bytes() == oracle.encode_bmp(bmp), (w, h, kind, flags)
