"""CPU-side checks of the product library: the C-ABI surface, host logic (BMP parsing, synthetic
generator, constant derivation) and loud failure without a device.  No compute calls are made here."""
from __future__ import annotations

import ctypes
import hashlib
import re
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

from conftest import PKG, ROOT, fixture_bmp, has_gpu

sys.path.insert(0, str(ROOT / "tools"))


def test_library_exports_every_declared_symbol(jpegamd):
    header = (ROOT / "include" / "jpeg_compression.h").read_text()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\([^;{]*\)\s*;", header))
    declared -= {"defined"}
    assert {"JpegCompression_Init", "convertToJpeg", "loadBMPImage", "saveJPEGGrayscale", "jpegamd_encode_async"} <= declared
    raw = ctypes.CDLL(str(jpegamd.LIB_PATH))
    missing = [s for s in sorted(declared) if not hasattr(raw, s)]
    assert not missing, f"declared in include/jpeg_compression.h but not exported: {missing}"
    assert set(jpegamd.EXPORTED) == declared


def test_library_exports_the_stage_interface(jpegamd):
    """include/natural_c_stages.h: every declared stage function / free function is exported, struct sizes match the reference's."""
    header = (ROOT / "include" / "natural_c_stages.h").read_text()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\([^;{]*\)\s*;", header))
    from jpegamd import stages
    assert declared == set(stages.STAGE_EXPORTS)
    raw = ctypes.CDLL(str(jpegamd.LIB_PATH))
    assert not [s for s in sorted(declared) if not hasattr(raw, s)]
    assert ctypes.sizeof(stages.RLESymbol) == 6 and ctypes.sizeof(stages.RLEData) == 24 and ctypes.sizeof(stages.ZigZagData) == 24
    assert ctypes.sizeof(stages.YImage) == 16 and ctypes.sizeof(stages.JpegEncoderBuffer) == 24


def test_struct_layouts_match_header(jpegamd):
    # natural_c/include/bmp_handler.h:37-41 and the DTO field order of jpeg_compression.h:32-64
    assert ctypes.sizeof(jpegamd.BMPImage) == 16
    assert [f[0] for f in jpegamd.DTO._fields_][:12] == ["width", "height", "r_phy_ptr", "gb_phy_ptr", "y_phy_ptr", "dct_phy_ptr",
                                                          "quant_phy_ptr", "zigzag_phy_ptr", "rle_phy_ptr", "rle_count",
                                                          "huff_phy_ptr", "huff_size"]
    assert ctypes.sizeof(jpegamd.Image) == 32 and ctypes.sizeof(jpegamd.Stats) == 64


def test_synth_generator_is_deterministic(jpegamd, manifest):
    for e in manifest:
        if e["source"] != "synth":
            continue
        bmp = fixture_bmp(e, jpegamd)
        assert hashlib.sha256(bmp).hexdigest() == e["bmp_sha256"], e["name"]
    b = jpegamd.synth_bmp(37, 11, 3, 0, 3)
    assert b[:2] == b"BM" and int.from_bytes(b[10:14], "little") == 138 and int.from_bytes(b[22:26], "little", signed=True) == -11
    assert len(b) == 138 + ((37 * 3 + 3) & ~3) * 11


def test_parse_bmp_follows_reference_rules(jpegamd):
    bmp = jpegamd.synth_bmp(37, 11, 3, 0, 1)
    img, off = jpegamd.parse_bmp(bmp)
    assert (img.width, img.height, img.row_stride, img.bottom_up, off) == (37, 11, 112, 0, 54)
    img, off = jpegamd.parse_bmp(jpegamd.synth_bmp(8, 8, 1, 0, 2))
    assert (img.bottom_up, off) == (1, 138)
    bad = bytearray(bmp)
    bad[28] = 8                                     # not 24 bpp (bmp_handler.c:44)
    with pytest.raises(jpegamd.JpegAmdError) as ei:
        jpegamd.parse_bmp(bytes(bad))
    assert ei.value.code == -7
    with pytest.raises(jpegamd.JpegAmdError):
        jpegamd.parse_bmp(bmp[:100])                # rows missing (bmp_handler.c:104)
    with pytest.raises(jpegamd.JpegAmdError):
        jpegamd.parse_bmp(b"PM" + bmp[2:])          # magic (bmp_handler.c:30)


def test_quant_table_for_quality(jpegamd, oracle):
    t50 = jpegamd.quant_table(50)
    assert list(t50[:8]) == [16, 11, 10, 16, 24, 40, 51, 61] and list(jpegamd.quant_table(0)) == list(t50)   # jpeg_tables.c:3-12
    for q in (1, 10, 50, 90, 100):
        assert list(jpegamd.quant_table(q)) == [int(x) for x in oracle.quant_table(q)]


def test_constants_derivation_is_reentrant(jpegamd):
    """Contexts may be driven from different host threads (one in-flight call per context): the table derivation must not
    share static scratch.  Two threads derive different qualities concurrently; results equal the single-threaded ones."""
    import threading
    want = {q: jpegamd.mfma_consts(q) for q in (10, 90)}
    bad = []

    def work(q):
        for _ in range(40):
            c = jpegamd.mfma_consts(q)
            if not (np.array_equal(c["qmul"], want[q]["qmul"]) and np.array_equal(c["qthr"], want[q]["qthr"]) and np.array_equal(c["bias"], want[q]["bias"])):
                bad.append(q)
    ts = [threading.Thread(target=work, args=(q,)) for q in (10, 90, 10, 90)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not bad


def test_mfma_constants_are_on_the_safe_side(jpegamd, oracle):
    """Matrix-pipe kernel, per zigzag position: bias - 0.5 >= delta (the band on either side of a rounding tie), the flag threshold is
    2 bias - 1 EXACTLY (the kernel derives it so, in float32) and not much wider than 2 delta; the multiplier is K/q (the MFMA output
    is the plain LUT sum)."""
    zz = [0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
          35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63]
    for q in (50, 10, 90):
        c = jpegamd.mfma_consts(q)
        table = oracle.quant_table(q)
        for z in range(64):
            k = zz[z]
            db = np.float64(np.float32(c["bias"][z])) - 0.5
            assert db >= c["delta"][k]
            assert np.float32(c["qthr"][z]) == np.float32(np.float32(2.0) * np.float32(c["bias"][z]) - np.float32(1.0))     # what fma(2, bias, -1) gives
            assert np.float64(c["qthr"][z]) == 2.0 * db                      # ... which is exact
            assert np.float64(c["qthr"][z]) >= db + c["delta"][k]
            assert np.float64(c["qthr"][z]) <= 2.01 * c["delta"][k] + 4e-7    # no wider than needed (one bias for all positions was 2.3 x as wide on average)
            u, v = divmod(k, 8)
            kk = np.float32(np.float32(np.float32(0.25) * (np.float32(0.707107) if u == 0 else np.float32(1))) * (np.float32(0.707107) if v == 0 else np.float32(1)))
            assert abs(float(c["qmul"][z]) * c["scale"] - float(kk) / float(table[k])) <= 1e-7 * float(kk)     # the accumulator holds kMfmaScale x LUT sum
            assert float(c["qthr"][z]) < 0.01                                # (2.0 would mean: the integer split failed, everything is flagged)
            # the uncentred operand's constant: five positions only, inside the band, and folded into what the fma adds
            assert (c["zoff"][z] != 0) == (z in (6, 9, 21, 24, 27)), z
            assert abs(float(c["zoff"][z])) <= 0.1 * c["delta"][k] and c["delta"][k] >= abs(float(c["zoff"][z])) * 1.0
            assert np.float32(c["qadd"][z]) == np.float32(np.float64(np.float32(c["bias"][z])) + np.float64(np.float32(c["zoff"][z])))
        assert c["scale"] == 2.0 ** -13 and c["dc_off"] == 1.0             # B operand = Y 2^-24, A terms = 2048 K: the DC row's surplus 64 x 128 is 1.0
        assert 1e-5 < c["delta"][1:].max() < 5e-3


def test_dc_closed_form(jpegamd):
    """The kernel never recomputes a DC coefficient: it quantises |S| (S = the block's centred pixel sum, an exact integer) as
    floor(|S| K/q + 0.5 + delta) and puts the sign back.  That rests on two facts, both checked here by exhaustion:
      (a) the reference's DC value -- fl(scale * S) with scale = fl(fl(0.25 * 0.707107) * 0.707107) (dct.c:87-93), a correctly
          rounded float32 division by q and roundf (quantization.c:34-36) -- equals sign(S) * floor((|S| + 4 q) / (8 q)) for EVERY
          sum of 64 values in [-128, 127] and EVERY q in 1 .. 255 (the scale lies above 1/8: all ties go away from zero);
      (b) the kernel's own evaluation, one float32 fma of kMfmaScale |S| (what is left of the DC row's accumulator once the
          uncentred operand's surplus, dc_off, is taken off: exact) with the stored multiplier and bias of zigzag 0, then floor,
          gives that value for every S at every quality 1 .. 100, with a margin far above a float32 step."""
    f32, f64 = np.float32, np.float64
    S = np.arange(-128 * 64, 127 * 64 + 1, dtype=np.int64)
    scale = f32(f32(f32(0.25) * f32(0.707107)) * f32(0.707107))
    for q in range(1, 256):
        z = ((scale * S.astype(f32)).astype(f32) / f32(q)).astype(f32).astype(f64)          # float32 products / quotients, correctly rounded
        ref = (np.sign(z) * np.floor(np.abs(z) + 0.5)).astype(np.int64)                     # roundf: half away from zero (exact in float64)
        assert np.array_equal(ref, np.sign(S) * ((np.abs(S) + 4 * q) // (8 * q))), q
    for quality in range(1, 101):
        c = jpegamd.mfma_consts(quality)
        q0 = int(jpegamd.quant_table(quality)[0])
        assert c["qadd"][0] == c["bias"][0]
        acc = (f32(c["scale"]) * (S + 8192).astype(f32)).astype(f32)                         # the DC row: kMfmaScale x the sum of 64 values in 0 .. 255 (exact)
        assert np.array_equal((acc - f32(c["dc_off"])).astype(f64), S.astype(f64) * c["scale"])       # ... minus the surplus: exact
        zc64 = np.abs(S).astype(f64) * c["scale"] * f64(c["qmul"][0]) + f64(f32(c["bias"][0]))  # 24 x 24 bits: the product is exact in float64
        n = np.floor(zc64.astype(f32)).astype(np.int64) * np.sign(S)
        assert np.array_equal(n, np.sign(S) * ((np.abs(S) + 4 * q0) // (8 * q0))), quality
        margin = np.abs(zc64 - np.rint(zc64)).min()                                          # the ties sit delta above an integer
        assert margin > 4 * np.spacing(f32(zc64.max())) and margin > 0.9 * (f64(f32(c["bias"][0])) - 0.5), quality


def test_group_zero_thresholds_are_safe(jpegamd):
    """Skipping a coefficient group is only legal if EVERY LUT sum it can hide quantises to an unflagged 0.  The kernel tests the
    hi accumulator chain alone, |hi| < thr in every site; the lo chain adds at most lo_bound, and the add that joins them rounds
    once: |a| <= (thr + lo_bound)(1 + 2^-24).  For each site of the group, in float32 exactly as the kernel evaluates it,
    fl(a * qmul + bias) must stay inside (qthr, 1) for a = +-that (monotone in a), so floor() is 0 and fract() is above the flag
    threshold.  lo_bound itself is checked against the split of the LUT products."""
    f32 = np.float32
    zz = [0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
          35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63]
    lut = jpegamd.cos_lut().astype(np.float64)
    scale = jpegamd.mfma_consts(50)["scale"]
    lo_true = np.zeros((4, 2))                                     # |lo-chain output| <= 128 x (sum |lo term| + |sum lo term|) (the operand is Y = p + 128), per group and lane half
    for z in range(64):
        u, v = divmod(zz[z], 8)
        K = np.outer(lut[:, u], lut[:, v]).reshape(64)
        lo = np.rint((K - np.rint(K * 2048.0) / 2048.0) * 4194304.0)            # in units of 2^-22
        lo_true[z >> 4, (z >> 3) & 1] = max(lo_true[z >> 4, (z >> 3) & 1], scale * 128.0 * (np.abs(lo).sum() + abs(lo.sum())) / 4194304.0)
    for q in (50, 10, 90, 1, 100):
        c = jpegamd.mfma_consts(q)
        thr, lob = jpegamd.group_thresholds(q, with_lo_bound=True)
        assert (lob.astype(np.float64) >= lo_true).all()
        assert (thr > 0).all() and (lob > 0).all() and (lob <= scale * 2.0 * 1.001).all()    # 64 terms x |p| <= 128 x |lo| <= 1024 units of 2^-22
        for g in range(4):
            for h in range(2):
                t = f32((np.float64(thr[g, h]) + np.float64(lob[g, h])) * (1.0 + 2.0 ** -23))
                for j in range(8):
                    z = 16 * g + 8 * h + j
                    for a in (t, -t, np.nextafter(t, f32(0)), -np.nextafter(t, f32(0))):
                        zc = f32(np.float64(a) * np.float64(c["qmul"][z]) + np.float64(f32(c["qadd"][z])))      # one rounding, like v_fma_f32
                        assert f32(c["qthr"][z]) < zc < f32(1.0), (q, g, h, j, float(a), float(zc))
        if 10 <= q <= 90:
            assert thr[0].min() <= thr[3].max()                     # coarser quantisation higher up: larger zero zone (Q=1 / 100: every step is 255 / 1)


def test_mfma_guard_band_holds_on_float32_emulation(jpegamd, oracle):
    """The matrix-pipe path on the CPU: the LUT-product matrix as two integer-valued binary16 terms (hi = round(2^11 K),
    lo = round(2^22 (K - hi 2^-11)) stored as lo 2^-11), the B operand the UNCENTRED luma Y = p + 128 as the binary16 subnormal
    Y 2^-24, one accumulator chain per term, float32 accumulation in three
    different orders (the hardware's order inside an MFMA is not specified).  Every chain must come out EXACT in every
    order; then one float32 add joins them, followed by the kernel's fma and flag test.  A coefficient the guard does NOT
    flag must equal the reference's quantised value, and the observed |z_fast - z_ref| must stay below delta (random, flat,
    extreme and basis-aligned blocks)."""
    f32, f64 = np.float32, np.float64
    zz = [0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
          35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63]
    lut = jpegamd.cos_lut()                                        # COS_LUT[x][u] as compiled into the kernels
    SCALE = 2048.0                                                 # the A terms are 2048 K (hi) and 2048 K-residual (lo 2^-11)

    K = np.zeros((64, 64))
    for k in range(64):
        u, v = divmod(k, 8)
        K[k] = np.outer(lut[:, u].astype(f64), lut[:, v].astype(f64)).reshape(64)     # K[k][x*8+y] = LUT[x][u] * LUT[y][v]
    hi = np.rint(K * SCALE); lo = np.rint((K - hi / SCALE) * 4194304.0) / SCALE
    for t in (hi, lo):                                             # both terms are binary16 values
        assert np.array_equal(t.astype(np.float16).astype(f64), t)
    assert np.abs(hi).max() <= 2048 and np.abs(lo * SCALE).max() <= 1024 and np.abs(K * SCALE - hi - lo).max() <= 2.0 ** -12
    terms = [lo, hi]
    rng = np.random.default_rng(11)
    blocks = [rng.integers(-128, 128, 64) for _ in range(150)] + [np.full(64, v) for v in (-128, 127, 3, -77)]
    blocks += [np.where(K[k] >= 0, 127, -128) for k in (1, 9, 27, 63)] + [np.where(K[k] >= 0, -128, -128) for k in (0,)]
    blocks += [rng.integers(-4, 5, 64) + 100 for _ in range(20)]
    P = np.array(blocks, dtype=np.int64)
    ref = oracle.dct_blocks(P.reshape(-1, 8, 8).astype(np.int8)).reshape(-1, 64)

    Y = (P + 128).astype(f64) * 2.0 ** -24                        # the B operand: 0 .. 255 as binary16 subnormals
    assert np.array_equal(((P + 128).astype(np.uint16).view(np.float16)).astype(f64), Y)          # the integer IS the bit pattern

    def chain(t, order):
        acc = np.zeros((len(P), 64), f32)
        prod = (t[None, :, :] * Y[:, None, :]).astype(f32)                          # exact in float32: 12 x 8 significant bits
        assert np.array_equal(prod.astype(f64), t[None, :, :] * Y[:, None, :])
        for s in range(4):
            idx = list(range(16 * s, 16 * s + 16))
            if order == "reverse":
                idx = idx[::-1]
            if order == "pairwise":
                part = prod[:, :, idx]
                while part.shape[2] > 1:
                    part = (part[:, :, 0::2] + part[:, :, 1::2]).astype(f32)
                acc = (acc + part[:, :, 0]).astype(f32)
            else:
                for i in idx:
                    acc = (acc + prod[:, :, i]).astype(f32)
        return acc

    exact = [(Y[:, None, :] * t[None, :, :]).sum(axis=2) for t in terms]
    for q in (50, 90):
        c = jpegamd.mfma_consts(q)
        table = oracle.quant_table(q).astype(f32)
        for order in ("forward", "reverse", "pairwise"):
            chains = [chain(t, order) for t in terms]
            for got, want in zip(chains, exact):
                assert np.array_equal(got.astype(f64), want), order                       # exact, whatever the order
            acc = (chains[1] + chains[0]).astype(f32)                                     # the kernel's one add
            acc[:, 0] = (acc[:, 0] - f32(c["dc_off"])).astype(f32)                        # the DC row: the surplus of the uncentred sum (exact)
            assert np.array_equal(acc[:, 0].astype(f64), P.sum(axis=1).astype(f64) * c["scale"])
            unflagged = 0
            for z in range(64):
                k = zz[z]
                zc = (acc[:, k].astype(f64) * f64(c["qmul"][z]) + f64(f32(c["qadd"][z]))).astype(f32)      # v_fma_f32: one rounding
                n = np.floor(zc).astype(np.int64)
                # the value the kernel keeps for an AC site: round-to-nearest-even of acc x qmul (the magic-number fma), WITHOUT the row's constant
                n_rne = np.rint((acc[:, k].astype(f64) * f64(c["qmul"][z]))).astype(np.int64)
                flagged = (zc - np.floor(zc)) <= f32(c["qthr"][z])
                want = np.array([int(np.float32(np.round(np.float32(r) / table[k]))) if abs(np.float32(r) / table[k]) % 1 != 0.5
                                 else int(np.sign(r) * np.ceil(abs(np.float32(r) / table[k]))) for r in ref[:, k]], np.int64)   # roundf: half away
                assert np.array_equal(n[~flagged], want[~flagged]), (q, order, z)
                if z:
                    assert np.array_equal(n_rne[~flagged], want[~flagged]), (q, order, z)
                z_fast = acc[:, k].astype(f64) * f64(c["qmul"][z]) + f64(f32(c["zoff"][z]))
                z_ref = ref[:, k].astype(f64) / f64(table[k])
                assert np.abs(z_fast - z_ref).max() <= c["delta"][k], (q, order, z)
                unflagged += int((~flagged).sum())
            assert unflagged > 0.9 * acc.size


def test_max_jfif_bytes_bounds_the_worst_case(jpegamd):
    assert jpegamd.max_jfif_bytes(8, 8) >= 328 + 2 + 2 * 216
    assert jpegamd.max_jfif_bytes(8192, 8192) > 2 * 1048576 * 215
    assert jpegamd.max_jfif_bytes(0, 5) == 0


@pytest.mark.skipif(has_gpu(), reason="checks the no-device failure mode")
def test_compute_fails_loudly_without_device(jpegamd, tmp_path):
    bmp = jpegamd.synth_bmp(16, 16, 1, 0, 0)
    with pytest.raises(jpegamd.JpegAmdError) as ei:
        jpegamd.encode_bmp_bytes(bmp)
    assert ei.value.code == -2                       # JPEGAMD_ERR_NO_DEVICE: there is no CPU fallback
    with pytest.raises(jpegamd.JpegAmdError):
        jpegamd.Encoder(64, 64)
    assert jpegamd.lib.JpegCompression_Init() == -2
    dto = jpegamd.DTO()
    assert jpegamd.lib.convertToJpeg(ctypes.byref(dto)) == -4     # not initialised
    # the batch entry refuses bad arguments before it touches a device (no context, no images, a count out of range)
    import ctypes as C
    img = jpegamd.Image(0, 8, 8, 24, 1, jpegamd.ORDER_BGR, 0)
    one = (C.c_void_p * 1)(None)
    assert jpegamd.lib.jpegamd_encode_batch_async(None, C.byref(img), 1, one, 1024, one, 1, None) == -1     # JPEGAMD_ERR_ARG
    assert jpegamd.MAX_BATCH == 32


def test_cli_argv_contract(tmp_path):
    app = PKG / "jpeg_compression_app"
    r = subprocess.run([str(app)], capture_output=True, text=True)
    assert r.returncode == 1 and "Usage:" in r.stderr and "<input_file_path> <output_file_path>" in r.stderr   # main.c:9-12
    r = subprocess.run([str(app), str(tmp_path / "missing.bmp"), str(tmp_path / "o.jpg")], capture_output=True, text=True)
    assert r.returncode == 1 and "Unable to open file" in r.stderr and "Failed to load image" in r.stderr         # main.c:29-31


def test_load_bmp_image_matches_reference_layout(jpegamd, tmp_path):
    bmp = jpegamd.synth_bmp(13, 7, 5, 0, 0)
    p = tmp_path / "a.bmp"
    p.write_bytes(bmp)
    img = jpegamd.lib.loadBMPImage(str(p).encode())
    assert img and (img.contents.width, img.contents.height) == (13, 7)
    data = np.ctypeslib.as_array(img.contents.data, shape=(7, 13, 3)).copy()
    stride = (13 * 3 + 3) & ~3
    px = np.frombuffer(bmp, np.uint8, offset=54).reshape(7, stride)[:, :39].reshape(7, 13, 3)
    assert np.array_equal(data, px[::-1, :, ::-1])   # flipped to top-down, BGR -> RGB (bmp_handler.c:109-122)
    jpegamd.lib.freeBMPImage(img)
    assert not jpegamd.lib.loadBMPImage(str(tmp_path / "nope.bmp").encode())
