"""The oracle (CPU restatement of natural_c) against everything that pins it:
golden JFIF bytes produced by the compiled reference, the compiled reference itself when
oracle/_ref is present, and the reference's table text when /root/reference is present."""
from __future__ import annotations

import hashlib
import json
import re
import struct
from pathlib import Path

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, fixture_bmp, golden_jpg

REF = Path("/root/reference")


def test_goldens_byte_exact(oracle, jpegamd, manifest):
    for e in manifest:
        bmp = fixture_bmp(e, jpegamd)
        assert hashlib.sha256(bmp).hexdigest() == e["bmp_sha256"], e["name"]
        got = oracle.encode_bmp(bmp)
        exp = golden_jpg(e)
        assert len(exp) == e["jpg_size"] and hashlib.sha256(exp).hexdigest() == e["jpg_sha256"]
        assert got == exp, f"{e['name']}: oracle differs from the reference's bytes"


def test_quality_goldens(oracle, jpegamd):
    q = json.loads((GOLDEN / "quality.json").read_text())
    for name, e in q.items():
        bmp = jpegamd.synth_bmp(e["width"], e["height"], e["seed"], e["kind"], e["flags"])
        got = oracle.encode_bmp(bmp, e["quality"])
        assert (len(got), hashlib.sha256(got).hexdigest()) == (e["jpg_size"], e["jpg_sha256"]), name


def test_large_golden_1080p(oracle, jpegamd):
    large = json.loads((GOLDEN / "large.json").read_text())
    e = large["1920x1080_seed1_kind0_q50"]
    got = oracle.encode_bmp(jpegamd.synth_bmp(1920, 1080, 1, 0, 0))
    assert (len(got), hashlib.sha256(got).hexdigest()) == (e["size"], e["sha256"])


def test_batch4096_golden_sample(oracle, jpegamd):
    """BASELINE configs[3] (64 x 4096^2): the restatement against the compiled reference's answer for two of the 64 seeds
    (the GPU suite hashes all 64 through the gather path)."""
    batch = json.loads((GOLDEN / "batch4096.json").read_text())
    assert len(batch) == 64 and len({e["sha256"] for e in batch.values()}) == 64
    for seed in (2000, 2063):
        e = batch[f"4096x4096_seed{seed}_kind0_q50"]
        bmp = jpegamd.synth_bmp(4096, 4096, seed, 0, 0)
        assert hashlib.sha256(bmp).hexdigest() == e["bmp_sha256"]
        got = oracle.encode_bmp(bmp)
        assert (len(got), hashlib.sha256(got).hexdigest()) == (e["size"], e["sha256"]), seed


def test_reference_assets_known_answers(oracle):
    """The reference's own four sample images (tests/golden/assets: data fixtures) -> SURVEY.md 8c known answers."""
    answers = json.loads((GOLDEN / "assets.json").read_text())
    for name, e in answers.items():
        bmp = (GOLDEN / "assets" / name).read_bytes()
        assert hashlib.sha256(bmp).hexdigest() == e["bmp_sha256"], name
        got = oracle.encode_bmp(bmp)
        assert (len(got), hashlib.sha256(got).hexdigest()) == (e["jpg_size"], e["jpg_sha256"]), name
    # SURVEY.md 8c known answers
    assert answers["lena.bmp"]["jpg_sha256"] == "95cf58feb23fabd6a71621548913f413a8bba9718ecad237cf0341155b83319a"
    assert answers["greenland.bmp"]["jpg_size"] == 40913


def test_against_compiled_reference(oracle, jpegamd, tmp_path):
    if not oracle.REF_APP.exists():
        pytest.skip("oracle/_ref not built")
    rng = np.random.default_rng(7)
    for i in range(12):
        w, h = int(rng.integers(1, 300)), int(rng.integers(1, 200))
        kind, flags = int(rng.integers(0, 4)), int(rng.integers(0, 4))
        bmp = jpegamd.synth_bmp(w, h, 100 + i, kind, flags)
        assert oracle.encode_bmp(bmp) == oracle.reference_app_encode(bmp, tmp_path), (w, h, kind, flags)


@pytest.mark.skipif(not (REF / "natural_c").exists(), reason="reference text only in the build container")
def test_tables_equal_reference_text(oracle):
    src = (REF / "natural_c/src/core/jpeg_tables.c").read_text()
    nums = lambda name: [int(x, 0) for x in re.findall(r"0x[0-9A-Fa-f]+|\d+", re.search(name + r"\[\d+\]\s*=\s*\{([^}]*)\}", src).group(1))]
    assert list(oracle.quant_table(50)) == nums("std_luminance_quant_tbl")
    prefix = oracle.jfif_prefix(640, 480)
    assert len(prefix) == 328
    assert list(prefix[107:123]) == nums("std_dc_luminance_nrcodes") and list(prefix[123:135]) == nums("std_dc_luminance_values")
    assert list(prefix[140:156]) == nums("std_ac_luminance_nrcodes") and list(prefix[156:318]) == nums("std_ac_luminance_values")
    # cosine LUT: compare the oracle's exact-order DCT of unit impulses against the literals
    dct = (REF / "natural_c/src/core/dct.c").read_text()
    lut = np.array([float(x[:-1]) for x in re.findall(r"-?\d\.\d{6}f", re.search(r"COS_LUT\[8\]\[8\]\s*=\s*\{(.*?)\};", dct, re.S).group(1))],
                   np.float32).reshape(8, 8)
    for x in range(8):
        blk = np.zeros((1, 8, 8), np.int8)
        blk[0, x, 0] = 1                                  # pixel (row x, col 0): F[u][0] = K * COS[x][u] * COS[0][0]
        f = oracle.dct_blocks(blk)[0]
        for u in range(8):
            k = np.float32(np.float32(np.float32(0.25) * (np.float32(0.707107) if u == 0 else np.float32(1))) * np.float32(0.707107))
            assert f[u, 0] == np.float32(k * np.float32(np.float32(np.float32(1) * lut[x, u]) * lut[0, 0]))


def test_stage_chain_is_consistent(oracle, jpegamd):
    bmp = jpegamd.synth_bmp(203, 117, 5, 1, 0)
    st = oracle.stages(bmp)
    full = oracle.encode_bmp(bmp)
    assert full[:328] == oracle.jfif_prefix(203, 117) and full[-2:] == b"\xff\xd9"
    assert oracle.entropy(st["zigzag"]) == full[328:-2]
    # SOF0 carries the ORIGINAL dims (jpeg_handler.c:226) while the data covers the padded image
    assert struct.unpack(">HH", full[94:98]) == (117, 203)
    assert st["y"].shape == (120, 208)
    syms = oracle.rle_symbols(st["zigzag"])
    assert len(syms) >= st["zigzag"].shape[0]            # at least a DC symbol per block


def test_entropy_properties(oracle):
    # all-zero blocks: DC size 0 (code 00) + EOB (1010) = 6 bits per block, zero padded at the end
    zz = np.zeros((4, 64), np.int16)
    assert oracle.entropy(zz) == bytes([0b00101000, 0b10100010, 0b10001010])
    assert oracle.entropy(zz[:3]) == bytes([0b00101000, 0b10100010, 0b10000000])   # 18 bits: zero-padded flush
    # a run of 0xFF is stuffed; trailing coefficient 63 != 0 suppresses EOB
    zz = np.zeros((1, 64), np.int16)
    zz[0, 63] = 1
    s = oracle.rle_symbols(zz)
    assert [x[0] for x in s] == [0, 0xF0, 0xF0, 0xF0, 0xE1]           # DC, 3 x ZRL, run 14 / size 1, no EOB


def test_bmp_errors(oracle, jpegamd):
    bmp = bytearray(jpegamd.synth_bmp(16, 16, 1, 0, 0))
    for mutate, code in ((lambda b: b.__setitem__(0, 0x41), -2), (lambda b: b.__setitem__(28, 32), -3),
                         (lambda b: b.__setitem__(30, 1), -4)):
        b = bytearray(bmp)
        mutate(b)
        with pytest.raises(oracle.OracleError) as ei:
            oracle.encode_bmp(bytes(b))
        assert ei.value.code == code
    with pytest.raises(oracle.OracleError) as ei:
        oracle.encode_bmp(bytes(bmp[:-10]))                            # short pixel data
    assert ei.value.code == -1
