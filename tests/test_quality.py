"""Decodability / quality harness: every emitted stream must open in an independent
decoder with the right geometry, and reconstruct the source luma to the PSNR a Q=50 baseline codec reaches."""
import json
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "jpeg-image-compression_amd" / "python"))
sys.path.insert(0, str(ROOT))
GOLD = ROOT / "tests" / "golden"

from jpegamd import quality  # noqa: E402


def _manifest():
    return json.loads((GOLD / "manifest.json").read_text())


def test_metric_definitions():
    a = np.full((16, 16), 100.0)
    b = a.copy()
    b[0, 0] = 116.0
    assert quality.mse(a, a) == 0 and quality.psnr(0) == float("inf")
    assert quality.mse(a, b) == pytest.approx(256.0 / 256.0)
    assert quality.psnr(1.0) == pytest.approx(20 * np.log10(255.0))
    rng = np.random.default_rng(1)
    x = rng.integers(0, 256, (64, 64)).astype(np.float64)
    assert quality.ssim(x, x) == pytest.approx(1.0)
    assert quality.ssim(x, 255.0 - x) < 0.0                      # inverted structure
    assert 0.0 < quality.ssim(x, np.clip(x + rng.normal(0, 20, x.shape), 0, 255)) < 1.0


@pytest.mark.parametrize("entry", _manifest(), ids=lambda e: e["name"])
def test_golden_streams_decode_with_an_independent_decoder(entry):
    """The reference's own outputs (the goldens) open in Pillow with the ORIGINAL dimensions (SOF0 carries them,
    jpeg_handler.c:42-60) -- which pins what the GPU path, byte-identical to them, emits."""
    img = quality.decode_jpeg((GOLD / f"{entry['name']}.jpg").read_bytes())
    assert img.shape == (entry["height"], entry["width"])


@pytest.mark.parametrize("name", ["lena_crop_128", "blackbuck_crop_96x200", "greenland_crop_130x75", "offset_sample_crop_160x120"])
def test_reconstruction_quality_of_reference_crops(name):
    r = quality.analyze((GOLD / f"{name}.bmp").read_bytes(), (GOLD / f"{name}.jpg").read_bytes())
    assert r["psnr"] > 28.0 and r["ssim"] > 0.80, r          # Q=50 luma; Pillow's 'L' weights differ slightly from (77,150,29)>>8
    assert r["compression_ratio"] > 3.0 and 0.0 < r["bpp"] < 8.0
    assert "PSNR" in quality.format_report(r)


def test_oracle_output_matches_source_luma_closely():
    """Against the codec's OWN luma definition the only loss is quantisation: PSNR is higher than against Pillow's 'L'."""
    import jpegamd
    from oracle import oracle
    bmp = jpegamd.synth_bmp(256, 192, 77, 0, 0)
    jpg = oracle.encode_bmp(bmp)
    dec = quality.decode_jpeg(jpg).astype(np.float64)
    y = oracle.stages(bmp)["y"][:192, :256].astype(np.float64) + 128.0
    assert quality.psnr(quality.mse(y, dec)) > 30.0


@pytest.mark.gpu
def test_gpu_streams_decode_and_reconstruct():
    import jpegamd
    for (w, h, seed, kind) in [(512, 384, 5, 0), (203, 117, 9, 1), (64, 64, 3, 3)]:
        bmp = jpegamd.synth_bmp(w, h, seed, kind, 0)
        jpg = jpegamd.encode_bmp_bytes(bmp)
        dec = quality.decode_jpeg(jpg)
        assert dec.shape == (h, w)
        r = quality.analyze(bmp, jpg)
        assert r["psnr"] > (20.0 if kind == 1 else 28.0), r      # uniform noise is the hard case at Q=50
