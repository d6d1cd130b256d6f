"""Quality / decodability harness for emitted JPEGs.

Reports what the reference's analysis script reports for an (original BMP, compressed JPEG) pair
(analyze_results.py:17-32 MSE / PSNR, :66-84 compression ratio, bits per pixel, SSIM), with an
independent decoder (Pillow) on the JPEG side, so a container or entropy-coding error that a byte
comparison against a wrong golden would miss shows up as an undecodable file or a collapsed PSNR.
SSIM is computed here (uniform 7x7 window, K1 = 0.01, K2 = 0.03, sample covariance: the defaults
of the library the reference calls) because scikit-image is not part of this environment.
Test / tooling code only: nothing on the encode path imports this module.
"""
import io
import math

import numpy as np


def _gray(img):
    return np.asarray(img.convert("L"), dtype=np.float64)


def mse(a: np.ndarray, b: np.ndarray) -> float:
    d = np.asarray(a, np.float64) - np.asarray(b, np.float64)
    return float(np.mean(d * d))


def psnr(mse_value: float, peak: float = 255.0) -> float:
    return float("inf") if mse_value == 0 else 20.0 * math.log10(peak / math.sqrt(mse_value))


def ssim(a: np.ndarray, b: np.ndarray, data_range: float = 255.0, win: int = 7) -> float:
    """Mean structural similarity (Wang et al. 2004), uniform window, borders cropped by win // 2."""
    from scipy.ndimage import uniform_filter

    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    if min(a.shape) < win:
        return float("nan")
    npix = win * win
    cov_norm = npix / (npix - 1.0)
    ua, ub = uniform_filter(a, win), uniform_filter(b, win)
    uaa, ubb, uab = uniform_filter(a * a, win), uniform_filter(b * b, win), uniform_filter(a * b, win)
    va, vb, vab = cov_norm * (uaa - ua * ua), cov_norm * (ubb - ub * ub), cov_norm * (uab - ua * ub)
    c1, c2 = (0.01 * data_range) ** 2, (0.03 * data_range) ** 2
    s = ((2 * ua * ub + c1) * (2 * vab + c2)) / ((ua * ua + ub * ub + c1) * (va + vb + c2))
    p = win // 2
    return float(np.mean(s[p:s.shape[0] - p, p:s.shape[1] - p]))


def decode_jpeg(jpeg: bytes) -> np.ndarray:
    """Decode with Pillow (independent of this repository's encoder and oracle) -> uint8 [H, W]."""
    from PIL import Image

    img = Image.open(io.BytesIO(jpeg))
    img.load()
    if img.format != "JPEG" or img.mode != "L":
        raise ValueError(f"not a grayscale baseline JPEG: format={img.format} mode={img.mode}")
    return np.asarray(img, dtype=np.uint8)


def analyze(bmp: bytes, jpeg: bytes) -> dict:
    """Metrics of the reference's report for one pair; the original is reduced to 'L' by Pillow like there."""
    from PIL import Image

    orig = Image.open(io.BytesIO(bmp))
    comp = Image.open(io.BytesIO(jpeg))
    comp.load()
    if orig.size != comp.size:                       # analyze_results.py:60-62 resizes; here it is an error
        raise ValueError(f"dimensions differ: original {orig.size}, compressed {comp.size}")
    go, gc = _gray(orig), _gray(comp)
    m = mse(go, gc)
    w, h = orig.size
    return {"width": w, "height": h, "size_original": len(bmp), "size_compressed": len(jpeg),
            "compression_ratio": len(bmp) / len(jpeg) if jpeg else 0.0, "bpp": 8.0 * len(jpeg) / (w * h),
            "mse": m, "psnr": psnr(m), "ssim": ssim(go, gc)}


def format_report(r: dict) -> str:
    lines = ["-" * 50, "ANALYSIS RESULTS", "-" * 50,
             f"File Size Orig : {r['size_original']} bytes", f"File Size Comp : {r['size_compressed']} bytes",
             f"Compression    : {r['compression_ratio']:.2f} : 1", f"Bits per pixel : {r['bpp']:.4f}",
             f"MSE            : {r['mse']:.4f}", f"PSNR           : {r['psnr']:.2f} dB", f"SSIM           : {r['ssim']:.4f}", "-" * 50]
    return "\n".join(lines)
