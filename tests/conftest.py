"""Shared fixtures.  `-m "not gpu"` covers the oracle, goldens, host logic and the C-ABI surface on CPU;
`-m gpu` holds the parity tests proper: every one of them calls through the C-ABI into the HIP kernels."""
from __future__ import annotations

import json
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
PKG = ROOT / "jpeg-image-compression_amd"
GOLDEN = ROOT / "tests" / "golden"
for p in (str(PKG / "python"), str(ROOT)):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box)")
    # Build what is missing (fresh checkout): product library and the oracle.  Building the
    # checker is not using it.
    if not (PKG / "libjpegamd.so").exists() or not (PKG / "jpeg_compression_app").exists():
        subprocess.run(["make", "-s", "-C", str(PKG)], check=True)
    if not (ROOT / "oracle" / "liboracle.so").exists():
        subprocess.run(["make", "-s", "-C", str(ROOT / "oracle"), "all"], check=True)


@pytest.fixture(scope="session")
def jpegamd():
    import jpegamd as m
    return m


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as o
    return o


@pytest.fixture(scope="session")
def manifest():
    return json.loads((GOLDEN / "manifest.json").read_text())


def fixture_bmp(entry, jpegamd) -> bytes:
    """Input bytes of a manifest entry: regenerated (synthetic) or read (asset crop)."""
    if entry["source"] == "synth":
        return jpegamd.synth_bmp(entry["width"], entry["height"], entry["seed"], entry["kind"], entry["flags"])
    return (GOLDEN / f"{entry['name']}.bmp").read_bytes()


def golden_jpg(entry) -> bytes:
    return (GOLDEN / f"{entry['name']}.jpg").read_bytes()


def has_gpu() -> bool:
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False
