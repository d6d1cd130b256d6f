"""ctypes wrapper of oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module.  It is the checker (a CPU restatement of the reference's natural_c path, pinned
against the compiled reference, see natural_oracle.h), never the product.
"""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
LIB = HERE / "liboracle.so"
REF_APP = HERE / "_ref" / "jpeg_compression_app"
REF_LIB = HERE / "_ref" / "libnatural_c_ref.so"
REF_LIB_O2 = HERE / "_ref" / "libnatural_c_ref_O2.so"      # the reference's sources with -O2 added (second CPU baseline)


def build(ref: bool = True) -> None:
    """Compile the restatement; when /root/reference is present also (re)build oracle/_ref."""
    targets = ["all"] + (["ref"] if ref else [])
    subprocess.run(["make", "-s", "-C", str(HERE)] + targets, check=True)


class _View(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("top_down", C.c_int32), ("row_stride", C.c_int32),
                ("pixels", C.c_void_p)]


class _Sym(C.Structure):
    _fields_ = [("symbol", C.c_uint8), ("code_bits", C.c_uint8), ("code", C.c_uint16)]


def _load():
    if not LIB.exists():
        build(ref=False)
    lib = C.CDLL(str(LIB))
    lib.oracle_parse_bmp.restype = C.c_int
    lib.oracle_parse_bmp.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(_View)]
    lib.oracle_encode_bmp.restype = C.c_long
    lib.oracle_encode_bmp.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_size_t]
    lib.oracle_max_jfif_bytes.restype = C.c_size_t
    lib.oracle_max_jfif_bytes.argtypes = [C.c_int, C.c_int]
    lib.oracle_quant_table.argtypes = [C.c_int, C.c_void_p]
    lib.oracle_luma_centered.argtypes = [C.POINTER(_View), C.c_void_p]
    lib.oracle_dct_image.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    lib.oracle_quant_image.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    lib.oracle_zigzag_image.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    lib.oracle_rle.restype = C.c_long
    lib.oracle_rle.argtypes = [C.c_void_p, C.c_long, C.c_void_p, C.c_long]
    lib.oracle_entropy.restype = C.c_long
    lib.oracle_entropy.argtypes = [C.c_void_p, C.c_long, C.c_void_p, C.c_size_t]
    lib.oracle_jfif_prefix.restype = C.c_size_t
    lib.oracle_jfif_prefix.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    return lib


_lib = _load()


class OracleError(RuntimeError):
    def __init__(self, code):
        super().__init__(f"oracle error {code}")
        self.code = code


def encode_bmp(bmp: bytes, quality: int = 50) -> bytes:
    """BMP file bytes -> JFIF file bytes, as natural_c's jpeg_compression_app would write them."""
    v = _View()
    rc = _lib.oracle_parse_bmp(bmp, len(bmp), C.byref(v))
    if rc:
        raise OracleError(rc)
    cap = min(_lib.oracle_max_jfif_bytes(v.width, v.height), 4096 + 4 * v.width * v.height)
    out = (C.c_uint8 * cap)()
    n = _lib.oracle_encode_bmp(bmp, len(bmp), quality, out, cap)
    if n < 0:
        raise OracleError(n)
    return bytes(out[:n])


def parse(bmp: bytes):
    v = _View()
    rc = _lib.oracle_parse_bmp(bmp, len(bmp), C.byref(v))
    if rc:
        raise OracleError(rc)
    return v


def quant_table(quality: int = 50) -> np.ndarray:
    t = np.zeros(64, np.uint8)
    _lib.oracle_quant_table(quality, t.ctypes.data)
    return t


def stages(bmp: bytes, quality: int = 50):
    """-> dict(y int8[PH,PW], dct f32[PH,PW], quant i16[PH,PW], zigzag i16[NB,64]) whole-image stage outputs."""
    buf = C.create_string_buffer(bmp, len(bmp))
    v = _View()
    rc = _lib.oracle_parse_bmp(buf, len(bmp), C.byref(v))
    if rc:
        raise OracleError(rc)
    pw, ph = (v.width + 7) & ~7, (v.height + 7) & ~7
    y = np.zeros((ph, pw), np.int8)
    _lib.oracle_luma_centered(C.byref(v), y.ctypes.data)
    d = np.zeros((ph, pw), np.float32)
    _lib.oracle_dct_image(y.ctypes.data, pw, ph, d.ctypes.data)
    qt = quant_table(quality)
    q = np.zeros((ph, pw), np.int16)
    _lib.oracle_quant_image(d.ctypes.data, pw, ph, qt.ctypes.data, q.ctypes.data)
    zz = np.zeros(((pw // 8) * (ph // 8), 64), np.int16)
    _lib.oracle_zigzag_image(q.ctypes.data, pw, ph, zz.ctypes.data)
    return dict(y=y, dct=d, quant=q, zigzag=zz, width=v.width, height=v.height)


def dct_blocks(blocks: np.ndarray) -> np.ndarray:
    """Exact-order float32 DCT of int8 blocks [N,8,8] -> float32 [N,8,8] (dct.c:63-96)."""
    blocks = np.ascontiguousarray(blocks, np.int8)
    n = blocks.shape[0]
    img = np.ascontiguousarray(blocks.reshape(n * 8, 8))          # a PW=8, PH=8n image
    out = np.zeros((n * 8, 8), np.float32)
    _lib.oracle_dct_image(img.ctypes.data, 8, n * 8, out.ctypes.data)
    return out.reshape(n, 8, 8)


def rle_symbols(zigzag: np.ndarray):
    zz = np.ascontiguousarray(zigzag, np.int16)
    nb = zz.shape[0]
    cap = nb * 70
    arr = (_Sym * cap)()
    n = _lib.oracle_rle(zz.ctypes.data, nb, arr, cap)
    if n < 0:
        raise OracleError(n)
    return [(arr[i].symbol, arr[i].code, arr[i].code_bits) for i in range(n)]


def entropy(zigzag: np.ndarray) -> bytes:
    zz = np.ascontiguousarray(zigzag, np.int16)
    nb = zz.shape[0]
    cap = nb * 2 * 216 + 64
    out = (C.c_uint8 * cap)()
    n = _lib.oracle_entropy(zz.ctypes.data, nb, out, cap)
    if n < 0:
        raise OracleError(n)
    return bytes(out[:n])


def jfif_prefix(width: int, height: int, quality: int = 50) -> bytes:
    qt = quant_table(quality)
    out = (C.c_uint8 * 400)()
    n = _lib.oracle_jfif_prefix(width, height, qt.ctypes.data, out)
    return bytes(out[:n])


def reference_app_encode(bmp: bytes, tmpdir) -> bytes:
    """Run the compiled reference (oracle/_ref/jpeg_compression_app) on a BMP; needs oracle/_ref."""
    tmpdir = Path(tmpdir)
    src, dst = tmpdir / "in.bmp", tmpdir / "out.jpg"
    src.write_bytes(bmp)
    subprocess.run([str(REF_APP), str(src), str(dst)], check=True, stdout=subprocess.DEVNULL)
    return dst.read_bytes()
