"""ctypes view of the GPU-backed natural_c stage functions (include/natural_c_stages.h).

Mirrors how natural_c's own driver chains them (src/io/jpeg_handler.c:119-282): every call takes the previous
stage's struct and returns a freshly allocated one (freed here as soon as its content is copied into numpy).
Used by the stage-level parity tests and for debugging; the production path is jpegamd.Encoder."""
import ctypes as C

import numpy as np

from . import BMPImage, lib


class YImage(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("data", C.POINTER(C.c_uint8))]


class CenteredYImage(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("data", C.POINTER(C.c_int8))]


class DCTImage(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("coefficients", C.POINTER(C.c_float))]


class QuantizedImage(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("data", C.POINTER(C.c_int16))]


class ZigZagData(C.Structure):
    _fields_ = [("numBlocksW", C.c_int), ("numBlocksH", C.c_int), ("totalBlocks", C.c_int), ("data", C.POINTER(C.c_int16))]


class RLESymbol(C.Structure):
    _fields_ = [("symbol", C.c_uint8), ("code", C.c_uint16), ("codeBits", C.c_uint8)]


class RLEData(C.Structure):
    _fields_ = [("data", C.POINTER(RLESymbol)), ("count", C.c_size_t), ("capacity", C.c_size_t)]


class JpegEncoderBuffer(C.Structure):
    _fields_ = [("data", C.POINTER(C.c_uint8)), ("size", C.c_size_t), ("capacity", C.c_size_t)]


STAGE_EXPORTS = ("convertBMPToJPEGGrayscale centerYImage performDCT computeDCTBlock quantizeImage performZigZag performRLE "
                 "encodeHuffman freeYImage freeCenteredYImage freeDCTImage freeQuantizedImage freeZigZagData freeRLEData "
                 "freeJpegEncoderBuffer").split()

_P = C.POINTER
for _name, _res, _args in (("convertBMPToJPEGGrayscale", _P(YImage), [_P(BMPImage)]), ("centerYImage", _P(CenteredYImage), [_P(YImage)]),
                           ("performDCT", _P(DCTImage), [_P(CenteredYImage)]), ("quantizeImage", _P(QuantizedImage), [_P(DCTImage)]),
                           ("performZigZag", _P(ZigZagData), [_P(QuantizedImage)]), ("performRLE", _P(RLEData), [_P(ZigZagData)]),
                           ("encodeHuffman", _P(JpegEncoderBuffer), [_P(RLEData), C.c_int]), ("computeDCTBlock", None, [C.c_void_p, C.c_void_p]),
                           ("freeYImage", None, [_P(YImage)]), ("freeCenteredYImage", None, [_P(CenteredYImage)]),
                           ("freeDCTImage", None, [_P(DCTImage)]), ("freeQuantizedImage", None, [_P(QuantizedImage)]),
                           ("freeZigZagData", None, [_P(ZigZagData)]), ("freeRLEData", None, [_P(RLEData)]),
                           ("freeJpegEncoderBuffer", None, [_P(JpegEncoderBuffer)])):
    _fn = getattr(lib, _name)
    _fn.restype, _fn.argtypes = _res, _args


def _np(ptr, n, dtype):
    return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dtype, copy=True) if n else np.zeros(0, dtype)


def run_stages(rgb: np.ndarray) -> dict:
    """rgb uint8 [H, W, 3] (a loaded BMPImage's layout) through all seven stage functions -> dict of numpy results."""
    rgb = np.ascontiguousarray(rgb, np.uint8)
    h, w = rgb.shape[:2]
    bmp = BMPImage(w, h, rgb.ctypes.data_as(C.POINTER(C.c_uint8)))
    out = {}
    y = lib.convertBMPToJPEGGrayscale(C.byref(bmp))
    if not y:
        raise RuntimeError("convertBMPToJPEGGrayscale returned NULL")
    pw, ph = y.contents.width, y.contents.height
    out["y"] = _np(y.contents.data, pw * ph, np.uint8).reshape(ph, pw)
    c = lib.centerYImage(y)
    out["centered"] = _np(c.contents.data, pw * ph, np.int8).reshape(ph, pw)
    d = lib.performDCT(c)
    out["dct"] = _np(d.contents.coefficients, pw * ph, np.float32).reshape(ph, pw)
    q = lib.quantizeImage(d)
    out["quant"] = _np(q.contents.data, pw * ph, np.int16).reshape(ph, pw)
    z = lib.performZigZag(q)
    nb = z.contents.totalBlocks
    out["blocks"] = (z.contents.numBlocksW, z.contents.numBlocksH, nb)
    out["zigzag"] = _np(z.contents.data, nb * 64, np.int16).reshape(nb, 64)
    r = lib.performRLE(z)
    n = r.contents.count
    out["rle"] = [(r.contents.data[i].symbol, r.contents.data[i].code, r.contents.data[i].codeBits) for i in range(n)] if n < 200000 else None
    out["rle_count"] = n
    e = lib.encodeHuffman(r, nb)
    out["entropy"] = bytes(_np(e.contents.data, e.contents.size, np.uint8))
    for fn, obj in ((lib.freeJpegEncoderBuffer, e), (lib.freeRLEData, r), (lib.freeZigZagData, z), (lib.freeQuantizedImage, q),
                    (lib.freeDCTImage, d), (lib.freeCenteredYImage, c), (lib.freeYImage, y)):
        fn(obj)
    return out
