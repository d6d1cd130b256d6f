"""ctypes binding of libjpegamd.so (include/jpeg_compression.h).

This is plumbing for tests, bench.py and __graft_entry__: every compute call goes through
the C-ABI into the HIP kernels.  There is no Python or CPU implementation of the codec
here; if the shared library is missing the import fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

_PKG_ROOT = Path(__file__).resolve().parents[2]          # .../jpeg-image-compression_amd
# JPEGAMD_LIB: A/B tooling only (tools/gpu_*.sh point it at a variant build of the same library)
LIB_PATH = Path(os.environ.get("JPEGAMD_LIB") or (_PKG_ROOT / "libjpegamd.so"))
HEADER_PATH = _PKG_ROOT.parent / "include" / "jpeg_compression.h"

ORDER_BGR, ORDER_RGB = 0, 1
JFIF_PREFIX_BYTES = 328

ERR_NAMES = {0: "OK", -1: "ERR_ARG", -2: "ERR_NO_DEVICE", -3: "ERR_HIP", -4: "ERR_NOT_INIT", -5: "ERR_TOO_LARGE",
             -6: "ERR_RLE_CAPACITY", -7: "ERR_BMP", -8: "ERR_HUFF_CAPACITY"}


class JpegAmdError(RuntimeError):
    def __init__(self, code: int, what: str):
        super().__init__(f"{what}: {ERR_NAMES.get(code, code)} ({code})")
        self.code = code


class Image(C.Structure):
    _fields_ = [("pixels", C.c_void_p), ("width", C.c_int32), ("height", C.c_int32), ("row_stride", C.c_int32),
                ("bottom_up", C.c_int32), ("channel_order", C.c_int32), ("quality", C.c_int32)]


class Stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("jfif_bytes", "entropy_bits", "stuffed_bytes", "exact_fallbacks",
                                          "ns_transform", "ns_entropy", "ns_pack", "ns_total")]


class DTO(C.Structure):
    """JPEG_COMPRESSION_DTO (dsp_port/jpeg_compression/include/jpeg_compression.h:32-64 + MI355X fields)."""
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("r_phy_ptr", C.c_uint64), ("gb_phy_ptr", C.c_uint64),
                ("y_phy_ptr", C.c_uint64), ("dct_phy_ptr", C.c_uint64), ("quant_phy_ptr", C.c_uint64),
                ("zigzag_phy_ptr", C.c_uint64), ("rle_phy_ptr", C.c_uint64), ("rle_count", C.c_uint32),
                ("huff_phy_ptr", C.c_uint64), ("huff_size", C.c_uint32),
                ("cycles_color_conversion", C.c_uint64), ("cycles_dct", C.c_uint64),
                ("cycles_quantization", C.c_uint64), ("cycles_zigzag", C.c_uint64), ("cycles_rle", C.c_uint64),
                ("cycles_huffman", C.c_uint64), ("cycles_total", C.c_uint64),
                ("row_stride", C.c_int32), ("bottom_up", C.c_int32), ("channel_order", C.c_int32),
                ("quality", C.c_int32)]


class BMPImage(C.Structure):
    """natural_c/include/bmp_handler.h:37-41"""
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("data", C.POINTER(C.c_uint8))]


def _load() -> C.CDLL:
    if not LIB_PATH.exists():
        raise ImportError(f"{LIB_PATH} is missing: build it with `make -C {_PKG_ROOT}` "
                          "(or __graft_entry__.build()); there is no fallback implementation")
    lib = C.CDLL(str(LIB_PATH), mode=getattr(os, "RTLD_NOW", 2))
    u64, i32, i64, vp = C.c_uint64, C.c_int32, C.c_int64, C.c_void_p
    sig = {
        "jpegamd_encoder_create": (i32, [C.POINTER(vp), i32, i32]),
        "jpegamd_encoder_destroy": (i32, [vp]),
        "jpegamd_max_jfif_bytes": (u64, [i32, i32]),
        "jpegamd_encode_async": (i32, [vp, C.POINTER(Image), vp, u64, vp, i32, vp]),
        "jpegamd_encode_batch_async": (i32, [vp, C.POINTER(Image), i32, C.POINTER(C.c_void_p), u64, C.POINTER(C.c_void_p), i32, vp]),
        "jpegamd_encoder_finish": (i32, [vp, C.POINTER(Stats)]),
        "jpegamd_encoder_set_pipeline": (i32, [vp, i32]),
        "jpegamd_encoder_set_profiling": (i32, [vp, i32]),
        "jpegamd_encoder_profile": (i32, [vp, i32, C.POINTER(Stats)]),
        "jpegamd_debug_stages": (i32, [vp, C.POINTER(Image), vp, vp, vp]),
        "jpegamd_debug_dct_exact": (i32, [vp, vp, vp, i64]),
        "jpegamd_synth_bmp": (u64, [i32, i32, C.c_uint32, i32, C.c_uint32, vp, u64]),
        "jpegamd_version": (C.c_char_p, []),
        "jpegamd_debug_quant_table": (i32, [i32, vp]),
        "jpegamd_segment_meta_words": (i32, []),
        "jpegamd_debug_mfma_consts": (i32, [i32, vp, vp, vp, vp]),
        "jpegamd_debug_group_thresholds": (i32, [i32, vp, vp]),
        "jpegamd_debug_mfma_offsets": (i32, [i32, vp, vp, vp, vp]),
        "jpegamd_debug_cos_lut": (i32, [vp]),
        "JpegCompression_Init": (i32, []),
        "JpegCompression_DeInit": (i32, []),
        "JpegCompression_Reserve": (i32, [i32, i32]),
        "convertToJpeg": (i32, [C.POINTER(DTO)]),
        "JpegCompression_RemoteServiceHandler": (i32, [C.c_char_p, C.c_uint32, vp, C.c_uint32, C.c_uint32]),
        "loadBMPImage": (C.POINTER(BMPImage), [C.c_char_p]),
        "freeBMPImage": (None, [C.POINTER(BMPImage)]),
        "saveJPEGGrayscale": (C.c_bool, [C.c_char_p, C.POINTER(BMPImage)]),
        "jpegamd_encode_bmp_memory": (i64, [vp, u64, i32, vp, u64]),
        "jpegamd_parse_bmp": (i32, [vp, u64, C.POINTER(Image), C.POINTER(u64)]),
        "jpegamd_gather_streams": (i32, [vp, i32, i32, i32, vp, u64, i32, vp, vp, u64, vp]),
        "jpegamd_encode_rows_async": (i32, [vp, C.POINTER(Image), i32, i32, vp]),
        "jpegamd_export_segments": (i32, [vp, C.POINTER(Image), i32, i32, vp, u64, vp, vp, vp]),
        "jpegamd_import_segments": (i32, [vp, C.POINTER(Image), i32, i32, vp, vp, vp]),
        "jpegamd_finalize_async": (i32, [vp, C.POINTER(Image), vp, u64, vp, i32, vp]),
    }
    for name, (res, args) in sig.items():
        if name in ("jpegamd_encoder_set_pipeline", "jpegamd_gather_streams", "jpegamd_debug_mfma_offsets") and os.environ.get("JPEGAMD_LIB") and not hasattr(lib, name):
            continue                                              # (A/B tooling: a variant build of an older round)
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    return lib


lib = _load()
EXPORTED = ("jpegamd_encoder_create jpegamd_encoder_destroy jpegamd_max_jfif_bytes jpegamd_encode_async jpegamd_encode_batch_async "
            "jpegamd_encoder_finish jpegamd_encoder_set_pipeline jpegamd_encoder_set_profiling jpegamd_encoder_profile jpegamd_debug_stages jpegamd_debug_dct_exact "
            "jpegamd_synth_bmp jpegamd_version jpegamd_debug_quant_table jpegamd_segment_meta_words jpegamd_debug_mfma_consts jpegamd_debug_group_thresholds jpegamd_debug_mfma_offsets jpegamd_debug_cos_lut JpegCompression_Init JpegCompression_DeInit JpegCompression_Reserve "
            "convertToJpeg JpegCompression_RemoteServiceHandler loadBMPImage freeBMPImage saveJPEGGrayscale "
            "jpegamd_encode_bmp_memory jpegamd_parse_bmp jpegamd_encode_files jpegamd_gather_streams "
            "jpegamd_encode_rows_async jpegamd_export_segments jpegamd_import_segments jpegamd_finalize_async").split()


def quant_table(quality: int = 50):
    """uint8[64], raster order: the reference's table (quality 50) or its libjpeg scaling."""
    import numpy as np
    table = np.zeros(64, np.uint8)
    lib.jpegamd_debug_quant_table(quality, table.ctypes.data)
    return table


PIPELINE_AUTO, PIPELINE_PAIR, PIPELINE_STITCH = 0, 1, 2                    # JPEGAMD_PIPELINE_*
MAX_BATCH = 32                                               # JPEGAMD_MAX_BATCH
SEG_META_WORDS = int(lib.jpegamd_segment_meta_words())      # metadata words per segment in the sharded-image exchange


def mfma_consts(quality: int = 50):
    """Constants of the matrix-pipe kernel: qmul/qthr/bias/zoff/qadd float32[64] by zigzag position, delta float64[64] by raster k,
    dc_off and scale (accumulator units)."""
    import numpy as np
    qmul, qthr = np.zeros(64, np.float32), np.zeros(64, np.float32)
    bias, delta = np.zeros(64, np.float32), np.zeros(64, np.float64)
    lib.jpegamd_debug_mfma_consts(quality, qmul.ctypes.data, qthr.ctypes.data, bias.ctypes.data, delta.ctypes.data)
    zoff, qadd = np.zeros(64, np.float32), np.zeros(64, np.float32)
    dc_off, scale = np.zeros(1, np.float32), np.zeros(1, np.float32)
    lib.jpegamd_debug_mfma_offsets(quality, zoff.ctypes.data, qadd.ctypes.data, dc_off.ctypes.data, scale.ctypes.data)
    return dict(qmul=qmul, qthr=qthr, bias=bias, delta=delta, zoff=zoff, qadd=qadd, dc_off=float(dc_off[0]), scale=float(scale[0]))


def group_thresholds(quality: int = 50, with_lo_bound: bool = False):
    """float32 [4][2]: zero threshold of coefficient group G for lane half h, on the hi chain (with_lo_bound: and the lo chain's bound)."""
    import numpy as np
    t, lo = np.zeros(8, np.float32), np.zeros(8, np.float32)
    lib.jpegamd_debug_group_thresholds(quality, t.ctypes.data, lo.ctypes.data)
    return (t.reshape(4, 2), lo.reshape(4, 2)) if with_lo_bound else t.reshape(4, 2)


def cos_lut():
    """float32 [8][8]: COS_LUT[x][u] as compiled into the kernels."""
    import numpy as np
    t = np.zeros(64, np.float32)
    lib.jpegamd_debug_cos_lut(t.ctypes.data)
    return t.reshape(8, 8)


def version() -> str:
    return lib.jpegamd_version().decode()


def synth_bmp(width: int, height: int, seed: int = 1, kind: int = 0, flags: int = 0) -> bytes:
    """Deterministic synthetic 24-bit BMP file (host code in the library; no device needed)."""
    n = lib.jpegamd_synth_bmp(width, height, seed, kind, flags, None, 0)
    if n == 0:
        raise ValueError("bad synth_bmp arguments")
    buf = (C.c_uint8 * n)()
    got = lib.jpegamd_synth_bmp(width, height, seed, kind, flags, buf, n)
    assert got == n
    return bytes(buf)


def synth_bmp_into(ptr: int, cap: int, width: int, height: int, seed: int = 1, kind: int = 0, flags: int = 0) -> int:
    """Generate straight into caller memory (e.g. a pinned torch tensor); returns file size."""
    return int(lib.jpegamd_synth_bmp(width, height, seed, kind, flags, C.c_void_p(ptr), cap))


def parse_bmp(data: bytes):
    """-> (Image view with pixels=offset, pixel_offset) using loadBMPImage's header rules."""
    img, off = Image(), C.c_uint64(0)
    rc = lib.jpegamd_parse_bmp(data, len(data), C.byref(img), C.byref(off))
    if rc:
        raise JpegAmdError(rc, "jpegamd_parse_bmp")
    return img, off.value


def max_jfif_bytes(width: int, height: int) -> int:
    return int(lib.jpegamd_max_jfif_bytes(width, height))


def encode_bmp_bytes(bmp: bytes, quality: int = 0) -> bytes:
    """BMP file bytes -> JFIF file bytes through the device (upload, encode, download)."""
    img, _ = parse_bmp(bmp)
    cap = 4096 + img.width * img.height * 2
    for _ in range(2):
        out = (C.c_uint8 * cap)()
        n = lib.jpegamd_encode_bmp_memory(bmp, len(bmp), quality, out, cap)
        if n == -8:
            cap = max_jfif_bytes(img.width, img.height)
            continue
        if n < 0:
            raise JpegAmdError(int(n), "jpegamd_encode_bmp_memory")
        return bytes(out[:n])
    raise JpegAmdError(-8, "jpegamd_encode_bmp_memory")


class BatchStats(C.Structure):
    _fields_ = [("files_ok", C.c_int32), ("files_failed", C.c_int32), ("bytes_in", C.c_uint64), ("bytes_out", C.c_uint64),
                ("seconds_total", C.c_double), ("seconds_read", C.c_double), ("seconds_write", C.c_double)]


def encode_files(in_paths, out_paths, quality: int = 0):
    """File-to-file batch with overlapped I/O and transfers (jpegamd_encode_files) -> (return code, [status per file], BatchStats)."""
    n = len(in_paths)
    if len(out_paths) != n:
        raise ValueError("in_paths and out_paths differ in length")
    ins = (C.c_char_p * n)(*[str(p).encode() for p in in_paths])
    outs = (C.c_char_p * n)(*[str(p).encode() for p in out_paths])
    status = (C.c_int32 * n)()
    st = BatchStats()
    fn = lib.jpegamd_encode_files
    fn.restype = C.c_int32
    fn.argtypes = [C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.POINTER(BatchStats)]
    rc = fn(ins, outs, n, quality, status, C.byref(st))
    return int(rc), list(status), st


class Encoder:
    """Level-1 context: device pointers in, device pointers out, stream-ordered."""

    def __init__(self, max_width: int, max_height: int):
        self._h = C.c_void_p()
        rc = lib.jpegamd_encoder_create(C.byref(self._h), max_width, max_height)
        if rc:
            raise JpegAmdError(rc, "jpegamd_encoder_create")

    def close(self):
        if self._h:
            lib.jpegamd_encoder_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_pipeline(self, pipeline: int):
        """PIPELINE_AUTO / PIPELINE_PAIR (k_segment_merge + k_finalize) / PIPELINE_STITCH (k_stitch): which kernels follow k_tile_encode."""
        rc = lib.jpegamd_encoder_set_pipeline(self._h, int(pipeline))
        if rc:
            raise JpegAmdError(rc, "jpegamd_encoder_set_pipeline")

    def set_profiling(self, slots: int):
        rc = lib.jpegamd_encoder_set_profiling(self._h, int(slots))
        if rc:
            raise JpegAmdError(rc, "jpegamd_encoder_set_profiling")

    def profile(self, slot: int) -> Stats:
        st = Stats()
        rc = lib.jpegamd_encoder_profile(self._h, slot, C.byref(st))
        if rc:
            raise JpegAmdError(rc, "jpegamd_encoder_profile")
        return st

    @staticmethod
    def image(pixels_ptr: int, width: int, height: int, row_stride: int, bottom_up: bool = True,
              channel_order: int = ORDER_BGR, quality: int = 0) -> Image:
        return Image(pixels_ptr, width, height, row_stride, 1 if bottom_up else 0, channel_order, quality)

    def encode_async(self, img: Image, out_ptr: int, out_cap: int, size_ptr: int, with_container: bool = True,
                     stream: int = 0):
        rc = lib.jpegamd_encode_async(self._h, C.byref(img), C.c_void_p(out_ptr), out_cap, C.c_void_p(size_ptr),
                                      1 if with_container else 0, C.c_void_p(stream))
        if rc:
            raise JpegAmdError(rc, "jpegamd_encode_async")

    def encode_batch_async(self, imgs, out_ptrs, out_cap: int, size_ptrs, with_container: bool = True, stream: int = 0):
        """`len(imgs)` images of one geometry (<= MAX_BATCH) through ONE launch of each kernel; the context must hold
        len(imgs) x the tiles and segments of one image (e.g. Encoder(W, len(imgs) * H))."""
        n = len(imgs)
        arr = (Image * n)(*imgs)
        outs = (C.c_void_p * n)(*[C.c_void_p(p) for p in out_ptrs])
        sizes = (C.c_void_p * n)(*[C.c_void_p(p) for p in size_ptrs])
        rc = lib.jpegamd_encode_batch_async(self._h, arr, n, outs, out_cap, sizes, 1 if with_container else 0, C.c_void_p(stream))
        if rc:
            raise JpegAmdError(rc, "jpegamd_encode_batch_async")

    # ---- one image sharded over GPUs by block rows (include/jpeg_compression.h) ----
    def encode_rows_async(self, img: Image, row_begin: int, row_end: int, stream: int = 0):
        rc = lib.jpegamd_encode_rows_async(self._h, C.byref(img), C.c_int32(row_begin), C.c_int32(row_end), C.c_void_p(stream))
        if rc:
            raise JpegAmdError(rc, "jpegamd_encode_rows_async")

    def export_segments(self, img: Image, row_begin: int, row_end: int, dense_ptr: int, dense_cap_words: int, meta_ptr: int,
                        total_ptr: int, stream: int = 0):
        rc = lib.jpegamd_export_segments(self._h, C.byref(img), C.c_int32(row_begin), C.c_int32(row_end), C.c_void_p(dense_ptr),
                                         C.c_uint64(dense_cap_words), C.c_void_p(meta_ptr), C.c_void_p(total_ptr), C.c_void_p(stream))
        if rc:
            raise JpegAmdError(rc, "jpegamd_export_segments")

    def import_segments(self, img: Image, row_begin: int, row_end: int, dense_ptr: int, meta_ptr: int, stream: int = 0):
        rc = lib.jpegamd_import_segments(self._h, C.byref(img), C.c_int32(row_begin), C.c_int32(row_end), C.c_void_p(dense_ptr),
                                         C.c_void_p(meta_ptr), C.c_void_p(stream))
        if rc:
            raise JpegAmdError(rc, "jpegamd_import_segments")

    def finalize_async(self, img: Image, out_ptr: int, out_cap: int, size_ptr: int, with_container: bool = True, stream: int = 0):
        rc = lib.jpegamd_finalize_async(self._h, C.byref(img), C.c_void_p(out_ptr), C.c_uint64(out_cap), C.c_void_p(size_ptr),
                                        1 if with_container else 0, C.c_void_p(stream))
        if rc:
            raise JpegAmdError(rc, "jpegamd_finalize_async")

    def finish(self) -> Stats:
        st = Stats()
        rc = lib.jpegamd_encoder_finish(self._h, C.byref(st))
        if rc:
            raise JpegAmdError(rc, "jpegamd_encoder_finish")
        return st

    def debug_stages(self, img: Image, y_ptr: int = 0, zz_ptr: int = 0, mask_ptr: int = 0):
        rc = lib.jpegamd_debug_stages(self._h, C.byref(img), C.c_void_p(y_ptr), C.c_void_p(zz_ptr),
                                      C.c_void_p(mask_ptr))
        if rc:
            raise JpegAmdError(rc, "jpegamd_debug_stages")

    def debug_dct_exact(self, blocks_ptr: int, coeffs_ptr: int, nblocks: int):
        rc = lib.jpegamd_debug_dct_exact(self._h, C.c_void_p(blocks_ptr), C.c_void_p(coeffs_ptr), nblocks)
        if rc:
            raise JpegAmdError(rc, "jpegamd_debug_dct_exact")
