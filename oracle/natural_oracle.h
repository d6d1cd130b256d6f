/*
 * natural_oracle.h -- CPU restatement of the reference's natural_c BMP -> grayscale
 * baseline-JPEG path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it; the product library (libjpegamd.so) never
 * links or calls anything in oracle/.
 *
 * Parity status: PINNED.  oracle/Makefile target `ref` compiles the reference's own
 * natural_c sources (from /root/reference, with natural_c/Makefile:4 flags) into
 * oracle/_ref/, tests/test_oracle_vs_ref.py compares this restatement byte-for-byte with
 * it, and tests/golden/ holds JFIF outputs produced by that reference build
 * (tests/golden/make_goldens.py is the generating script).
 *
 * All file:line citations are into /root/reference/natural_c/.
 */
#ifndef NATURAL_ORACLE_H
#define NATURAL_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* A parsed view of a 24-bit uncompressed BMP held in memory (src/io/bmp_handler.c:15-129). */
typedef struct {
    int32_t width;        /* biWidth */
    int32_t height;       /* |biHeight| */
    int32_t top_down;     /* 1 when biHeight < 0 (bmp_handler.c:68-72) */
    int32_t row_stride;   /* (3*width + 3) & ~3 (bmp_handler.c:75) */
    const uint8_t *pixels;/* file + bfOffBits (bmp_handler.c:88), BGR order */
} OracleBmpView;

/* Error codes shared by every entry point (0 = ok). */
enum {
    ORACLE_OK = 0,
    ORACLE_ERR_SHORT = -1,       /* truncated file / insufficient row data */
    ORACLE_ERR_MAGIC = -2,       /* bfType != 'BM' (bmp_handler.c:30) */
    ORACLE_ERR_BITCOUNT = -3,    /* biBitCount != 24 (bmp_handler.c:44) */
    ORACLE_ERR_COMPRESSED = -4,  /* biCompression != 0 (bmp_handler.c:49) */
    ORACLE_ERR_CAPACITY = -5,    /* output buffer too small */
    ORACLE_ERR_ARG = -6
};

int oracle_parse_bmp(const uint8_t *file, size_t file_len, OracleBmpView *out);

/* Quantisation table for a libjpeg-style quality (extension; Q=50 == reference table,
 * src/core/jpeg_tables.c:3-12).  Raster (u*8+v) order. */
void oracle_quant_table(int quality, uint8_t table[64]);

/* Stage outputs (whole image), for per-stage parity tests.  PW/PH = dims padded to 8. */
void oracle_padded_dims(const OracleBmpView *v, int *pw, int *ph);
/* converter.c:4-58 + :60-90 : edge-replicated luma minus 128, int8 [PH][PW]. */
void oracle_luma_centered(const OracleBmpView *v, int8_t *out);
/* dct.c:63-96 : float32 coefficients, image layout [PH][PW]. */
void oracle_dct_image(const int8_t *centered, int pw, int ph, float *out);
/* quantization.c:34-36 : int16 image layout [PH][PW]. */
void oracle_quant_image(const float *dct, int pw, int ph, const uint8_t qt[64], int16_t *out);
/* zigzag.c:51-61 : int16 [NB][64] block-major. */
void oracle_zigzag_image(const int16_t *quant, int pw, int ph, int16_t *out);

/* RLE symbol record laid out like include/rle.h:8-14 (packed by hand: 4 bytes here). */
typedef struct {
    uint8_t symbol;
    uint8_t code_bits;
    uint16_t code;
} OracleRleSymbol;
/* rle.c:51-127.  Returns number of symbols, or ORACLE_ERR_CAPACITY. */
long oracle_rle(const int16_t *zz, long nblocks, OracleRleSymbol *out, long cap);

/* huffman.c:121-193 : entropy-coded segment (stuffed, zero-bit flushed) from zigzag data.
 * Returns byte count or a negative error. */
long oracle_entropy(const int16_t *zz, long nblocks, uint8_t *out, size_t cap);

/* jpeg_handler.c:7-110 : the 328-byte APP0+DQT+SOF0+DHT+DHT+SOS prefix. */
size_t oracle_jfif_prefix(int width, int height, const uint8_t qt[64], uint8_t *out);

/* Whole path: BMP file bytes -> JFIF file bytes (prefix + segment + EOI).
 * Streams block row by block row; no whole-image intermediates.
 * Returns total byte count or a negative error. */
long oracle_encode_bmp(const uint8_t *file, size_t file_len, int quality,
                       uint8_t *out, size_t cap);

/* Upper bound on oracle_encode_bmp output for a WxH image. */
size_t oracle_max_jfif_bytes(int width, int height);

#ifdef __cplusplus
}
#endif
#endif
