/*
 * natural_oracle.c -- CPU restatement of the reference's natural_c encoder.
 *
 * TEST INFRASTRUCTURE ONLY (see natural_oracle.h).  Parity status: PINNED against the
 * compiled reference (oracle/_ref) and the goldens under tests/golden/.
 *
 * Structure is this repo's own: one streaming pass per block row feeding a 64-bit bit
 * sink, plus whole-image "stage" helpers used only by per-stage parity tests.  The
 * arithmetic follows the reference statement by statement where float32 rounding order
 * matters; every such place cites the reference line.  Build with -ffp-contract=off and
 * never with -ffast-math.
 */
#include "natural_oracle.h"

#include <math.h>
#include <string.h>

/* ------------------------------------------------------------------------------------
 * Tables (data; values must equal the reference's, tests/test_oracle.py::test_tables_equal_reference_text
 * checks them against the reference text when /root/reference is present).
 * ---------------------------------------------------------------------------------- */

/* Annex-K luminance table, raster order (src/core/jpeg_tables.c:3-12). */
static const uint8_t kBaseQuant[64] = {
    16, 11, 10, 16, 24,  40,  51,  61,  12, 12, 14, 19, 26,  58,  60,  55,
    14, 13, 16, 24, 40,  57,  69,  56,  14, 17, 22, 29, 51,  87,  80,  62,
    18, 22, 37, 56, 68,  109, 103, 77,  24, 35, 55, 64, 81,  104, 113, 92,
    49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99};

/* zigzag position -> raster index (src/core/zigzag.c:7-15; jpeg_handler.c:25-34). */
static const uint8_t kZigzag[64] = {
    0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
    41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
    30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

/* Huffman specs (src/core/jpeg_tables.c:14-48). */
static const uint8_t kDcCounts[16] = {0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0};
static const uint8_t kDcSymbols[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
static const uint8_t kAcCounts[16] = {0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7D};
static const uint8_t kAcSymbols[162] = {
    0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51,
    0x61, 0x07, 0x22, 0x71, 0x14, 0x32, 0x81, 0x91, 0xA1, 0x08, 0x23, 0x42, 0xB1, 0xC1,
    0x15, 0x52, 0xD1, 0xF0, 0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0A, 0x16, 0x17, 0x18,
    0x19, 0x1A, 0x25, 0x26, 0x27, 0x28, 0x29, 0x2A, 0x34, 0x35, 0x36, 0x37, 0x38, 0x39,
    0x3A, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4A, 0x53, 0x54, 0x55, 0x56, 0x57,
    0x58, 0x59, 0x5A, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6A, 0x73, 0x74, 0x75,
    0x76, 0x77, 0x78, 0x79, 0x7A, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8A, 0x92,
    0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9A, 0xA2, 0xA3, 0xA4, 0xA5, 0xA6, 0xA7,
    0xA8, 0xA9, 0xAA, 0xB2, 0xB3, 0xB4, 0xB5, 0xB6, 0xB7, 0xB8, 0xB9, 0xBA, 0xC2, 0xC3,
    0xC4, 0xC5, 0xC6, 0xC7, 0xC8, 0xC9, 0xCA, 0xD2, 0xD3, 0xD4, 0xD5, 0xD6, 0xD7, 0xD8,
    0xD9, 0xDA, 0xE1, 0xE2, 0xE3, 0xE4, 0xE5, 0xE6, 0xE7, 0xE8, 0xE9, 0xEA, 0xF1, 0xF2,
    0xF3, 0xF4, 0xF5, 0xF6, 0xF7, 0xF8, 0xF9, 0xFA};

/* The reference's cosine LUT (src/core/dct.c:9-18), stored frequency-major here:
 * kCos[u][x] == COS_LUT[x][u].  The six-decimal literals are NOT symmetric
 * (-0.382684 / 0.195091 / -0.923879) and that asymmetry is part of the parity target. */
static const float kCos[8][8] = {
    {1.000000f, 1.000000f, 1.000000f, 1.000000f, 1.000000f, 1.000000f, 1.000000f, 1.000000f},
    {0.980785f, 0.831470f, 0.555570f, 0.195090f, -0.195090f, -0.555570f, -0.831470f, -0.980785f},
    {0.923880f, 0.382683f, -0.382683f, -0.923880f, -0.923880f, -0.382684f, 0.382684f, 0.923880f},
    {0.831470f, -0.195090f, -0.980785f, -0.555570f, 0.555570f, 0.980785f, 0.195091f, -0.831470f},
    {0.707107f, -0.707107f, -0.707107f, 0.707107f, 0.707107f, -0.707107f, -0.707107f, 0.707107f},
    {0.555570f, -0.980785f, 0.195090f, 0.831470f, -0.831470f, -0.195090f, 0.980785f, -0.555570f},
    {0.382683f, -0.923880f, 0.923880f, -0.382683f, -0.382684f, 0.923880f, -0.923879f, 0.382684f},
    {0.195090f, -0.555570f, 0.831470f, -0.980785f, 0.980785f, -0.831470f, 0.555570f, -0.195090f}};

/* C(0) = 0.707107f, C(k>0) = 1 (src/core/dct.c:4-6). */
static inline float c_scale(int k) { return k == 0 ? 0.707107f : 1.000000f; }

/* ------------------------------------------------------------------------------------
 * BMP view (src/io/bmp_handler.c:15-129)
 * ---------------------------------------------------------------------------------- */
static uint32_t rd32(const uint8_t *p) {
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}
static uint16_t rd16(const uint8_t *p) { return (uint16_t)(p[0] | (p[1] << 8)); }

int oracle_parse_bmp(const uint8_t *file, size_t file_len, OracleBmpView *out) {
    if (!file || !out) return ORACLE_ERR_ARG;
    if (file_len < 14) return ORACLE_ERR_SHORT;              /* fread of file header fails */
    if (rd16(file) != 0x4D42) return ORACLE_ERR_MAGIC;       /* bmp_handler.c:30 */
    if (file_len < 54) return ORACLE_ERR_SHORT;              /* fread of info header fails */
    if (rd16(file + 28) != 24) return ORACLE_ERR_BITCOUNT;   /* :44 */
    if (rd32(file + 30) != 0) return ORACLE_ERR_COMPRESSED;  /* :49 */
    int32_t w = (int32_t)rd32(file + 18);
    int32_t h = (int32_t)rd32(file + 22);
    out->top_down = 0;
    if (h < 0) { h = -h; out->top_down = 1; }                /* :68-72 */
    if (w <= 0 || h <= 0) return ORACLE_ERR_ARG;
    out->width = w;
    out->height = h;
    out->row_stride = (w * 3 + 3) & ~3;                      /* :75 */
    uint32_t off = rd32(file + 10);                          /* bfOffBits, :88 */
    if ((uint64_t)off + (uint64_t)out->row_stride * (uint64_t)h > (uint64_t)file_len)
        return ORACLE_ERR_SHORT;                             /* :104 "Insufficient data" */
    out->pixels = file + off;
    return ORACLE_OK;
}

void oracle_padded_dims(const OracleBmpView *v, int *pw, int *ph) {
    *pw = (v->width + 7) & ~7;   /* converter.c:15 */
    *ph = (v->height + 7) & ~7;  /* converter.c:16 */
}

/* Luma of the pixel the reference would see at top-down coordinates (x, y) after its
 * loader flipped rows and swapped BGR->RGB (bmp_handler.c:109-122), with the converter's
 * edge clamp (converter.c:31,36) and integer weights (converter.c:51). */
static inline int luma_at(const OracleBmpView *v, int x, int y) {
    if (x > v->width - 1) x = v->width - 1;
    if (y > v->height - 1) y = v->height - 1;
    int file_row = v->top_down ? y : (v->height - 1 - y);
    const uint8_t *px = v->pixels + (size_t)file_row * (size_t)v->row_stride + (size_t)x * 3u;
    uint32_t b = px[0], g = px[1], r = px[2];
    return (int)((77u * r + 150u * g + 29u * b) >> 8);
}

void oracle_luma_centered(const OracleBmpView *v, int8_t *out) {
    int pw, ph;
    oracle_padded_dims(v, &pw, &ph);
    for (int y = 0; y < ph; ++y)
        for (int x = 0; x < pw; ++x)
            out[(size_t)y * pw + x] = (int8_t)(luma_at(v, x, y) - 128); /* converter.c:84-86 */
}

/* ------------------------------------------------------------------------------------
 * DCT / quantisation of one block, in the reference's float32 evaluation order.
 * p[x][y]: x = row, y = column (dct.c:72-93).
 * ---------------------------------------------------------------------------------- */
static void dct_block_exact(const int8_t p[8][8], float f[8][8]) {
    for (int u = 0; u < 8; ++u) {
        for (int v = 0; v < 8; ++v) {
            float acc = 0.0f;
            for (int x = 0; x < 8; ++x) {
                const float cx = kCos[u][x];
                for (int y = 0; y < 8; ++y) {
                    /* dct.c:84 : sum += pixel * cosX * cosY  ==  sum + ((pixel*cosX)*cosY),
                     * each product and the add rounded to float32 separately. */
                    float t = (float)p[x][y] * cx;
                    t = t * kCos[v][y];
                    acc = acc + t;
                }
            }
            /* dct.c:93 : 0.25f * cu * cv * sum, left-associated. */
            float k = 0.25f * c_scale(u);
            k = k * c_scale(v);
            f[u][v] = k * acc;
        }
    }
}

static inline int16_t quant_one(float coef, uint8_t q) {
    /* quantization.c:34-36 : float division, roundf (half away from zero), int16 cast. */
    float step = (float)q;
    return (int16_t)roundf(coef / step);
}

void oracle_dct_image(const int8_t *centered, int pw, int ph, float *out) {
    for (int by = 0; by + 8 <= ph; by += 8)
        for (int bx = 0; bx + 8 <= pw; bx += 8) {
            int8_t p[8][8];
            float f[8][8];
            for (int r = 0; r < 8; ++r)
                memcpy(p[r], centered + (size_t)(by + r) * pw + bx, 8);
            dct_block_exact(p, f);
            for (int r = 0; r < 8; ++r)
                memcpy(out + (size_t)(by + r) * pw + bx, f[r], 8 * sizeof(float));
        }
}

void oracle_quant_image(const float *dct, int pw, int ph, const uint8_t qt[64], int16_t *out) {
    for (int y = 0; y < ph; ++y)
        for (int x = 0; x < pw; ++x)
            out[(size_t)y * pw + x] = quant_one(dct[(size_t)y * pw + x], qt[(y & 7) * 8 + (x & 7)]);
}

void oracle_zigzag_image(const int16_t *quant, int pw, int ph, int16_t *out) {
    size_t b = 0;
    for (int by = 0; by < ph; by += 8)
        for (int bx = 0; bx < pw; bx += 8, ++b)
            for (int i = 0; i < 64; ++i) {
                int r = kZigzag[i] >> 3, c = kZigzag[i] & 7; /* zigzag.c:54-58 */
                out[b * 64 + i] = quant[(size_t)(by + r) * pw + bx + c];
            }
}

void oracle_quant_table(int quality, uint8_t table[64]) {
    /* Extension: libjpeg scaling of the Annex-K table; Q=50 -> identity. */
    if (quality < 1) quality = 1;
    if (quality > 100) quality = 100;
    int s = quality < 50 ? 5000 / quality : 200 - 2 * quality;
    for (int i = 0; i < 64; ++i) {
        int q = (kBaseQuant[i] * s + 50) / 100;
        if (q < 1) q = 1;
        if (q > 255) q = 255;
        table[i] = (uint8_t)q;
    }
}

/* ------------------------------------------------------------------------------------
 * Entropy coding
 * ---------------------------------------------------------------------------------- */
typedef struct { uint16_t code[256]; uint8_t len[256]; } CodeBook;

/* Canonical code assignment (huffman.c:89-104).  Symbols absent from the spec keep
 * len 0, and a len-0 code emits no bits -- the reference's putBits returns early on
 * numBits == 0 (huffman.c:36). */
static void build_codebook(const uint8_t counts[16], const uint8_t *symbols, CodeBook *cb) {
    memset(cb, 0, sizeof(*cb));
    unsigned next = 0, idx = 0;
    for (int bits = 1; bits <= 16; ++bits) {
        for (unsigned n = 0; n < counts[bits - 1]; ++n, ++next) {
            cb->code[symbols[idx]] = (uint16_t)next;
            cb->len[symbols[idx]] = (uint8_t)bits;
            ++idx;
        }
        next <<= 1;
    }
}

typedef struct {
    uint8_t *dst;
    size_t cap, n;
    uint64_t acc;   /* pending bits, right-aligned */
    int pending;    /* < 8 between calls */
    int overflow;
} BitSink;

static inline void sink_byte(BitSink *s, uint8_t b) {
    /* huffman.c:26-32 : every 0xFF is followed by a stuffed 0x00. */
    if (s->n + 2 > s->cap) { s->overflow = 1; return; }
    s->dst[s->n++] = b;
    if (b == 0xFF) s->dst[s->n++] = 0x00;
}

static inline void sink_bits(BitSink *s, unsigned value, int nbits) {
    if (nbits == 0) return;                       /* huffman.c:36 */
    value &= (1u << nbits) - 1u;                  /* huffman.c:39 */
    s->acc = (s->acc << nbits) | value;           /* MSB-first (huffman.c:53-61) */
    s->pending += nbits;
    while (s->pending >= 8) {
        s->pending -= 8;
        sink_byte(s, (uint8_t)(s->acc >> s->pending));
    }
}

static inline void sink_flush(BitSink *s) {
    /* huffman.c:65-81 : the last partial byte is padded with ZERO bits. */
    if (s->pending > 0) {
        sink_byte(s, (uint8_t)((s->acc << (8 - s->pending)) & 0xFF));
        s->pending = 0;
    }
}

static inline int magnitude_bits(int v) {         /* rle.c:9-22 */
    unsigned a = (unsigned)(v < 0 ? -v : v);
    int n = 0;
    while (a) { ++n; a >>= 1; }
    return n;
}
static inline unsigned amplitude(int v) {         /* rle.c:24-35 */
    return (unsigned)(uint16_t)(v > 0 ? v : v - 1);
}

/* One block's symbols, pushed to `emit` in stream order (rle.c:62-124). */
typedef void (*SymbolFn)(void *ctx, int is_dc, uint8_t symbol, uint16_t code, uint8_t bits);

static void block_symbols(const int16_t zz[64], int16_t *pred, SymbolFn emit, void *ctx) {
    int16_t diff = (int16_t)(zz[0] - *pred);      /* rle.c:68-70 */
    *pred = zz[0];
    int nb = magnitude_bits(diff);
    emit(ctx, 1, (uint8_t)nb, (uint16_t)amplitude(diff), (uint8_t)nb);
    int run = 0;
    for (int k = 1; k < 64; ++k) {
        int v = zz[k];
        if (v == 0) { ++run; continue; }
        while (run >= 16) { emit(ctx, 0, 0xF0, 0, 0); run -= 16; }   /* rle.c:99-103 */
        nb = magnitude_bits(v);
        emit(ctx, 0, (uint8_t)((run << 4) | nb), (uint16_t)amplitude(v), (uint8_t)nb); /* :110 */
        run = 0;
    }
    /* Trailing zeros <=> last non-zero index < 63 <=> EOB (rle.c:121-123). */
    if (run > 0) emit(ctx, 0, 0x00, 0, 0);
}

typedef struct { OracleRleSymbol *out; long n, cap; } RleCtx;
static void rle_emit(void *c, int is_dc, uint8_t symbol, uint16_t code, uint8_t bits) {
    (void)is_dc;
    RleCtx *r = (RleCtx *)c;
    if (r->n < r->cap) {
        r->out[r->n].symbol = symbol;
        r->out[r->n].code = code;
        r->out[r->n].code_bits = bits;
    }
    r->n++;
}

long oracle_rle(const int16_t *zz, long nblocks, OracleRleSymbol *out, long cap) {
    RleCtx ctx = {out, 0, cap};
    int16_t pred = 0;                              /* rle.c:59 */
    for (long b = 0; b < nblocks; ++b) block_symbols(zz + b * 64, &pred, rle_emit, &ctx);
    return ctx.n <= cap ? ctx.n : ORACLE_ERR_CAPACITY;
}

typedef struct { BitSink *sink; const CodeBook *dc, *ac; } HuffCtx;
static void huff_emit(void *c, int is_dc, uint8_t symbol, uint16_t code, uint8_t bits) {
    HuffCtx *h = (HuffCtx *)c;
    const CodeBook *cb = is_dc ? h->dc : h->ac;    /* huffman.c:145-153 / :165-174 */
    sink_bits(h->sink, cb->code[symbol], cb->len[symbol]);
    sink_bits(h->sink, code, bits);
}

long oracle_entropy(const int16_t *zz, long nblocks, uint8_t *out, size_t cap) {
    CodeBook dc, ac;
    build_codebook(kDcCounts, kDcSymbols, &dc);
    build_codebook(kAcCounts, kAcSymbols, &ac);
    BitSink sink = {out, cap, 0, 0, 0, 0};
    HuffCtx ctx = {&sink, &dc, &ac};
    int16_t pred = 0;
    for (long b = 0; b < nblocks; ++b) block_symbols(zz + b * 64, &pred, huff_emit, &ctx);
    sink_flush(&sink);
    return sink.overflow ? ORACLE_ERR_CAPACITY : (long)sink.n;
}

/* ------------------------------------------------------------------------------------
 * JFIF container (io/jpeg_handler.c:7-117)
 * ---------------------------------------------------------------------------------- */
static uint8_t *put16be(uint8_t *p, unsigned v) { p[0] = (uint8_t)(v >> 8); p[1] = (uint8_t)v; return p + 2; }

size_t oracle_jfif_prefix(int width, int height, const uint8_t qt[64], uint8_t *out) {
    uint8_t *p = out;
    /* SOI + APP0, 20 bytes (jpeg_handler.c:7-22): JFIF 1.01, units = dpi, 96x96, no thumb. */
    p = put16be(p, 0xFFD8); p = put16be(p, 0xFFE0); p = put16be(p, 16);
    memcpy(p, "JFIF", 5); p += 5;
    p = put16be(p, 0x0101); *p++ = 1; p = put16be(p, 96); p = put16be(p, 96); *p++ = 0; *p++ = 0;
    /* DQT, 69 bytes, table in zigzag order (jpeg_handler.c:36-49). */
    p = put16be(p, 0xFFDB); p = put16be(p, 67); *p++ = 0x00;
    for (int i = 0; i < 64; ++i) *p++ = qt[kZigzag[i]];
    /* SOF0, 13 bytes, ORIGINAL (unpadded) dimensions truncated to 16 bits (:52-67, :226). */
    p = put16be(p, 0xFFC0); p = put16be(p, 11); *p++ = 8;
    p = put16be(p, (uint16_t)height); p = put16be(p, (uint16_t)width);
    *p++ = 1; *p++ = 1; *p++ = 0x11; *p++ = 0;
    /* DHT DC (33 bytes, :70-80) and DHT AC (183 bytes, :83-93). */
    p = put16be(p, 0xFFC4); p = put16be(p, 31); *p++ = 0x00;
    memcpy(p, kDcCounts, 16); p += 16; memcpy(p, kDcSymbols, 12); p += 12;
    p = put16be(p, 0xFFC4); p = put16be(p, 181); *p++ = 0x10;
    memcpy(p, kAcCounts, 16); p += 16; memcpy(p, kAcSymbols, 162); p += 162;
    /* SOS, 10 bytes (:96-110). */
    p = put16be(p, 0xFFDA); p = put16be(p, 8); *p++ = 1; *p++ = 1; *p++ = 0x00;
    *p++ = 0; *p++ = 63; *p++ = 0;
    return (size_t)(p - out);
}

size_t oracle_max_jfif_bytes(int width, int height) {
    size_t nb = (size_t)((width + 7) / 8) * (size_t)((height + 7) / 8);
    /* <= 20 + 63*26 bits per block, every byte possibly stuffed. */
    return 328 + 2 + nb * 2 * 208 + 16;
}

long oracle_encode_bmp(const uint8_t *file, size_t file_len, int quality, uint8_t *out, size_t cap) {
    OracleBmpView v;
    int rc = oracle_parse_bmp(file, file_len, &v);
    if (rc) return rc;
    if (cap < 330) return ORACLE_ERR_CAPACITY;
    uint8_t qt[64];
    oracle_quant_table(quality, qt);
    size_t n = oracle_jfif_prefix(v.width, v.height, qt, out);

    CodeBook dc, ac;
    build_codebook(kDcCounts, kDcSymbols, &dc);
    build_codebook(kAcCounts, kAcSymbols, &ac);
    BitSink sink = {out + n, cap - n - 2, 0, 0, 0, 0};
    HuffCtx ctx = {&sink, &dc, &ac};
    int16_t pred = 0;

    int pw, ph;
    oracle_padded_dims(&v, &pw, &ph);
    for (int by = 0; by < ph; by += 8) {
        for (int bx = 0; bx < pw; bx += 8) {
            int8_t p[8][8];
            float f[8][8];
            int16_t zz[64];
            for (int r = 0; r < 8; ++r)
                for (int c = 0; c < 8; ++c) p[r][c] = (int8_t)(luma_at(&v, bx + c, by + r) - 128);
            dct_block_exact(p, f);
            for (int i = 0; i < 64; ++i) {
                int pos = kZigzag[i];
                zz[i] = quant_one(f[pos >> 3][pos & 7], qt[pos]);
            }
            block_symbols(zz, &pred, huff_emit, &ctx);
        }
        if (sink.overflow) return ORACLE_ERR_CAPACITY;
    }
    sink_flush(&sink);
    if (sink.overflow) return ORACLE_ERR_CAPACITY;
    n += sink.n;
    out[n++] = 0xFF; out[n++] = 0xD9;              /* EOI (jpeg_handler.c:113-117) */
    return (long)n;
}
