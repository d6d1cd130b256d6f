/*
 * synth_bmp.c -- deterministic, integer-only synthetic 24-bit BMP generator.
 *
 * The reference ships four sample BMPs and no benchmark inputs; the
 * configs in BASELINE.json call for synthetic 1920x1080 / 4096^2 / 8192^2 images.  This
 * generator produces them bit-identically on every machine (no floating point, no libc
 * rand), position-hashed so any tile can be produced independently.
 *
 * File layout follows what natural_c/src/io/bmp_handler.c:15-129 accepts: 14+40 byte
 * headers, 24 bpp, BI_RGB, rows padded to 4 bytes, bottom-up unless flags bit0.
 */
#include <stdint.h>
#include <string.h>

#include "jpeg_compression.h"

static inline uint32_t mix32(uint32_t h) {
    h ^= h >> 16; h *= 0x7feb352dU;
    h ^= h >> 15; h *= 0x846ca68bU;
    h ^= h >> 16;
    return h;
}

/* Triangle wave in 0..255 with period `p` (p >= 2). */
static inline int tri(uint32_t t, uint32_t p) {
    uint32_t m = t % p;
    uint32_t d = 2 * m;
    uint32_t a = d > p ? d - p : p - d;     /* p..0..p */
    return (int)((a * 255u) / p);
}

static inline uint8_t clamp8(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

static void pixel_photo(uint32_t x, uint32_t y, uint32_t seed, uint8_t bgr[3]) {
    /* Smooth illumination: three incommensurate triangle waves per channel. */
    uint32_t s0 = seed * 2654435761u;
    uint32_t p1 = 701 + (s0 & 255), p2 = 523 + ((s0 >> 8) & 255), p3 = 1231 + ((s0 >> 16) & 255);
    /* Texture mask: slowly varying; high in roughly a quarter of the area. */
    int mask = tri(x * 3u + y * 5u + (s0 >> 3), 2909);
    int rough = mask > 200 ? 22 : (mask > 150 ? 6 : 2);
    /* Fine detail shared by the channels (luma noise) plus a little chroma noise. */
    uint32_t h = mix32((y * 0x9E3779B1u) ^ (x * 0x85EBCA77u) ^ seed);
    int luma_n = (int)(h & 0xFF) - 128;               /* -128..127 */
    luma_n = (luma_n * rough) / 128;
    /* Medium-scale structure: 16-pixel cells with a per-cell offset (edges between cells). */
    uint32_t hc = mix32(((y >> 4) * 0x27D4EB2Fu) ^ ((x >> 4) * 0x165667B1u) ^ (seed + 17u));
    int cell = mask > 100 ? ((int)(hc & 31) - 16) : 0;
    for (int c = 0; c < 3; ++c) {
        int base = (tri(x + 97u * (uint32_t)c, p1) * 3 + tri(y + 53u * (uint32_t)c, p2) * 3 +
                    tri(x + 2u * y + 31u * (uint32_t)c, p3) * 2) >> 3;
        int chroma_n = (int)((h >> (8 + 8 * c)) & 7) - 3;
        bgr[c] = clamp8(base + luma_n + cell + chroma_n);
    }
}

static void pixel_at(int kind, uint32_t x, uint32_t y, uint32_t w, uint32_t h, uint32_t seed,
                     uint8_t bgr[3]) {
    switch (kind) {
    case 1: {   /* uniform noise: symbol-dense stress (no-EOB blocks, ZRL, 0xFF stuffing) */
        uint32_t r = mix32((y * 0x9E3779B1u) ^ (x * 0x85EBCA77u) ^ seed);
        bgr[0] = (uint8_t)r; bgr[1] = (uint8_t)(r >> 8); bgr[2] = (uint8_t)(r >> 16);
        break;
    }
    case 2:     /* flat grey: every block's DC sits on a rounding tie when the level is odd */
        bgr[0] = bgr[1] = bgr[2] = (uint8_t)(seed & 255u);
        break;
    case 3: {   /* gradients */
        bgr[0] = (uint8_t)((x * 255u) / (w > 1 ? w - 1 : 1));
        bgr[1] = (uint8_t)((y * 255u) / (h > 1 ? h - 1 : 1));
        bgr[2] = (uint8_t)(((x + y) * 255u) / (w + h > 2 ? w + h - 2 : 1));
        break;
    }
    default:
        pixel_photo(x, y, seed, bgr);
    }
}

static void wr16(uint8_t *p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); }
static void wr32(uint8_t *p, uint32_t v) { wr16(p, v); wr16(p + 2, v >> 16); }

uint64_t jpegamd_synth_bmp(int32_t width, int32_t height, uint32_t seed, int32_t kind,
                           uint32_t flags, uint8_t *out, uint64_t cap) {
    if (width <= 0 || height <= 0) return 0;
    const uint32_t w = (uint32_t)width, h = (uint32_t)height;
    const uint64_t stride = ((uint64_t)w * 3u + 3u) & ~(uint64_t)3u;
    const uint32_t off = (flags & 2u) ? 138u : 54u;
    const uint64_t total = off + stride * h;
    if (!out) return total;
    if (cap < total) return 0;

    memset(out, 0, off);
    out[0] = 'B'; out[1] = 'M';
    wr32(out + 2, (uint32_t)total);
    wr32(out + 10, off);
    wr32(out + 14, off - 14u);                       /* biSize: 40, or 124 for the V5-style pad */
    wr32(out + 18, w);
    wr32(out + 22, (flags & 1u) ? (uint32_t)(-(int32_t)h) : h);
    wr16(out + 26, 1);
    wr16(out + 28, 24);
    wr32(out + 30, 0);
    wr32(out + 34, (uint32_t)(stride * h));
    wr32(out + 38, 2835); wr32(out + 42, 2835);

    for (uint32_t fr = 0; fr < h; ++fr) {
        const uint32_t y = (flags & 1u) ? fr : (h - 1u - fr);   /* image row stored in file row fr */
        uint8_t *row = out + off + stride * fr;
        for (uint32_t x = 0; x < w; ++x) pixel_at(kind, x, y, w, h, seed, row + 3u * x);
        /* Row padding bytes are deliberately non-zero: a correct reader never looks at them. */
        for (uint64_t k = (uint64_t)w * 3u; k < stride; ++k) row[k] = 0xA5;
    }
    return total;
}
