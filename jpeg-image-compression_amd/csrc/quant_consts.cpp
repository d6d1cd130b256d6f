// quant_consts.cpp -- host-side derivation of the kernels' constants: quantisation table for a quality, the LUT-product
// matrix as MFMA A fragments, the guard band of the fast quantiser, Huffman code words, the JFIF prefix.
//
// Guard band.  The fast path evaluates z = S * (K/q) where S is the LUT sum of a coefficient (matrix pipe) and trusts its
// rounding as the reference's  roundf(F[u][v] / q)  (natural_c/src/core/dct.c:63-96, quantization.c:34-36) only when z is
// further than delta from a rounding tie.  delta bounds |z_fast - r_ref| rigorously:
//
//   delta = (K/q) * (E_ref + E_mfma + E_split) + 4u (zmax + 1)          (per zigzag position: q, K and the LUT weights differ)
//
//   E_ref    reference evaluation error: two roundings per product, one per sequential add, worst case over |p| <= 128
//            with the actual |COS_LUT products| as weights
//   E_mfma   what the matrix pipe adds: its two accumulator chains are EXACT in any summation order (integer-valued binary16
//            terms, every partial sum below 2^24 units -- see derive_mfma_tables), so only the ONE add that joins them rounds
//   E_split  residual of the two-term split of the LUT products (2^-23 absolute per product)
//   last     rounding of K/q, of the fma, and of the reference's final scale and division
//
// tests/test_host.py::test_mfma_constants_are_on_the_safe_side pins the stored constants against the oracle's arithmetic.
#include <cmath>
#include <cstring>
#include <vector>

#include "jpegamd_internal.h"

namespace jpegamd {

const uint8_t kZigzagHost[64] = {
    0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
    41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
    30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

// Annex-K luminance table, raster order (natural_c/src/core/jpeg_tables.c:3-12).
static const uint8_t kBaseQuant[64] = {
    16, 11, 10, 16, 24,  40,  51,  61,  12, 12, 14, 19, 26,  58,  60,  55,
    14, 13, 16, 24, 40,  57,  69,  56,  14, 17, 22, 29, 51,  87,  80,  62,
    18, 22, 37, 56, 68,  109, 103, 77,  24, 35, 55, 64, 81,  104, 113, 92,
    49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99};

// Huffman specs (natural_c/src/core/jpeg_tables.c:14-48).
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

// COS_LUT[x][u] as the float32 literals of natural_c/src/core/dct.c:9-18.
static const float kCosLut[8][8] = {
    {1.000000f, 0.980785f, 0.923880f, 0.831470f, 0.707107f, 0.555570f, 0.382683f, 0.195090f},
    {1.000000f, 0.831470f, 0.382683f, -0.195090f, -0.707107f, -0.980785f, -0.923880f, -0.555570f},
    {1.000000f, 0.555570f, -0.382683f, -0.980785f, -0.707107f, 0.195090f, 0.923880f, 0.831470f},
    {1.000000f, 0.195090f, -0.923880f, -0.555570f, 0.707107f, 0.831470f, -0.382683f, -0.980785f},
    {1.000000f, -0.195090f, -0.923880f, 0.555570f, 0.707107f, -0.831470f, -0.382684f, 0.980785f},
    {1.000000f, -0.555570f, -0.382684f, 0.980785f, -0.707107f, -0.195090f, 0.923880f, -0.831470f},
    {1.000000f, -0.831470f, 0.382684f, 0.195091f, -0.707107f, 0.980785f, -0.923879f, 0.555570f},
    {1.000000f, -0.980785f, 0.923880f, -0.831470f, 0.707107f, -0.555570f, 0.382684f, -0.195090f}};

void quant_table_for_quality(int quality, uint8_t table[64]) {
    // quality 0/50 -> the reference's table; otherwise libjpeg scaling (extension).
    if (quality <= 0) quality = 50;
    if (quality > 100) quality = 100;
    const int s = quality < 50 ? 5000 / quality : 200 - 2 * quality;
    for (int i = 0; i < 64; ++i) {
        int q = (kBaseQuant[i] * s + 50) / 100;
        if (q < 1) q = 1;
        if (q > 255) q = 255;
        table[i] = (uint8_t)q;
    }
}

void build_huffman_words(uint32_t words[272]) {
    // Canonical codes (natural_c/src/core/huffman.c:89-104) packed as len<<16 | code.
    // Symbols the spec does not list stay 0 (length 0: the reference emits no code bits
    // for them, huffman.c:36).
    std::memset(words, 0, 272 * sizeof(uint32_t));
    auto fill = [](const uint8_t counts[16], const uint8_t *symbols, uint32_t *dst) {
        uint32_t code = 0;
        int idx = 0;
        for (int len = 1; len <= 16; ++len) {
            for (int i = 0; i < counts[len - 1]; ++i) dst[symbols[idx++]] = ((uint32_t)len << 16) | (code++ & 0xFFFFu);
            code <<= 1;
        }
    };
    fill(kAcCounts, kAcSymbols, words);
    fill(kDcCounts, kDcSymbols, words + 256);
}

void build_code_table(uint32_t words[kCodeWords]) {
    // The coder's view of the same codes (jpegamd_internal.h: entry (row, fb)).  rle.c:9-35 (size, amplitude), rle.c:99-123
    // (ZRL, run/size symbol, EOB), huffman.c:145-188 (code, then amplitude bits).
    uint32_t hw[272];
    build_huffman_words(hw);
    std::memset(words, 0, kCodeWords * sizeof(uint32_t));
    auto entry = [](uint32_t w, uint32_t size, uint32_t zrl) -> uint32_t {
        const uint32_t clen = w >> 16, code = w & 0xFFFFu;
        // a symbol the tables have no code for gets no code bits, but its amplitude bits are still written (huffman.c:36,176-186)
        return ((clen ? code << (16u - clen) : 0u) << 16) | ((clen + size) << 8) | (zrl << 5) | clen;
    };
    for (int r = 0; r < kCodeRows; ++r)
        for (int fb = -1; fb <= 30; ++fb) {
            if (fb >= 0 && fb < 18) continue;                       // sizes above 13 do not occur
            const uint32_t size = fb < 0 ? 0u : (uint32_t)(31 - fb);
            uint32_t e = 0;
            if (r == 0) {
                if (size <= 11) e = entry(hw[256 + size], size, 0);
                else if (size == 13) e = entry(hw[0x00], 0, 0);    // EOB: the code alone
            } else if (size >= 1 && size <= 11) {                   // a non-zero AC coefficient
                const uint32_t run = (uint32_t)(r - 1);
                e = entry(hw[((run & 15u) << 4) | size], size, run >> 4);
            }
            words[kCodeLead + kCodeRowStride * r + fb] = e;
        }
}

size_t build_jfif_prefix(int width, int height, const uint8_t table[64], uint8_t out[328]) {
    // natural_c/src/io/jpeg_handler.c:7-110 (byte layout of the six marker segments).
    uint8_t *p = out;
    auto be16 = [&](unsigned v) { *p++ = (uint8_t)(v >> 8); *p++ = (uint8_t)v; };
    be16(0xFFD8); be16(0xFFE0); be16(16);
    std::memcpy(p, "JFIF", 5); p += 5;
    be16(0x0101); *p++ = 1; be16(96); be16(96); *p++ = 0; *p++ = 0;
    be16(0xFFDB); be16(67); *p++ = 0;
    for (int i = 0; i < 64; ++i) *p++ = table[kZigzagHost[i]];
    be16(0xFFC0); be16(11); *p++ = 8; be16((uint16_t)height); be16((uint16_t)width);
    *p++ = 1; *p++ = 1; *p++ = 0x11; *p++ = 0;
    be16(0xFFC4); be16(31); *p++ = 0x00;
    std::memcpy(p, kDcCounts, 16); p += 16; std::memcpy(p, kDcSymbols, 12); p += 12;
    be16(0xFFC4); be16(181); *p++ = 0x10;
    std::memcpy(p, kAcCounts, 16); p += 16; std::memcpy(p, kAcSymbols, 162); p += 162;
    be16(0xFFDA); be16(8); *p++ = 1; *p++ = 1; *p++ = 0; *p++ = 0; *p++ = 63; *p++ = 0;
    return (size_t)(p - out);
}

namespace {
constexpr double kU = 5.9604644775390625e-08;   // 2^-24
constexpr double kPmax = 128.0;
}  // namespace

// ---- matrix-pipe tables ----------------------------------------------------------------------
// out_z = sum_p Kmat[c][p] * pix[p], Kmat[c][p] = COS_LUT[x][u] * COS_LUT[y][v] (exact product of the two float32
// literals), computed by two chains of 4 v_mfma_f32_32x32x16_f16 per row half: Kmat = hi 2^-11 + lo 2^-22 with INTEGER
// hi = round(2^11 Kmat) (|hi| <= 2048) and lo = round(2^22 (Kmat - hi 2^-11)) (|lo| <= 1024); the hi chain's A entries
// are hi, the lo chain's are lo 2^-11 (both exact in binary16).
// The B operand is the UNCENTRED luma y = p + 128 in 0 .. 255 as a binary16 SUBNORMAL: the integer y IS the bit pattern of y 2^-24,
// so the kernel packs two dot-product bytes into a register with one v_perm instead of converting each with v_cvt_f16_i16 (the matrix
// pipe takes subnormal inputs at face value: tools/ubench/mfma_denorm.hip, profiles/r04_ubench_mfma_denorm.txt).  What makes this
// free: for every AC position the hi terms of a row sum to ZERO exactly (the LUT's symmetry), so sum hi y = sum hi p, and the lo
// terms sum to T_lo = 0 for all but five positions (32 units of 2^-22 at (0,3), (3,0), (0,6), (6,0), 1 at (3,3)), whose constant
// 128 T_lo 2^-22 goes into the quantiser's additive constant (zoff).  The DC row (hi = 2048 everywhere, lo = 0) yields the
// plain pixel sum, 64 * 128 too large: the kernel takes 1.0 off its accumulator (dc_off; exact).
// Exactness: with 0 <= y <= 255 every partial sum of a chain lies between -255 * (sum of the negative terms) and 255 * (sum of the
// positive ones), multiples of the chain's unit below 2^24 units (AC rows: at most 0.71 * 2^24; the DC row's partial sums are
// multiples of 2048 units), so the float32 accumulation is EXACT in any order the hardware adds; the kernel joins the chains with
// one float add: acc = kMfmaScale * (LUT sum over y with Kmat replaced by hi 2^-11 + lo 2^-22), one rounding, kMfmaScale = 2^-13.
// derive_mfma_tables verifies the representability, the zero row sums and the 2^24 bounds (split_ok) and falls back to "flag
// everything" if they ever failed.
// (History: round 1 used three bf16 terms in one accumulator, round 2's first kernel two binary16 terms in one accumulator
// with E_mfma = 2 * 16u * sum_m (|acc before MFMA m| + S_m): 2.3 x the reference's own evaluation error E_ref.)
namespace {
// double -> IEEE binary16 bits, round to nearest even, subnormals kept; |x| < 65520
uint16_t to_f16(double x) {
    const uint16_t sign = x < 0 ? 0x8000u : 0u;
    double a = std::fabs(x);
    if (a == 0.0) return sign;
    int e;
    (void)std::frexp(a, &e);                       // a = m * 2^e, 0.5 <= m < 1  ->  exponent of the leading bit: e - 1
    int ex = e - 1;
    if (ex < -14) ex = -14;                        // subnormal range: fixed quantum 2^-24
    const double quantum = std::ldexp(1.0, ex - 10);
    double n = std::nearbyint(a / quantum);        // default rounding mode: to nearest even
    if (n >= 2048.0) { n *= 0.5; ++ex; }           // rounded up into the next binade
    const uint32_t mant = (uint32_t)n;             // 1024..2047 normal, 0..1023 subnormal
    if (mant < 1024u) return (uint16_t)(sign | mant);
    return (uint16_t)(sign | ((uint32_t)(ex + 15) << 10) | (mant - 1024u));
}
double from_f16(uint16_t b) {
    const int ex = (b >> 10) & 31, mant = b & 1023;
    const double v = ex ? std::ldexp((double)(1024 + mant), ex - 25) : std::ldexp((double)mant, -24);
    return (b & 0x8000u) ? -v : v;
}
}  // namespace

void derive_mfma_tables(const uint8_t table[64], MfmaTables *mt, double delta_out[64]) {
    std::memset(mt, 0, sizeof(*mt));
    uint16_t *af = reinterpret_cast<uint16_t *>(mt->afrag);
    double delta_z[64], lo_abs[64];
    double dmax = 0;
    bool split_ok = true;
    const double S = (double)kMfmaScale;
    for (int z = 0; z < 64; ++z) {
        const int k = kZigzagHost[z], u = k >> 3, v = k & 7;
        double kmat[64], term[2][64], split_res = 0, hi_units = 0, lo_units = 0;     // terms in units of Kmat
        double hi_pos = 0, hi_neg = 0, lo_pos = 0, lo_neg = 0, hi_sum = 0, lo_sum = 0; // signed parts of a row's terms, in units
        for (int x = 0; x < 8; ++x)
            for (int y = 0; y < 8; ++y) {
                const int p = x * 8 + y;
                kmat[p] = (double)kCosLut[x][u] * (double)kCosLut[y][v];
                // hi = round(2^11 K) (|hi| <= 2048: an integer binary16 holds exactly), lo = round(2^22 (K - hi 2^-11)) (|lo| <= 1024),
                // stored as lo 2^-11 (a multiple of 2^-11 with an 11-bit numerator: exact as well)
                const double hi_i = std::nearbyint(kmat[p] * 2048.0);
                const double lo_i = std::nearbyint((kmat[p] - hi_i / 2048.0) * 4194304.0);
                const uint16_t hi = to_f16(hi_i), lo = to_f16(lo_i / 2048.0);
                if (from_f16(hi) != hi_i || from_f16(lo) != lo_i / 2048.0 || std::fabs(hi_i) > 2048.0 || std::fabs(lo_i) > 1024.0) split_ok = false;
                term[0][p] = lo_i / 4194304.0; term[1][p] = hi_i / 2048.0;
                split_res += std::fabs(kmat[p] - (term[0][p] + term[1][p]));
                hi_units += std::fabs(hi_i); lo_units += std::fabs(lo_i);
                (hi_i > 0 ? hi_pos : hi_neg) += std::fabs(hi_i); (lo_i > 0 ? lo_pos : lo_neg) += std::fabs(lo_i);
                hi_sum += hi_i; lo_sum += lo_i;
                // scatter into the A-operand order: term t, chain H, matrix row R, k-step s, lane (hk, R), element j;
                // lane half h = (z >> 3) & 1 holds z = 16G + 8h + j at site 8G + j = 16H + r
                const int site = 8 * (z >> 4) + (z & 7);
                const int h = (z >> 3) & 1, H = site >> 4, r = site & 15;
                const int R = (r & 3) + 8 * (r >> 2) + 4 * h;
                const int s = p >> 4, hk = (p >> 3) & 1, j = p & 7;
                const int lane = 32 * hk + R;
                const uint16_t t2[2] = {lo, hi};
                for (int t = 0; t < 2; ++t) af[((((size_t)t * 2 + H) * 4 + s) * 64 + lane) * 8 + j] = t2[t];
            }
        // reference evaluation error: two roundings per product, 63 sequential float32 additions (dct.c:84)
        double wsum = 0, run = 0, adds = 0;
        for (int j = 0; j < 64; ++j) { const double w = std::fabs(kmat[j]); wsum += w; run += w; if (j >= 1) adds += run; }
        const double e_ref = (2.0 * kU * kPmax * wsum + kU * kPmax * adds) * 1.001;
        // Each chain is exact when every partial sum stays below 2^24 of its units (hi: 1, lo: 2^-11 of the hi unit) whatever the
        // summation order: with 0 <= y <= 255 a partial sum lies in [-255 * negative part, 255 * positive part].  (The DC row: every
        // term is 2048 units, every partial sum a multiple of 2048 below 2^24 * 2048.)  What is left of the matrix pipe is the ONE
        // rounding of the add that joins the chains, relative to |S| <= kPmax * wsum (the sums themselves are those of the centred
        // pixels: the hi terms of an AC row add up to zero).
        const bool dc_row = z == 0;
        if (dc_row ? (hi_pos != 64.0 * 2048.0 || hi_neg != 0.0 || lo_units != 0.0) : (hi_sum != 0.0)) split_ok = false;
        if ((!dc_row && 255.0 * std::fmax(hi_pos, hi_neg) > 16777216.0) || 255.0 * std::fmax(lo_pos, lo_neg) > 16777216.0) split_ok = false;   // (integers up to 2^24 inclusive are float32 values)
        // the constant the uncentred operand leaves in an AC row: 128 * T_lo units of 2^-22 (in units of Kmat)
        const double c_row = dc_row ? 0.0 : 128.0 * lo_sum / 4194304.0;
        lo_abs[z] = S * (kPmax * lo_units + 128.0 * std::fabs(lo_sum)) / 4194304.0;               // |lo-chain output| <= this, in accumulator units
        const double e_mfma = kU * kPmax * wsum * 1.0001;
        const float cu = u == 0 ? 0.707107f : 1.0f, cv = v == 0 ? 0.707107f : 1.0f;
        const double K = (double)((0.25f * cu) * cv);
        const double q = (double)table[k];
        const double zmax = K * kPmax * wsum / q;
        // (the rounded value comes from fma(acc, qmul, 1.5 * 2^23), which cannot carry the row's constant: it rounds z + (K/q) c_row, and
        //  that agrees with the rounding of z wherever z is further than |(K/q) c_row| from a tie -- so the band includes it)
        const double delta = (K / q) * (e_ref + e_mfma + kPmax * split_res + std::fabs(c_row)) + 4.0 * kU * (zmax + 1.0);
        mt->zoff[z] = (float)(-(K / q) * c_row);
        delta_z[z] = delta;
        if (z > 0 && delta > dmax) dmax = delta;
        mt->qmul[z] = (float)(K / (q * S));                 // the accumulator holds kMfmaScale * LUT sum (a power of two: K / q rounds the same)
        mt->qstep[z] = (float)table[k];
        if (delta_out) delta_out[k] = delta;
    }
    // One bias PER POSITION: zc = z + 0.5 + b_z with b_z >= delta_z, flagged when fract(zc) <= 2 b_z, i.e. z within b_z of a tie on
    // either side.  (Rounds 1-2 used one bias for all positions, 0.5 + max delta: the band below a tie was then max delta wide
    // everywhere -- 3.5 x the mean delta, 2.3 x the exact-order events of this layout.)  2 b - 1 is exact in float32 for b in [0.5, 1).
    (void)dmax;
    for (int z = 0; z < 64; ++z) {
        mt->bias[z] = (float)(0.5 + delta_z[z] * 1.001 + 1.0e-7);
        mt->qthr[z] = 2.0f * mt->bias[z] - 1.0f;
    }
    for (int z = 0; z < 64; ++z) mt->qadd[z] = (float)((double)mt->bias[z] + (double)mt->zoff[z]);      // what the kernel's fma adds
    mt->dc_off = (float)(S * 64.0 * 128.0);                // what the DC row's accumulator holds beyond the centred sum: kMfmaScale * sum of 64 x 128 (1.0 at 2^-13)
    if (!split_ok)                                          // cannot happen with the reference's LUT; if it did, EVERY coefficient takes the exact-order path
        for (int z = 0; z < 64; ++z) { mt->bias[z] = 1.5f; mt->qadd[z] = 1.5f; mt->qthr[z] = 2.0f; }      // (fract <= 2 always; the group thresholds come out negative: no group is skipped)
    // zero threshold of a group: qthr_z < fl(a * qmul_z + qadd_z) < 1 (i.e. floor = 0, not flagged) for every |a| below t;
    // the 2^-18 relative margin covers the single rounding of the kernel's fma
    for (int g = 0; g < 4; ++g)
        for (int h = 0; h < 2; ++h) {
            double t = 1.0e30;
            for (int j = 0; j < 8; ++j) {
                const int z = 16 * g + 8 * h + j;
                const double up = 1.0 - (double)mt->qadd[z], dn = (double)mt->qadd[z] - (double)mt->qthr[z];
                t = std::fmin(t, std::fmin(up, dn) / (double)mt->qmul[z]);
            }
            float fmax = 0.0f;
            for (int j = 0; j < 8; ++j) fmax = std::fmax(fmax, mt->qthr[16 * g + 8 * h + j]);
            mt->flag_thr[2 * g + h] = fmax;
            // The kernel tests the hi chain alone (the add that joins the chains is then spent on active groups only):
            // |hi| < thr and |lo| <= lo_bound give |fl(hi + lo)| <= (thr + lo_bound) (1 + 2^-24) < t.
            double lob = 0;
            for (int j = 0; j < 8; ++j) lob = std::fmax(lob, lo_abs[16 * g + 8 * h + j]);
            mt->lo_bound[2 * g + h] = (float)(lob * (1.0 + 1.0 / 1048576.0));
            const double th = t * (1.0 - 1.0 / 262144.0) - (double)mt->lo_bound[2 * g + h] * (1.0 + 1.0 / 1048576.0);
            mt->grp_thr[2 * g + h] = th > 0.0 ? (float)th : 0.0f;                  // (0: |hi| >= 0 always -- the group is never skipped)
            if (th > 0.0 && (double)mt->grp_thr[2 * g + h] > th) mt->grp_thr[2 * g + h] = std::nextafterf(mt->grp_thr[2 * g + h], 0.0f);
        }
}

void cos_lut_copy(float out[64]) {
    for (int x = 0; x < 8; ++x)
        for (int u = 0; u < 8; ++u) out[x * 8 + u] = kCosLut[x][u];
}

}  // namespace jpegamd
