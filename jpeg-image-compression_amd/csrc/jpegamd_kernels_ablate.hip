// jpegamd_kernels.hip -- CDNA4 (gfx950) kernels of the BMP -> grayscale baseline-JPEG path.
//
// Replaces the per-stage whole-image passes of the reference's natural_c pipeline
// (natural_c/src/io/jpeg_handler.c:133-201) with:
//
//   k_transform   one wavefront per 64-block segment, one 8x8 block per lane:
//                 luma + level shift (converter.c:51,84-86), 2-D DCT (dct.c:63-96),
//                 quantisation (quantization.c:34-36), zigzag (zigzag.c:51-61), run/size
//                 symbols (rle.c:51-127) and their Huffman codes (huffman.c:145-188), fused.
//                 Each wave emits its segment's contiguous MSB-first bit string + bit count.
//   k_scan_*      single-workgroup exclusive prefix sums (bit offsets, stuffing offsets).
//   k_count_ff    per segment: how many 0xFF bytes its OWNED output bytes contain.
//   k_pack        per segment: stitch at the scanned bit offset, stuff 0xFF -> 0xFF00
//                 (huffman.c:26-32), zero-pad the last byte (huffman.c:65-81), JFIF
//                 prefix and EOI (jpeg_handler.c:220-262).
//
// Bit-exactness: the DCT is evaluated with a fast separable AAN flow
// graph; a coefficient's rounding is trusted only when z = coef/q is farther than a
// rigorous per-coefficient bound delta_k from every half-integer (quant_consts.cpp derives
// delta_k).  Otherwise the wave recomputes that one coefficient in the reference's exact
// float32 order (64 sequential, separately rounded multiply-multiply-add terms), one term
// per lane and a 63-step DPP chain for the ordered sum.  The DC coefficient is always
// exact (its sum is an integer).
//
// No MFMA: the path is byte/integer work plus a 16-FLOP/pixel transform; it is bound by
// HBM reads and VALU issue, not by dense contraction.

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "jpegamd_internal.h"

#ifndef JPEGAMD_ABLATE
#define JPEGAMD_ABLATE 0
#endif
namespace jpegamd {

#include "std_table_consts.inc"

// ------------------------------------------------------------------------------------
// Tables
// ------------------------------------------------------------------------------------

// zigzag position -> raster index (zigzag.c:7-15)
__device__ constexpr uint8_t kZZ[64] = {
    0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
    41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
    30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

// The reference's cosine LUT (dct.c:9-18) stored frequency-major: kCosFM[u*8+x] = COS_LUT[x][u].
__constant__ float kCosFM[64] = {
    1.000000f, 1.000000f, 1.000000f, 1.000000f, 1.000000f, 1.000000f, 1.000000f, 1.000000f,
    0.980785f, 0.831470f, 0.555570f, 0.195090f, -0.195090f, -0.555570f, -0.831470f, -0.980785f,
    0.923880f, 0.382683f, -0.382683f, -0.923880f, -0.923880f, -0.382684f, 0.382684f, 0.923880f,
    0.831470f, -0.195090f, -0.980785f, -0.555570f, 0.555570f, 0.980785f, 0.195091f, -0.831470f,
    0.707107f, -0.707107f, -0.707107f, 0.707107f, 0.707107f, -0.707107f, -0.707107f, 0.707107f,
    0.555570f, -0.980785f, 0.195090f, 0.831470f, -0.831470f, -0.195090f, 0.980785f, -0.555570f,
    0.382683f, -0.923880f, 0.923880f, -0.382683f, -0.382684f, 0.923880f, -0.923879f, 0.382684f,
    0.195090f, -0.555570f, 0.831470f, -0.980785f, 0.980785f, -0.831470f, 0.555570f, -0.195090f};

// 0.25f * C(u) * C(v), left-associated float32 products (dct.c:87-93).
__device__ __forceinline__ float ref_scale(int u, int v) {
    const float cu = (u == 0) ? 0.707107f : 1.000000f;
    const float cv = (v == 0) ? 0.707107f : 1.000000f;
    return __fmul_rn(__fmul_rn(0.25f, cu), cv);
}

// quantization.c:34-36: float32 division, roundf (half away from zero).
__device__ __forceinline__ int ref_quantise(float coef, float qstep) {
    return (int)roundf(__fdiv_rn(coef, qstep));
}

// ------------------------------------------------------------------------------------
// Pixel access
// ------------------------------------------------------------------------------------
__device__ __forceinline__ const uint8_t *row_ptr(const ImageDesc &im, int y) {
    const int stored = im.bottom_up ? (im.height - 1 - y) : y;      // bmp_handler.c:109
    return im.pixels + (size_t)stored * (size_t)im.row_stride;
}

// Luma of image pixel (x, y) with the converter's edge clamp (converter.c:31,36,51).
__device__ __forceinline__ int luma_clamped(const ImageDesc &im, int x, int y) {
    x = min(x, im.width - 1);
    y = min(y, im.height - 1);
    const uint8_t *p = row_ptr(im, y) + 3 * (size_t)x;
    const uint32_t w = im.weights;
    return (int)(((w & 0xFF) * p[0] + ((w >> 8) & 0xFF) * p[1] + ((w >> 16) & 0xFF) * p[2]) >> 8);
}

// Luma = bits 15:8 of the dot product (the sum is < 2^16).  Shift + convert: the one-instruction
// v_cvt_f32_ubyte1 form costs ~40 more live VGPRs in hipcc's schedule (197 vs 160, one wave per
// SIMD less), and as inline asm right behind v_dot4 it reads a stale register on gfx950 (the
// compiler's DOT->VALU hazard padding does not cover asm operands).
__device__ __forceinline__ float ubyte1_f32(uint32_t x) { return (float)(int)(x >> 8); }

// 8 pixels (24 bytes, 4-byte aligned) -> 8 luma values via v_dot4_u32_u8.  The dot product is
// 256*Y + fraction (< 2^16), so Y = byte 1 of the result: v_cvt_f32_ubyte1 converts it in one op.
__device__ __forceinline__ void luma_row8(const uint32_t *__restrict__ src, uint32_t w, float *y) {
    const uint32_t d0 = src[0], d1 = src[1], d2 = src[2], d3 = src[3], d4 = src[4], d5 = src[5];
    const uint32_t c0 = w & 0xFFu, c1 = (w >> 8) & 0xFFu, c2 = (w >> 16) & 0xFFu;
    const uint32_t wA = w;                         // pixel in bytes 0..2
    const uint32_t wB0 = c0 << 24, wB1 = c1 | (c2 << 8);          // byte 3 | bytes 0..1
    const uint32_t wC0 = (c0 << 16) | (c1 << 24), wC1 = c2;       // bytes 2..3 | byte 0
    const uint32_t wD = w << 8;                    // pixel in bytes 1..3
    y[0] = ubyte1_f32(__builtin_amdgcn_udot4(d0, wA, 0u, false));
    y[1] = ubyte1_f32(__builtin_amdgcn_udot4(d1, wB1, __builtin_amdgcn_udot4(d0, wB0, 0u, false), false));
    y[2] = ubyte1_f32(__builtin_amdgcn_udot4(d2, wC1, __builtin_amdgcn_udot4(d1, wC0, 0u, false), false));
    y[3] = ubyte1_f32(__builtin_amdgcn_udot4(d2, wD, 0u, false));
    y[4] = ubyte1_f32(__builtin_amdgcn_udot4(d3, wA, 0u, false));
    y[5] = ubyte1_f32(__builtin_amdgcn_udot4(d4, wB1, __builtin_amdgcn_udot4(d3, wB0, 0u, false), false));
    y[6] = ubyte1_f32(__builtin_amdgcn_udot4(d5, wC1, __builtin_amdgcn_udot4(d4, wC0, 0u, false), false));
    y[7] = ubyte1_f32(__builtin_amdgcn_udot4(d5, wD, 0u, false));
}

// ------------------------------------------------------------------------------------
// Fast 8-point DCT (Arai-Agui-Nakajima flow graph), in place, stride S.
// Output k is the true DCT-II sum scaled by a known factor folded into QuantConsts::mult.
// Output 0 is the plain sum of the inputs (adds only -> exact for integer inputs).
// ------------------------------------------------------------------------------------
template <int S>
__device__ __forceinline__ void aan8(float *d) {
    constexpr float A1 = 0.70710678118654752f;   // cos(pi/4)
    constexpr float A2 = 0.54119610014619698f;   // sqrt2 * cos(3pi/8)
    constexpr float A4 = 1.30656296487637653f;   // sqrt2 * cos(pi/8)
    constexpr float A5 = 0.38268343236508977f;   // cos(3pi/8)
    const float t0 = d[0 * S] + d[7 * S], t7 = d[0 * S] - d[7 * S];
    const float t1 = d[1 * S] + d[6 * S], t6 = d[1 * S] - d[6 * S];
    const float t2 = d[2 * S] + d[5 * S], t5 = d[2 * S] - d[5 * S];
    const float t3 = d[3 * S] + d[4 * S], t4 = d[3 * S] - d[4 * S];
    const float e0 = t0 + t3, e3 = t0 - t3;
    const float e1 = t1 + t2, e2 = t1 - t2;
    d[0 * S] = e0 + e1;
    d[4 * S] = e0 - e1;
    const float z1 = (e2 + e3) * A1;
    d[2 * S] = e3 + z1;
    d[6 * S] = e3 - z1;
    const float o0 = t4 + t5, o1 = t5 + t6, o2 = t6 + t7;
    const float z5 = (o0 - o2) * A5;
    const float z2 = fmaf(o0, A2, z5);
    const float z4 = fmaf(o2, A4, z5);
    const float z3 = o1 * A1;
    const float z11 = t7 + z3, z13 = t7 - z3;
    d[5 * S] = z13 + z2;
    d[3 * S] = z13 - z2;
    d[1 * S] = z11 + z4;
    d[7 * S] = z11 - z4;
}

// ------------------------------------------------------------------------------------
// Wave helpers (wave = 64 lanes)
// ------------------------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63u); }

__device__ __forceinline__ int wave_sum_i32(int v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// inclusive prefix sum across the wave
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v, int lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t t = __shfl_up(v, off, 64);
        if (lane >= off) v += t;
    }
    return v;
}

__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

// ------------------------------------------------------------------------------------
// Exact-order coefficient (dct.c:72-93 + quantization.c:34-36), cooperative: lane j owns
// term j = x*8+y of block (bx, by); the ordered float32 sum s_j = fl(s_{j-1} + t_j) is a
// 63-step DPP wave_shr chain.  Must be called with all 64 lanes active; (bx, by, u, v)
// wave-uniform.  Returns the quantised value in every lane.
// ------------------------------------------------------------------------------------
__device__ __forceinline__ float exact_term_sum(float t) {
    // After step i every lane j <= i holds fl(...fl(t_0 + t_1)... + t_j).
    float acc = t;
#pragma unroll 1
    for (int i = 1; i < 64; ++i) {
        const float prev = __builtin_bit_cast(
            float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, acc), 0x138 /*wave_shr:1*/, 0xF,
                                               0xF, false));
        // lane 0 receives 0.0f: fl(t_0 + 0) == t_0, so lane 0 stays t_0 (dct.c:68 starts at 0.0f).
        acc = __fadd_rn(t, prev);
    }
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, acc), 63));
}

__device__ __forceinline__ float exact_coef_float(float pixel /*lane j: p[x=j>>3][y=j&7]*/, int u, int v,
                                                  const float *s_cos, int lane) {
    const float cx = s_cos[u * 8 + (lane >> 3)];     // COS_LUT[x][u]
    const float cy = s_cos[v * 8 + (lane & 7)];      // COS_LUT[y][v]
    const float t = __fmul_rn(__fmul_rn(pixel, cx), cy);             // dct.c:84
    const float s = exact_term_sum(t);
    return __fmul_rn(ref_scale(u, v), s);                            // dct.c:93
}

// ------------------------------------------------------------------------------------
// k_transform
// ------------------------------------------------------------------------------------
struct WaveLds {
    uint32_t priv[(kPrivWords + 1) * 64];   // [word][lane] private bit words; also edge-tile staging
    uint32_t offs[66];                       // exclusive bit offsets of the 64 blocks + total
    uint32_t ev_k[64];                       // flagged sites
    uint64_t ev_mask[64];
};

__device__ __forceinline__ void priv_store(WaveLds &w, uint32_t *ovf, int lane, int idx, uint32_t val) {
    if (idx < kPrivWords) w.priv[idx * 64 + lane] = val;
    else ovf[(size_t)(idx - kPrivWords) * 64 + lane] = val;
}
__device__ __forceinline__ uint32_t priv_load(const WaveLds &w, const uint32_t *ovf, int blk, int idx) {
    if (idx < kPrivWords) return w.priv[idx * 64 + blk];
    return ovf[(size_t)(idx - kPrivWords) * 64 + blk];
}

struct BitAcc {
    uint64_t acc = 0;   // low `cnt` bits pending
    int cnt = 0;        // < 32 between appends
    int nwords = 0;
};

__device__ __forceinline__ void append_bits(BitAcc &b, WaveLds &w, uint32_t *ovf, int lane, uint32_t bits,
                                            int nbits /*0..27*/) {
    b.acc = (b.acc << nbits) | bits;
    b.cnt += nbits;
    if (b.cnt >= 32) {
        b.cnt -= 32;
        priv_store(w, ovf, lane, b.nwords, (uint32_t)(b.acc >> b.cnt));
        b.nwords++;
    }
}

// size category and amplitude bits (rle.c:9-35)
__device__ __forceinline__ int mag_bits(int v) { return 32 - __clz(abs(v)); }   // v != 0
__device__ __forceinline__ uint32_t amp_bits(int v, int nbits) {
    return (uint32_t)(v + (v >> 31)) & ((1u << nbits) - 1u);
}

// floor to int in one instruction (the compiler emits v_floor_f32 + v_cvt_i32_f32)
__device__ __forceinline__ int cvt_floor_i32(float x) {
    int r;
    asm("v_cvt_flr_i32_f32_e32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}

// kStd: constants of the reference's own table baked in as instruction literals (no scalar
// loads in the 63 quantisation sites); otherwise they come from the kernel argument block.
#ifndef JPEGAMD_WAVES_PER_EU
#define JPEGAMD_WAVES_PER_EU 0
#endif
#if JPEGAMD_WAVES_PER_EU > 0
#define JPEGAMD_OCCUPANCY __attribute__((amdgpu_waves_per_eu(JPEGAMD_WAVES_PER_EU, JPEGAMD_WAVES_PER_EU)))
#else
#define JPEGAMD_OCCUPANCY
#endif
template <bool kTaps, bool kStd>
__global__ __launch_bounds__(64 * kWavesPerGroup) JPEGAMD_OCCUPANCY void k_transform(const ImageDesc im, const QuantConsts qc,
                                                                   const TransformOut out) {
    __shared__ uint32_t s_ac[256];
    __shared__ uint32_t s_dc[16];
    __shared__ float s_cos[64];
    __shared__ WaveLds s_wave[kWavesPerGroup];

    {
        const int t = (int)threadIdx.x;
        s_ac[t] = out.huff[t];
        if (t < 16) s_dc[t] = out.huff[256 + t];
        if (t < 64) s_cos[t] = kCosFM[t];
    }
    __syncthreads();

    const int lane = lane_id();
    const int wave = (int)(threadIdx.x >> 6);
    const int seg = (int)blockIdx.x * kWavesPerGroup + wave;
    if (seg >= im.num_segs) return;       // whole wave; no block-level sync below
    WaveLds &wl = s_wave[wave];

    const int by = seg / im.segs_per_row;
    const int bx0 = (seg - by * im.segs_per_row) * kSegBlocks;
    const int nblk = min(kSegBlocks, im.blocks_w - bx0);
    const bool active = lane < nblk;
    const int bx = bx0 + min(lane, nblk - 1);      // idle lanes shadow the last block
    const int px0 = bx * 8, py0 = by * 8;

    // ---- 1. load + luma -----------------------------------------------------------------
    // d[] holds UNSHIFTED luma (0..255).  The level shift (converter.c:84-86) only moves the
    // DC term: every other output of the flow graph is a difference of exact integer sums, so
    // the AC values are bit-identical with or without it, and DC = sum - 64*128 exactly.
    float d[64];
    const bool interior = im.fast_ok && ((bx0 + nblk) * 8 <= im.width) && (py0 + 8 <= im.height);
    if (interior) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
            luma_row8(reinterpret_cast<const uint32_t *>(row_ptr(im, py0 + r) + 3 * (size_t)px0), im.weights, &d[r * 8]);
    } else {
        // Edge tile (right/bottom replication, converter.c:31,36) or unaligned source:
        // byte-wise gather staged through LDS so the register file keeps static indices.
        uint8_t *stage = reinterpret_cast<uint8_t *>(wl.priv);
#pragma unroll 1
        for (int i = 0; i < 64; ++i) {
            const int yv = luma_clamped(im, px0 + (i & 7), py0 + (i >> 3));
            stage[(i >> 2) * 256 + lane * 4 + (i & 3)] = (uint8_t)yv;
        }
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            const uint32_t wv = wl.priv[g * 64 + lane];
#pragma unroll
            for (int j = 0; j < 4; ++j) d[g * 4 + j] = (float)((wv >> (8 * j)) & 0xFFu);
        }
    }

    if (kTaps && active && out.tap_y) {
        int8_t *ty = out.tap_y + ((size_t)by * im.blocks_w + bx) * 64;
#pragma unroll
        for (int i = 0; i < 64; ++i) ty[i] = (int8_t)((int)d[i] - 128);   // converter.c:84-86
    }

    // ---- 2. fast 2-D DCT ----------------------------------------------------------------
#if JPEGAMD_ABLATE < 3
#pragma unroll
    for (int r = 0; r < 8; ++r) aan8<1>(&d[r * 8]);
#pragma unroll
    for (int c = 0; c < 8; ++c) aan8<8>(&d[c]);
#endif

    // ---- 3. quantise with guard band; record sites needing the exact order --------------
    int n[64];
    // DC: d[0] - 8192 is the exact integer sum of the 64 centred pixels, so the reference's
    // sequential float sum equals it and fl(K00 * S) / q reproduces dct.c:93 + quantization.c:36.
    n[0] = ref_quantise(__fmul_rn(ref_scale(0, 0), d[0] - 8192.0f), qc.qstep[0]);
    const unsigned long long active_mask = __ballot(active);
    float bias_v = kStdBias;
    if (kStd) asm volatile("v_mov_b32 %0, %1" : "=v"(bias_v) : "s"(kStdBias));   // keep it in a VGPR
    int nev = 0;
#pragma unroll
    for (int k = 1; k < 64; ++k) {
        // zc = z + 0.5 + delta; floor(zc) is the rounded quotient unless fract(zc) <= thr_k,
        // i.e. unless z is within delta_k of a tie.
        const float zc = kStd ? fmaf(d[k], kStdMult[k], bias_v) : fmaf(d[k], qc.mult[k], qc.bias[k]);
        const float g = __builtin_amdgcn_fractf(zc);
        n[k] = cvt_floor_i32(zc);
        const unsigned long long m = __ballot(g <= (kStd ? kStdThr[k] : qc.thr[k])) & active_mask;
        if (__builtin_expect(m != 0ull, 0)) {
            if (lane == 0) {
                wl.ev_k[nev] = (uint32_t)k;
                wl.ev_mask[nev] = m;
            }
            ++nev;
        }
    }

    // ---- 4. exact-order recomputation of the flagged coefficients -----------------------
    int nexact = 0;
    uint64_t exact_mask = 0;
    if (__builtin_expect(nev != 0, 0)) {
#pragma unroll 1
        for (int e = 0; e < nev; ++e) {
            const int k = uniform((int)wl.ev_k[e]);
            unsigned long long m = wl.ev_mask[e];
            m = ((unsigned long long)(uint32_t)uniform((int)(m >> 32)) << 32) | (uint32_t)uniform((int)m);
            const int u = k >> 3, v = k & 7;
            while (m) {
                const int b = __ffsll((long long)m) - 1;
                m &= m - 1;
                const float pix = (float)(luma_clamped(im, (bx0 + b) * 8 + (lane & 7), py0 + (lane >> 3)) - 128);
                const float coef = exact_coef_float(pix, u, v, s_cos, lane);
                const int val = ref_quantise(coef, qc.qstep[k]);
                ++nexact;
                if (kTaps && lane == b) exact_mask |= 1ull << k;
                // Write-back into the statically indexed register array: a select chain (a
                // 64-way switch here made hipcc keep ~300 VGPRs live; this keeps 155).
#pragma unroll
                for (int K = 1; K < 64; ++K) n[K] = (k == K && lane == b) ? val : n[K];
            }
        }
    }

    if (kTaps && active) {
        const size_t blk = (size_t)by * im.blocks_w + bx;
        if (out.tap_zz) {
#pragma unroll
            for (int i = 0; i < 64; ++i) out.tap_zz[blk * 64 + i] = (int16_t)n[kZZ[i]];
        }
        if (out.tap_mask) out.tap_mask[blk] = exact_mask;
    }

    // ---- 5. DC prediction (rle.c:59-70): predecessor in raster block order ---------------
    int pred_first = 0;      // wave-uniform: quantised DC of the block before this segment
    {
        int pbx = bx0 - 1, pby = by;
        if (pbx < 0) { pbx = im.blocks_w - 1; pby = by - 1; }
        if (pby >= 0) {
            const int yv = luma_clamped(im, pbx * 8 + (lane & 7), pby * 8 + (lane >> 3)) - 128;
            const int s = wave_sum_i32(yv);
            pred_first = ref_quantise(__fmul_rn(ref_scale(0, 0), (float)s), qc.qstep[0]);
        }
    }
    int pred = __shfl_up(n[0], 1, 64);
    if (lane == 0) pred = pred_first;

    // ---- 6. run/size symbols + Huffman codes into the lane's private bit string -----------
    uint32_t *ovf = out.ovf_words + (size_t)seg * (kOvfWords * 64);
    BitAcc ba;
    int nsym = 0;
    if (active) {
        {   // DC (rle.c:68-76, huffman.c:145-153)
            const int diff = n[0] - pred;
            const int nb = diff ? mag_bits(diff) : 0;
            const uint32_t hc = s_dc[nb];
            const uint32_t code = ((hc & 0xFFFFu) << nb) | (nb ? amp_bits(diff, nb) : 0u);
            append_bits(ba, wl, ovf, lane, code, (int)(hc >> 16) + nb);
            ++nsym;
        }
#if JPEGAMD_ABLATE >= 1
        int last = 0; { int acc_ = 0;
#pragma unroll
        for (int i = 1; i < 64; ++i) acc_ |= n[kZZ[i]]; if (acc_ == 12345) last = 5; }
#else
        int last = 0;
#pragma unroll
        for (int i = 1; i < 64; ++i) {
            const int v = n[kZZ[i]];
            if (v != 0) {
                int run = i - last - 1;
                last = i;
                while (run >= 16) {                       // ZRL (rle.c:99-103): 11111111001
                    const uint32_t hz = s_ac[0xF0];
                    append_bits(ba, wl, ovf, lane, hz & 0xFFFFu, (int)(hz >> 16));
                    run -= 16;
                    ++nsym;
                }
                const int nb = mag_bits(v);
                const uint32_t hc = s_ac[(run << 4) | nb];        // rle.c:110, huffman.c:165
                const uint32_t code = ((hc & 0xFFFFu) << nb) | amp_bits(v, nb);
                append_bits(ba, wl, ovf, lane, code, (int)(hc >> 16) + nb);
                ++nsym;
            }
        }
#endif
        if (last != 63) {                                 // EOB (rle.c:121-123)
            const uint32_t he = s_ac[0x00];
            append_bits(ba, wl, ovf, lane, he & 0xFFFFu, (int)(he >> 16));
            ++nsym;
        }
        if (ba.cnt > 0) priv_store(wl, ovf, lane, ba.nwords, (uint32_t)(ba.acc << (32 - ba.cnt)));
    }
    const uint32_t my_bits = active ? (uint32_t)(ba.nwords * 32 + ba.cnt) : 0u;

    // ---- 7. segment assembly: scan block lengths, gather words, store ---------------------
    const uint32_t incl = wave_incl_scan_u32(my_bits, lane);
    wl.offs[lane] = incl - my_bits;
    const uint32_t total = __shfl(incl, 63, 64);
    if (lane == 0) { wl.offs[64] = total; wl.offs[65] = total; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0);   // private words (LDS + HBM overflow) and offsets are visible to the wave
    __builtin_amdgcn_wave_barrier();

#if JPEGAMD_ABLATE < 2
    uint32_t *segw = out.seg_words + (size_t)seg * kSegCapWords;
    const uint32_t nwords = (total + 31u) >> 5;
#pragma unroll 1
    for (uint32_t j = (uint32_t)lane; j < nwords; j += 64) {
        const uint32_t pos0 = j * 32u;
        int b = 0;
#pragma unroll
        for (int step = 32; step > 0; step >>= 1)
            if (wl.offs[b + step] <= pos0) b += step;
        uint32_t word = 0;
        int filled = 0;
        while (filled < 32 && b < 64) {
            const uint32_t ob = wl.offs[b];
            const int rel = (int)(pos0 + (uint32_t)filled - ob);
            const int avail = (int)(wl.offs[b + 1] - ob) - rel;
            if (avail <= 0) { ++b; continue; }
            const int take = min(avail, 32 - filled);
            const int wi = rel >> 5, sh = rel & 31;
            const uint64_t win = ((uint64_t)priv_load(wl, ovf, b, wi) << 32) | priv_load(wl, ovf, b, wi + 1);
            const uint32_t top = (uint32_t)((win << sh) >> 32);
            const uint32_t chunk = top >> (32 - take);
            word |= chunk << (32 - filled - take);
            filled += take;
            if (take == avail) ++b;
        }
        segw[j] = word;
    }

#endif
    const int seg_syms = wave_sum_i32(nsym);
    if (lane == 0) {
        out.seg_bits[seg] = total;
        out.seg_syms[seg] = (uint32_t)seg_syms;
        out.seg_exact[seg] = (uint32_t)nexact;
    }
}

int launch_transform(const ImageDesc &im, const QuantConsts &qc, const TransformOut &out, bool taps, bool std_table,
                     void *stream) {
    const dim3 grid((im.num_segs + kWavesPerGroup - 1) / kWavesPerGroup), block(64 * kWavesPerGroup);
    hipStream_t s = (hipStream_t)stream;
    if (taps && std_table) hipLaunchKernelGGL((k_transform<true, true>), grid, block, 0, s, im, qc, out);
    else if (taps) hipLaunchKernelGGL((k_transform<true, false>), grid, block, 0, s, im, qc, out);
    else if (std_table) hipLaunchKernelGGL((k_transform<false, true>), grid, block, 0, s, im, qc, out);
    else hipLaunchKernelGGL((k_transform<false, false>), grid, block, 0, s, im, qc, out);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------
// Exclusive prefix sums over segments (one workgroup; n is a few thousand)
// ------------------------------------------------------------------------------------
constexpr int kScanThreads = 1024;

__device__ __forceinline__ uint64_t wave_incl_scan_u64(uint64_t v, int lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint64_t t = __shfl_up(v, off, 64);
        if (lane >= off) v += t;
    }
    return v;
}

// Block-wide exclusive scan of one u64 per thread; returns exclusive prefix, *total for all.
__device__ __forceinline__ uint64_t block_excl_scan_u64(uint64_t v, uint64_t *s_wave_tot /*[17]*/, uint64_t *total) {
    const int lane = lane_id(), wave = (int)(threadIdx.x >> 6);
    const uint64_t incl = wave_incl_scan_u64(v, lane);
    if (lane == 63) s_wave_tot[wave] = incl;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t run = 0;
        for (int w = 0; w < kScanThreads / 64; ++w) { const uint64_t t = s_wave_tot[w]; s_wave_tot[w] = run; run += t; }
        s_wave_tot[16] = run;
    }
    __syncthreads();
    *total = s_wave_tot[16];
    return s_wave_tot[wave] + incl - v;
}

// Tiles of kScanThreads*16 elements: every thread owns 16 consecutive values (4 coalesced
// 16-byte loads, all issued before the first use), scans them in registers, the block scans
// the per-thread sums, and the running base carries to the next tile.
__global__ __launch_bounds__(kScanThreads) void k_scan_segments(const uint32_t *__restrict__ in,
                                                                const uint32_t *__restrict__ aux0,
                                                                const uint32_t *__restrict__ aux1,
                                                                uint64_t *__restrict__ out, int n,
                                                                ScanStats *stats, int which) {
    __shared__ uint64_t s_tot[17];
    __shared__ uint64_t s_aux[2][kScanThreads / 64];
    constexpr int kPer = 16;
    uint64_t base = 0, a0 = 0, a1 = 0;
    for (int tile = 0; tile < n; tile += kScanThreads * kPer) {
        const int begin = tile + (int)threadIdx.x * kPer;
        uint32_t v[kPer];
        if (begin + kPer <= n) {
            const uint4 *p = reinterpret_cast<const uint4 *>(in + begin);     // hipMalloc base, begin % 16 == 0
#pragma unroll
            for (int j = 0; j < kPer / 4; ++j) {
                const uint4 q = p[j];
                v[4 * j] = q.x; v[4 * j + 1] = q.y; v[4 * j + 2] = q.z; v[4 * j + 3] = q.w;
            }
            if (aux0) {
                const uint4 *pa = reinterpret_cast<const uint4 *>(aux0 + begin), *pb = reinterpret_cast<const uint4 *>(aux1 + begin);
#pragma unroll
                for (int j = 0; j < kPer / 4; ++j) {
                    const uint4 qa = pa[j], qb = pb[j];
                    a0 += (uint64_t)qa.x + qa.y + qa.z + qa.w;
                    a1 += (uint64_t)qb.x + qb.y + qb.z + qb.w;
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < kPer; ++j) {
                const int i = begin + j;
                v[j] = i < n ? in[i] : 0u;
                if (aux0 && i < n) { a0 += aux0[i]; a1 += aux1[i]; }
            }
        }
        uint64_t sum = 0;
#pragma unroll
        for (int j = 0; j < kPer; ++j) sum += v[j];
        uint64_t total;
        uint64_t run = base + block_excl_scan_u64(sum, s_tot, &total);
#pragma unroll
        for (int j = 0; j < kPer; ++j) {
            if (begin + j < n) out[begin + j] = run;
            run += v[j];
        }
        base += total;
        __syncthreads();                       // s_tot is reused by the next tile
    }
    if (threadIdx.x == 0) out[n] = base;
    const int lane = lane_id(), wave = (int)(threadIdx.x >> 6);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { a0 += __shfl_xor(a0, off, 64); a1 += __shfl_xor(a1, off, 64); }
    if (lane == 0) { s_aux[0][wave] = a0; s_aux[1][wave] = a1; }
    __syncthreads();
    if (threadIdx.x == 0 && stats) {
        uint64_t t0 = 0, t1 = 0;
        for (int w = 0; w < kScanThreads / 64; ++w) { t0 += s_aux[0][w]; t1 += s_aux[1][w]; }
        if (which == 0) { stats->total_bits = base; stats->total_syms = t0; stats->total_exact = t1; stats->status = 0u; }
        else stats->total_ff = base;
    }
}

int launch_scan_bits(const uint32_t *seg_bits, const uint32_t *seg_syms, const uint32_t *seg_exact,
                     uint64_t *seg_bitstart, int num_segs, ScanStats *stats, void *stream) {
    hipLaunchKernelGGL(k_scan_segments, dim3(1), dim3(kScanThreads), 0, (hipStream_t)stream, seg_bits, seg_syms,
                       seg_exact, seg_bitstart, num_segs, stats, 0);
    return (int)hipGetLastError();
}
int launch_scan_ff(const uint32_t *seg_ff, uint64_t *seg_ffstart, int num_segs, ScanStats *stats, void *stream) {
    hipLaunchKernelGGL(k_scan_segments, dim3(1), dim3(kScanThreads), 0, (hipStream_t)stream, seg_ff,
                       (const uint32_t *)nullptr, (const uint32_t *)nullptr, seg_ffstart, num_segs, stats, 1);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------
// Byte extraction shared by k_count_ff and k_pack.
//
// Output byte i of the unstuffed stream (bits 8i .. 8i+7) is OWNED by the segment that
// contains its last bit.  Segment s with bits [B0, B1) therefore owns bytes
// [B0>>3, B1>>3); its first owned byte may start with `lead = B0 & 7` bits of earlier
// segments.
// ------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t seg_bits_at(const uint32_t *__restrict__ w, uint32_t pos, int nbits /*1..8*/) {
    const uint32_t i = pos >> 5, sh = pos & 31u;
    const uint64_t win = ((uint64_t)w[i] << 32) | w[i + 1];
    return (uint32_t)((win << sh) >> (64 - nbits));
}

// The `need` (1..7) bits that precede segment `s` in the stream (s may equal num_segs).
__device__ __forceinline__ uint32_t tail_bits_before(const PackArgs &a, int s, int need) {
    uint32_t val = 0;
    int got = 0;
    int sp = s - 1;
    while (got < need && sp >= 0) {
        const uint32_t tp = a.seg_bits[sp];
        const int take = min(need - got, (int)tp);
        if (take > 0) {
            const uint32_t bits = seg_bits_at(a.seg_words + (size_t)sp * kSegCapWords, tp - (uint32_t)take, take);
            val |= bits << got;
            got += take;
        }
        --sp;
    }
    return val;
}

struct SegView {
    const uint32_t *words;
    uint64_t b0, b1;
    uint32_t nown;
    int lead;
    uint32_t leadbits;
};

__device__ __forceinline__ SegView seg_view(const PackArgs &a, int s) {
    SegView v;
    v.words = a.seg_words + (size_t)s * kSegCapWords;
    v.b0 = a.seg_bitstart[s];
    v.b1 = v.b0 + a.seg_bits[s];
    v.nown = (uint32_t)((v.b1 >> 3) - (v.b0 >> 3));
    v.lead = (int)(v.b0 & 7u);
    v.leadbits = v.lead ? tail_bits_before(a, s, v.lead) : 0u;
    return v;
}

__device__ __forceinline__ uint32_t owned_byte(const SegView &v, uint32_t r) {
    if (r == 0 && v.lead) return (v.leadbits << (8 - v.lead)) | seg_bits_at(v.words, 0, 8 - v.lead);
    return seg_bits_at(v.words, 8u * r - (uint32_t)v.lead, 8);
}

__global__ __launch_bounds__(256) void k_count_ff(const PackArgs a) {
    const int lane = lane_id();
    const int s = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6);
    if (s >= a.num_segs) return;
    const SegView v = seg_view(a, s);
    int cnt = 0;
    for (uint32_t r = (uint32_t)lane; r < v.nown; r += 64) cnt += (owned_byte(v, r) == 0xFFu);
    cnt = wave_sum_i32(cnt);
    if (lane == 0) a.seg_ff[s] = (uint32_t)cnt;
}

__global__ __launch_bounds__(256) void k_pack(const PackArgs a) {
    const int lane = lane_id();
    const int s = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6);
    if (s >= a.num_segs) return;

    if (s == 0 && a.prefix_len > 0) {            // JFIF prefix (jpeg_handler.c:220-233)
        for (int i = lane; i < a.prefix_len; i += 64)
            if ((uint64_t)i < a.out_capacity) a.out[i] = a.prefix[i];
    }

    const SegView v = seg_view(a, s);
    const uint64_t base = (uint64_t)a.prefix_len + (v.b0 >> 3) + a.seg_ffstart[s];
    uint32_t running = 0;
    bool overflow = false;
    for (uint32_t r0 = 0; r0 < v.nown; r0 += 64) {
        const uint32_t r = r0 + (uint32_t)lane;
        const bool valid = r < v.nown;
        const uint32_t byte = valid ? owned_byte(v, r) : 0u;
        const bool isff = valid && byte == 0xFFu;
        const unsigned long long m = __ballot(isff);
        const uint32_t before = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        const uint64_t pos = base + r + running + before;
        if (valid) {
            if (pos + (isff ? 2u : 1u) <= a.out_capacity) {
                a.out[pos] = (uint8_t)byte;
                if (isff) a.out[pos + 1] = 0x00;          // huffman.c:29-31
            } else {
                overflow = true;
            }
        }
        running += (uint32_t)__popcll(m);
    }
    if (__any(overflow) && lane == 0) atomicOr(&a.stats->status, 1u);

    if (s == a.num_segs - 1 && lane == 0) {
        uint64_t end = base + v.nown + running;
        const int rem = (int)(v.b1 & 7u);
        bool ok = true;
        if (rem) {                                          // zero-padded flush (huffman.c:65-81)
            const uint32_t bits = tail_bits_before(a, a.num_segs, rem);
            if (end < a.out_capacity) a.out[end] = (uint8_t)(bits << (8 - rem)); else ok = false;
            ++end;
        }
        if (a.write_eoi) {                                  // jpeg_handler.c:113-117
            if (end + 2 <= a.out_capacity) { a.out[end] = 0xFF; a.out[end + 1] = 0xD9; } else ok = false;
            end += 2;
        }
        if (!ok) atomicOr(&a.stats->status, 1u);
        *a.out_size = end;
        a.stats->out_size = end;
    }
}

int launch_count_ff(const PackArgs &a, void *stream) {
    hipLaunchKernelGGL(k_count_ff, dim3((a.num_segs + 3) / 4), dim3(256), 0, (hipStream_t)stream, a);
    return (int)hipGetLastError();
}
int launch_pack(const PackArgs &a, void *stream) {
    hipLaunchKernelGGL(k_pack, dim3((a.num_segs + 3) / 4), dim3(256), 0, (hipStream_t)stream, a);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------
// Exact-order DCT of arbitrary blocks (parity tap for dct.c:63-96): one wave per block.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_dct_exact(const int8_t *__restrict__ blocks, float *__restrict__ coeffs,
                                                  long long nblocks) {
    __shared__ float s_cos[64];
    const int lane = lane_id();
    s_cos[lane] = kCosFM[lane];
    __syncthreads();
    for (long long blk = blockIdx.x; blk < nblocks; blk += gridDim.x) {
        const float pix = (float)blocks[blk * 64 + lane];
#pragma unroll 1
        for (int k = 0; k < 64; ++k) {
            const float c = exact_coef_float(pix, k >> 3, k & 7, s_cos, lane);
            if (lane == 0) coeffs[blk * 64 + k] = c;
        }
    }
}

int launch_dct_exact(const int8_t *blocks, float *coeffs, int64_t nblocks, void *stream) {
    const int grid = (int)(nblocks < 4096 ? nblocks : 4096);
    if (grid <= 0) return 0;
    hipLaunchKernelGGL(k_dct_exact, dim3(grid), dim3(64), 0, (hipStream_t)stream, blocks, coeffs, (long long)nblocks);
    return (int)hipGetLastError();
}

}  // namespace jpegamd
