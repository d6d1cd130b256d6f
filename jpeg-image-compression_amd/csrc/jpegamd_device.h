// jpegamd_device.h -- device-side helpers shared by the transform kernels.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "jpegamd_internal.h"

namespace jpegamd {

// ------------------------------------------------------------------------------------
// Tables
// ------------------------------------------------------------------------------------

// zigzag position -> raster index (zigzag.c:7-15)
__device__ constexpr uint8_t kZZ[64] = {
    0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
    41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
    30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

// The reference's cosine LUT (dct.c:9-18) stored frequency-major: kCosFM[u*8+x] = COS_LUT[x][u].
static __constant__ float kCosFM[64] = {
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
__device__ __forceinline__ const uint8_t *row_ptr(const ImageDesc &im, const uint8_t *pixels, int y) {
    const int stored = im.bottom_up ? (im.height - 1 - y) : y;      // bmp_handler.c:109
    return pixels + (size_t)stored * (size_t)im.row_stride;
}

// Luma of image pixel (x, y) with the converter's edge clamp (converter.c:31,36,51).
__device__ __forceinline__ int luma_clamped(const ImageDesc &im, const uint8_t *pixels, int x, int y) {
    x = min(x, im.width - 1);
    y = min(y, im.height - 1);
    const uint8_t *p = row_ptr(im, pixels, y) + 3 * (size_t)x;
    const uint32_t w = im.weights;
    return (int)(((w & 0xFF) * p[0] + ((w >> 8) & 0xFF) * p[1] + ((w >> 16) & 0xFF) * p[2]) >> 8);
}

// ------------------------------------------------------------------------------------
// Wave helpers (wave = 64 lanes)
// ------------------------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63u); }

// Cross-lane helpers on DPP / v_readlane (no LDS crossbar round trips: __shfl_* compiles to
// ds_bpermute_b32, ~100+ cycles per step on the critical path of a wave).
template <int kCtrl, int kRowMask = 0xF>
__device__ __forceinline__ uint32_t dpp_or_zero(uint32_t v) {        // lanes without a source (or masked rows) read 0
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, kCtrl, kRowMask, 0xF, false);
}
// inclusive prefix sum inside each 32-lane half (rows of 16: row_shr 1,2,4,8, then row_bcast:15 into rows 1 and 3)
__device__ __forceinline__ uint32_t half_incl_scan_dpp(uint32_t v) {
    v += dpp_or_zero<0x111>(v);
    v += dpp_or_zero<0x112>(v);
    v += dpp_or_zero<0x114>(v);
    v += dpp_or_zero<0x118>(v);
    v += dpp_or_zero<0x142, 0xA>(v);
    return v;
}
// inclusive prefix sum across the wave
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v, int /*lane*/ = 0) {
    v = half_incl_scan_dpp(v);
    v += dpp_or_zero<0x143, 0xC>(v);                                  // row_bcast:31 into rows 2 and 3
    return v;
}
// maximum across the wave (DPP row_shr steps + row broadcasts, as the prefix sums above)
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
    v = max(v, dpp_or_zero<0x111>(v));
    v = max(v, dpp_or_zero<0x112>(v));
    v = max(v, dpp_or_zero<0x114>(v));
    v = max(v, dpp_or_zero<0x118>(v));
    v = max(v, dpp_or_zero<0x142, 0xA>(v));
    v = max(v, dpp_or_zero<0x143, 0xC>(v));
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ int wave_sum_i32(int v) {
    return __builtin_amdgcn_readlane((int)wave_incl_scan_u32((uint32_t)v), 63);
}
// value of lane - 1 (lane 0 reads 0): wave_shr:1
__device__ __forceinline__ int lane_shift_up1(int v) { return (int)dpp_or_zero<0x138>((uint32_t)v); }
// value of lane ^ 32 (v_permlane32_swap)
__device__ __forceinline__ uint32_t other_half(uint32_t v, int lane) {
    const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    return lane < 32 ? r[1] : r[0];
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

// floor(x) as int32 in one instruction (v_cvt_flr_i32_f32); x comes from an ordinary VALU op (no software hazard)
__device__ __forceinline__ int floor_to_int(float x) {
    int r;
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}

// Eight predicates folded into a register with 2 instructions each and no hazard padding: a compare writes a lane
// mask, an add-with-carry consumes it.  gfx950 needs 2 wait states between a VALU writing a lane mask and a VALU
// reading it (hipcc pads its own compare/select pairs with s_nop -- 3 issue slots per site); here three mask
// registers rotate so that two independent instructions always sit between a compare and its consumer.
//   shift_in_le8: bits = (bits << 8) | sum_j (g[j] <= t[j]) << j        (sites consumed 7 .. 0)
//   count_ne8   : c += number of non-zero v[j]
#define JPEGAMD_FOLD8(CMP, ADD, X7, X6, X5, X4, X3, X2, X1, X0)                                                   \
    CMP("%[A]", X7) CMP("%[B]", X6) CMP("vcc", X5) ADD("%[A]") CMP("%[A]", X4) ADD("%[B]") CMP("%[B]", X3) ADD("vcc") \
    CMP("vcc", X2) ADD("%[A]") CMP("%[A]", X1) ADD("%[B]") CMP("%[B]", X0) ADD("vcc") ADD("%[A]") ADD("%[B]")

__device__ __forceinline__ uint32_t shift_in_le8(uint32_t bits, const float (&g)[8], const float (&t)[8]) {
    unsigned long long ma, mb;
#define JPEGAMD_CMP(M, J) "v_cmp_le_f32 " M ", %[g" #J "], %[t" #J "]\n\t"
#define JPEGAMD_ADD(M) "v_addc_co_u32 %[b], " M ", %[b], %[b], " M "\n\t"
    asm(JPEGAMD_FOLD8(JPEGAMD_CMP, JPEGAMD_ADD, 7, 6, 5, 4, 3, 2, 1, 0)
        : [b] "+v"(bits), [A] "=&s"(ma), [B] "=&s"(mb)
        : [g0] "v"(g[0]), [g1] "v"(g[1]), [g2] "v"(g[2]), [g3] "v"(g[3]), [g4] "v"(g[4]), [g5] "v"(g[5]), [g6] "v"(g[6]), [g7] "v"(g[7]),
          [t0] "v"(t[0]), [t1] "v"(t[1]), [t2] "v"(t[2]), [t3] "v"(t[3]), [t4] "v"(t[4]), [t5] "v"(t[5]), [t6] "v"(t[6]), [t7] "v"(t[7])
        : "vcc");
#undef JPEGAMD_CMP
#undef JPEGAMD_ADD
    return bits;
}

__device__ __forceinline__ uint32_t count_ne8(uint32_t c, const int (&v)[8]) {
    unsigned long long ma, mb;
#define JPEGAMD_CMP(M, J) "v_cmp_ne_u32 " M ", 0, %[v" #J "]\n\t"
#define JPEGAMD_ADD(M) "v_addc_co_u32 %[c], " M ", %[c], 0, " M "\n\t"
    asm(JPEGAMD_FOLD8(JPEGAMD_CMP, JPEGAMD_ADD, 7, 6, 5, 4, 3, 2, 1, 0)
        : [c] "+v"(c), [A] "=&s"(ma), [B] "=&s"(mb)
        : [v0] "v"(v[0]), [v1] "v"(v[1]), [v2] "v"(v[2]), [v3] "v"(v[3]), [v4] "v"(v[4]), [v5] "v"(v[5]), [v6] "v"(v[6]), [v7] "v"(v[7])
        : "vcc");
#undef JPEGAMD_CMP
#undef JPEGAMD_ADD
    return c;
}

__device__ __forceinline__ float exact_coef_float(float pixel /*lane j: p[x=j>>3][y=j&7]*/, int u, int v,
                                                  const float *s_cos, int lane) {
    const float cx = s_cos[u * 8 + (lane >> 3)];     // COS_LUT[x][u]
    const float cy = s_cos[v * 8 + (lane & 7)];      // COS_LUT[y][v]
    const float t = __fmul_rn(__fmul_rn(pixel, cx), cy);             // dct.c:84
    const float s = exact_term_sum(t);
    return __fmul_rn(ref_scale(u, v), s);                            // dct.c:93
}

// The same ordered sum through LDS: every lane publishes its term, then EVERY lane adds the 64 terms in order
// (broadcast reads, 63 dependent v_add_f32 with hardware forwarding).  ~90 issue slots; the DPP chain above is a
// rolled loop of ~6 slots per step (DPP needs wait states behind the add that feeds it) -- ~380 slots per event,
// which at 12 k events per 8192^2 image was 12 % of the whole pipeline's instructions.
__device__ __forceinline__ float exact_term_sum_lds(float t, float *terms /*LDS, 64 floats owned by this wave*/, int lane) {
    terms[lane] = t;
    float s = 0.0f;                                                  // dct.c:68
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const float4 a = *reinterpret_cast<const float4 *>(&terms[8 * c]);
        const float4 b = *reinterpret_cast<const float4 *>(&terms[8 * c + 4]);
        s = __fadd_rn(s, a.x); s = __fadd_rn(s, a.y); s = __fadd_rn(s, a.z); s = __fadd_rn(s, a.w);
        s = __fadd_rn(s, b.x); s = __fadd_rn(s, b.y); s = __fadd_rn(s, b.z); s = __fadd_rn(s, b.w);
    }
    return s;
}

__device__ __forceinline__ float exact_coef_float_lds(float pixel /*lane j: p[x=j>>3][y=j&7]*/, int u, int v, const float *s_cos,
                                                      float *terms, int lane) {
    const float cx = s_cos[u * 8 + (lane >> 3)];     // COS_LUT[x][u]
    const float cy = s_cos[v * 8 + (lane & 7)];      // COS_LUT[y][v]
    const float t = __fmul_rn(__fmul_rn(pixel, cx), cy);             // dct.c:84
    return __fmul_rn(ref_scale(u, v), exact_term_sum_lds(t, terms, lane));   // dct.c:93
}

}  // namespace jpegamd
