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

#include "jpegamd_device.h"

namespace jpegamd {

#include "std_table_consts.inc"

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

// ------------------------------------------------------------------------------------
// Symbol-parallel entropy back-end.
//
// The per-lane loop of the first back-end keeps ~16 % of the lanes busy (a lane only works at
// the zigzag positions where ITS block is non-zero).  Here each lane only *records* its
// non-zeros -- (run, value) items appended to its own range of an LDS list (ranges come from
// a wave prefix sum of per-lane counts) -- and then the wave walks the list 64 items at a
// time with one lane per SYMBOL: size, amplitude, Huffman code and length (rle.c:9-35,99-123;
// huffman.c:145-188), a wave prefix sum of the lengths gives every symbol its bit offset in
// the segment, and the bits are OR-ed into a small LDS window that is flushed to the
// segment's word array as it fills.  Items per block: DC, the non-zero ACs in zigzag order,
// EOB when coefficient 63 is zero.  A run >= 16 becomes 1..3 ZRL prefixes on the item.
//
// List capacity is bounded (kItemCap); segments with more symbols are processed in groups of
// consecutive blocks, so dense / adversarial content stays correct, only slower.
// ------------------------------------------------------------------------------------
constexpr int kItemCap = (kPrivWords + 1) * 64;     // 1024 items in the former private-word area
constexpr uint32_t kItemDc = 0x80000000u;

__device__ __forceinline__ uint32_t entropy_symbol_parallel(const int (&n)[64], int pred, bool active, int lane,
                                                            WaveLds &wl, const uint32_t *s_huff /*ac[256] dc[16]*/,
                                                            uint32_t *__restrict__ segw, int &nsym_out) {
    uint32_t *items = wl.priv;                      // [kItemCap]
    uint32_t *win = reinterpret_cast<uint32_t *>(wl.ev_mask);   // 128-word bit window (events are done)

    // ---- per-lane symbol count and its prefix sum --------------------------------------
    int nnz = 0;
#pragma unroll
    for (int i = 1; i < 64; ++i) nnz += (n[kZZ[i]] != 0) ? 1 : 0;
    const bool eob = n[kZZ[63]] == 0;                               // rle.c:121-123
    const uint32_t cnt = active ? (uint32_t)(1 + nnz + (eob ? 1 : 0)) : 0u;
    const uint32_t incl = wave_incl_scan_u32(cnt, lane);
    const uint32_t base = incl - cnt;
    const uint32_t t_all = __shfl(incl, 63, 64);

    uint32_t carry_bits = 0;          // bits emitted so far in this segment (wave-uniform)
    uint32_t wbase = 0;               // word index of win[0] in the segment
    int zrl_total = 0;
    if (lane == 0) win[0] = 0u;

    uint32_t gbase = 0;               // first item of the current group
    while (gbase < t_all) {
        // group = maximal run of consecutive lanes whose items fit the list (every block <= 65 items)
        const bool fits = (incl - gbase) <= (uint32_t)kItemCap;
        const bool mine = active && base >= gbase && fits;
        const unsigned long long gm = __ballot(mine);
        const uint32_t gend_incl = __shfl(incl, 63 - __builtin_clzll(gm), 64);   // last lane of the group
        const uint32_t gt = gend_incl - gbase;                                   // items in this group

        // ---- scatter: every lane appends its items to its own range ---------------------
        if (mine) {
            uint32_t ptr = base - gbase;
            items[ptr++] = kItemDc | (uint32_t)((n[0] - pred) & 0xFFFF);         // rle.c:68-76
            int last = 0;
#pragma unroll
            for (int i = 1; i < 64; ++i) {
                const int v = n[kZZ[i]];
                if (v != 0) {
                    items[ptr++] = ((uint32_t)(i - last - 1) << 16) | (uint32_t)(v & 0xFFFF);
                    last = i;
                }
            }
            if (eob) items[ptr] = 0u;                                             // value 0, not DC = EOB
        }

        // ---- one lane per symbol -------------------------------------------------------
        for (uint32_t b0 = 0; b0 < gt; b0 += 64) {
            const uint32_t idx = b0 + (uint32_t)lane;
            const bool valid = idx < gt;
            const uint32_t it = valid ? items[idx] : 0u;
            const int v = (int)(short)(it & 0xFFFFu);
            const bool isdc = (it & kItemDc) != 0u;
            const int run = (int)((it >> 16) & 0x7FFFu);
            const int nb = v ? (32 - __clz(abs(v))) : 0;                          // rle.c:9-22
            const uint32_t amp = (uint32_t)(v + (v >> 31)) & ((1u << nb) - 1u);   // rle.c:24-35
            const uint32_t hc = s_huff[isdc ? (256 + nb) : (((run & 15) << 4) | nb)];
            uint32_t hi = ((hc & 0xFFFFu) << nb) | amp;                           // code bits then amplitude bits
            uint32_t lo = 0;
            int len = valid ? (int)(hc >> 16) + nb : 0;                           // <= 27
            const int zrl = (valid && !isdc) ? (run >> 4) : 0;                    // rle.c:99-103
            hi <<= (32 - len) & 31;                                               // left-align (len == 0 -> hi == 0 anyway)
            if (len == 0) hi = 0;
            if (__builtin_expect(__any(zrl != 0), 0)) {
                // prepend zrl x "11111111001" (huffman code of 0xF0): up to 33 + 27 bits
                const uint32_t z = s_huff[0xF0];
                const uint32_t zc = z & 0xFFFFu;
                const int zl = (int)(z >> 16);
                unsigned long long acc = ((unsigned long long)hi << 32);
                int tot = len;
                for (int r = 0; r < 3; ++r) {
                    if (r < zrl) { acc = (acc >> zl) | ((unsigned long long)zc << (64 - zl)); tot += zl; }
                }
                hi = (uint32_t)(acc >> 32);
                lo = (uint32_t)acc;
                len = tot;
                zrl_total += zrl;
            }
            const uint32_t incl_b = wave_incl_scan_u32((uint32_t)len, lane);
            const uint32_t batch_bits = __shfl(incl_b, 63, 64);
            const uint32_t rel = carry_bits + incl_b - (uint32_t)len - wbase * 32u;   // bit offset inside the window
            // clear the part of the window this batch can reach (win[0] holds the carried partial word)
            win[1 + lane] = 0u;
            if (lane < 63) win[65 + lane] = 0u;
            if (len) {
                const uint32_t w = rel >> 5, sh = rel & 31u;
                atomicOr(&win[w], __builtin_amdgcn_alignbit(0u, hi, sh));
                const uint32_t w1 = __builtin_amdgcn_alignbit(hi, lo, sh);
                if (w1) atomicOr(&win[w + 1], w1);
                const uint32_t w2 = __builtin_amdgcn_alignbit(lo, 0u, sh);
                if (w2) atomicOr(&win[w + 2], w2);
            }
            carry_bits += batch_bits;
            const uint32_t done = (carry_bits >> 5) - wbase;        // complete words now in the window (<= 121)
            if ((uint32_t)lane < done) segw[wbase + lane] = win[lane];
            if ((uint32_t)lane + 64u < done) segw[wbase + 64u + lane] = win[64 + lane];
            const uint32_t part = win[done];                        // every lane reads it before lane 0 overwrites win[0]
            if (lane == 0) win[0] = part;
            wbase += done;
        }
        gbase = gend_incl;
    }
    if ((carry_bits & 31u) && lane == 0) segw[wbase] = win[0];      // last partial word, zero-padded
    nsym_out = (int)cnt + zrl_total;      // per-lane: symbols of this block (+ the ZRLs this lane coded); summed by the caller
    return carry_bits;
}

#if JPEGAMD_WAVES_PER_EU > 0
#define JPEGAMD_OCCUPANCY __attribute__((amdgpu_waves_per_eu(JPEGAMD_WAVES_PER_EU, JPEGAMD_WAVES_PER_EU)))
#else
#define JPEGAMD_OCCUPANCY
#endif
template <bool kTaps, bool kStd, int kEntropy>
__global__ __launch_bounds__(64 * kWavesPerGroup) JPEGAMD_OCCUPANCY void k_transform(const ImageDesc im, const QuantConsts qc,
                                                                   const TransformOut out) {
    __shared__ uint32_t s_huff[272];          // AC codes [0,256) then DC codes [256,272): len<<16 | code
    uint32_t *const s_ac = s_huff;
    uint32_t *const s_dc = s_huff + 256;
    __shared__ float s_cos[64];
    __shared__ WaveLds s_wave[kWavesPerGroup];

    {
        const int t = (int)threadIdx.x;
        s_huff[t] = out.huff[t];
        if (blockIdx.x == 0 && t == 0 && out.reset.stats) out.reset.stats->status = 0u;   // cleared for this call's finalize kernels
        if (t < 16) s_huff[256 + t] = out.huff[256 + t];
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
#pragma unroll
    for (int r = 0; r < 8; ++r) aan8<1>(&d[r * 8]);
#pragma unroll
    for (int c = 0; c < 8; ++c) aan8<8>(&d[c]);

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

    uint32_t *segw = out.seg_words + (size_t)seg * kSegCapWords;
    uint32_t total = 0;
    int nsym = 0;
    if constexpr (kEntropy == 1) {
        total = entropy_symbol_parallel(n, pred, active, lane, wl, s_huff, segw, nsym);
    } else {
    // ---- 6. run/size symbols + Huffman codes into the lane's private bit string -----------
    uint32_t *ovf = out.ovf_words + (size_t)seg * (kOvfWords * 64);
    BitAcc ba;
    if (active) {
        {   // DC (rle.c:68-76, huffman.c:145-153)
            const int diff = n[0] - pred;
            const int nb = diff ? mag_bits(diff) : 0;
            const uint32_t hc = s_dc[nb];
            const uint32_t code = ((hc & 0xFFFFu) << nb) | (nb ? amp_bits(diff, nb) : 0u);
            append_bits(ba, wl, ovf, lane, code, (int)(hc >> 16) + nb);
            ++nsym;
        }
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
    total = __shfl(incl, 63, 64);
    if (lane == 0) { wl.offs[64] = total; wl.offs[65] = total; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0);   // private words (LDS + HBM overflow) and offsets are visible to the wave
    __builtin_amdgcn_wave_barrier();

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

    }
    const int seg_syms = wave_sum_i32(nsym);
    if (lane == 0) {
        out.seg_bits[seg] = total;
        out.seg_syms[seg] = (uint32_t)seg_syms;
        out.seg_exact[seg] = (uint32_t)nexact;
    }
}

int launch_transform(const ImageDesc &im, const QuantConsts &qc, const TransformOut &out, bool taps, bool std_table,
                     int entropy_backend, void *stream) {
    const dim3 grid((im.num_segs + kWavesPerGroup - 1) / kWavesPerGroup), block(64 * kWavesPerGroup);
    hipStream_t s = (hipStream_t)stream;
#define JPEGAMD_LAUNCH(T, S, E) hipLaunchKernelGGL((k_transform<T, S, E>), grid, block, 0, s, im, qc, out)
    if (entropy_backend == 1) {
        if (taps && std_table) JPEGAMD_LAUNCH(true, true, 1);
        else if (taps) JPEGAMD_LAUNCH(true, false, 1);
        else if (std_table) JPEGAMD_LAUNCH(false, true, 1);
        else JPEGAMD_LAUNCH(false, false, 1);
    } else {
        if (taps && std_table) JPEGAMD_LAUNCH(true, true, 0);
        else if (taps) JPEGAMD_LAUNCH(true, false, 0);
        else if (std_table) JPEGAMD_LAUNCH(false, true, 0);
        else JPEGAMD_LAUNCH(false, false, 0);
    }
#undef JPEGAMD_LAUNCH
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
            const uint32_t bits = seg_bits_at(a.seg_words + (size_t)sp * a.seg_stride, tp - (uint32_t)take, take);
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
    v.words = a.seg_words + (size_t)s * a.seg_stride;
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
