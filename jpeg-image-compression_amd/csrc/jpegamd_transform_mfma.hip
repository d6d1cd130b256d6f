// jpegamd_transform_mfma.hip -- the transform + entropy kernel with the DCT on the matrix pipe.
//
// Why: the register-resident AAN kernel (jpegamd_kernels.hip) keeps 64 coefficients per lane
// (160 VGPRs, 3 waves/SIMD) and rocprofv3 shows it latency/issue-bound at ~30 % VALU utilisation while
// the matrix pipe idles.  Here the 2-D DCT of a block is ONE 64x64 matrix-vector product with the
// reference's own LUT products,  out[c] = sum_p COS_LUT[x][u]*COS_LUT[y][v] * pix[p]  (dct.c:72-93
// without the order: the guard band of quant_consts.cpp decides when the order matters), evaluated
// for 32 blocks at a time by 24 v_mfma_f32_32x32x16_bf16: the matrix is split into three bf16 terms
// (24 bits), the centred pixels (int8) are exact in bf16, every product is exact in f32.  Each lane
// then owns 32 coefficients (2 lanes per block), which halves the live registers, and VALU work
// (luma, quantisation, symbol compaction) overlaps with the MFMAs of other waves.
//
// Lane l = (h = l >> 5, b = l & 31):
//   B operand, k-step s: row 2s+h of block b (8 pixels)        -> each half-wave reads whole image rows
//   D result, chain H, register r: zigzag position 32h + 16H + r of block b
// so lane (0,b) holds zigzag 0..31 and lane (1,b) zigzag 32..63 of the same block, both in zigzag
// order: the symbol list of a block is "lane 0's items, then lane 1's".
//
// One wavefront = one segment = up to kSegTiles tiles (128 blocks) of one block row; the bit offset,
// the bit window and the DC predictor carry from tile to tile.  Entropy coding is symbol-parallel as
// in jpegamd_kernels.hip (items -> one lane per symbol -> scan -> OR into an LDS window).
#include "jpegamd_device.h"

namespace jpegamd {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#ifndef JPEGAMD_MFMA_WAVES
#define JPEGAMD_MFMA_WAVES 4
#endif
constexpr int kWavesM = 8;            // 512-thread workgroups share the 24 KiB matrix image in LDS
constexpr int kItemCapM = 512;
constexpr uint32_t kItemDcM = 0x80000000u;

struct WaveLdsM {
    uint32_t items[kItemCapM + 64 + 2];   // [0] = 0 sentinel, then (DC flag | zigzag position << 16 | value16); +64 read slack
    uint32_t win[128];                // bit window being assembled
};

// 8 pixels (24 bytes, 4-byte aligned) -> 8 centred luma values as bf16.  The -128 rides in the dot
// product's accumulator (C = -32768 = -128 * 256), so (int)dot >> 8 is Y - 128 (converter.c:51,84-86).
struct RawRow { uint32_t d[6]; };      // one block row: 8 pixels x 3 bytes

__device__ __forceinline__ RawRow load_raw_row(const uint32_t *__restrict__ src) {
    RawRow r;
#pragma unroll
    for (int i = 0; i < 6; ++i) r.d[i] = src[i];
    return r;
}

__device__ __forceinline__ bf16x8 luma_row8_bf16(const RawRow &raw, uint32_t w) {
    const uint32_t d0 = raw.d[0], d1 = raw.d[1], d2 = raw.d[2], d3 = raw.d[3], d4 = raw.d[4], d5 = raw.d[5];
    const uint32_t c0 = w & 0xFFu, c1 = (w >> 8) & 0xFFu, c2 = (w >> 16) & 0xFFu;
    const uint32_t wA = w, wB0 = c0 << 24, wB1 = c1 | (c2 << 8), wC0 = (c0 << 16) | (c1 << 24), wC1 = c2, wD = w << 8;
    const uint32_t kC = 0xFFFF8000u;
    int y[8];
    y[0] = (int)__builtin_amdgcn_udot4(d0, wA, kC, false) >> 8;
    y[1] = (int)__builtin_amdgcn_udot4(d1, wB1, __builtin_amdgcn_udot4(d0, wB0, kC, false), false) >> 8;
    y[2] = (int)__builtin_amdgcn_udot4(d2, wC1, __builtin_amdgcn_udot4(d1, wC0, kC, false), false) >> 8;
    y[3] = (int)__builtin_amdgcn_udot4(d2, wD, kC, false) >> 8;
    y[4] = (int)__builtin_amdgcn_udot4(d3, wA, kC, false) >> 8;
    y[5] = (int)__builtin_amdgcn_udot4(d4, wB1, __builtin_amdgcn_udot4(d3, wB0, kC, false), false) >> 8;
    y[6] = (int)__builtin_amdgcn_udot4(d5, wC1, __builtin_amdgcn_udot4(d4, wC0, kC, false), false) >> 8;
    y[7] = (int)__builtin_amdgcn_udot4(d5, wD, kC, false) >> 8;
    bf16x8 r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = (__bf16)(float)y[j];      // |y| <= 128: exact in bf16
    return r;
}

// In-kernel phase stamps (diagnostic builds only: -DJPEGAMD_STAMPS; the shipped kernel executes none).
#ifdef JPEGAMD_STAMPS
#define STAMP(i)                                                                              \
    do {                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                    \
        unsigned long long t_;                                                                \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); \
        st_sum[i] += t_ - st_last;                                                            \
        st_last = t_;                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                    \
    } while (0)
#else
#define STAMP(i) do { } while (0)
#endif

template <bool kTaps>
__global__ __launch_bounds__(64 * kWavesM) __attribute__((amdgpu_waves_per_eu(JPEGAMD_MFMA_WAVES, JPEGAMD_MFMA_WAVES))) void k_transform_mfma(const ImageDesc im, const TransformOutM out) {
    __shared__ __attribute__((aligned(16))) uint32_t s_afrag[kAFragWords];
    __shared__ float2 s_q[64];                 // (multiplier, threshold) by zigzag position
    __shared__ float s_qstep[64];
    __shared__ uint32_t s_huff[272];
    __shared__ float s_cos[64];
    __shared__ WaveLdsM s_wave[kWavesM];

    {
        const int t = (int)threadIdx.x;
        const uint4 *src = reinterpret_cast<const uint4 *>(out.tables->afrag);
        uint4 *dst = reinterpret_cast<uint4 *>(s_afrag);
        for (int i = t; i < kAFragWords / 4; i += 64 * kWavesM) dst[i] = src[i];
        if (t < 64) {
            s_q[t] = make_float2(out.tables->qmul[t], out.tables->qthr[t]);
            s_qstep[t] = out.tables->qstep[t];
            s_cos[t] = kCosFM[t];
        }
        if (t < 272) s_huff[t] = out.huff[t];
        if (blockIdx.x == 0 && t == 0 && out.reset.stats) out.reset.stats->status = 0u;   // cleared for this call's finalize kernels
    }
    __syncthreads();

    const int lane = lane_id();
    const int wave = (int)(threadIdx.x >> 6);
    const int seg = (int)blockIdx.x * kWavesM + wave;
    if (seg >= im.num_segs) return;            // whole wave; no workgroup-level sync below
    WaveLdsM &wl = s_wave[wave];
    const int h = lane >> 5, b = lane & 31;

    const int by = seg / im.segs_per_row;
    const int sbx0 = (seg - by * im.segs_per_row) * kSegBlocksM;
    const int seg_nblk = min(kSegBlocksM, im.blocks_w - sbx0);
    const int py0 = by * 8;
    const float bias = out.tables->bias;
    const float2 *sq_lane = &s_q[32 * h];

    // DC of the block that precedes the segment in raster block order (rle.c:59-70)
    int prev_dc = 0;
    {
        int pbx = sbx0 - 1, pby = by;
        if (pbx < 0) { pbx = im.blocks_w - 1; pby = by - 1; }
        if (pby >= 0) {
            const int yv = luma_clamped(im, pbx * 8 + (lane & 7), pby * 8 + (lane >> 3)) - 128;
            prev_dc = ref_quantise(__fmul_rn(ref_scale(0, 0), (float)wave_sum_i32(yv)), s_qstep[0]);
        }
    }

#ifdef JPEGAMD_STAMPS
    unsigned long long st_sum[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, st_last;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_last)::"memory");
#endif
    uint32_t *segw = out.seg_words + (size_t)seg * kSegCapWordsM;
    uint32_t carry_bits = 0, wbase = 0, last_word = 0;          // last_word: the most recent COMPLETE word of the bit string
    int nsym = 0, nexact = 0;
    if (lane == 0) { wl.win[0] = 0u; wl.items[0] = 0u; }

    // Pixel rows of the NEXT tile are requested while the current tile is being coded (the raw registers
    // are free once the luma conversion is done), so HBM latency hides behind the entropy phases.
    const auto tile_interior = [&](int tb0) {
        const int nb = min(kTileBlocks, seg_nblk - tb0);
        return im.fast_ok && ((sbx0 + tb0 + nb) * 8 <= im.width) && (py0 + 8 <= im.height);
    };
    const auto request_rows = [&](int tb0, RawRow (&raw)[4]) {
        const int nb = min(kTileBlocks, seg_nblk - tb0);
        const int pbx = sbx0 + tb0 + min(b, nb - 1);
#pragma unroll
        for (int s = 0; s < 4; ++s)
            raw[s] = load_raw_row(reinterpret_cast<const uint32_t *>(row_ptr(im, py0 + 2 * s + h) + 24 * (size_t)pbx));
    };
    RawRow raw[4];
    if (tile_interior(0)) request_rows(0, raw);

#pragma unroll 1
    for (int tb0 = 0; tb0 < seg_nblk; tb0 += kTileBlocks) {
        const int nblk = min(kTileBlocks, seg_nblk - tb0);
        const bool active = b < nblk;
        const int bx = sbx0 + tb0 + min(b, nblk - 1);       // idle columns shadow the last block
        const int px0 = bx * 8;

        STAMP(0);   // tile prologue / previous tile's tail
        // ---- 1. pixels -> B fragments ----------------------------------------------------------
        bf16x8 bfrag[4];
        const bool interior = tile_interior(tb0);
        if (interior) {
#pragma unroll
            for (int s = 0; s < 4; ++s) bfrag[s] = luma_row8_bf16(raw[s], im.weights);
        } else {
            // edge tile (right/bottom replication, converter.c:31,36) or unaligned source: clamped byte gather
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    bfrag[s][j] = (__bf16)(float)(luma_clamped(im, px0 + j, py0 + 2 * s + h) - 128);
        }
        if (tb0 + kTileBlocks < seg_nblk && tile_interior(tb0 + kTileBlocks)) request_rows(tb0 + kTileBlocks, raw);
        if (kTaps && active && out.tap_y) {
            int8_t *ty = out.tap_y + ((size_t)by * im.blocks_w + bx) * 64;
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int j = 0; j < 8; ++j) ty[(2 * s + h) * 8 + j] = (int8_t)(int)(float)bfrag[s][j];
        }

        STAMP(1);   // loads + luma
        // ---- 2. the 64x64 transform on the matrix pipe: small terms first ----------------------
        f32x16 acc[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; }
#pragma unroll 1
        for (int t = 0; t < 3; ++t) {          // rolled: one term's 8 A fragments (32 VGPRs) are fetched together,
            const uint32_t *at = &s_afrag[(t * 2 * 4 * 64 + lane) * 4];     // so the 8 MFMAs issue back to back
            bf16x8 afr[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) afr[i] = *reinterpret_cast<const bf16x8 *>(&at[(i * 64) * 4]);   // i = H * 4 + s
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int H = 0; H < 2; ++H) acc[H] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[H * 4 + s], bfrag[s], acc[H], 0, 0, 0);
        }

        { float ck_ = acc[0][0] + acc[1][15]; asm volatile("" ::"v"(ck_)); }
        STAMP(2);   // MFMA chain
        // ---- 3. quantise with the guard band ---------------------------------------------------
        // Branch-free: a lane's flagged sites are collected in a 32-bit mask (bit 16H + r) so the 32 LDS
        // constant reads can be batched by the compiler; the rare exact-order path runs once afterwards.
        int n[2][16];
        // DC (zigzag 0, lanes h == 0): the sum is an exact integer, so the reference's value is reproducible directly
        const int dc_exact = ref_quantise(__fmul_rn(ref_scale(0, 0), acc[0][0]), s_qstep[0]);
        uint32_t flagbits = 0;
#pragma unroll
        for (int H = 1; H >= 0; --H)
#pragma unroll
            for (int r = 15; r >= 0; --r) {                       // descending: the shift-in below leaves site s at bit s
                const float2 q = sq_lane[16 * H + r];
                const float zc = fmaf(acc[H][r], q.x, bias);      // z + 0.5 + delta
                const float g = __builtin_amdgcn_fractf(zc);
                n[H][r] = (int)floorf(zc);
                flagbits = (flagbits << 1) | ((g <= q.y) ? 1u : 0u);   // within delta of a rounding tie
                if ((r & 7) == 0) __builtin_amdgcn_sched_barrier(0);   // at most 8 constant pairs in flight
            }
        if (h == 0) { n[0][0] = dc_exact; flagbits &= ~1u; }     // DC lanes never need the fallback
        if (!active) flagbits = 0;

        STAMP(3);   // quantise
        // ---- 4. exact-order recomputation of flagged coefficients ------------------------------
        uint64_t exact_mask = 0;
        unsigned long long fm = __ballot(flagbits != 0u);
        if (__builtin_expect(fm != 0ull, 0)) {
            while (fm) {
                const int fl = __ffsll((long long)fm) - 1;
                fm &= fm - 1;
                uint32_t bits = (uint32_t)__builtin_amdgcn_readlane((int)flagbits, fl);
                while (bits) {
                    const int site = __ffs((int)bits) - 1;
                    bits &= bits - 1;
                    const int z = 32 * (fl >> 5) + site;
                    const int k = kZZ[z], u = k >> 3, v = k & 7;
                    const int ebx = sbx0 + tb0 + (fl & 31);
                    const float pix = (float)(luma_clamped(im, ebx * 8 + (lane & 7), py0 + (lane >> 3)) - 128);
                    const float coef = exact_coef_float(pix, u, v, s_cos, lane);
                    const int val = ref_quantise(coef, s_qstep[z]);
                    ++nexact;
                    if (kTaps && lane == fl) exact_mask |= 1ull << k;
#pragma unroll
                    for (int H = 0; H < 2; ++H)
#pragma unroll
                        for (int r = 0; r < 16; ++r) n[H][r] = (site == 16 * H + r && lane == fl) ? val : n[H][r];
                }
            }
        }
        if (kTaps && active) {
            const size_t blk = (size_t)by * im.blocks_w + bx;
            if (out.tap_zz) {
#pragma unroll
                for (int H = 0; H < 2; ++H)
#pragma unroll
                    for (int r = 0; r < 16; ++r) out.tap_zz[blk * 64 + 32 * h + 16 * H + r] = (int16_t)n[H][r];
            }
            if (out.tap_mask) atomicOr((unsigned long long *)&out.tap_mask[blk], (unsigned long long)exact_mask);
        }

        STAMP(4);   // exact fallback
        // ---- 5. per-block symbol counts, DC prediction ------------------------------------------
        int nnz = 0;
#pragma unroll
        for (int H = 0; H < 2; ++H)
#pragma unroll
            for (int r = 0; r < 16; ++r) nnz += (n[H][r] != 0) ? 1 : 0;
        if (h == 0) nnz -= (n[0][0] != 0) ? 1 : 0;                                  // DC is not an AC symbol
        const bool eob = (h == 1) && (n[1][15] == 0);                               // rle.c:121-123 (zigzag 63)
        const uint32_t cnt = active ? (uint32_t)(nnz + (h == 0 ? 1 : (eob ? 1 : 0))) : 0u;
        const uint32_t partner = other_half(cnt, lane);
        const uint32_t tb = cnt + partner;                                          // symbols of block b
        const uint32_t incl = half_incl_scan_dpp(tb);                            // same in both halves
        const uint32_t base_b = incl - tb;
        const uint32_t my_base = base_b + (h ? partner : 0u);
        const uint32_t t_all = (uint32_t)__builtin_amdgcn_readlane((int)incl, 31);

        int pred = lane_shift_up1(n[0][0]);                                        // lanes h == 0: DC of block b - 1
        if (b == 0) pred = prev_dc;
        const int dc_diff = n[0][0] - pred;
        prev_dc = __builtin_amdgcn_readlane(n[0][0], nblk - 1);
        nsym += (int)cnt;

        STAMP(5);   // counts + scans + DC prediction
        // ---- 6./7. items, then one lane per symbol ----------------------------------------------
        uint32_t gbase = 0;
        while (gbase < t_all) {
            const bool fits = (incl - gbase) <= (uint32_t)kItemCapM;
            const bool mine = active && base_b >= gbase && fits;
            const unsigned long long gm = __ballot(mine) & 0xFFFFFFFFull;
            const uint32_t gend = (uint32_t)__builtin_amdgcn_readlane((int)incl, 31 - __builtin_clz((uint32_t)gm));
            const uint32_t gt = gend - gbase;

            if (mine) {
                uint32_t ptr = my_base - gbase + 1u;           // items[0] is the "previous item" of the first one
                const uint32_t zhi = (uint32_t)(32 * h) << 16;
                if (h == 0) wl.items[ptr++] = kItemDcM | (uint32_t)(dc_diff & 0xFFFF);      // rle.c:68-76
#pragma unroll
                for (int H = 0; H < 2; ++H)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int v = (H == 0 && r == 0 && h == 0) ? 0 : n[H][r];
                        if (v != 0) wl.items[ptr++] = (zhi + ((uint32_t)(16 * H + r) << 16)) | (uint32_t)(v & 0xFFFF);
                    }
                if (eob) wl.items[ptr] = 64u << 16;                                          // value 0, not DC = EOB
            }

            STAMP(6);   // scatter
            for (uint32_t b0 = 0; b0 < gt; b0 += 64) {
                const uint32_t idx = b0 + (uint32_t)lane;
                const bool valid = idx < gt;
                const uint32_t itp = wl.items[idx];              // previous item (sentinel 0 in front of the first)
                const uint32_t it = wl.items[idx + 1];           // reads past the list are masked by `valid`
                const int v = (int)(short)(it & 0xFFFFu);
                const bool isdc = (it & kItemDcM) != 0u;
                const int run = v ? (int)((it >> 16) & 0x7Fu) - (int)((itp >> 16) & 0x7Fu) - 1 : 0;   // EOB: symbol 0x00
                const int nb = v ? (32 - __clz(abs(v))) : 0;                                          // rle.c:9-22
                const uint32_t amp = (uint32_t)(v + (v >> 31)) & ((1u << nb) - 1u);                   // rle.c:24-35
                const uint32_t hc = s_huff[isdc ? (256 + nb) : (((run & 15) << 4) | nb)];
                uint32_t hi = ((hc & 0xFFFFu) << nb) | amp;
                uint32_t lo = 0;
                int len = valid ? (int)(hc >> 16) + nb : 0;
                const int zrl = (valid && !isdc) ? (run >> 4) : 0;                                    // rle.c:99-103
                hi <<= (32 - len) & 31;
                if (len == 0) hi = 0;
                const bool any_zrl = __any(zrl != 0);
                if (__builtin_expect(any_zrl, 0)) {
                    const uint32_t zw = s_huff[0xF0];
                    const uint32_t zc = zw & 0xFFFFu;
                    const int zl = (int)(zw >> 16);
                    unsigned long long a64 = ((unsigned long long)hi << 32);
                    int tot = len;
                    for (int q = 0; q < 3; ++q)
                        if (q < zrl) { a64 = (a64 >> zl) | ((unsigned long long)zc << (64 - zl)); tot += zl; }
                    hi = (uint32_t)(a64 >> 32);
                    lo = (uint32_t)a64;
                    len = tot;
                    nsym += zrl;
                }
                const uint32_t incl_b = wave_incl_scan_u32((uint32_t)len);
                const uint32_t batch_bits = (uint32_t)__builtin_amdgcn_readlane((int)incl_b, 63);
                const uint32_t rel = carry_bits + incl_b - (uint32_t)len - wbase * 32u;
                wl.win[1 + lane] = 0u;
                wl.win[65 + (lane < 63 ? lane : 62)] = 0u;
                {   // empty symbols OR zeros into an in-range word: no divergence
                    const uint32_t w = rel >> 5, sh = rel & 31u;
                    atomicOr(&wl.win[w], __builtin_amdgcn_alignbit(0u, hi, sh));
                    atomicOr(&wl.win[w + 1], __builtin_amdgcn_alignbit(hi, lo, sh));
                    if (__builtin_expect(any_zrl, 0)) atomicOr(&wl.win[w + 2], __builtin_amdgcn_alignbit(lo, 0u, sh));
                }
                carry_bits += batch_bits;
                const uint32_t done = (carry_bits >> 5) - wbase;
                const uint32_t out0 = wl.win[lane], out1 = wl.win[64 + lane], part = wl.win[done];   // one LDS round trip
                if ((uint32_t)lane < done) segw[wbase + lane] = out0;
                if ((uint32_t)lane + 64u < done) segw[wbase + 64u + lane] = out1;
                if (done) last_word = done > 64u ? (uint32_t)__builtin_amdgcn_readlane((int)out1, (int)done - 65)
                                                 : (uint32_t)__builtin_amdgcn_readlane((int)out0, (int)done - 1);
                if (lane == 0) wl.win[0] = part;
                wbase += done;
            }
            STAMP(7);   // symbol-parallel batches
            gbase = gend;
        }
    }

    if ((carry_bits & 31u) && lane == 0) segw[wbase] = wl.win[0];
#ifdef JPEGAMD_STAMPS
    STAMP(8);
    if (lane == 0 && out.stamps) for (int i = 0; i < 10; ++i) out.stamps[(size_t)seg * 16 + i] = st_sum[i];
#endif
    const int seg_syms = wave_sum_i32(nsym);
    if (lane == 0) {
        // last 7 bits of the string = what the next segment's first output byte may start with
        const uint32_t p = carry_bits & 31u, w0 = wl.win[0];
        const uint32_t tail = p ? ((last_word << p) | (w0 >> (32u - p))) : last_word;
        if (out.seg_tail) out.seg_tail[seg] = (uint8_t)(tail & 0x7Fu);
        out.seg_bits[seg] = carry_bits;
        out.seg_syms[seg] = (uint32_t)seg_syms;
        out.seg_exact[seg] = (uint32_t)nexact;
    }
}

int launch_transform_mfma(const ImageDesc &im, const TransformOutM &out, bool taps, void *stream) {
    const dim3 grid((im.num_segs + kWavesM - 1) / kWavesM), block(64 * kWavesM);
    if (taps) hipLaunchKernelGGL(k_transform_mfma<true>, grid, block, 0, (hipStream_t)stream, im, out);
    else hipLaunchKernelGGL(k_transform_mfma<false>, grid, block, 0, (hipStream_t)stream, im, out);
    return (int)hipGetLastError();
}

}  // namespace jpegamd
