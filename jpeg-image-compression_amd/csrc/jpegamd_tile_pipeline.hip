// jpegamd_tile_pipeline.hip -- the transform and the entropy coder as TWO kernels.
//
// Measurement that drove the split (profiles/, tools/stamp_profile.py): the fused kernel is bound by the
// dependent latency of ONE wave walking through luma -> MFMA -> quantise -> compaction -> symbol batches
// (a lone wave needs ~30 us for its 128 blocks), and its 128 live registers allow only 4 waves per SIMD
// to overlap that latency.  Here
//
//   k_tile_transform  one wave = one tile of 32 blocks: luma, the 64x64 DCT on the matrix pipe, guard-band
//                     quantisation, exact-order fallback (identical to jpegamd_transform_mfma.hip), then
//                     every lane APPENDS its block's symbols-to-be -- (zigzag position, value) items: DC,
//                     non-zero ACs, EOB -- to the tile's list in HBM (plain stores at addresses from a
//                     DPP prefix sum of per-lane counts).  No LDS list, no bit window, no segment state.
//   k_entropy         one wave = one segment (4 tiles = 128 blocks): streams the lists 64 items at a time,
//                     one lane per SYMBOL (size / amplitude / Huffman code, rle.c:9-35,99-123,
//                     huffman.c:145-188), a wave prefix sum gives bit offsets, bits are OR-ed into an LDS
//                     window and flushed as whole words.  ~30 registers, 8 waves per SIMD.
//
// Extra HBM traffic: 4 bytes per symbol written and read once (~25 MB per 8192^2 photo-like image, vs
// 201 MB of pixels).  Lists are reserved at the worst case (65 items per block) so any content fits.
#include "jpegamd_device.h"

namespace jpegamd {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int kWavesT = 8;                      // 512-thread workgroups share the 24 KiB matrix image in LDS
constexpr uint32_t kItDc = 0x80000000u;         // item is a DC difference
constexpr uint32_t kItFirst = 0x40000000u;      // ... of the first block of its tile: value is the absolute DC

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


#ifndef JPEGAMD_TILE_WAVES
#define JPEGAMD_TILE_WAVES 4
#endif

template <bool kTaps>
__global__ __launch_bounds__(64 * kWavesT) __attribute__((amdgpu_waves_per_eu(JPEGAMD_TILE_WAVES, JPEGAMD_TILE_WAVES)))
void k_tile_transform(const ImageDesc im, const TransformOutM out) {
    __shared__ __attribute__((aligned(16))) uint32_t s_afrag[kAFragWords];
    __shared__ float2 s_q[64];                 // (multiplier, threshold) by zigzag position
    __shared__ float s_qstep[64];
    __shared__ float s_cos[64];

    {
        const int t = (int)threadIdx.x;
        const uint4 *src = reinterpret_cast<const uint4 *>(out.tables->afrag);
        uint4 *dst = reinterpret_cast<uint4 *>(s_afrag);
        for (int i = t; i < kAFragWords / 4; i += 64 * kWavesT) dst[i] = src[i];
        if (t < 64) {
            s_q[t] = make_float2(out.tables->qmul[t], out.tables->qthr[t]);
            s_qstep[t] = out.tables->qstep[t];
            s_cos[t] = kCosFM[t];
        }
        if (blockIdx.x == 0 && t == 0 && out.reset.stats) out.reset.stats->status = 0u;   // cleared for this call's finalize kernels
    }
    __syncthreads();

    const int lane = lane_id();
    const int wave = (int)(threadIdx.x >> 6);
    const int h = lane >> 5, b = lane & 31;
    const float bias = out.tables->bias;
    const float2 *sq_lane = &s_q[32 * h];

    // Persistent waves: tile = first, first + stride, ...  The matrix image is loaded once per workgroup and the
    // NEXT tile's pixel rows are requested as soon as the current ones are converted, so their HBM latency
    // hides behind the MFMA / quantise / append phases of the current tile.
    const int stride = (int)gridDim.x * kWavesT;
    const int first = (int)blockIdx.x * kWavesT + wave;
    struct TileGeo { int by, tbx0, nblk, bx; bool interior; };
    const auto geo = [&](int tile) {
        TileGeo g;
        g.by = tile / im.tiles_per_row;
        g.tbx0 = (tile - g.by * im.tiles_per_row) * kTileBlocks;
        g.nblk = min(kTileBlocks, im.blocks_w - g.tbx0);
        g.bx = g.tbx0 + min(b, g.nblk - 1);                     // idle columns shadow the last block
        g.interior = im.fast_ok && ((g.tbx0 + g.nblk) * 8 <= im.width) && (g.by * 8 + 8 <= im.height);
        return g;
    };
    const auto request_rows = [&](const TileGeo &g, RawRow (&raw)[4]) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
            raw[s] = load_raw_row(reinterpret_cast<const uint32_t *>(row_ptr(im, g.by * 8 + 2 * s + h) + 24 * (size_t)g.bx));
    };
    RawRow raw[4];
    if (first < im.num_tiles) { const TileGeo g0 = geo(first); if (g0.interior) request_rows(g0, raw); }

#pragma unroll 1
    for (int tile = first; tile < im.num_tiles; tile += stride) {
        const TileGeo tg = geo(tile);
        const int by = tg.by, tbx0 = tg.tbx0, nblk = tg.nblk, bx = tg.bx;
        const int py0 = by * 8, px0 = bx * 8;
        const bool active = b < nblk, interior = tg.interior;
        int nexact = 0;
        // ---- 1. pixels -> B fragments ----------------------------------------------------------
        bf16x8 bfrag[4];
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
        if (tile + stride < im.num_tiles) { const TileGeo gn = geo(tile + stride); if (gn.interior) request_rows(gn, raw); }
        if (kTaps && active && out.tap_y) {
            int8_t *ty = out.tap_y + ((size_t)by * im.blocks_w + bx) * 64;
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int j = 0; j < 8; ++j) ty[(2 * s + h) * 8 + j] = (int8_t)(int)(float)bfrag[s][j];
        }

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
            __builtin_amdgcn_sched_barrier(0);     // keep the 8 LDS reads ahead of the MFMAs (hipcc otherwise sinks each read next to its use)
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int H = 0; H < 2; ++H) acc[H] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[H * 4 + s], bfrag[s], acc[H], 0, 0, 0);
        }

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
                    const int ebx = tbx0 + (fl & 31);
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


        // ---- 5. per-lane symbol counts -> list positions (DPP prefix sums, both halves agree per block) ----
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
        const uint32_t incl = half_incl_scan_dpp(tb);
        const uint32_t my_base = incl - tb + (h ? partner : 0u);
        const uint32_t t_all = (uint32_t)__builtin_amdgcn_readlane((int)incl, 31);

        // DC prediction inside the tile (rle.c:59-70); the first block keeps its absolute value, the entropy
        // kernel subtracts the previous tile's last DC.
        const int pred = lane_shift_up1(n[0][0]);
        const uint32_t dc_item = (b == 0) ? (kItDc | kItFirst | (uint32_t)(n[0][0] & 0xFFFF))
                                          : (kItDc | (uint32_t)((n[0][0] - pred) & 0xFFFF));

        // ---- 6. append the items: slot 0 of the list is a zero sentinel ("previous item" of the first) ----
        uint32_t *list = out.tile_items + (size_t)tile * kTileItemCap;
        if (active) {
            uint32_t *p = list + 1 + my_base;
            const uint32_t zhi = (uint32_t)(32 * h) << 16;
            if (h == 0) *p++ = dc_item;
#pragma unroll
            for (int H = 0; H < 2; ++H)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int v = (H == 0 && r == 0 && h == 0) ? 0 : n[H][r];
                    if (v != 0) *p++ = (zhi + ((uint32_t)(16 * H + r) << 16)) | (uint32_t)(v & 0xFFFF);
                }
            if (eob) *p = 64u << 16;                                                // value 0, not DC = EOB
        }
        if (lane == 0) {
            list[0] = 0u;
            out.tile_count[tile] = t_all;
            out.tile_lastdc[tile] = __builtin_amdgcn_readlane(n[0][0], nblk - 1);
            out.tile_exact[tile] = (uint32_t)nexact;
        }
    }
}

int launch_tile_transform(const ImageDesc &im, const TransformOutM &out, bool taps, void *stream) {
    // persistent: at most 2 workgroups per CU (16 waves/CU at 4 waves/SIMD), fewer for small images
    const int wgs = (im.num_tiles + kWavesT - 1) / kWavesT;
    const dim3 grid(wgs < 512 ? wgs : 512), block(64 * kWavesT);
    if (taps) hipLaunchKernelGGL(k_tile_transform<true>, grid, block, 0, (hipStream_t)stream, im, out);
    else hipLaunchKernelGGL(k_tile_transform<false>, grid, block, 0, (hipStream_t)stream, im, out);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------
// k_entropy: one wave per segment, one lane per symbol
// ------------------------------------------------------------------------------------
constexpr int kWavesE = 4;
constexpr int kSegBufWords = 512;               // LDS bit buffer per wave (typical segment: ~90 words); flushed when nearly full

__global__ __launch_bounds__(64 * kWavesE) void k_entropy(const EntropyArgs a) {
    __shared__ uint32_t s_huff[288];                // 272 used; masked garbage items may index a little past it
    __shared__ uint32_t s_win[kWavesE][kSegBufWords];
    {
        const int t = (int)threadIdx.x;
        s_huff[t] = a.huff[t];
        if (t < 32) s_huff[256 + t] = t < 16 ? a.huff[256 + t] : 0u;
    }
    __syncthreads();
    const int lane = lane_id(), wave = (int)(threadIdx.x >> 6);
    const int seg = (int)blockIdx.x * kWavesE + wave;
    if (seg >= a.num_segs) return;
    uint32_t *win = s_win[wave];

    const int by = seg / a.segs_per_row;
    const int tx0 = (seg - by * a.segs_per_row) * kSegTiles;
    const int ntiles = min(kSegTiles, a.tiles_per_row - tx0);
    const int tile0 = by * a.tiles_per_row + tx0;

    uint32_t *segw = a.seg_words + (size_t)seg * kSegCapWordsM;
    uint32_t carry_bits = 0, wbase = 0, last_word = 0;
    int nsym = 0;
#pragma unroll
    for (int i = 0; i < kSegBufWords / 64; ++i) win[i * 64 + lane] = 0u;
    // everything whose address is known up front is requested now: counts and predecessor DCs of the segment's tiles
    const int tcount = lane < ntiles ? (int)a.tile_count[tile0 + lane] : 0;
    const int tprev = (lane < ntiles && tile0 + lane > 0) ? a.tile_lastdc[tile0 + lane - 1] : 0;
    uint32_t nexact = lane < ntiles ? a.tile_exact[tile0 + lane] : 0u;

    // Flat walk over (tile, batch of 64 items) with the NEXT batch's two loads already in flight.
    int ti = 0;
    uint32_t b0 = 0;
    uint32_t gt = (uint32_t)__builtin_amdgcn_readlane(tcount, 0);
    const uint32_t *items = a.tile_items + (size_t)tile0 * kTileItemCap;
    uint32_t nx_itp = items[lane], nx_it = items[lane + 1];      // slot 0 = sentinel ("previous item" of the first)
#pragma unroll 1
    while (ti < ntiles) {
        const uint32_t itp = nx_itp, it = nx_it;
        const uint32_t cur_gt = gt, cur_b0 = b0;
        const int prev_dc = __builtin_amdgcn_readlane(tprev, ti);
        // advance and request
        b0 += 64;
        if (b0 >= gt) {
            nsym += (lane == 0) ? (int)gt : 0;
            ++ti;
            b0 = 0;
            if (ti < ntiles) { gt = (uint32_t)__builtin_amdgcn_readlane(tcount, ti); items += kTileItemCap; }
        }
        if (ti < ntiles) { nx_itp = items[b0 + lane]; nx_it = items[b0 + lane + 1]; }   // past-the-list reads stay inside the reservation
        {
            const uint32_t idx = cur_b0 + (uint32_t)lane;
            const bool valid = idx < cur_gt;
            int v = (int)(short)(it & 0xFFFFu);
            const bool isdc = (it & kItDc) != 0u;
            if (it & kItFirst) v -= prev_dc;                 // first block of a tile: DC difference against the previous tile
            const int run = (v && !isdc) ? (int)((it >> 16) & 0x7Fu) - (int)((itp >> 16) & 0x7Fu) - 1 : 0;   // EOB: symbol 0x00
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
            // The loop body holds NO global store: waiting for the next batch's loads (vmcnt) would otherwise
            // also wait for every younger store to be acknowledged.  Stores happen only in the rare flush.
            if (__builtin_expect((carry_bits >> 5) - wbase + 124u > (uint32_t)kSegBufWords, 0)) {
                const uint32_t done = (carry_bits >> 5) - wbase;            // complete words in the buffer
                uint32_t part = win[done];
                for (uint32_t j = (uint32_t)lane; j < done; j += 64) segw[wbase + j] = win[j];
                last_word = win[done - 1];
#pragma unroll
                for (int i = 0; i < kSegBufWords / 64; ++i) win[i * 64 + lane] = 0u;
                if (lane == 0) win[0] = part;
                wbase += done;
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // leave no store pending behind the branch
            }
            const uint32_t rel = carry_bits + incl_b - (uint32_t)len - wbase * 32u;
            {   // empty symbols OR zeros into an in-range word: no divergence
                const uint32_t w = rel >> 5, sh = rel & 31u;
                atomicOr(&win[w], __builtin_amdgcn_alignbit(0u, hi, sh));
                atomicOr(&win[w + 1], __builtin_amdgcn_alignbit(hi, lo, sh));
                if (__builtin_expect(any_zrl, 0)) atomicOr(&win[w + 2], __builtin_amdgcn_alignbit(lo, 0u, sh));
            }
            carry_bits += batch_bits;
        }
    }
    {   // final flush: complete words, then the zero-padded partial word
        const uint32_t done = (carry_bits >> 5) - wbase;
        for (uint32_t j = (uint32_t)lane; j < done; j += 64) segw[wbase + j] = win[j];
        if (done) last_word = win[done - 1];
        const uint32_t part = win[done];
        wbase += done;
        if ((carry_bits & 31u) && lane == 0) segw[wbase] = part;
        if (lane == 0) win[0] = part;
    }
    const int seg_syms = wave_sum_i32(nsym);
    const int seg_exact = wave_sum_i32((int)nexact);
    if (lane == 0) {
        const uint32_t p = carry_bits & 31u, w0 = win[0];
        const uint32_t tail = p ? ((last_word << p) | (w0 >> (32u - p))) : last_word;
        a.seg_tail[seg] = (uint8_t)(tail & 0x7Fu);           // what the next segment's first output byte may start with
        a.seg_bits[seg] = carry_bits;
        a.seg_syms[seg] = (uint32_t)seg_syms;
        a.seg_exact[seg] = (uint32_t)seg_exact;
    }
}

int launch_entropy(const EntropyArgs &a, void *stream) {
    hipLaunchKernelGGL(k_entropy, dim3((a.num_segs + kWavesE - 1) / kWavesE), dim3(64 * kWavesE), 0, (hipStream_t)stream, a);
    return (int)hipGetLastError();
}

}  // namespace jpegamd
