// jpegamd_entropy.hip -- k_entropy: per-tile symbol lists -> per-segment bit strings (rle.c:51-127, huffman.c:121-193).
//
// One wave codes one segment (8 tiles of one block row, <= 256 blocks).  The items of the segment's tiles form ONE
// stream (every list holds an even number of items, so a lane's two items never come from two lists); the wave walks
// it 128 items at a time, TWO symbols per lane:
//   * size / amplitude / run / Huffman code per symbol (rle.c:9-35,83-123; huffman.c:145-188), the run of an AC symbol
//     being the gap to the item in front of it (same lane, or the neighbour lane's second item);
//   * the two codes of a lane are joined into one left-aligned string of <= 54 bits, ONE wave prefix sum over the
//     lanes' bit counts places it, and it is OR-ed into an LDS bit window (a word is shared by ~2.5 lanes instead of ~5
//     symbols: round 1's same-word ds_or serialisation, 14 conflict cycles per LDS instruction, kept the LDS pipe busy
//     for 22 of the kernel's 24 us -- profiles/r01_pmc_sq.txt);
//   * tile boundaries, the DC predictor of a tile's first block, list padding and the stream's end are handled on the
//     scalar unit (they are wave-uniform) and touch single lanes through EXEC;
//   * ZRL symbols (runs >= 16, rle.c:99-103) are rare: a batch that has one takes a slower path, symbol by symbol.
// When the window is written out (once, for ordinary segments) the wave also counts, for each of the 8 byte phases the
// segment's first bit may end up at, the 0xFF bytes that lie wholly inside the segment: the finalize kernel then knows
// every stuffing offset from per-segment numbers alone and the separate counting kernel of round 1 is gone.
#include <hip/hip_ext.h>
#include "jpegamd_device.h"

namespace jpegamd {

constexpr int kWavesE = 4;
constexpr int kSegBufWords = 512;               // LDS bit window per wave (typical segment: ~180 words); flushed when nearly full
constexpr int kBatchItems = 128;


// lanes >= first (first in 0..64) of a full wave
__device__ __forceinline__ unsigned long long lanes_from(int first) {
    return first >= 64 ? 0ull : (~0ull << first);
}

// x += add in the lanes of `mask` (wave-uniform mask and addend: one VALU instruction under a scalar-set EXEC).
// Only used where all 64 lanes are active.
__device__ __forceinline__ void add_in_lanes(uint32_t &x, unsigned long long mask, uint32_t add) {
    asm volatile("s_mov_b64 exec, %1\n\tv_add_u32 %0, %2, %0\n\ts_mov_b64 exec, -1" : "+v"(x) : "s"(mask), "s"(add));
}
__device__ __forceinline__ void set_in_lanes(uint32_t &x, unsigned long long mask, uint32_t value) {
    asm volatile("s_mov_b64 exec, %1\n\tv_mov_b32 %0, %2\n\ts_mov_b64 exec, -1" : "+v"(x) : "s"(mask), "s"(value));
}

// v_ffbh_i32: position of the first bit that differs from the sign bit, counted from the top
__device__ __forceinline__ int leading_sign_bits(int x) {
    int r;
    asm("v_ffbh_i32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}

struct Sym {            // one coded symbol: left-aligned code + amplitude bits, their count, ZRLs in front of it
    uint32_t bits;      // left-aligned in 32 bits
    uint32_t len;       // <= 27
    uint32_t zrl;       // 0..3
};

// Code table in LDS, by symbol (AC: run << 4 | size; DC: 256 + size): bits 31..5 = the Huffman code LEFT-ALIGNED,
// bits 4..0 = code length + size, i.e. the length of code + amplitude bits.  Symbols without a code hold 0.
__device__ __forceinline__ uint32_t table_entry(uint32_t word /*(len << 16) | code, quant_consts.cpp*/, uint32_t size) {
    const uint32_t clen = word >> 16, code = word & 0xFFFFu;
    return clen ? ((code << (27u - clen)) << 5) | (clen + size) : 0u;
}

// item -> symbol.  `prev` is the item in front of it in the stream (only its position field is used, and only for a
// non-zero AC item, whose predecessor is always an item of the same block).
__device__ __forceinline__ Sym code_item(uint32_t it, int v /*its value: (int16) it, minus the DC predictor for a tile's first item*/,
                                         uint32_t prev, const uint32_t *s_tab) {
    const bool isdc = (int)it < 0;
    const int w = v + (v >> 31);                                              // rle.c:24-35: v, or v - 1 when negative
    const int nb = 31 - leading_sign_bits((w << 1) | 1);                      // rle.c:9-22 without the abs / zero special cases
    const uint32_t amp = __builtin_amdgcn_ubfe((uint32_t)w, 0u, (uint32_t)nb);
    const int gap = (int)((it >> 16) & 0x7Fu) - (int)((prev >> 16) & 0x7Fu) - 1;
    const int run = isdc ? 0 : gap;                                           // a non-DC item is a non-zero AC coefficient (EOB: kItEobValue)
    const uint32_t sym = (isdc ? 256u : (uint32_t)((run & 15) << 4)) | (uint32_t)nb;
    const uint32_t e = s_tab[sym];
    Sym s;
    s.len = e & 31u;
    s.bits = (e & ~31u) | (amp << ((32u - s.len) & 31u));                     // amplitude right behind the code (huffman.c:145-153,176-186)
    s.zrl = (uint32_t)run >> 4;                                               // rle.c:99-103
    return s;
}

// OR a left-aligned string (hi:lo, <= 64 bits) into the window at bit `rel`.
__device__ __forceinline__ void window_or(uint32_t *win, uint32_t rel, uint32_t hi, uint32_t lo, bool third) {
    const uint32_t w = rel >> 5, sh = rel & 31u;
    atomicOr(&win[w], __builtin_amdgcn_alignbit(0u, hi, sh));
    atomicOr(&win[w + 1], __builtin_amdgcn_alignbit(hi, lo, sh));
    if (third) atomicOr(&win[w + 2], __builtin_amdgcn_alignbit(lo, 0u, sh));
}

// 0xFF bytes wholly inside the bit string, by byte phase: for the word `cur` followed by `nxt` (MSB-first), bit (31 - o) of
// the result is set when the 8 stream bits from offset o of `cur` are all ones.
__device__ __forceinline__ uint32_t ones8_starts(uint32_t cur, uint32_t nxt) {
    uint32_t hi = cur & __builtin_amdgcn_alignbit(cur, nxt, 31u), lo = nxt & (nxt << 1);
    hi &= __builtin_amdgcn_alignbit(hi, lo, 30u); lo &= lo << 2;
    hi &= __builtin_amdgcn_alignbit(hi, lo, 28u);
    return hi;
}

__global__ __launch_bounds__(64 * kWavesE) void k_entropy(const EntropyArgs a) {
    __shared__ uint32_t s_huff[288];                // [0,256) AC, [256,272) DC sizes 0..15 (12: padding item, no code; 13: EOB)
    __shared__ uint32_t s_win[kWavesE][kSegBufWords + 8];
    __shared__ uint32_t s_gmeta[kWavesE][12];       // {bits, edge, ff[8]} of the workgroup's segments, for the group aggregate
    const int lane = lane_id(), wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int seg = a.seg_begin + (int)blockIdx.x * kWavesE + wave;
    // A wave without a segment (the last workgroup) walks the same code with an empty stream -- zero tiles, zero batches, its
    // loads fall to the range checks, its stores are guarded by `have` -- because the workgroup meets at two barriers.
    const bool have = seg < a.seg_end;
    const int segc = have ? seg : a.seg_end - 1;
    // Every load whose address does not depend on data goes out FIRST: the code-table words and the tiles' records.  (The table
    // used to be built, behind its own load and a barrier, before the records were even requested: three serial round trips
    // -- table, records, first items -- in front of the first symbol; now two.)
    const int tcode = (int)threadIdx.x;
    const uint32_t hword = a.huff[tcode];
    const uint32_t hword2 = tcode == 13 ? a.huff[0] /*EOB, symbol 0x00*/ : tcode < 16 ? a.huff[256 + tcode] : 0u;
    uint32_t *win = s_win[wave];

    const int image = a.tiles_per_image ? segc / a.num_segs : 0;    // a batch: every image has num_segs segments and tiles_per_image tiles
    const int sl = segc - image * a.num_segs;
    const int by = sl / a.segs_per_row;
    const int tx0 = (sl - by * a.segs_per_row) * kSegTiles;
    const int ntiles = have ? min(kSegTiles, a.tiles_per_row - tx0) : 0;
    const int tile_in_image = by * a.tiles_per_row + tx0;
    const int tile0 = image * a.tiles_per_image + tile_in_image;

#pragma unroll
    for (int i = 0; i < kSegBufWords / 64; ++i) win[i * 64 + lane] = 0u;
    if (lane < 8) win[kSegBufWords + lane] = 0u;

    // The segment's tiles: lane t < ntiles holds tile t's numbers.
    const uint32_t *trec = a.tile_items + (size_t)(tile0 + lane) * kTileItemCap + kTileRecord;      // {items, last DC, exact count, 0}
    const uint4 rec = lane < ntiles ? *reinterpret_cast<const uint4 *>(trec) : make_uint4(0u, 0u, 0u, 0u);
    const uint32_t tprev = (lane < ntiles && tile_in_image + lane > 0) ? trec[1 - kTileItemCap] : 0u;   // the tile before (of the same image): its last DC (rle.c:59-70)
    const uint32_t pcnt = (rec.x + 1u) & ~1u;                                                      // list length incl. the padding item
    const uint32_t pincl = wave_incl_scan_u32(pcnt);
    const uint32_t pstart = pincl - pcnt;                                                           // stream index of the tile's first item
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)pincl, 63);                     // items in the stream (even)
    const uint32_t kdelta = ((uint32_t)lane * (uint32_t)kTileItemCap - pstart) * 4u;                // byte address of stream item g of tile t: 4 g + kdelta_t
    const int seg_syms_items = wave_sum_i32((int)rec.x);
    const int seg_exact = wave_sum_i32((int)rec.z);

    const __amdgpu_buffer_rsrc_t irsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint32_t *>(a.tile_items + (size_t)tile0 * kTileItemCap), 0, ntiles * kTileItemCap * 4, 0x00020000);
    const uint32_t lane8 = (uint32_t)lane * 8u;

    // Plans of the batches.  Everything about a batch of 128 stream items that is wave-uniform -- the tile its window starts
    // in, the lanes at which the (at most two) later tiles begin, the address steps there, the DC predictors of the tiles
    // that start inside it -- is computed for 64 batches at once, batch b of the chunk in lane b, and fetched with six
    // v_readlane per batch.  (Only a row's last tile can hold fewer than 64 items, so a window never meets a third start.)
    uint32_t f_soff, f_dk1, f_dk2, f_lanes, f_d01, f_d2;
    uint32_t tstart[kSegTiles];                                   // stream index of every tile's first item (total beyond the last)
#pragma unroll
    for (int t = 0; t < kSegTiles; ++t) tstart[t] = (uint32_t)__builtin_amdgcn_readlane((int)pstart, t);
    const auto make_plans = [&](uint32_t first_batch) {
        const uint32_t g0 = (first_batch + (uint32_t)lane) * (uint32_t)kBatchItems;
        uint32_t t0 = 0;
#pragma unroll
        for (int t = 1; t < kSegTiles; ++t) t0 += (g0 >= tstart[t] && t < ntiles) ? 1u : 0u;
        const auto of_tile = [&](uint32_t v, uint32_t t) { return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(t * 4u), (int)v); };
        const uint32_t s0 = of_tile(pstart, t0), b1 = of_tile(pstart, t0 + 1u), b2 = of_tile(pstart, t0 + 2u);   // lanes >= ntiles hold `total`
        const uint32_t k0 = of_tile(kdelta, t0), k1 = of_tile(kdelta, t0 + 1u), k2 = of_tile(kdelta, t0 + 2u);
        const uint32_t end = g0 + (uint32_t)kBatchItems;
        const uint32_t l1 = (b1 < end && b1 < total) ? (b1 - g0) >> 1 : 64u;
        const uint32_t l2 = (b2 < end && b2 < total) ? (b2 - g0) >> 1 : 64u;
        const uint32_t left = total - g0;                         // only batches with g0 < total are ever fetched
        const uint32_t nvalid = left >= (uint32_t)kBatchItems ? 64u : left >> 1;
        f_soff = g0 * 4u + k0;
        f_dk1 = k1 - k0;
        f_dk2 = k2 - k1;
        f_lanes = l1 | (l2 << 8) | (nvalid << 16) | ((s0 == g0 ? 1u : 0u) << 24);
        f_d01 = (of_tile(tprev, t0) & 0xFFFFu) | (of_tile(tprev, t0 + 1u) << 16);
        f_d2 = of_tile(tprev, t0 + 2u);
    };
    struct Plan {
        uint32_t soff, dk1, dk2;    // scalar byte offset of the window's first item; address steps at the later tiles' first lanes
        int l1, l2, nvalid, f0;     // first lane of the 2nd / 3rd tile in the window (64: none); lanes holding items; lane 0 starts a tile
        int d0, d1, d2;             // DC predictors of the tiles starting at lane 0 / l1 / l2 (rle.c:59-70)
    };
    const auto fetch_plan = [&](uint32_t b /*batch index inside the chunk*/) {
        Plan p;
        p.soff = (uint32_t)__builtin_amdgcn_readlane((int)f_soff, (int)b);
        p.dk1 = (uint32_t)__builtin_amdgcn_readlane((int)f_dk1, (int)b);
        p.dk2 = (uint32_t)__builtin_amdgcn_readlane((int)f_dk2, (int)b);
        const uint32_t ln = (uint32_t)__builtin_amdgcn_readlane((int)f_lanes, (int)b);
        const uint32_t d01 = (uint32_t)__builtin_amdgcn_readlane((int)f_d01, (int)b);
        p.l1 = (int)(ln & 0xFFu); p.l2 = (int)((ln >> 8) & 0xFFu); p.nvalid = (int)((ln >> 16) & 0xFFu); p.f0 = (int)(ln >> 24);
        p.d0 = (int)(short)(d01 & 0xFFFFu); p.d1 = (int)(short)(d01 >> 16);
        p.d2 = (int)(short)((uint32_t)__builtin_amdgcn_readlane((int)f_d2, (int)b) & 0xFFFFu);
        return p;
    };
    const auto request = [&](const Plan &p) {
        uint32_t voff = lane8;
        if (p.l1 < 64) {
            add_in_lanes(voff, lanes_from(p.l1), p.dk1);
            if (p.l2 < 64) add_in_lanes(voff, lanes_from(p.l2), p.dk2);
        }
        return __builtin_amdgcn_raw_buffer_load_b64(irsrc, (int)voff, (int)p.soff, 0);
    };

    uint32_t carry_bits = 0, wbase = 0, last_word = 0, first_word = 0, nzrl = 0, seg_edge = 0;
    bool flushed = false, any_ff = false;
    uint32_t ffc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    // 0xFF census of the words [0, n) of the window; every word's successor is final (n == complete words and the rest is
    // counted later, or the stream has ended and the window is zero behind it).  `last_word` (the word in front of
    // win[0], when the window was written out before) still needs its windows that reach into win[0].
    const auto census = [&](uint32_t n, bool with_carried) {
        const uint32_t c = with_carried ? 1u : 0u;                        // index 0 is then the carried word, i - 1 the window word
        for (uint32_t i0 = 0; i0 < n + c; i0 += 64) {
            const uint32_t i = i0 + (uint32_t)lane;
            uint32_t cw = 0, nw = 0;
            if (i < n + c) {
                const uint32_t j = i - c;                                 // window index of the word (0xFFFFFFFF: the carried one)
                cw = (c && i == 0u) ? last_word : win[j];
                nw = win[j + 1u];
            }
            const uint32_t m = ones8_starts(cw, nw);
            if (__ballot(m != 0u) != 0ull) {
                any_ff = true;
#pragma unroll
                for (int p = 0; p < 8; ++p) ffc[p] += (uint32_t)__popc(m & (0x80808080u >> ((8 - p) & 7)));
            }
        }
    };

    uint32_t *const segw = a.seg.words + (size_t)seg * kSegCapWords;
    const uint32_t nbatches = (total + (uint32_t)kBatchItems - 1u) / (uint32_t)kBatchItems;
    make_plans(0u);
    Plan cur = fetch_plan(0u);
    auto nx = request(cur);
    // the code table (its words were requested at the top), then the workgroup's first barrier, behind the first items' request
    s_huff[tcode] = table_entry(hword, (uint32_t)tcode & 15u);
    if (tcode < 32) s_huff[256 + tcode] = tcode == 13 ? table_entry(hword2, 0u) : tcode < 16 ? table_entry(hword2, (uint32_t)tcode) : 0u;
    __syncthreads();
    uint32_t prev_b = 0;                                          // second item of the lane before lane 0: the previous batch's last item
#pragma unroll 1
    for (uint32_t batch = 0; batch < nbatches; ++batch) {
        const auto items = nx;
        const Plan pl = cur;
        if (batch + 1u < nbatches) {                              // next batch's loads in flight while this one is coded
            if (((batch + 1u) & 63u) == 0u) make_plans(batch + 1u);
            cur = fetch_plan((batch + 1u) & 63u);
            nx = request(cur);
        }
        uint32_t ia = (uint32_t)items[0], ib = (uint32_t)items[1];
        // single-lane fix-ups, all decided on the scalar unit
        if (pl.nvalid < 64) { set_in_lanes(ia, lanes_from(pl.nvalid), kItPadValue); set_in_lanes(ib, lanes_from(pl.nvalid), kItPadValue); }
        uint32_t va = (uint32_t)(int)(short)(ia & 0xFFFFu);
        const int vb = (int)(short)(ib & 0xFFFFu);
        if (pl.f0) add_in_lanes(va, 1ull, (uint32_t)-pl.d0);          // first block of a tile: DC difference against the tile before (rle.c:68-70)
        if (pl.l1 < 64) {
            add_in_lanes(va, 1ull << pl.l1, (uint32_t)-pl.d1);
            if (pl.l2 < 64) add_in_lanes(va, 1ull << pl.l2, (uint32_t)-pl.d2);
        }
        const uint32_t pb = (uint32_t)lane_shift_up1((int)ib);
        const uint32_t prev_a = lane == 0 ? prev_b : pb;
        prev_b = (uint32_t)__builtin_amdgcn_readlane((int)ib, 63);
        const Sym sa = code_item(ia, (int)va, prev_a, s_huff), sb = code_item(ib, vb, ia, s_huff);

        const bool any_zrl = __any((sa.zrl | sb.zrl) != 0u);
        uint32_t la = sa.len, lb = sb.len;
        uint32_t zl = 0, zc = 0;
        if (__builtin_expect(any_zrl, 0)) {
            const uint32_t zw = s_huff[0xF0];
            zc = zw & ~31u;                                       // left-aligned
            zl = zw & 31u;
            la += sa.zrl * zl;
            lb += sb.zrl * zl;
            nzrl += sa.zrl + sb.zrl;
        }
        const uint32_t lab = la + lb;
        const uint32_t incl_b = wave_incl_scan_u32(lab);
        const uint32_t batch_bits = (uint32_t)__builtin_amdgcn_readlane((int)incl_b, 63);
        // The loop body holds NO global store (waiting for the next batch's loads would otherwise also wait for every
        // younger store to be acknowledged): the window is written out only when the next batch might not fit.
        if (__builtin_expect(((carry_bits + batch_bits) >> 5) - wbase + 3u > (uint32_t)kSegBufWords, 0)) {
            const uint32_t done = (carry_bits >> 5) - wbase;            // complete words in the window
            if (done) {
                census(done - 1u, flushed);                              // the last complete word waits for its successor
                const uint32_t part = win[done];
                for (uint32_t j = (uint32_t)lane; j < done; j += 64) segw[wbase + j] = win[j];
                if (!flushed) first_word = win[0];
                last_word = win[done - 1];
#pragma unroll
                for (int i = 0; i < kSegBufWords / 64; ++i) win[i * 64 + lane] = 0u;
                if (lane == 0) win[0] = part;
                wbase += done;
                flushed = true;
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // leave no store pending behind the branch
            }
        }
        const uint32_t rel = carry_bits + incl_b - lab - wbase * 32u;
        if (__builtin_expect(!any_zrl, 1)) {
            // join the lane's two strings: (bits_a : bits_b >> len_a), <= 54 bits
            const uint32_t hi = sa.bits | __builtin_amdgcn_alignbit(0u, sb.bits, la);
            const uint32_t lo = __builtin_amdgcn_alignbit(sb.bits, 0u, la);
            window_or(win, rel, hi, lo, __any((rel & 31u) + lab > 64u));
        } else {
            // symbol by symbol, each with its ZRLs in front (huffman.c:158-188 codes them as ordinary symbols)
            const auto with_zrl = [&](const Sym &s, uint32_t &hi, uint32_t &lo) {
                unsigned long long a64 = (unsigned long long)s.bits << 32;
                for (uint32_t q = 0; q < 3; ++q)
                    if (q < s.zrl) a64 = (a64 >> zl) | ((unsigned long long)zc << 32);
                hi = (uint32_t)(a64 >> 32);
                lo = (uint32_t)a64;
            };
            uint32_t hi, lo;
            with_zrl(sa, hi, lo);
            window_or(win, rel, hi, lo, true);
            with_zrl(sb, hi, lo);
            window_or(win, rel + la, hi, lo, true);
        }
        carry_bits += batch_bits;
    }
    {   // the stream has ended: census of everything still in the window, then write it out
        const uint32_t done = (carry_bits >> 5) - wbase;
        const uint32_t nw = done + ((carry_bits & 31u) ? 1u : 0u);           // words holding bits; the window is zero behind them
        census(nw, flushed);
        for (uint32_t j = (uint32_t)lane; j < nw; j += 64) segw[wbase + j] = win[j];     // (nw == 0 without a segment)
        if (!flushed) first_word = win[0];
        if (done) last_word = win[done - 1];
        const uint32_t part = win[done];
        {
            const uint32_t p = carry_bits & 31u;
            const uint32_t tail = p ? ((last_word << p) | (part >> (32u - p))) : last_word;
            seg_edge = ((first_word >> 24) << 8) | (tail & 0x7Fu);           // first 8 bits | last 7 bits of the segment's string
        }
        if (lane == 0 && have) {
            a.seg.edge[seg] = seg_edge;
            a.seg.bits[seg] = carry_bits;
            a.seg.syms[seg] = (uint32_t)seg_syms_items;                      // + ZRLs below
            a.seg.exact[seg] = (uint32_t)seg_exact;
        }
    }
    const uint32_t zsum = (uint32_t)wave_sum_i32((int)nzrl);
    if (zsum && lane == 0 && have) a.seg.syms[seg] = (uint32_t)seg_syms_items + zsum;
    uint32_t mine = 0;                                                       // lane p < 8 stores the count of phase p
    if (any_ff) {
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const uint32_t t = (uint32_t)wave_sum_i32((int)ffc[p]);
            if (lane == p) mine = t;
        }
    }
    if (lane < 8) {
        if (have) a.seg.ffin[(size_t)seg * 8 + lane] = (uint16_t)min(mine, 65535u);
        s_gmeta[wave][2 + lane] = min(mine, 65535u);
    }
    if (lane == 0) { s_gmeta[wave][0] = carry_bits; s_gmeta[wave][1] = have ? seg_edge : 0u; }   // (no segment: no bits, no ones at either end)
    // Group aggregate (SegArrays::grp_bits / grp_ff): lane i * 8 + p of wave 0 takes segment i of the group at group phase p.
    __syncthreads();
    static_assert(kWavesE == kSegGroup, "one workgroup of k_entropy = one segment group");
    if (wave == 0) {
        const int i = (lane >> 3) & 3, p = lane & 7;
        uint32_t pre = 0;                           // bits of the group in front of segment i
#pragma unroll
        for (int j = 0; j < kWavesE - 1; ++j) pre += (j < i) ? s_gmeta[j][0] : 0u;
        const uint32_t pi = ((uint32_t)p + pre) & 7u;
        uint32_t c = s_gmeta[i][2 + pi];
        if (pi && i > 0) {                          // the byte straddling the start of segment i (fin_owned_ff, jpegamd_finalize.hip)
            const uint32_t tail_ones = (uint32_t)__builtin_ctz(~(s_gmeta[i - 1][1] & 0x7Fu));
            const uint32_t lead_ones = (uint32_t)__clz(~((s_gmeta[i][1] >> 8) << 24));
            c += (tail_ones >= pi && lead_ones >= 8u - pi) ? 1u : 0u;
        }
        if (lane >= 32) c = 0u;
        c += (uint32_t)__builtin_amdgcn_ds_bpermute((lane + 8) * 4, (int)c);     // i: 0+1, 2+3 (lanes 0..7 and 16..23 matter)
        c += (uint32_t)__builtin_amdgcn_ds_bpermute((lane + 16) * 4, (int)c);    // lanes 0..7: all four
        if (lane < 8) a.seg.grp_ff[(size_t)blockIdx.x * 8 + lane] = (uint16_t)min(c, 65535u);
        if (lane == 0) a.seg.grp_bits[blockIdx.x] = s_gmeta[0][0] + s_gmeta[1][0] + s_gmeta[2][0] + s_gmeta[3][0];
    }
}

int launch_entropy(const EntropyArgs &a, void *stream, void *const *ev) {
    if (a.seg_end <= a.seg_begin) return 0;
    const dim3 grid((a.seg_end - a.seg_begin + kWavesE - 1) / kWavesE), block(64 * kWavesE);
    if (ev) hipExtLaunchKernelGGL(k_entropy, grid, block, 0, (hipStream_t)stream, (hipEvent_t)ev[0], (hipEvent_t)ev[1], 0, a);
    else hipLaunchKernelGGL(k_entropy, grid, block, 0, (hipStream_t)stream, a);
    return (int)hipGetLastError();
}

}  // namespace jpegamd
