// jpegamd_entropy.hip -- k_segment_merge: the bit strings of a segment's tiles -> ONE bit string per segment (second of the
// pipeline's three kernels; rle.c:59-70, huffman.c:145-153).
//
// k_tile_encode leaves, per tile, an 8-word record and the Huffman bit string of every symbol of the tile except ONE: the DC
// difference of its first block, whose predictor is the last block of the tile before (rle.c:68-70 chains the prediction
// across all blocks of the picture) -- a tile that another wave, usually of another workgroup, is coding at the same time.
// One wave per segment (8 tiles of one block row, <= 256 blocks):
//   * lane t < 8 codes that symbol for tile t from the two records (size, amplitude, DC table);
//   * a prefix sum over (symbol + string) lengths places the 16 pieces; every tile's string is shifted to its bit offset by
//     one funnel shift per word and OR-ed into a bit window in LDS (all eight strings are requested up front);
//   * when the window is written out (once, for ordinary segments) the wave counts, for each of the 8 byte phases the
//     segment's first bit may end up at, the 0xFF bytes that lie wholly inside the segment: k_finalize then knows every
//     stuffing offset from per-segment numbers alone.
// Round 2's k_entropy did all of the coding here, from item lists in HBM: 5.7 M vector instructions per 8192^2 picture,
// this kernel ~1 M.
#include <hip/hip_ext.h>
#include "jpegamd_device.h"

namespace jpegamd {

constexpr int kWavesE = 4;               // segments per workgroup = one segment group (SegArrays::grp_bits / grp_ff)
constexpr int kSegBufWords = 512;               // LDS bit window per wave and 8 tiles (typical segment: ~180 words); flushed when nearly full
constexpr int kPieceWords = 128;                // words of a tile's string moved per step (8 bytes per lane)

// 0xFF bytes wholly inside the bit string, by byte phase: for the word `cur` followed by `nxt` (MSB-first), bit (31 - o) of
// the result is set when the 8 stream bits from offset o of `cur` are all ones.
__device__ __forceinline__ uint32_t ones8_starts(uint32_t cur, uint32_t nxt) {
    uint32_t hi = cur & __builtin_amdgcn_alignbit(cur, nxt, 31u), lo = nxt & (nxt << 1);
    hi &= __builtin_amdgcn_alignbit(hi, lo, 30u); lo &= lo << 2;
    hi &= __builtin_amdgcn_alignbit(hi, lo, 28u);
    return hi;
}

template <int kTiles>
__global__ __launch_bounds__(64 * kWavesE) void k_segment_merge(const MergeArgs a) {
    constexpr int kBuf = kSegBufWords * kTiles / 8;               // LDS bit window per wave: a typical segment is ~23 words per tile
    constexpr int kLanesPerTile = 64 / kTiles;                    // lanes that share a tile's string in the lane-parallel placement
    constexpr int kStepWords = kLanesPerTile * 4;                 // words of every tile's string moved per step
    __shared__ uint32_t s_dc[16];                   // DC table by size: (length << 16) | code
    __shared__ uint32_t s_win[kWavesE][kBuf + 8];
    __shared__ uint32_t s_gmeta[kWavesE][12];       // {bits, edge, ff[8]} of the workgroup's segments, for the group aggregate
    const int lane = lane_id(), wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int seg = a.seg_begin + (int)blockIdx.x * kWavesE + wave;
    // A wave without a segment (the last workgroup) walks the same code with no tiles -- its loads fall to the range checks,
    // its stores are guarded by `have` -- because the workgroup meets at two barriers.
    const bool have = seg < a.seg_end;
    const int segc = have ? seg : a.seg_end - 1;
    const uint32_t dcword = threadIdx.x < 16 ? a.huff[256 + threadIdx.x] : 0u;
    uint32_t *win = s_win[wave];

    const int image = a.tiles_per_image ? segc / a.num_segs : 0;    // a batch: every image has num_segs segments and tiles_per_image tiles
    const int sl = segc - image * a.num_segs;
    const int by = sl / a.segs_per_row;
    const int tx0 = (sl - by * a.segs_per_row) * kTiles;
    const int ntiles = have ? min(kTiles, a.tiles_per_row - tx0) : 0;
    const int tile_in_image = by * a.tiles_per_row + tx0;
    const int tile0 = image * a.tiles_per_image + tile_in_image;

    // Every load whose address does not depend on data goes out first: the tiles' records, the record of the tile in front
    // of the segment (its last DC), and the first piece of every tile's string.
    const uint32_t *tbase = a.tile_head + (size_t)tile0 * kTileHeadWords;
    const uint32_t *trec = tbase + (size_t)lane * kTileHeadWords;        // lane t: tile t
    uint4 rec = make_uint4(0u, 0u, 0u, 0u);
    uint32_t rsyms = 0, prev_last = 0;
    if (lane < ntiles) {
        rec = *reinterpret_cast<const uint4 *>(trec);                  // {string bits, first DC, last DC, exact-order fallbacks}
        rsyms = trec[4];
        if (lane == 0 && tile_in_image > 0) prev_last = *(trec + 2 - kTileHeadWords);   // the tile before (of the same image): its last DC
    }
    typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
    typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
    // (tiles beyond the segment fall to the range checks: zeros.  Everything sits in the vector offset: the range check does
    //  not see the scalar one.)
    const __amdgpu_buffer_rsrc_t hrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint32_t *>(tbase), 0, ntiles * kTileHeadWords * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint32_t *>(a.tile_over + (size_t)tile0 * kTileOverCap), 0, ntiles * kTileOverCap * 4, 0x00020000);
    // The strings, lane-parallel over the segment's tiles: with 8 tiles lane (t = l >> 3, i = l & 7) takes words 32 k + 4 i .. + 3
    // of tile t's string in step k (with 16 tiles: t = l >> 2, words 16 k + 4 i ..).  The first two steps (64 / 32 words: all or
    // most of a photo-like tile) are requested before anything is known.
    const uint32_t lt = (uint32_t)lane / (uint32_t)kLanesPerTile, li4 = ((uint32_t)lane % (uint32_t)kLanesPerTile) * 4u;
    const int head_off = (int)((lt * (uint32_t)kTileHeadWords + (uint32_t)kTileRecWords + li4) * 4u);
    u32x4 quad[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) quad[k] = __builtin_amdgcn_raw_buffer_load_b128(hrsrc, head_off + k * kStepWords * 4, 0, 0);

#pragma unroll
    for (int i = 0; i < kBuf / 64; ++i) win[i * 64 + lane] = 0u;
    if (lane < 8) win[kBuf + lane] = 0u;
    if (threadIdx.x < 16) s_dc[threadIdx.x] = dcword;
    __syncthreads();

    // The DC symbol of every tile's first block (rle.c:66-76, huffman.c:145-153): lane t, tile t.
    const uint32_t rbits = min(rec.x, (uint32_t)(kTileBlocks * kMaxBlockBits));            // a record is trusted only up to what 32 blocks can hold:
                                                                                           // the segment then fits kSegCapWords whatever the records say
    const bool bad_record = rbits != rec.x;
    const int left_last = lane_shift_up1((int)rec.z);          // (a cross-lane read must not sit inside a lane-dependent branch)
    const int pred = lane == 0 ? (int)prev_last : left_last;
    const int diff = (int)(short)(((int)rec.y - pred) & 0xFFFF);
    const int w = diff + (diff >> 31);                                            // rle.c:24-35
    int fbw;
    asm("v_ffbh_i32 %0, %1" : "=v"(fbw) : "v"(w << 1));
    const uint32_t nb = (uint32_t)(31 - fbw) & 31u;                                // rle.c:9-22 (fbw = -1 for a zero difference)
    const uint32_t dcw = s_dc[nb & 15u];
    const uint32_t dclen = lane < ntiles ? (dcw >> 16) + nb : 0u;
    const uint32_t dcsym = ((dcw & 0xFFFFu) << nb) | __builtin_amdgcn_ubfe((uint32_t)w, 0u, nb);   // code, then amplitude bits: <= 20 bits
    const uint32_t tbits = lane < ntiles ? dclen + rbits : 0u;
    const uint32_t tincl = wave_incl_scan_u32(tbits);
    const uint32_t seg_bits = (uint32_t)__builtin_amdgcn_readlane((int)tincl, 63);
    const uint32_t toff = tincl - tbits;                                           // bit offset of tile t's DC symbol in the segment
    const int seg_syms = wave_sum_i32((int)rsyms);
    const int seg_exact = wave_sum_i32((int)rec.w);
    const bool any_bad = __ballot(bad_record) != 0ull;

    uint32_t wbase = 0, last_word = 0, first_word = 0, seg_edge = 0;
    bool flushed = false, any_ff = false;
    uint32_t ffc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    // 0xFF census of the words [0, n) of the window; every word's successor is final (n == complete words and the rest is
    // counted later, or the string has ended and the window is zero behind it).  `last_word` (the word in front of
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
    uint32_t *const segw = a.seg.words + (size_t)seg * a.seg.words_stride;
    // Everything below bit `upto` of the segment is final: write the complete words out when the next piece might not fit.
    const auto make_room = [&](uint32_t upto /*bits*/, uint32_t need_end /*bit behind the next piece*/) {
        if (__builtin_expect((need_end >> 5) - wbase + 3u > (uint32_t)kBuf, 0)) {
            const uint32_t done = (upto >> 5) - wbase;                    // complete words in the window
            if (done) {
                census(done - 1u, flushed);                                // the last complete word waits for its successor
                const uint32_t part = win[done];
                if (have) for (uint32_t j = (uint32_t)lane; j < done; j += 64) segw[wbase + j] = win[j];
                if (!flushed) first_word = win[0];
                last_word = win[done - 1];
#pragma unroll
                for (int i = 0; i < kBuf / 64; ++i) win[i * 64 + lane] = 0u;
                if (lane < 8) win[kBuf + lane] = 0u;
                if (lane == 0) win[0] = part;
                wbase += done;
                flushed = true;
            }
        }
    };

    const uint32_t nwords_t = (rbits + 31u) >> 5;                                 // lane t < 8: words of tile t's string
    const uint32_t max_words = (uint32_t)wave_max_u32(nwords_t);
    if (__builtin_expect(seg_bits <= (uint32_t)((kBuf - 8) * 32) && max_words <= (uint32_t)kTileHeadStr, 1)) {
        // The whole segment fits the window and every string its tile's head (nearly always): all tiles at once.
        const uint32_t my_start = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(lt * 4u), (int)(toff + dclen));   // bit offset of my tile's string
        const uint32_t my_words = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(lt * 4u), (int)nwords_t);
        if (lane < ntiles) {                                          // the DC symbols: lane t, <= 20 bits each
            const uint32_t wi = toff >> 5, sh = toff & 31u;
            const unsigned long long s64 = ((unsigned long long)dcsym << (64u - dclen)) >> sh;     // left-aligned at bit sh of a word pair
            atomicOr(&win[wi], (uint32_t)(s64 >> 32));
            atomicOr(&win[wi + 1], (uint32_t)s64);
        }
        for (uint32_t k = 0; k * (uint32_t)kStepWords < max_words; ++k) {
            u32x4 q = k == 0 ? quad[0] : quad[1];
            if (k >= 2) q = __builtin_amdgcn_raw_buffer_load_b128(hrsrc, head_off + (int)k * kStepWords * 4, 0, 0);
            const uint32_t j0 = k * (uint32_t)kStepWords + li4;       // my first word of this step
            if (j0 < my_words) {
                // words beyond the string are not zero in memory: mask by count
                const uint32_t x0 = q[0], x1 = j0 + 1u < my_words ? q[1] : 0u, x2 = j0 + 2u < my_words ? q[2] : 0u, x3 = j0 + 3u < my_words ? q[3] : 0u;
                const uint32_t rel = my_start + j0 * 32u, wi = rel >> 5, sh = rel & 31u;
                atomicOr(&win[wi], __builtin_amdgcn_alignbit(0u, x0, sh));
                atomicOr(&win[wi + 1], __builtin_amdgcn_alignbit(x0, x1, sh));
                atomicOr(&win[wi + 2], __builtin_amdgcn_alignbit(x1, x2, sh));
                atomicOr(&win[wi + 3], __builtin_amdgcn_alignbit(x2, x3, sh));
                atomicOr(&win[wi + 4], __builtin_amdgcn_alignbit(x3, 0u, sh));
            }
        }
    } else {
        // The 16 pieces, in order (a segment that does not fit the window, or a tile whose string reaches beyond its head).  A DC symbol is <= 20 bits: the lane that made it ORs it in.  A string is moved 128 words at
        // a time: lane l holds words 2 l and 2 l + 1 of the piece; shifted by the piece's bit offset they land in three window
        // words, the outer two shared with the neighbour lanes (ds_or).
    #pragma unroll
        for (int t = 0; t < kTiles; ++t) {
            if (t >= ntiles) break;
            const uint32_t off = (uint32_t)__builtin_amdgcn_readlane((int)toff, t);
            const uint32_t dl = (uint32_t)__builtin_amdgcn_readlane((int)dclen, t);
            const uint32_t sb = (uint32_t)__builtin_amdgcn_readlane((int)rbits, t);
            make_room(off, off + dl + min(sb, (uint32_t)(kPieceWords * 32)));
            if (lane == t && dl) {
                const uint32_t rel = off - wbase * 32u, wi = rel >> 5, sh = rel & 31u;
                const unsigned long long s64 = ((unsigned long long)dcsym << (64u - dl)) >> sh;     // left-aligned at bit sh of a word pair
                atomicOr(&win[wi], (uint32_t)(s64 >> 32));
                atomicOr(&win[wi + 1], (uint32_t)s64);
            }
            const uint32_t nwords = (sb + 31u) >> 5;
            u32x2 pc = __builtin_amdgcn_raw_buffer_load_b64(hrsrc, lane < kTileHeadStr / 2 ? (int)((uint32_t)lane * 8u) + (t * kTileHeadWords + kTileRecWords) * 4 : 0x7FFFFFF0, 0, 0);
            for (uint32_t w0 = 0; w0 < nwords; w0 += w0 ? (uint32_t)kPieceWords : (uint32_t)kTileHeadStr) {
                const uint32_t start = off + dl + w0 * 32u;                              // bit offset of the piece in the segment
                if (w0) {                                                                // (only strings beyond the tile's head come here)
                    make_room(start, start + min(sb - w0 * 32u, (uint32_t)(kPieceWords * 32)));
                    pc = __builtin_amdgcn_raw_buffer_load_b64(orsrc, (int)((uint32_t)lane * 8u) + (t * kTileOverCap + (int)w0 - kTileHeadStr) * 4, 0, 0);
                }
                // (a piece's words beyond the string are not zero: mask by count)
                const uint32_t left = min(nwords - w0, w0 ? (uint32_t)kPieceWords : (uint32_t)kTileHeadStr);
                const uint32_t x0 = 2u * (uint32_t)lane < left ? pc[0] : 0u, x1 = 2u * (uint32_t)lane + 1u < left ? pc[1] : 0u;
                const uint32_t rel = start - wbase * 32u, wi = (rel >> 5) + 2u * (uint32_t)lane, sh = rel & 31u;
                if (2u * (uint32_t)lane < left) {
                    atomicOr(&win[wi], __builtin_amdgcn_alignbit(0u, x0, sh));
                    atomicOr(&win[wi + 1], __builtin_amdgcn_alignbit(x0, x1, sh));
                    atomicOr(&win[wi + 2], __builtin_amdgcn_alignbit(x1, 0u, sh));
                }
            }
        }
    }
    {   // the segment has ended: census of everything still in the window, then write it out
        const uint32_t done = (seg_bits >> 5) - wbase;
        const uint32_t nw = done + ((seg_bits & 31u) ? 1u : 0u);           // words holding bits; the window is zero behind them
        census(nw, flushed);
        if (have) for (uint32_t j = (uint32_t)lane; j < nw; j += 64) segw[wbase + j] = win[j];
        if (!flushed) first_word = win[0];
        if (done) last_word = win[done - 1];
        const uint32_t part = win[done];
        {
            const uint32_t p = seg_bits & 31u;
            const uint32_t tail = p ? ((last_word << p) | (part >> (32u - p))) : last_word;
            seg_edge = ((first_word >> 24) << 8) | (tail & 0x7Fu);           // first 8 bits | last 7 bits of the segment's string
        }
        if (lane == 0 && have) {
            a.seg.edge[seg] = seg_edge;
            a.seg.bits[seg] = seg_bits;
            a.seg.syms[seg] = (uint32_t)seg_syms;
            a.seg.exact[seg] = (uint32_t)seg_exact;
            if (any_bad) atomicOr(a.status, 2u);
        }
    }
    uint32_t mine = 0;                                                       // lane p < 8 stores the count of phase p
    if (any_ff) {
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const uint32_t t = (uint32_t)wave_sum_i32((int)ffc[p]);
            if (lane == p) mine = t;
        }
    }
    if (lane < 8) {
        // (16-bit counts: a segment's Huffman string cannot hold that many 0xFF bytes -- every code but the rare run-15 ones has
        //  a zero in it -- but a count that did not fit must not pass silently)
        if (have) a.seg.ffin[(size_t)seg * 8 + lane] = (uint16_t)min(mine, 65535u);
        if (have && mine > 65535u) atomicOr(a.status, 2u);
        s_gmeta[wave][2 + lane] = mine;
    }
    if (lane == 0) { s_gmeta[wave][0] = seg_bits; s_gmeta[wave][1] = have ? seg_edge : 0u; }   // (no segment: no bits, no ones at either end)
    // Group aggregate (SegArrays::grp_bits / grp_ff): lane i * 8 + p of wave 0 takes segment i of the group at group phase p.
    __syncthreads();
    static_assert(kWavesE == kSegGroup && kWavesE == 4, "one workgroup of k_segment_merge = one segment group of four");
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
        if (lane < 8) a.seg.grp_ff[(size_t)blockIdx.x * 8 + lane] = c;
        if (lane == 0) a.seg.grp_bits[blockIdx.x] = s_gmeta[0][0] + s_gmeta[1][0] + s_gmeta[2][0] + s_gmeta[3][0];
    }
}

int launch_segment_merge(const MergeArgs &a, void *stream, void *const *ev) {
    if (a.seg_end <= a.seg_begin) return 0;
    const dim3 grid((a.seg_end - a.seg_begin + kWavesE - 1) / kWavesE), block(64 * kWavesE);
    if (a.seg_tiles == kSegTilesBatch) {
        if (ev) hipExtLaunchKernelGGL(k_segment_merge<kSegTilesBatch>, grid, block, 0, (hipStream_t)stream, (hipEvent_t)ev[0], (hipEvent_t)ev[1], 0, a);
        else hipLaunchKernelGGL(k_segment_merge<kSegTilesBatch>, grid, block, 0, (hipStream_t)stream, a);
    } else {
        if (ev) hipExtLaunchKernelGGL(k_segment_merge<kSegTiles>, grid, block, 0, (hipStream_t)stream, (hipEvent_t)ev[0], (hipEvent_t)ev[1], 0, a);
        else hipLaunchKernelGGL(k_segment_merge<kSegTiles>, grid, block, 0, (hipStream_t)stream, a);
    }
    return (int)hipGetLastError();
}

}  // namespace jpegamd
