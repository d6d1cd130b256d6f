// jpegamd_stitch.hip -- k_stitch: the tiles' bit strings -> the finished entropy-coded segment, in ONE pass (second and last
// kernel of the pipeline for whole images; rle.c:59-70, huffman.c:26-81,145-153, jpeg_handler.c:220-262).
//
// Rounds 1-3 ran two kernels here: k_segment_merge joined the strings of a segment's tiles into seg.words (HBM) and counted, for
// each of the 8 byte phases a segment might start at, the 0xFF bytes inside it; k_finalize -- behind a kernel boundary, the
// global barrier -- scanned every predecessor's numbers and shifted / stuffed the strings out of seg.words.  Both were chains of
// dependent memory round trips at one picture per launch (6.9 + 7.3 us), and 20 MB per picture went to HBM and back.
//
// Here a wave owns a segment from the tiles' records to the output bytes:
//   1. records of its tiles -> the DC symbol of every tile's first block (its predictor is the tile before: rle.c:66-76,
//      huffman.c:145-153), bit offsets of the tiles inside the segment, the segment's bit count;
//   2. the workgroup's bit count is PUBLISHED, and the bit offset of the workgroup in the picture comes from a decoupled
//      look-back over the earlier workgroups' counts (one wave; every lane polls four predecessors);
//      meanwhile all tile strings are shifted into the wave's bit window in LDS;
//   3. with the byte phase known, the 0xFF bytes the segment owns are counted ONCE (an output byte is owned by the segment
//      holding its last bit) -- not for eight phases -- and a second look-back (a scalar sum again) gives the stuffed bytes in
//      front of the workgroup;
//   4. the window goes out as aligned 16-byte pieces, 0x00 behind every 0xFF (huffman.c:29-31).
// Hand-offs between workgroups are 8-byte {epoch, status, value} granules written and polled with agent-scope (sc1) accesses;
// nothing else written in this launch is read by another workgroup.  A workgroup only waits for workgroups of LOWER index,
// which the dispatcher started earlier; every spin is bounded and ends in a status bit, not a hang.
// Segments too long for the window (noise, very high qualities) run as several PARTS -- whole tiles, or pieces of one tile --
// twice: once to count, once to write.
#include <hip/hip_ext.h>
#include "jpegamd_device.h"

namespace jpegamd {

constexpr int kStStripPieces = 130;              // 16-byte pieces of a wave's staging strip (15 + 2 x 1024 bytes at most)
constexpr int kStCarry = 4;                      // window word j lives at index kStCarry + j; kStCarry - 1 holds the bits in front
constexpr uint32_t kSpinLimit = 1u << 18;        // polls before a look-back gives up (status bit 2): a fraction of a second, not a hang

typedef unsigned long long u64;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((address_space(1))) uint32_t gu32;

// ---- the hand-off granule: 16 bytes per workgroup, written by ONE store and read by ONE load, both write-through / L1-bypassing
// (sc1).  AGGREGATE (status 1: what the workgroup's own segments add, whatever lies in front of it):
//   d0  31..18 epoch, 17..16 status, 15..8 the workgroup's first 8 bits, 7..1 its last 7 bits, 0 "counts saturated"
//   d1  31..24 check byte, 23..0 bits of the workgroup's segments
//   d2, d3     for each byte phase p = 0..7 the workgroup's first bit may land on: the 0xFF bytes lying wholly inside the
//              workgroup's bits, 8 bits each, saturated at 255 (then bit 0 of d0 is set and StitchArgs::desc_ffx holds them in full)
// INCLUSIVE (status 2: everything from the picture's first bit to the workgroup's last):
//   d0  epoch, status, 7..1 the stream's last 7 bits     d1 bits (low 32)     d2 0xFF bytes owned (low 32)
//   d3  31..24 check byte, 23..12 bits (high 12), 11..0 0xFF bytes (high 12)
// The check byte ties the four words to one store (a 16-byte store was never seen torn on gfx950, but nothing promises it).
__device__ __forceinline__ uint32_t gran_check(uint32_t epoch, uint32_t x, uint32_t y) {
    return (epoch + __builtin_amdgcn_sad_u8(x, 0u, 0u) + __builtin_amdgcn_sad_u8(y, 0u, 0u)) & 0xFFu;
}
__device__ __forceinline__ bool gran_valid(const u32x4 &v, uint32_t epoch) {
    const uint32_t st = (v[0] >> 16) & 3u;
    if ((v[0] >> 18) != epoch || st == 0u) return false;
    return st == 1u ? (v[1] >> 24) == gran_check(epoch, v[2], v[3]) : (v[3] >> 24) == gran_check(epoch, v[1], v[2]);
}
// a byte straddling the boundary between a string ending in `tail7` and one starting with `first8` (left-aligned, zeros
// behind a string shorter than 8 bits), `a` of its bits in front of the boundary: is it 0xFF?
__device__ __forceinline__ uint32_t straddle_ff(uint32_t tail7, uint32_t first8, uint32_t a) {
    const uint32_t tail_ones = (uint32_t)__builtin_ctz(~(tail7 & 0x7Fu));          // 0..7
    const uint32_t lead_ones = (uint32_t)__clz(~(first8 << 24));                   // 0..8
    return (a != 0u && tail_ones >= a && lead_ones >= 8u - a) ? 1u : 0u;
}

// last min(nbits, 7) bits of a bit string of nbits bits whose words start at w (right-aligned; zeros above them)
__device__ __forceinline__ uint32_t tail7_of(const uint32_t *w, uint32_t nbits) {
    if (nbits == 0u) return 0u;
    const uint32_t take = min(nbits, 7u), pos = nbits - take, i = pos >> 5, sh = pos & 31u;
    const u64 win = ((u64)w[i] << 32) | w[i + 1];
    return (uint32_t)((win << sh) >> (64u - take));
}
// the last 7 bits of (a string ending in `before`) followed by a string of nbits bits ending in `tail`
__device__ __forceinline__ uint32_t tail7_join(uint32_t before, uint32_t tail, uint32_t nbits) {
    return nbits >= 7u ? tail : ((before << nbits) | tail) & 0x7Fu;
}
// 0xFF bytes by byte phase (jpegamd_entropy.hip has the same census): for the word `cur` followed by `nxt` (MSB-first), bit
// (31 - o) of the result is set when the 8 stream bits from offset o of `cur` are all ones.
__device__ __forceinline__ uint32_t st_ones8_starts(uint32_t cur, uint32_t nxt) {
    uint32_t hi = cur & __builtin_amdgcn_alignbit(cur, nxt, 31u), lo = nxt & (nxt << 1);
    hi &= __builtin_amdgcn_alignbit(hi, lo, 30u); lo &= lo << 2;
    hi &= __builtin_amdgcn_alignbit(hi, lo, 28u);
    return hi;
}

template <int kTiles, int kStWaves>
__global__ __launch_bounds__(64 * kStWaves) void k_stitch(const StitchArgs a) {
    constexpr int kBuf = 64 * kTiles;                             // bit window per wave (words): a typical segment is ~23 words per tile
    constexpr int kLanesPerTile = 64 / kTiles;                    // lanes that share a tile's string in the lane-parallel placement
    constexpr int kStepWords = kLanesPerTile * 4;                 // words of every tile's string moved per step
    constexpr uint32_t kPartBits = (uint32_t)(kBuf - 8) * 32u;    // a part's bits: the window, less a margin for the placement's trailing ds_or
    constexpr uint32_t kPieceWords = 128;                         // words of one tile's string moved per step of the tile-by-tile placement
    static_assert(kStWaves <= 16, "the waves' numbers are scanned in one DPP row");
    __shared__ __attribute__((aligned(16))) uint32_t s_win[kStWaves][kStCarry + kBuf + 12];
    __shared__ __attribute__((aligned(16))) uint32_t s_strip[kStWaves][4 * kStStripPieces];
    // what a wave tells the workgroup about its segment: bits, first 8 / last 7 bits, 0xFF bytes inside it by byte phase ...
    __shared__ uint32_t s_bits[kStWaves], s_first8[kStWaves], s_last7[kStWaves], s_ffc[kStWaves][8];
    // ... and what wave 0 tells it back once the look-back is through: stream bit offset, stuffed bytes in front, owned 0xFF bytes, the 7 bits in front
    __shared__ u64 s_b0s[kStWaves], s_ffoff[kStWaves];
    __shared__ uint32_t s_myff[kStWaves], s_tailin[kStWaves];
    const int lane = lane_id(), wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), tid = (int)threadIdx.x;
    const int image = a.batch > 1 ? (int)blockIdx.x / a.wgs_per_image : 0;
    const int g = (int)blockIdx.x - image * a.wgs_per_image;      // workgroup inside its picture
    const int s = g * kStWaves + wave;                            // segment inside its picture
    const bool have = s < a.num_segs;                             // (a wave without a segment walks on with no tiles: the workgroup meets at barriers)
    const int sc = have ? s : a.num_segs - 1;
    uint32_t *const wv = &s_win[wave][kStCarry];                  // wv[-1]: the (up to 7) stream bits in front of the window's first bit
    uint8_t *const out = a.out[image];
    const u64 out_capacity = a.out_capacity;

    if (g == 0 && a.prefix_len > 0)                               // JFIF prefix (jpeg_handler.c:220-233)
        for (int i = tid; i < a.prefix_len; i += 64 * kStWaves)
            if ((u64)i < out_capacity) out[i] = a.prefix[i];

    // ---- 1. the segment's tiles: records, the DC symbol of every tile's first block, offsets ---------------------------------
    const int by = sc / a.segs_per_row;
    const int tx0 = (sc - by * a.segs_per_row) * kTiles;
    const int ntiles = have ? min(kTiles, a.tiles_per_row - tx0) : 0;
    const int tile_in_image = by * a.tiles_per_row + tx0;
    const int tile0 = image * a.tiles_per_image + tile_in_image;
    const uint32_t *tbase = a.tile_head + (size_t)tile0 * kTileHeadWords;
    const uint32_t *trec = tbase + (size_t)lane * kTileHeadWords;        // lane t: tile t
    const uint32_t dcword = lane < 16 ? a.huff[256 + lane] : 0u;         // DC table by size: (length << 16) | code
    uint4 rec = make_uint4(0u, 0u, 0u, 0u);
    uint32_t rsyms = 0, prev_last = 0;
    if (lane < ntiles) {
        rec = *reinterpret_cast<const uint4 *>(trec);                    // {string bits, first DC, last DC, exact-order fallbacks}
        rsyms = trec[4];
        if (lane == 0 && tile_in_image > 0) prev_last = *(trec + 2 - kTileHeadWords);   // the tile before (of the same picture): its last DC
    }
    typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
    typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
    const __amdgpu_buffer_rsrc_t hrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(tbase), 0, ntiles * kTileHeadWords * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint32_t *>(a.tile_over + (size_t)tile0 * kTileOverCap), 0, ntiles * kTileOverCap * 4, 0x00020000);
    // the strings, lane-parallel over the tiles (lane (t = l / lanes per tile, i): words step * k + 4 i .. + 3 of tile t's string in
    // step k); the first two steps are requested before anything is known about the tiles
    const uint32_t lt = (uint32_t)lane / (uint32_t)kLanesPerTile, li4 = ((uint32_t)lane % (uint32_t)kLanesPerTile) * 4u;
    const int head_off = (int)((lt * (uint32_t)kTileHeadWords + (uint32_t)kTileRecWords + li4) * 4u);
    u32x4 quad[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) quad[k] = __builtin_amdgcn_raw_buffer_load_b128(hrsrc, head_off + k * kStepWords * 4, 0, 0);

    const auto zero_window = [&]() {
        const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int i = 0; i < (kStCarry + kBuf + 12) / 4; i += 64)
            if (i + lane < (kStCarry + kBuf + 12) / 4) reinterpret_cast<u32x4 *>(&s_win[wave][0])[i + lane] = z;
    };
    zero_window();

    const uint32_t rbits = min(rec.x, (uint32_t)(kTileBlocks * kMaxBlockBits));            // a record is trusted only up to what 32 blocks can hold
    const bool bad_record = rbits != rec.x;
    const int left_last = lane_shift_up1((int)rec.z);
    const int pred = lane == 0 ? (int)prev_last : left_last;
    const int diff = (int)(short)(((int)rec.y - pred) & 0xFFFF);
    const int w = diff + (diff >> 31);                                            // rle.c:24-35
    int fbw;
    asm("v_ffbh_i32 %0, %1" : "=v"(fbw) : "v"(w << 1));
    const uint32_t nb = (uint32_t)(31 - fbw) & 31u;                                // rle.c:9-22 (fbw = -1 for a zero difference)
    const uint32_t dcw = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((nb & 15u) * 4u), (int)dcword);
    const uint32_t dclen = lane < ntiles ? (dcw >> 16) + nb : 0u;
    const uint32_t dcsym = ((dcw & 0xFFFFu) << nb) | __builtin_amdgcn_ubfe((uint32_t)w, 0u, nb);   // code, then amplitude bits: <= 20 bits
    const uint32_t tbits = lane < ntiles ? dclen + rbits : 0u;
    const uint32_t tincl = wave_incl_scan_u32(tbits);
    const uint32_t seg_bits = (uint32_t)__builtin_amdgcn_readlane((int)tincl, 63);
    const uint32_t toff = tincl - tbits;                                           // bit offset of tile t's DC symbol in the segment
    const int seg_syms = wave_sum_i32((int)rsyms);
    const int seg_exact = wave_sum_i32((int)rec.w);
    const bool any_bad = __ballot(bad_record) != 0ull;
    const uint32_t nwords_t = (rbits + 31u) >> 5;                                  // lane t: words of tile t's string
    const uint32_t max_words = (uint32_t)wave_max_u32(nwords_t);
    const bool single = seg_bits <= kPartBits && max_words <= (uint32_t)kTileHeadStr;   // (uniform) everything fits the window: ONE part, placed once
    if (lane == 0 && have) {
        const size_t sg = (size_t)image * (size_t)a.num_segs + (size_t)s;
        a.seg_syms[sg] = (uint32_t)seg_syms;
        a.seg_exact[sg] = (uint32_t)seg_exact;
        if (any_bad) atomicOr(a.status, 2u);
    }

    // One tile's string word j (head, then the sparse reservation), for the slow paths.
    const auto str_word_off = [&](int t, uint32_t j, bool &over) -> int {
        over = j >= (uint32_t)kTileHeadStr;
        return over ? (t * kTileOverCap + (int)j - kTileHeadStr) * 4 : (t * kTileHeadWords + kTileRecWords + (int)j) * 4;
    };
    // Lane-parallel placement of the tiles [0, ntiles) at bit 0 of the window (only when `single`).
    const auto place_all = [&]() {
        const uint32_t my_start = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(lt * 4u), (int)(toff + dclen));   // bit offset of my tile's string
        const uint32_t my_words = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(lt * 4u), (int)nwords_t);
        if (lane < ntiles) {                                          // the DC symbols: lane t, <= 20 bits each
            const uint32_t wi = toff >> 5, sh = toff & 31u;
            const u64 s64 = ((u64)dcsym << (64u - dclen)) >> sh;     // left-aligned at bit sh of a word pair
            if (dclen) { atomicOr(&wv[wi], (uint32_t)(s64 >> 32)); atomicOr(&wv[wi + 1], (uint32_t)s64); }
        }
        for (uint32_t k = 0; k * (uint32_t)kStepWords < max_words; ++k) {
            u32x4 q = k == 0 ? quad[0] : quad[1];
            if (k >= 2) q = __builtin_amdgcn_raw_buffer_load_b128(hrsrc, head_off + (int)k * kStepWords * 4, 0, 0);
            const uint32_t j0 = k * (uint32_t)kStepWords + li4;       // my first word of this step
            if (j0 < my_words) {
                // words beyond the string are not zero in memory: mask by count
                const uint32_t x0 = q[0], x1 = j0 + 1u < my_words ? q[1] : 0u, x2 = j0 + 2u < my_words ? q[2] : 0u, x3 = j0 + 3u < my_words ? q[3] : 0u;
                const uint32_t rel = my_start + j0 * 32u, wi = rel >> 5, sh = rel & 31u;
                atomicOr(&wv[wi], __builtin_amdgcn_alignbit(0u, x0, sh));
                atomicOr(&wv[wi + 1], __builtin_amdgcn_alignbit(x0, x1, sh));
                atomicOr(&wv[wi + 2], __builtin_amdgcn_alignbit(x1, x2, sh));
                atomicOr(&wv[wi + 3], __builtin_amdgcn_alignbit(x2, x3, sh));
                atomicOr(&wv[wi + 4], __builtin_amdgcn_alignbit(x3, 0u, sh));
            }
        }
    };
    // Placement of ONE piece at bit `at` of the window: tile t's DC symbol (with_dc) and words [w0, w0 + n) of its string, the
    // string's bit count sb masking the last word.  n <= kPieceWords: lane l holds words 2 l, 2 l + 1 of the piece.
    const auto place_piece = [&](int t, uint32_t at, bool with_dc, uint32_t w0, uint32_t n, uint32_t nwords) {
        uint32_t pos = at;
        if (with_dc) {
            const uint32_t dl = (uint32_t)__builtin_amdgcn_readlane((int)dclen, t), ds = (uint32_t)__builtin_amdgcn_readlane((int)dcsym, t);
            if (lane == 0 && dl) {
                const uint32_t wi = pos >> 5, sh = pos & 31u;
                const u64 s64 = ((u64)ds << (64u - dl)) >> sh;
                atomicOr(&wv[wi], (uint32_t)(s64 >> 32)); atomicOr(&wv[wi + 1], (uint32_t)s64);
            }
            pos += dl;
        }
        if (n == 0u) return;
        const uint32_t j = w0 + 2u * (uint32_t)lane;                  // my first word of the string
        u32x2 pc = {0u, 0u};
        if (2u * (uint32_t)lane < n) {
            bool over;
            const int off = str_word_off(t, j, over);
            // (a piece never straddles head and reservation: kTileHeadStr and the piece size are both even, pieces start at even words)
            pc = over ? __builtin_amdgcn_raw_buffer_load_b64(orsrc, off, 0, 0) : __builtin_amdgcn_raw_buffer_load_b64(hrsrc, off, 0, 0);
        }
        const uint32_t x0 = (2u * (uint32_t)lane < n && j < nwords) ? pc[0] : 0u, x1 = (2u * (uint32_t)lane + 1u < n && j + 1u < nwords) ? pc[1] : 0u;
        const uint32_t rel = pos + 64u * (uint32_t)lane, wi = rel >> 5, sh = rel & 31u;
        if (2u * (uint32_t)lane < n) {
            atomicOr(&wv[wi], __builtin_amdgcn_alignbit(0u, x0, sh));
            atomicOr(&wv[wi + 1], __builtin_amdgcn_alignbit(x0, x1, sh));
            atomicOr(&wv[wi + 2], __builtin_amdgcn_alignbit(x1, 0u, sh));
        }
    };

    // The segment as PARTS.  A part is what the window holds at once: the tiles [t0, t1) whole, or -- a tile whose string alone
    // is longer than the window -- a run of words of ONE tile.  Cursor: (pt, pw) = next tile, next word of its string.
    struct Part { int t0, t1; uint32_t w0, w1; uint32_t bits; };     // tiles [t0, t1) (w0 = 0) or words [w0, w1) of tile t0 (t1 == t0 + 1)
    const auto next_part = [&](int pt, uint32_t pw) -> Part {
        Part p;
        p.t0 = pt; p.w0 = pw;
        const uint32_t tb0 = (uint32_t)__builtin_amdgcn_readlane((int)tbits, pt), sb0 = (uint32_t)__builtin_amdgcn_readlane((int)rbits, pt);
        const uint32_t dl0 = tb0 - sb0;
        if (pw == 0u && tb0 <= kPartBits) {                           // whole tiles, as many as fit
            uint32_t bits = tb0;
            int t1 = pt + 1;
            while (t1 < ntiles) {
                const uint32_t tb = (uint32_t)__builtin_amdgcn_readlane((int)tbits, t1);
                if (bits + tb > kPartBits) break;
                bits += tb; ++t1;
            }
            p.t1 = t1; p.w1 = 0u; p.bits = bits;
        } else {                                                      // a run of words of one long string (multiples of the piece size)
            const uint32_t nwords = (sb0 + 31u) >> 5;
            const uint32_t room = ((kPartBits - 32u) >> 5) / kPieceWords * kPieceWords;      // words per part (the DC symbol's 20 bits fit the margin)
            const uint32_t w1 = min(nwords, pw + room);
            p.t1 = pt + 1; p.w1 = w1;
            p.bits = (pw == 0u ? dl0 : 0u) + (w1 == nwords ? sb0 - 32u * pw : 32u * (w1 - pw));
        }
        return p;
    };
    const auto place_part = [&](const Part &p) {
        if (p.w1 == 0u) {
            uint32_t at = 0;
            for (int t = p.t0; t < p.t1; ++t) {
                const uint32_t sb = (uint32_t)__builtin_amdgcn_readlane((int)rbits, t), nwords = (sb + 31u) >> 5;
                bool first = true;
                uint32_t pos = at;
                for (uint32_t w0 = 0; first || w0 < nwords; w0 += kPieceWords) {
                    place_piece(t, pos, first, w0, min(nwords - w0, kPieceWords), nwords);
                    if (first) pos += (uint32_t)__builtin_amdgcn_readlane((int)dclen, t);
                    pos += 32u * kPieceWords;
                    first = false;
                }
                at += (uint32_t)__builtin_amdgcn_readlane((int)tbits, t);
            }
        } else {
            const uint32_t sb = (uint32_t)__builtin_amdgcn_readlane((int)rbits, p.t0), nwords = (sb + 31u) >> 5;
            uint32_t pos = 0;
            bool first = p.w0 == 0u;
            for (uint32_t w0 = p.w0; w0 < p.w1; w0 += kPieceWords) {
                place_piece(p.t0, pos, first, w0, min(p.w1 - w0, kPieceWords), nwords);
                if (first) pos += (uint32_t)__builtin_amdgcn_readlane((int)dclen, p.t0);
                pos += 32u * kPieceWords;
                first = false;
            }
        }
    };
    const auto advance = [&](const Part &p, int &pt, uint32_t &pw) {
        if (p.w1 == 0u) { pt = p.t1; pw = 0u; }
        else {
            const uint32_t sb = (uint32_t)__builtin_amdgcn_readlane((int)rbits, p.t0), nwords = (sb + 31u) >> 5;
            if (p.w1 >= nwords) { pt = p.t0 + 1; pw = 0u; } else { pw = p.w1; }
        }
    };

    // 0xFF bytes among the `nown` owned bytes of what the window holds, `lead` borrowed bits in wv[-1]'s low end.
    // Owned byte q = stream bits [8 q - lead, 8 q - lead + 8) of the window.
    const auto group_words = [&](uint32_t idx, uint32_t lead, uint32_t (&v)[4]) {  // the 16 owned bytes 16 idx .. + 15, big-endian in 4 words
        const int wi = 4 * (int)idx - (lead ? 1 : 0);
        const uint32_t x0 = wv[wi], x1 = wv[wi + 1], x2 = wv[wi + 2], x3 = wv[wi + 3], x4 = wv[wi + 4];
        v[0] = lead ? __builtin_amdgcn_alignbit(x0, x1, lead) : x0;
        v[1] = lead ? __builtin_amdgcn_alignbit(x1, x2, lead) : x1;
        v[2] = lead ? __builtin_amdgcn_alignbit(x2, x3, lead) : x2;
        v[3] = lead ? __builtin_amdgcn_alignbit(x3, x4, lead) : x3;
    };
    const auto mask_tail = [&](uint32_t (&v)[4], int nvl /*this lane's owned bytes: <= 0 none, >= 16 all*/) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int kb = min(max(nvl - 4 * k, 0), 4);
            v[k] = kb ? v[k] & (0xFFFFFFFFu << (8 * (4 - kb))) : 0u;
        }
    };
    const auto census = [&](uint32_t lead, uint32_t nown) -> uint32_t {
        uint32_t c = 0;
        for (uint32_t i0 = 0; 16u * i0 < nown; i0 += 64u) {
            const uint32_t idx = i0 + (uint32_t)lane;
            if (16u * idx < nown) {
                uint32_t v[4];
                group_words(idx, lead, v);
                mask_tail(v, (int)nown - 16 * (int)idx);
#pragma unroll
                for (int k = 0; k < 4; ++k) c += (uint32_t)__popc(((v[k] & 0x7F7F7F7Fu) + 0x01010101u) & v[k] & 0x80808080u);
            }
        }
        return (uint32_t)wave_sum_i32((int)c);
    };

    // ---- 2. the strings go into the window; the segment's numbers -----------------------------------------------------------
    // 0xFF bytes lying wholly inside the segment for each byte phase p its first bit may land on: a byte starts at bit i of the
    // segment when (i + p) % 8 == 0.  `at` = bit offset of the window's first bit in the segment (parts).
    uint32_t ffc[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
    bool any_ff = false;
    const auto census8 = [&](uint32_t nbits, uint32_t at) {
        const uint32_t nw = (nbits + 31u) >> 5;
        for (uint32_t i0 = 0; i0 < nw; i0 += 64u) {
            const uint32_t i = i0 + (uint32_t)lane;
            uint32_t cw = 0, nx = 0;
            if (i < nw) { cw = wv[i]; nx = wv[i + 1]; }
            const uint32_t m = st_ones8_starts(cw, nx);       // (the window is zero behind the string: no run of ones reaches beyond it)
            if (__ballot(m != 0u) != 0ull) {
                any_ff = true;
#pragma unroll
                for (int p = 0; p < 8; ++p) ffc[p] += (uint32_t)__popc(m & (0x80808080u >> ((16u - (uint32_t)p - (at & 7u)) & 7u)));
            }
        }
    };
    uint32_t seg_tail = 0, seg_first8 = 0;                            // last min(seg_bits, 7) bits (right-aligned), first 8 (left-aligned, zeros behind a shorter string)
    uint32_t xff = 0;                                                 // lane p < 8: bytes straddling two parts that are 0xFF at phase p
    if (single) {
        if (ntiles > 0) place_all();
        seg_tail = tail7_of(wv, seg_bits);
        seg_first8 = wv[0] >> 24;
        census8(seg_bits, 0u);
    } else {
        int pt = 0; uint32_t pw = 0, at = 0, got8 = 0;
        bool fresh = true;                                            // the window is still zero from the start
        while (pt < ntiles) {
            const Part p = next_part(pt, pw);
            if (!fresh) zero_window();
            fresh = false;
            place_part(p);
            census8(p.bits, at);
            const uint32_t f8 = wv[0] >> 24;
            if (at) xff += straddle_ff(seg_tail, f8, ((uint32_t)lane + at) & 7u);      // (zeros in front of the segment: a byte that starts before it is not inside it)
            if (got8 < 8u) { seg_first8 |= (f8 >> got8) & 0xFFu; got8 += min(p.bits, 8u - got8); }
            seg_tail = tail7_join(seg_tail, tail7_of(wv, p.bits), p.bits);
            at += p.bits;
            advance(p, pt, pw);
        }
        zero_window();                                                // (the output pass places the parts again)
    }
    {
        uint32_t mine = lane < 8 ? xff : 0u;                          // lane p < 8 ends up with the count of phase p
        if (any_ff) {
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                const uint32_t t = (uint32_t)wave_sum_i32((int)ffc[p]);
                if (lane == p) mine += t;
            }
        }
        if (lane < 8) s_ffc[wave][lane] = mine;
        if (lane == 0) { s_bits[wave] = seg_bits; s_first8[wave] = seg_first8; s_last7[wave] = seg_tail; }
    }
    __syncthreads();                                                  // ---- barrier 1: every segment's numbers

    // ---- 3. wave 0: the workgroup's aggregate goes out; its offsets come from the workgroups in front (look-back) ------------
    if (wave == 0) {
        const __amdgpu_buffer_rsrc_t drsrc = __builtin_amdgcn_make_buffer_rsrc(a.desc + 4 * (size_t)((int)blockIdx.x - g), 0, a.wgs_per_image * 16, 0x00020000);
        uint32_t *const ffx_mine = a.desc_ffx + 8 * (size_t)blockIdx.x;
        const uint32_t P = (uint32_t)lane & 7u;
        // aggregate: lane P (0..7) walks the segments with the workgroup's first bit at phase P
        uint32_t agg = 0, wg_bits = 0, wg_tail0 = 0, wg_first8 = 0, got8 = 0;
        for (int j = 0; j < kStWaves; ++j) {
            const uint32_t bj = s_bits[j], ph = (P + wg_bits) & 7u;
            uint32_t c = s_ffc[j][ph];
            if (j > 0) c += straddle_ff(wg_tail0, s_first8[j], ph);      // (the byte straddling the workgroup's start is the look-back's)
            agg += bj ? c : 0u;
            if (got8 < 8u) { wg_first8 |= (s_first8[j] >> got8) & 0xFFu; got8 += min(bj, 8u - got8); }
            wg_tail0 = tail7_join(wg_tail0, s_last7[j], bj);
            wg_bits += bj;
        }
        const bool esc = __ballot(lane < 8 && agg > 255u) != 0ull;
        const uint32_t sat = min(agg, 255u);
        uint32_t d2 = 0, d3 = 0;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            d2 |= (uint32_t)__builtin_amdgcn_readlane((int)sat, p) << (8 * p);
            d3 |= (uint32_t)__builtin_amdgcn_readlane((int)sat, p + 4) << (8 * p);
        }
        if (esc) {                                                    // (rare: dense content) the counts in full, in memory before the granule that points at them
            if (lane < 8) __hip_atomic_store((gu32 *)(ffx_mine + lane), agg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        const auto store_gran = [&](uint32_t x0, uint32_t x1, uint32_t x2, uint32_t x3) {
            const u32x4 gv = {x0, x1, x2, x3};
            if (lane == 0) __builtin_amdgcn_raw_buffer_store_b128(gv, drsrc, g * 16, 0, 16 /*sc1: write-through*/);
        };
        const auto store_inclusive = [&](u64 bits, u64 ff, uint32_t last7) {
            const uint32_t x1 = (uint32_t)bits, x2 = (uint32_t)ff;
            store_gran((a.epoch << 18) | (2u << 16) | (last7 << 1), x1, x2,
                       (gran_check(a.epoch, x1, x2) << 24) | (((uint32_t)(bits >> 32) & 0xFFFu) << 12) | ((uint32_t)(ff >> 32) & 0xFFFu));
        };
        u64 B0 = 0, FF0 = 0;
        uint32_t prevtail = 0;
        if (g > 0) {
            store_gran((a.epoch << 18) | (1u << 16) | (wg_first8 << 8) | (wg_tail0 << 1) | (esc ? 1u : 0u), (gran_check(a.epoch, d2, d3) << 24) | wg_bits, d2, d3);
#ifndef JPEGAMD_ST_NO_LOOKBACK     // (timing-only builds skip the wait: wrong offsets, every write still inside the output)
            // Round r reads the predecessors hi - 1 .. hi - 256 (lane l, k = 0..3: hi - 1 - (4 l + k)), nearest first, until one of
            // them is INCLUSIVE; workgroup 0's always is.
            u32x4 v[4];
            const auto load_round = [&](int hi) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int idx = hi - 1 - (4 * lane + k);
                    if (idx >= 0) v[k] = __builtin_amdgcn_raw_buffer_load_b128(drsrc, idx * 16, 0, 16 /*sc1*/);
                    else { const u32x4 z = {(a.epoch << 18) | (2u << 16), 0u, 0u, gran_check(a.epoch, 0u, 0u) << 24}; v[k] = z; }   // in front of the picture: nothing
                }
            };
            uint32_t ok = 0, inc = 0;
            int fl = 64, fk = 4;                                    // the nearest inclusive granule of the round: lane fl, k = fk
            const auto classify = [&]() -> bool {                   // -> every granule up to the nearest inclusive one is valid
                ok = 0; inc = 0;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const bool valid = gran_valid(v[k], a.epoch);
                    ok |= valid ? 1u << k : 0u;
                    inc |= (valid && ((v[k][0] >> 16) & 3u) == 2u) ? 1u << k : 0u;
                }
                const u64 mi = __ballot(inc != 0u);
                fl = mi ? __ffsll((long long)mi) - 1 : 64;
                const uint32_t inc_fl = fl < 64 ? (uint32_t)__builtin_amdgcn_readlane((int)inc, fl) : 0u;
                fk = fl < 64 ? __ffs((int)inc_fl) - 1 : 4;
                const uint32_t need = lane < fl ? 15u : (lane == fl ? (2u << fk) - 1u : 0u);
                return __ballot((ok & need) != need) == 0ull;
            };
            const auto sum64 = [&](u64 x) -> u64 {                  // wave sum of values below 2^40
                return (u64)(uint32_t)wave_sum_i32((int)((uint32_t)x & 0xFFFFFu)) + ((u64)(uint32_t)wave_sum_i32((int)(uint32_t)(x >> 20)) << 20);
            };
            const auto agg_bits = [&](int k) -> uint32_t {          // bits of granule k when it is an aggregate in front of the inclusive one
                const bool nearer = lane < fl || (lane == fl && k < fk);
                return nearer ? v[k][1] & 0xFFFFFFu : 0u;
            };
            const auto incl_sel = [&](uint32_t x0, uint32_t x1, uint32_t x2, uint32_t x3) -> uint32_t {      // a word of the round's inclusive granule (fl < 64)
                const uint32_t x = fk == 0 ? x0 : fk == 1 ? x1 : fk == 2 ? x2 : x3;
                return (uint32_t)__builtin_amdgcn_readlane((int)x, fl);
            };
#define JPEGAMD_INCL_WORD(W) incl_sel(v[0][W], v[1][W], v[2][W], v[3][W])
            // phase 1: the bit offset.  (A sum: no order needed.)
            int hi = g, rounds = 0;
            uint32_t spins = 0;
            bool failed = false;
            for (;;) {
                load_round(hi);
                if (!classify()) {
                    if (++spins > kSpinLimit) { failed = true; break; }
                    __builtin_amdgcn_s_sleep(2);
                    continue;
                }
                B0 += sum64((u64)agg_bits(0) + agg_bits(1) + agg_bits(2) + agg_bits(3));
                ++rounds;
                if (fl < 64) { B0 += (u64)JPEGAMD_INCL_WORD(1) | ((u64)((JPEGAMD_INCL_WORD(3) >> 12) & 0xFFFu) << 32); break; }
                hi -= 256;
            }
            // phase 2: the 0xFF bytes in front, walking from the nearest workgroup again: with the bit offset known every
            // predecessor's byte phase is, and with it the count it contributes.  (A granule may have turned inclusive since phase
            // 1: the walk then simply ends earlier.)
            hi = g;
            uint32_t near_bits = 0;                                 // bits of the predecessors already walked (mod 2^32: phases need 3 bits)
            bool first_round = true;
            while (!failed) {
                if (!(first_round && rounds == 1)) {                // (one round: phase 1's registers are the snapshot)
                    load_round(hi);
                    if (!classify()) {
                        if (++spins > kSpinLimit) { failed = true; break; }
                        __builtin_amdgcn_s_sleep(2);
                        continue;
                    }
                }
                if (first_round) prevtail = ((uint32_t)__builtin_amdgcn_readlane((int)v[0][0], 0) >> 1) & 0x7Fu;
                first_round = false;
                // the workgroup in front of the round's farthest one, for lane 63's last granule
                uint32_t far_tail = 0;
                if (fl == 64 && hi - 257 >= 0) {
                    const u32x4 e = __builtin_amdgcn_raw_buffer_load_b128(drsrc, (hi - 257) * 16, 0, 16);
                    far_tail = (e[0] >> 1) & 0x7Fu;              // (valid: every workgroup publishes its aggregate before it looks back, and the round was complete)
                    if (!gran_valid(e, a.epoch)) { if (++spins > kSpinLimit) { failed = true; break; } __builtin_amdgcn_s_sleep(2); continue; }
                }
                uint32_t bk[4], cum = 0;
#pragma unroll
                for (int k = 0; k < 4; ++k) { bk[k] = agg_bits(k); cum += bk[k]; }
                const uint32_t lane_incl = wave_incl_scan_u32(cum);
                const uint32_t round_bits = (uint32_t)__builtin_amdgcn_readlane((int)lane_incl, 63);
                uint32_t before = near_bits + lane_incl - cum;      // bits of the predecessors nearer than my granule 0
                const uint32_t next_tail0 = (uint32_t)__builtin_amdgcn_ds_bpermute(((lane + 1) & 63) * 4, (int)((v[0][0] >> 1) & 0x7Fu));   // last 7 bits of lane + 1's granule 0
                const bool any_esc = __ballot(((v[0][0] | v[1][0] | v[2][0] | v[3][0]) & 1u) != 0u) != 0ull;
                u64 contrib = 0;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const bool nearer = lane < fl || (lane == fl && k < fk);
                    before += bk[k];
                    const uint32_t ph = ((uint32_t)B0 - before) & 7u;                       // byte phase of this workgroup's first bit
                    uint32_t c = ((ph & 4u ? v[k][3] : v[k][2]) >> (8u * (ph & 3u))) & 0xFFu;
                    if (any_esc && nearer && (v[k][0] & 1u)) {                              // saturated: the count in full
                        const int idx = hi - 1 - (4 * lane + k);
                        c = __hip_atomic_load((gu32 *)(a.desc_ffx + 8 * (size_t)((int)blockIdx.x - g + idx) + ph), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    const uint32_t tail_far = k < 3 ? (v[k < 3 ? k + 1 : 3][0] >> 1) & 0x7Fu : (lane == 63 ? far_tail : next_tail0);
                    c += straddle_ff(tail_far, (v[k][0] >> 8) & 0xFFu, ph);
                    contrib += nearer ? c : 0u;
                }
                FF0 += sum64(contrib);
                if (fl < 64) { FF0 += (u64)JPEGAMD_INCL_WORD(2) | ((u64)(JPEGAMD_INCL_WORD(3) & 0xFFFu) << 32); break; }
                near_bits += round_bits;
                hi -= 256;
            }
            if (failed && lane == 0) atomicOr(a.status, 4u);
#undef JPEGAMD_INCL_WORD
#endif
        }
        // every segment's offsets (lane j: segment j of the workgroup), and the inclusive granule for the workgroups that follow
        uint32_t pre = 0, tb = prevtail, bj = 0, cj = 0;
        {
            uint32_t run = 0;
            for (int j = 0; j < kStWaves; ++j) {                    // (uniform walk; lane j keeps what belongs to segment j)
                const uint32_t b = s_bits[j];
                if (lane == j) { pre = run; bj = b; }
                run += b;
            }
            for (int j = 0; j < lane && j < kStWaves; ++j) tb = tail7_join(tb, s_last7[j], s_bits[j]);
        }
        if (lane < kStWaves && bj) {
            const uint32_t ph = (uint32_t)((B0 + pre) & 7ull);
            cj = s_ffc[lane][ph] + straddle_ff(tb, s_first8[lane], ph);
        }
        const uint32_t cincl = half_incl_scan_dpp(lane < kStWaves ? cj : 0u);
        const uint32_t wg_ff = (uint32_t)__builtin_amdgcn_readlane((int)cincl, kStWaves - 1);
        if (lane < kStWaves) {
            s_b0s[lane] = B0 + pre;
            s_ffoff[lane] = FF0 + (cincl - cj);
            s_myff[lane] = cj;
            s_tailin[lane] = tb;
        }
        uint32_t tail_end_wg = prevtail;
        for (int j = 0; j < kStWaves; ++j) tail_end_wg = tail7_join(tail_end_wg, s_last7[j], s_bits[j]);
        store_inclusive(B0 + wg_bits, FF0 + wg_ff, tail_end_wg);
    }
    __syncthreads();                                                  // ---- barrier 2: every segment's offsets
    if (!have) return;                                                             // no workgroup-wide synchronisation below

    // ---- 4. the owned bytes go out ------------------------------------------------------------------------------------------
    const u64 b0s = s_b0s[wave], ffoff = s_ffoff[wave];             // stream bit offset of the segment; stuffed bytes in front of it
    const uint32_t my_ff = s_myff[wave], tail_in = s_tailin[wave];   // 0xFF bytes it owns; the 7 stream bits in front of it
    bool overflow = false;
    uint8_t *const strip = reinterpret_cast<uint8_t *>(&s_strip[wave][0]);
    u32x4 *const strip16 = reinterpret_cast<u32x4 *>(&s_strip[wave][0]);
    // The `nown` owned bytes of what the window holds go to out + base; part_ff of them are 0xFF.
    const auto emit = [&](uint32_t lead, uint32_t nown, u64 base, uint32_t part_ff) {
        if (nown == 0u) return;
        if (base + nown + part_ff > out_capacity) { overflow = true; return; }
        uint8_t *dst = out + base;
        if (part_ff == 0u) {
            // No 0xFF among the owned bytes: they land contiguously -- the middle as ALIGNED 16-byte pieces, lane k building output
            // dwords 4 k .. 4 k + 3 from five window words with four funnel shifts -- and at most 15 + 15 bytes at the two ends.
            const uint32_t hd = min((16u - (uint32_t)((uintptr_t)dst & 15u)) & 15u, nown);
            const uint32_t nq = (nown - hd) >> 4;
            const int q = (int)(8u * hd) - (int)lead;                               // bit offset of piece 0 in the window: -7 .. 120
            const int wq = q >> 5;                                                  // its first window word (-1: the borrowed bits)
            const uint32_t sh = (uint32_t)q & 31u;
            for (uint32_t k = (uint32_t)lane; k < nq; k += 64) {
                const int wi = wq + 4 * (int)k;
                const uint32_t v0 = wv[wi], v1 = wv[wi + 1], v2 = wv[wi + 2], v3 = wv[wi + 3], v4 = wv[wi + 4];
                u32x4 o;
                o[0] = sh ? __builtin_amdgcn_alignbit(v0, v1, 32u - sh) : v0;
                o[1] = sh ? __builtin_amdgcn_alignbit(v1, v2, 32u - sh) : v1;
                o[2] = sh ? __builtin_amdgcn_alignbit(v2, v3, 32u - sh) : v2;
                o[3] = sh ? __builtin_amdgcn_alignbit(v3, v4, 32u - sh) : v3;
#pragma unroll
                for (int c = 0; c < 4; ++c) o[c] = __builtin_bswap32(o[c]);
                *reinterpret_cast<u32x4 *>(dst + hd + 16u * k) = o;
            }
            const uint32_t ntail = nown - hd - 16u * nq;
            if ((uint32_t)lane < hd + ntail) {                                      // the bytes at the two ends, one lane each
                const uint32_t r = (uint32_t)lane < hd ? (uint32_t)lane : 16u * nq + (uint32_t)lane;
                const int bp = 8 * (int)r - (int)lead;                              // first bit of byte r in the window: >= -7
                const int wi = bp >> 5;
                const uint32_t bs = (uint32_t)bp & 31u;
                const u64 win = ((u64)wv[wi] << 32) | wv[wi + 1];
                dst[r] = (uint8_t)((win << bs) >> 56);
            }
            return;
        }
        // 0xFF among the owned bytes.  Sixteen owned bytes per lane and pass; a wave prefix sum over the lanes' 0xFF counts says where
        // each lane's bytes go; the lane writes them one by one into a ZEROED staging strip in LDS -- the stuffed 0x00 behind an
        // 0xFF (huffman.c:29-31) is simply left out -- and the strip leaves as aligned 16-byte pieces.  Strip offset == output
        // address mod 16: what does not fill a piece stays as the head of the next pass.
        const u32x4 zero4 = {0u, 0u, 0u, 0u};
        for (int p = lane; p < kStStripPieces; p += 64) strip16[p] = zero4;
        const uint32_t fill0 = (uint32_t)((uintptr_t)dst & 15u);                    // the bytes in front of it in the first piece belong to earlier segments
        uint32_t fill = fill0;
        uint8_t *gdst = dst - fill0;                                                // where strip byte 0 goes: 16-byte aligned
        bool first = true;
        for (uint32_t done = 0; done < nown; done += 1024u) {
            const uint32_t idx = (done >> 4) + (uint32_t)lane;
            uint32_t v[4] = {0u, 0u, 0u, 0u};
            if (16u * idx < nown) group_words(idx, lead, v);
            const uint32_t left = nown - done;                                      // owned bytes from this pass on
            if (left < 1024u) mask_tail(v, (int)left - 16 * lane);                  // (uniform) the last pass: bytes beyond the owned ones count as zeros
            uint32_t m[4], c = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                m[k] = ((v[k] & 0x7F7F7F7Fu) + 0x01010101u) & v[k] & 0x80808080u;   // bit 7 of a byte: the byte is 0xFF
                c += (uint32_t)__popc(m[k]);
            }
            const uint32_t incl = wave_incl_scan_u32(c);
            const uint32_t nfill = fill + min(left, 1024u) + (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);     // <= 15 + 2048
            uint32_t at = fill + 16u * (uint32_t)lane + incl - c;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                strip[at] = (uint8_t)(v[k] >> 24); at += 1u + (m[k] >> 31);
                strip[at] = (uint8_t)(v[k] >> 16); at += 1u + ((m[k] >> 23) & 1u);
                strip[at] = (uint8_t)(v[k] >> 8);  at += 1u + ((m[k] >> 15) & 1u);
                strip[at] = (uint8_t)v[k];         at += 1u + ((m[k] >> 7) & 1u);
            }
            const uint32_t npieces = nfill >> 4;                                    // complete pieces: <= 128
            const u32x4 rest = strip16[npieces];                                    // (one address for the wave) the incomplete piece
            if (first && fill0 && (uint32_t)lane >= fill0 && (uint32_t)lane < min(16u, nfill)) gdst[lane] = strip[lane];   // the first piece: its own bytes one by one
            for (uint32_t p = (uint32_t)lane; p < npieces; p += 64u) {
                const u32x4 piece = strip16[p];
                if (!(first && fill0 && p == 0u)) *reinterpret_cast<u32x4 *>(gdst + 16u * p) = piece;
                strip16[p] = zero4;
            }
            if (lane == 0) {
                strip16[npieces] = zero4;
                strip16[0] = rest;
            }
            gdst += 16u * npieces;
            fill = nfill & 15u;
            first = first && npieces == 0u;
        }
        if ((uint32_t)lane < fill && !(first && (uint32_t)lane < fill0)) gdst[lane] = strip[lane];     // what is left of the last piece
    };

    const u64 base0 = (u64)a.prefix_len + (b0s >> 3) + ffoff;                        // where the segment's first owned byte goes
    uint32_t tail_end = tail_in;                                                   // the 7 stream bits in front of the segment's end
    if (single) {
        const uint32_t lead = (uint32_t)(b0s & 7ull);
        if (lane == 0) wv[-1] = tail_in;
        emit(lead, (uint32_t)(((b0s + seg_bits) >> 3) - (b0s >> 3)), base0, my_ff);
        tail_end = tail7_join(tail_in, seg_tail, seg_bits);
    } else {
        int pt = 0; uint32_t pw = 0, ffrun = 0;
        u64 pos = b0s;
        while (pt < ntiles) {
            const Part p = next_part(pt, pw);
            zero_window();
            place_part(p);
            if (lane == 0) wv[-1] = tail_end;
            const uint32_t lead = (uint32_t)(pos & 7ull), nown = (uint32_t)(((pos + p.bits) >> 3) - (pos >> 3));
            const uint32_t pff = census(lead, nown);                               // (counted again: where the next part's bytes go)
            emit(lead, nown, base0 + ((pos >> 3) - (b0s >> 3)) + ffrun, pff);
            ffrun += pff;
            tail_end = tail7_join(tail_end, tail7_of(wv, p.bits), p.bits);
            pos += p.bits;
            advance(p, pt, pw);
        }
    }
    if (__any(overflow) && lane == 0) atomicOr(a.status, 1u);

    if (s == a.num_segs - 1 && lane == 0) {
        const u64 b1 = b0s + seg_bits;
        u64 end = (u64)a.prefix_len + (b1 >> 3) + ffoff + my_ff;
        const uint32_t rem = (uint32_t)(b1 & 7ull);
        bool ok = true;
        if (rem) {                                              // zero-padded flush (huffman.c:65-81)
            if (end < out_capacity) out[end] = (uint8_t)((tail_end & ((1u << rem) - 1u)) << (8u - rem)); else ok = false;
            ++end;
        }
        if (a.write_eoi) {                                      // jpeg_handler.c:113-117
            if (end + 2 <= out_capacity) { out[end] = 0xFF; out[end + 1] = 0xD9; } else ok = false;
            end += 2;
        }
        if (!ok) atomicOr(a.status, 1u);
        *a.out_size[image] = end;
        if (image == a.batch - 1) {
            a.stats->out_size = end;
            a.stats->total_bits = b1;
            a.stats->total_ff = ffoff + my_ff;
        }
    }
}

// Segments of 16 tiles, eight per workgroup: an 8192^2 picture is 2 048 segments = 256 workgroups, so every look-back is ONE round of
// four granules per lane.  (Measured: 8-tile segments with sixteen waves per workgroup 15.7 us per single 8192^2 picture against
// 13.8; sixteen waves per workgroup at eight pictures per launch 79 us against 57 -- profiles/r04_notes_experiments.txt.)
#ifndef JPEGAMD_ST_WAVES
#define JPEGAMD_ST_WAVES 8
#endif
constexpr int kStWaves = JPEGAMD_ST_WAVES;
int launch_stitch(const StitchArgs &a, void *stream, void *const *ev) {
    if (a.wgs_per_image <= 0 || a.batch <= 0) return 0;
    if (a.seg_tiles != kSegTilesBatch) return (int)hipErrorInvalidValue;
    const dim3 grid((unsigned)(a.wgs_per_image * a.batch));
    if (ev) hipExtLaunchKernelGGL((k_stitch<kSegTilesBatch, kStWaves>), grid, dim3(64 * kStWaves), 0, (hipStream_t)stream, (hipEvent_t)ev[0], (hipEvent_t)ev[1], 0, a);
    else hipLaunchKernelGGL((k_stitch<kSegTilesBatch, kStWaves>), grid, dim3(64 * kStWaves), 0, (hipStream_t)stream, a);
    return (int)hipGetLastError();
}
int stitch_workgroups(int num_segs) { return (num_segs + kStWaves - 1) / kStWaves; }

}  // namespace jpegamd
