// jpegamd_finalize.hip -- everything after the entropy coder, in ONE launch: global bit offsets of the segments, 0xFF
// stuffing offsets, byte stitching and stuffing (huffman.c:26-62), zero-padded flush (huffman.c:65-81), container
// (jpeg_handler.c:220-262).  No inter-workgroup waits.
//
// An output byte is OWNED by the segment that holds its last bit, so every output byte is written by exactly one lane
// with a plain store: no atomics, no pre-zeroing.  Where a segment's bytes go depends on (a) the bits of all earlier
// segments -- a direct sum of their bit counts, final when this kernel starts -- and (b) the 0xFF bytes owned by all
// earlier segments, because each is followed by a stuffed 0x00.  (b) depends on every earlier segment's byte phase, which
// is why round 1 needed a separate counting kernel between two dependent launches.  Now k_segment_merge leaves, per segment,
// the number of 0xFF bytes lying wholly inside it for each of the 8 phases, plus its first 8 and last 7 bits (what the byte
// straddling two segments is made of); a workgroup scans the earlier segments' bit counts (phase of each), picks the matching
// counts, adds the straddling bytes, and has both offsets -- chain-free, from per-segment numbers alone.
//
//   workgroup = 16 consecutive segments, one per wave;  grid = ceil(num_segs / 16)
#include <cstdlib>
#include <cstring>

#include <hip/hip_ext.h>
#include "jpegamd_device.h"

namespace jpegamd {

constexpr int kFinWaves = 16;                                  // segments per workgroup
constexpr int kFinStagePieces = 130;                           // 16-byte pieces of a wave's staging strip

__device__ __forceinline__ uint32_t fin_bits_at(const uint32_t *__restrict__ w, uint32_t pos, int nbits /*1..8*/) {
    const uint32_t i = pos >> 5, sh = pos & 31u;
    const uint64_t win = ((uint64_t)w[i] << 32) | w[i + 1];
    return (uint32_t)((win << sh) >> (64 - nbits));
}

// The `need` (1..7) bits that precede segment `s` in the stream (s may equal num_segs): only for predecessors shorter
// than a byte, and for the final flush.
template <class View>
__device__ __forceinline__ uint32_t fin_tail_bits(const View &a, int s, int need) {
    uint32_t val = 0;
    int got = 0;
    for (int sp = s - 1; got < need && sp >= 0; --sp) {
        const uint32_t tp = a.seg.bits[sp];
        const int take = min(need - got, (int)tp);
        if (take > 0) {
            val |= fin_bits_at(a.seg.words + (size_t)sp * a.seg.words_stride, tp - (uint32_t)take, take) << got;
            got += take;
        }
    }
    return val;
}

// Owned 0xFF bytes of segment t given its byte phase p (= bit offset & 7): those wholly inside it (counted by k_segment_merge)
// plus the byte that straddles its start -- 0xFF iff the last p bits in front of it and its first 8 - p bits are all ones.
// edge = (first 8 bits << 8) | last 7 bits of a segment's own string.  A segment shorter than 8 bits is "00 1010" (one
// flat block): it has no leading one, and its tail ends in 0, so neither side can complete an 0xFF across it.
__device__ __forceinline__ uint32_t fin_owned_ff(uint4 ffin /*the segment's 8 counts, 16 bits each*/, int t, uint32_t p, uint32_t edge_t, uint32_t edge_prev) {
    const uint32_t lo = (p & 4u) ? ffin.z : ffin.x, hi = (p & 4u) ? ffin.w : ffin.y;
    const uint32_t w = (p & 2u) ? hi : lo;
    uint32_t c = (p & 1u) ? w >> 16 : w & 0xFFFFu;
    if (p && t > 0) {
        const uint32_t tail_ones = (uint32_t)__builtin_ctz(~(edge_prev & 0x7Fu));             // 0..7
        const uint32_t lead_ones = (uint32_t)__clz(~((edge_t >> 8) << 24));                    // 0..8
        c += (tail_ones >= p && lead_ones >= 8u - p) ? 1u : 0u;
    }
    return c;
}

// The same for a GROUP's aggregate (eight 32-bit counts; the bytes straddling the group's inner boundaries are in the counts).
__device__ __forceinline__ uint32_t fin_owned_ff_group(uint4 lo /*phases 0..3*/, uint4 hi /*4..7*/, int t, uint32_t p, uint32_t edge_t, uint32_t edge_prev) {
    const uint4 h4 = (p & 4u) ? hi : lo;
    const uint32_t a2 = (p & 2u) ? h4.z : h4.x, b2 = (p & 2u) ? h4.w : h4.y;
    uint32_t c = (p & 1u) ? b2 : a2;
    if (p && t > 0) {
        const uint32_t tail_ones = (uint32_t)__builtin_ctz(~(edge_prev & 0x7Fu));
        const uint32_t lead_ones = (uint32_t)__clz(~((edge_t >> 8) << 24));
        c += (tail_ones >= p && lead_ones >= 8u - p) ? 1u : 0u;
    }
    return c;
}

// bit 8k set <=> byte k (little-endian numbering) of w is 0xFF
__device__ __forceinline__ uint32_t fin_ff_mask(uint32_t w) {
    uint32_t t = w & (w >> 4);
    t &= t >> 2;
    t &= t >> 1;
    return t & 0x01010101u;
}

// One image of the launch as the kernel body sees it: its own segment arrays (indices from 0), output and size word.
struct FinalizeView {
    SegArrays seg;
    int32_t num_segs;
    uint8_t *out;
    uint64_t out_capacity;
    uint64_t *out_size;
    ScanStats *stats;
    const uint8_t *prefix;
    int32_t prefix_len, write_eoi;
    bool last_image;                // its totals go to the per-call record
};

__global__ __launch_bounds__(64 * kFinWaves) void k_finalize(const FinalizeArgs args) {
    __shared__ uint32_t s_wbits[2][kFinWaves], s_wff[2][kFinWaves];
    __shared__ __attribute__((aligned(16))) uint32_t s_stage[kFinWaves][4 * kFinStagePieces];   // per wave: the 0xFF branch's staging strip (15 + 2 x 1024 bytes at most)
    static_assert(kFinWaves == 16, "the waves' totals are scanned in one DPP row");
    const int lane = lane_id(), wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), tid = (int)threadIdx.x;
    // A batch: workgroup -> (image, chunk inside the image); everything below works on that image alone (bit offsets, byte
    // phases and stuffing counts restart with every image).
    const int image = args.batch > 1 ? (int)blockIdx.x / args.num_chunks : 0;
    const int g = (int)blockIdx.x - image * args.num_chunks;
    FinalizeView a;
    {
        const size_t s0 = (size_t)image * (size_t)args.num_segs;
        a.seg.words = args.seg.words + s0 * args.seg.words_stride; a.seg.words_stride = args.seg.words_stride;
        a.seg.bits = args.seg.bits + s0; a.seg.syms = args.seg.syms + s0; a.seg.exact = args.seg.exact + s0;
        a.seg.edge = args.seg.edge + s0; a.seg.ffin = args.seg.ffin + s0 * 8;
        a.seg.grp_bits = args.seg.grp_bits + s0 / kSegGroup; a.seg.grp_ff = args.seg.grp_ff + (s0 / kSegGroup) * 8;   // (use_groups: s0 is a multiple)
        a.num_segs = args.num_segs;
        a.out = args.out[image]; a.out_capacity = args.out_capacity; a.out_size = args.out_size[image];
        a.stats = args.stats; a.prefix = args.prefix; a.prefix_len = args.prefix_len; a.write_eoi = args.write_eoi;
        a.last_image = image == args.batch - 1;
    }
    const int s = g * kFinWaves + wave;
    const bool have = s < a.num_segs;

    if (g == 0 && a.prefix_len > 0)                             // JFIF prefix (jpeg_handler.c:220-233)
        for (int i = tid; i < a.prefix_len; i += 64 * kFinWaves)
            if ((uint64_t)i < a.out_capacity) a.out[i] = a.prefix[i];

    // Every load whose address does not depend on data is issued up front (this kernel is one chain of dependent round trips
    // otherwise): the chunk's own per-segment numbers, and below the predecessors' -- all 8 phase counts of a segment are
    // fetched (16 bytes) and the right one picked in registers once the phase is known.
    const int sp = g * kFinWaves + (lane & 15);
    const bool in = lane < kFinWaves && sp < a.num_segs;
    const uint32_t vb = in ? a.seg.bits[sp] : 0u;
    const uint32_t ve = in ? a.seg.edge[sp] : 0u;
    const uint32_t ve_prev = (in && sp > 0) ? a.seg.edge[sp - 1] : 0u;
    const uint4 vffin = in ? *reinterpret_cast<const uint4 *>(a.seg.ffin + (size_t)sp * 8) : make_uint4(0u, 0u, 0u, 0u);
    const uint32_t pbits = (have && s > 0) ? a.seg.bits[s - 1] : 0u;

    // ---- 1. everything in front of this chunk: bits (64-bit) and owned 0xFF bytes of the segments [0, 16 g) -------------
    // A block-wide exclusive scan of the bit counts gives every entry its byte phase (the sum is needed mod 8 only, so 32-bit
    // wrap-around is harmless).
#ifdef JPEGAMD_FIN_SKIP_SCAN       // timing-only build: no scan over the predecessors (wrong offsets, every write still inside the output)
    const int n_before = 0;
#else
    const int n_before = g * kFinWaves;
#endif
    unsigned long long chunk_b0 = 0, chunk_ff0 = 0;
    const bool use_groups = args.use_groups != 0;
    int par = 0;
    // A thread takes kSegGroup consecutive segments: as ONE entry (k_segment_merge's aggregate of its workgroup) or one by one.
    // A wave whose entries all lie beyond n_before (half the waves on average) only joins the two barriers and reads the
    // round's totals.
    static_assert(kSegGroup == 4, "a thread's four segments are one group of k_segment_merge");
    const int per_thread = kSegGroup;
    for (int base = 0; base < n_before; base += per_thread * 64 * kFinWaves) {
        const int i = base + per_thread * tid;
        const bool wave_has = base + per_thread * 64 * wave < n_before;
        uint4 b = make_uint4(0u, 0u, 0u, 0u), e = make_uint4(0u, 0u, 0u, 0u), f0 = b, f1 = b, f2 = b, f3 = b;
        uint32_t eprev = 0, tot = 0, incl = 0;
        if (wave_has) {
            if (i < n_before) {                                  // n_before is a multiple of 16: a thread's segments are all in front or none
                if (use_groups) {
                    b.x = a.seg.grp_bits[i / kSegGroup];
                    f0 = *reinterpret_cast<const uint4 *>(a.seg.grp_ff + (size_t)(i / kSegGroup) * 8);
                    f1 = *reinterpret_cast<const uint4 *>(a.seg.grp_ff + (size_t)(i / kSegGroup) * 8 + 4);
                    e.x = a.seg.edge[i];
                } else {
                    b = *reinterpret_cast<const uint4 *>(a.seg.bits + i);
                    e = *reinterpret_cast<const uint4 *>(a.seg.edge + i);
                    const uint4 *fp = reinterpret_cast<const uint4 *>(a.seg.ffin + (size_t)i * 8);
                    f0 = fp[0]; f1 = fp[1]; f2 = fp[2]; f3 = fp[3];
                }
                if (i > 0) eprev = a.seg.edge[i - 1];
            }
            tot = b.x + b.y + b.z + b.w;
            incl = wave_incl_scan_u32(tot);
        }
        uint32_t *wb = s_wbits[par], *wf = s_wff[par];            // double-buffered by round: no barrier before the next round's writes
        if (lane == 63) wb[wave] = incl;
        __syncthreads();
        // the 16 waves' totals: every wave scans them in its own lanes 0..15 (one LDS read + a row scan, instead of 16 reads)
        const uint32_t wt = lane < kFinWaves ? wb[lane] : 0u;
        const uint32_t wincl = half_incl_scan_dpp(wt);            // kFinWaves == 16: one DPP row
        const uint32_t round_bits = (uint32_t)__builtin_amdgcn_readlane((int)wincl, kFinWaves - 1);
        uint32_t wff = 0;
        if (wave_has) {
            const uint32_t woff = (uint32_t)__builtin_amdgcn_readlane((int)(wincl - wt), wave);
            const uint32_t x0 = (uint32_t)chunk_b0 + woff + incl - tot;             // bit offset of segment i, mod 2^32
            uint32_t ff = 0;
            if (i < n_before) {
                if (use_groups)
                    ff = fin_owned_ff_group(f0, f1, i, x0 & 7u, e.x, eprev);
                else
                    ff = fin_owned_ff(f0, i, x0 & 7u, e.x, eprev) + fin_owned_ff(f1, i + 1, (x0 + b.x) & 7u, e.y, e.x) + fin_owned_ff(f2, i + 2, (x0 + b.x + b.y) & 7u, e.z, e.y) +
                          fin_owned_ff(f3, i + 3, (x0 + b.x + b.y + b.z) & 7u, e.w, e.z);
            }
            wff = (uint32_t)wave_sum_i32((int)ff);
        }
        if (lane == 0) wf[wave] = wff;
        __syncthreads();
        const uint32_t ft = lane < kFinWaves ? wf[lane] : 0u;
        const uint32_t round_ff = (uint32_t)__builtin_amdgcn_readlane((int)half_incl_scan_dpp(ft), kFinWaves - 1);
        chunk_b0 += round_bits;
        chunk_ff0 += round_ff;
        par ^= 1;
    }
    if (!have) return;                                            // no workgroup-wide synchronisation below

    // ---- 2. inside the chunk: lanes 0..15 of every wave hold the chunk's 16 segments ---------------------------------------
    const uint32_t ib = wave_incl_scan_u32(vb);
    const uint32_t off_in = ib - vb;                                                // bits of the chunk in front of segment sp
    const uint32_t vf = in ? fin_owned_ff(vffin, sp, ((uint32_t)chunk_b0 + off_in) & 7u, ve, ve_prev) : 0u;
    const uint32_t iff = wave_incl_scan_u32(vf);
    const uint32_t my_bits = (uint32_t)__builtin_amdgcn_readlane((int)vb, wave);
    const uint32_t my_ff = (uint32_t)__builtin_amdgcn_readlane((int)vf, wave);
    const uint32_t ff_in = (uint32_t)__builtin_amdgcn_readlane((int)(iff - vf), wave);
    const uint32_t prev_edge = (uint32_t)__builtin_amdgcn_readlane((int)ve_prev, wave);

    const uint32_t *words = a.seg.words + (size_t)s * a.seg.words_stride;
    const unsigned long long b0 = chunk_b0 + (uint32_t)__builtin_amdgcn_readlane((int)off_in, wave);
    const unsigned long long b1 = b0 + my_bits;
    const uint32_t nown = (uint32_t)((b1 >> 3) - (b0 >> 3));                       // bytes whose last bit lies in this segment
    const uint32_t lead = (uint32_t)(b0 & 7u);
    uint32_t leadbits = 0;                                                          // the `lead` bits in front of the segment
    if (lead && s > 0) {
        leadbits = pbits >= 7u ? (prev_edge & ((1u << lead) - 1u)) : fin_tail_bits(a, s, (int)lead);
    }

    // ---- 3. the segment's owned bytes ---------------------------------------------------------------------------------------
    // The segment prefixed by its borrowed bits is a byte string; any 32 consecutive bits of it are one funnel shift of two
    // adjacent words.  0x00 goes behind every 0xFF (huffman.c:29-31).
    const uint64_t base = (uint64_t)a.prefix_len + (b0 >> 3) + chunk_ff0 + ff_in;
    uint32_t running = 0;
    bool overflow = false;
#ifdef JPEGAMD_FIN_NOFF          // timing-only build: every segment takes the no-0xFF path (wrong bytes behind the first 0xFF)
    if (true) {
#else
    if (my_ff == 0u) {
#endif
        // No 0xFF among the owned bytes (nearly every segment): the bytes land contiguously, so the middle goes out as ALIGNED
        // 16-byte pieces -- lane k builds output dwords 4 k .. 4 k + 3 straight from the bit string: five string words, four
        // funnel shifts -- and at most 15 + 15 bytes at the two ends.  (One dword per lane cost 45 instructions per 64 dwords.)
        if (base + nown <= a.out_capacity) {
            uint8_t *dst = a.out + base;
            const uint32_t hd = min((16u - (uint32_t)((uintptr_t)dst & 15u)) & 15u, nown);
            const uint32_t nq = (nown - hd) >> 4;
            const int q = (int)(8u * hd) - (int)lead;                               // bit offset of piece 0 in the segment: -7 .. 120
            const int wq = q >> 5;                                                  // its first string word (-1: the borrowed bits)
            const uint32_t sh = (uint32_t)q & 31u;
            typedef uint32_t u32x4a __attribute__((ext_vector_type(4), aligned(4)));
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            for (uint32_t k = (uint32_t)lane; k < nq; k += 64) {
                const int wi = wq + 4 * (int)k, wl = max(wi, 0);
                const u32x4a x = *reinterpret_cast<const u32x4a *>(words + wl);
                const uint32_t x4 = words[wl + 4];
                const bool borrow = wi < 0;
                const uint32_t v0 = borrow ? leadbits : x[0], v1 = borrow ? x[0] : x[1], v2 = borrow ? x[1] : x[2], v3 = borrow ? x[2] : x[3],
                               v4 = borrow ? x[3] : x4;
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
                dst[r] = (r == 0u && lead) ? (uint8_t)((leadbits << (8u - lead)) | fin_bits_at(words, 0u, 8 - (int)lead))
                                           : (uint8_t)fin_bits_at(words, 8u * r - lead, 8);
            }
        } else {
            overflow = true;
        }
    } else if (base + nown + my_ff > a.out_capacity) {
        overflow = true;
        running = my_ff;                                            // (the would-be size stays exact: callers size their second attempt from it)
    } else {
        // 0xFF among the owned bytes (one byte in ~300 of photo-like content: nearly every segment of 16 tiles has some).  Sixteen
        // owned bytes per lane and pass, built from the bit string as in the branch above; a wave prefix sum over the lanes' 0xFF
        // counts says where each lane's bytes go; the lane writes them one by one into a ZEROED staging strip in LDS -- the stuffed
        // 0x00 behind an 0xFF (huffman.c:29-31) is simply left out -- and the strip leaves as aligned 16-byte pieces.  Strip
        // offset == output address mod 16: what does not fill a piece stays as the head of the next pass.  (Round 2 stored every
        // byte to HBM by itself, four bytes per lane and pass, behind 64-bit address arithmetic and a branch per byte: ~600
        // instructions per KiB against ~120, and k_finalize at 2.7 x the duration the no-0xFF branch would have.)
        typedef uint32_t u32x4a __attribute__((ext_vector_type(4), aligned(4)));
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        uint8_t *const stage = reinterpret_cast<uint8_t *>(&s_stage[wave][0]);
        u32x4 *const stage16 = reinterpret_cast<u32x4 *>(&s_stage[wave][0]);
        const u32x4 zero4 = {0u, 0u, 0u, 0u};
        for (int p = lane; p < kFinStagePieces; p += 64) stage16[p] = zero4;
        const uint32_t fill0 = (uint32_t)((uintptr_t)(a.out + base) & 15u);         // the bytes in front of it in the first piece belong to earlier segments
        uint32_t fill = fill0;
        uint8_t *gdst = a.out + base - fill0;                                       // where strip byte 0 goes: 16-byte aligned
        bool first = true;
        const uint32_t sh = (32u - lead) & 31u;                                     // a lane's 16 bytes start `lead` bits in front of a word boundary
        const auto request = [&](uint32_t idx, u32x4a &x, uint32_t &x4) {           // string words 4 idx - (lead ? 1 : 0) .. + 4 of 16-byte group idx
            if (16u * idx < nown) {
                const int wi = 4 * (int)idx - (lead ? 1 : 0), wl = max(wi, 0);
                x = *reinterpret_cast<const u32x4a *>(words + wl);
                x4 = words[wl + 4];
            }
        };
        u32x4a nx = {0u, 0u, 0u, 0u};
        uint32_t nx4 = 0;
        request((uint32_t)lane, nx, nx4);
        for (uint32_t done = 0; done < nown; done += 1024u) {                       // (the next pass's words are requested ahead of this pass's work)
            const uint32_t idx = (done >> 4) + (uint32_t)lane;
            const u32x4a x = nx;
            const uint32_t x4 = nx4;
            request(idx + 64u, nx, nx4);
            const bool borrow = idx == 0u && lead != 0u;
            const uint32_t v0 = borrow ? leadbits : x[0], v1 = borrow ? x[0] : x[1], v2 = borrow ? x[1] : x[2], v3 = borrow ? x[2] : x[3],
                           v4 = borrow ? x[3] : x4;
            uint32_t v[4];
            v[0] = sh ? __builtin_amdgcn_alignbit(v0, v1, 32u - sh) : v0;
            v[1] = sh ? __builtin_amdgcn_alignbit(v1, v2, 32u - sh) : v1;
            v[2] = sh ? __builtin_amdgcn_alignbit(v2, v3, 32u - sh) : v2;
            v[3] = sh ? __builtin_amdgcn_alignbit(v3, v4, 32u - sh) : v3;
            const uint32_t left = nown - done;                                      // owned bytes from this pass on
            if (left < 1024u) {                                                     // (uniform) the last pass: bytes beyond the owned ones count as zeros
                const int nvl = (int)left - 16 * lane;                              // this lane's owned bytes: <= 0 none, >= 16 all
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int kb = min(max(nvl - 4 * k, 0), 4);
                    v[k] = kb ? v[k] & (0xFFFFFFFFu << (8 * (4 - kb))) : 0u;
                }
            }
            uint32_t m[4], c = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                m[k] = ((v[k] & 0x7F7F7F7Fu) + 0x01010101u) & v[k] & 0x80808080u;   // bit 7 of a byte: the byte is 0xFF (0x7F + 1 carries into bit 7 only)
                c += (uint32_t)__popc(m[k]);
            }
            const uint32_t incl = wave_incl_scan_u32(c);
            const uint32_t nfill = fill + min(left, 1024u) + (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);     // <= 15 + 2048
            // byte t of the lane goes to (strip offset of the lane) + t + (0xFF bytes among its bytes in front of t); bytes beyond the owned
            // ones are zeros that land beyond the data, where the strip is zero anyway
            uint32_t at = fill + 16u * (uint32_t)lane + incl - c;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                stage[at] = (uint8_t)(v[k] >> 24); at += 1u + (m[k] >> 31);
                stage[at] = (uint8_t)(v[k] >> 16); at += 1u + ((m[k] >> 23) & 1u);
                stage[at] = (uint8_t)(v[k] >> 8);  at += 1u + ((m[k] >> 15) & 1u);
                stage[at] = (uint8_t)v[k];         at += 1u + ((m[k] >> 7) & 1u);
            }
            const uint32_t npieces = nfill >> 4;                                    // complete pieces: <= 128
            const u32x4 rest = stage16[npieces];                                    // (one address for the wave) the incomplete piece
            if (first && fill0 && (uint32_t)lane >= fill0 && (uint32_t)lane < min(16u, nfill)) gdst[lane] = stage[lane];   // the segment's first piece: its own bytes one by one
            for (uint32_t p = (uint32_t)lane; p < npieces; p += 64u) {
                const u32x4 piece = stage16[p];
                if (!(first && fill0 && p == 0u)) *reinterpret_cast<u32x4 *>(gdst + 16u * p) = piece;
                stage16[p] = zero4;
            }
            if (lane == 0) {
                stage16[npieces] = zero4;
                stage16[0] = rest;
            }
            gdst += 16u * npieces;
            fill = nfill & 15u;
            first = first && npieces == 0u;
        }
        if ((uint32_t)lane < fill && !(first && (uint32_t)lane < fill0)) gdst[lane] = stage[lane];     // what is left of the last piece
        running = my_ff;
    }
    if (__any(overflow) && lane == 0) atomicOr(&a.stats->status, 1u);

    if (s == a.num_segs - 1 && lane == 0) {
        uint64_t end = base + nown + running;
        const int rem = (int)(b1 & 7u);
        bool ok = true;
        if (rem) {                                              // zero-padded flush (huffman.c:65-81)
            const uint32_t bits = fin_tail_bits(a, a.num_segs, rem);
            if (end < a.out_capacity) a.out[end] = (uint8_t)(bits << (8 - rem)); else ok = false;
            ++end;
        }
        if (a.write_eoi) {                                      // jpeg_handler.c:113-117
            if (end + 2 <= a.out_capacity) { a.out[end] = 0xFF; a.out[end + 1] = 0xD9; } else ok = false;
            end += 2;
        }
        if (!ok) atomicOr(&a.stats->status, 1u);
        *a.out_size = end;
        if (a.last_image) {
            a.stats->out_size = end;
            a.stats->total_bits = b1;
            a.stats->total_ff = chunk_ff0 + ff_in + running;
        }
    }
}

int launch_finalize(const FinalizeArgs &a, void *stream, void *const *ev) {
    if (a.num_chunks <= 0 || a.batch <= 0) return 0;
    const dim3 grid((unsigned)(a.num_chunks * a.batch));
    if (ev) hipExtLaunchKernelGGL(k_finalize, grid, dim3(64 * kFinWaves), 0, (hipStream_t)stream, (hipEvent_t)ev[0], (hipEvent_t)ev[1], 0, a);
    else hipLaunchKernelGGL(k_finalize, grid, dim3(64 * kFinWaves), 0, (hipStream_t)stream, a);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------
// Segment exchange (one image sharded over GPUs by block rows): used words of segments [s0, s1) packed densely +
// kSegMetaWords metadata words per segment {bits, word offset, edge, symbols, exact-path count, 0, 0, 0, ffin[8] as 4 words};
// and the inverse.
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_seg_offsets(const SegExchange x) {
    __shared__ uint32_t s_w[16];
    __shared__ uint32_t s_carry;
    const int n = x.s1 - x.s0, lane = lane_id(), wave = (int)(threadIdx.x >> 6);
    if (threadIdx.x == 0) s_carry = 0u;
    __syncthreads();
    for (int base = 0; base < n; base += 1024) {
        const int i = base + (int)threadIdx.x;
        const uint32_t bits = i < n ? x.seg.bits[x.s0 + i] : 0u;
        const uint32_t nw = (bits + 31u) >> 5;
        const uint32_t incl = wave_incl_scan_u32(nw);
        if (lane == 63) s_w[wave] = incl;
        __syncthreads();
        uint32_t before = s_carry;
        for (int w = 0; w < wave; ++w) before += s_w[w];
        if (i < n) {
            uint32_t *m = x.meta + (size_t)i * kSegMetaWords;
            const uint32_t *f = reinterpret_cast<const uint32_t *>(x.seg.ffin + (size_t)(x.s0 + i) * 8);
            m[0] = bits; m[1] = before + incl - nw; m[2] = x.seg.edge[x.s0 + i];
            m[3] = x.seg.syms[x.s0 + i]; m[4] = x.seg.exact[x.s0 + i]; m[5] = m[6] = m[7] = 0u;
            m[8] = f[0]; m[9] = f[1]; m[10] = f[2]; m[11] = f[3];
        }
        __syncthreads();
        if (threadIdx.x == 1023) s_carry = before + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        *x.total_words = s_carry;
        if ((uint64_t)s_carry > x.dense_cap_words) atomicOr(x.status, 1u);
    }
}

template <bool kExport>
__global__ __launch_bounds__(256) void k_seg_copy(const SegExchange x) {
    const int lane = lane_id();
    const int i = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6);
    if (i >= x.s1 - x.s0) return;
    const uint32_t *m = x.meta + (size_t)i * kSegMetaWords;
    const uint32_t bits = m[0], off = m[1], nw = (bits + 31u) >> 5;
    uint32_t *strided = x.seg.words + (size_t)(x.s0 + i) * x.seg.words_stride;
    if (kExport) {
        if ((uint64_t)off + nw > x.dense_cap_words) return;           // flagged by k_seg_offsets
        for (uint32_t j = (uint32_t)lane; j < nw; j += 64) x.dense[off + j] = strided[j];
    } else {
        for (uint32_t j = (uint32_t)lane; j < nw; j += 64) strided[j] = x.dense[off + j];
        if (lane == 0) {
            x.seg.bits[x.s0 + i] = bits; x.seg.edge[x.s0 + i] = m[2];
            x.seg.syms[x.s0 + i] = m[3]; x.seg.exact[x.s0 + i] = m[4];
            uint32_t *f = reinterpret_cast<uint32_t *>(x.seg.ffin + (size_t)(x.s0 + i) * 8);
            f[0] = m[8]; f[1] = m[9]; f[2] = m[10]; f[3] = m[11];
        }
    }
}

int launch_seg_export(const SegExchange &x, void *stream) {
    const int n = x.s1 - x.s0;
    hipLaunchKernelGGL(k_seg_offsets, dim3(1), dim3(1024), 0, (hipStream_t)stream, x);
    if (n > 0) hipLaunchKernelGGL(k_seg_copy<true>, dim3((n + 3) / 4), dim3(256), 0, (hipStream_t)stream, x);
    return (int)hipGetLastError();
}

int launch_seg_import(const SegExchange &x, void *stream) {
    const int n = x.s1 - x.s0;
    if (n > 0) hipLaunchKernelGGL(k_seg_copy<false>, dim3((n + 3) / 4), dim3(256), 0, (hipStream_t)stream, x);
    return (int)hipGetLastError();
}

// Sums of the per-segment symbol / exact-path counters, on request (jpegamd_encoder_finish with stats).
__global__ __launch_bounds__(1024) void k_sum_stats(const uint32_t *__restrict__ seg_syms, const uint32_t *__restrict__ seg_exact,
                                                     int n, ScanStats *stats) {
    __shared__ unsigned long long s_part[2][16];
    unsigned long long a = 0, b = 0;
    for (int i = (int)threadIdx.x; i < n; i += 1024) { a += seg_syms[i]; b += seg_exact[i]; }
    const int lane = lane_id(), wave = (int)(threadIdx.x >> 6);
    for (int off = 32; off > 0; off >>= 1) { a += __shfl_xor(a, off, 64); b += __shfl_xor(b, off, 64); }
    if (lane == 0) { s_part[0][wave] = a; s_part[1][wave] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long ta = 0, tb = 0;
        for (int w = 0; w < 16; ++w) { ta += s_part[0][w]; tb += s_part[1][w]; }
        stats->total_syms = ta;
        stats->total_exact = tb;
    }
}

int launch_sum_stats(const uint32_t *seg_syms, const uint32_t *seg_exact, int n, ScanStats *stats, void *stream) {
    hipLaunchKernelGGL(k_sum_stats, dim3(1), dim3(1024), 0, (hipStream_t)stream, seg_syms, seg_exact, n, stats);
    return (int)hipGetLastError();
}

int finalize_chunks(int num_segs) { return (num_segs + kFinWaves - 1) / kFinWaves; }

// ------------------------------------------------------------------------------------
// Exact-order DCT of arbitrary blocks (parity tap for dct.c:63-96): one wave per block.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_dct_exact(const int8_t *__restrict__ blocks, float *__restrict__ coeffs, long long nblocks) {
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
