// jpegamd_finalize.hip -- everything after the transform: bit offsets of the segments, 0xFF counting,
// stuffing offsets, byte stitching / stuffing, container.  Two launches, no inter-workgroup waits:
//
//   k_fin_count  workgroup = 16 consecutive segments (one per wave).  All bit counts are final when the
//                kernel starts, so the chunk's bit offset is simply the sum of every earlier segment's
//                count (a few tens of KB from L2, no scan chain).  With the offset -- hence the byte
//                phase -- known, each wave counts the 0xFF bytes it OWNS (a byte belongs to the segment
//                holding its last bit) and the workgroup records its bit offset and 0xFF total.
//   k_fin_write  sums the earlier chunks' 0xFF totals, then every wave writes its owned bytes at
//                prefix + byte index + stuffing offset, 0x00 after each 0xFF (huffman.c:26-32); the wave
//                owning the last segment adds the zero-padded final byte (huffman.c:65-81), EOI
//                (jpeg_handler.c:113-117) and the size.  Every output byte is written by exactly one
//                lane with a plain store: no atomics, no pre-zeroing.
//
// Both kernels are latency-bound (3 MB of payload), so they front-load every global read whose address
// does not depend on data (own / previous segment's bit count, the segment's words into LDS, the
// predecessor sums) and keep dependent round trips to two.
#include <cstdlib>
#include <cstring>

#include "jpegamd_device.h"

namespace jpegamd {

constexpr int kFinWaves = 16;                                  // segments per workgroup
constexpr int kFinCache = 256;                                 // segment words kept in LDS per wave (typical segment: ~90)

__device__ __forceinline__ uint32_t fin_bits_at(const uint32_t *__restrict__ w, uint32_t pos, int nbits /*1..8*/) {
    const uint32_t i = pos >> 5, sh = pos & 31u;
    const uint64_t win = ((uint64_t)w[i] << 32) | w[i + 1];
    return (uint32_t)((win << sh) >> (64 - nbits));
}

// The `need` (1..7) bits that precede segment `s` in the stream (s may equal num_segs).
__device__ __forceinline__ uint32_t fin_tail_bits(const FinalizeArgs &a, int s, int need) {
    uint32_t val = 0;
    int got = 0;
    for (int sp = s - 1; got < need && sp >= 0; --sp) {
        const uint32_t tp = a.seg_bits[sp];
        const int take = min(need - got, (int)tp);
        if (take > 0) {
            val |= fin_bits_at(a.seg_words + (size_t)sp * a.seg_stride, tp - (uint32_t)take, take) << got;
            got += take;
        }
    }
    return val;
}

// 64-bit block sum of one value per thread (values < 2^40), result to every thread.
__device__ __forceinline__ unsigned long long block_sum_u64(unsigned long long part, unsigned long long *s_part /*[kFinWaves]*/) {
    const int lane = lane_id(), wave = (int)(threadIdx.x >> 6);
    const uint32_t plo = (uint32_t)wave_sum_i32((int)(uint32_t)(part & 0xFFFFFFu));
    const uint32_t phi = (uint32_t)wave_sum_i32((int)(uint32_t)(part >> 24));
    if (lane == 0) s_part[wave] = (unsigned long long)plo + ((unsigned long long)phi << 24);
    __syncthreads();
    unsigned long long t = 0;
#pragma unroll
    for (int w = 0; w < kFinWaves; ++w) t += s_part[w];
    __syncthreads();
    return t;
}

// Per-wave view of one segment: its place in the stream and byte access through the LDS copy.
struct FinSeg {
    const uint32_t *words;      // global
    const uint32_t *cache;      // LDS copy of the first kFinCache words
    unsigned long long b0, b1;
    uint32_t nown, leadbits;
    int lead;
    __device__ __forceinline__ uint32_t bits_at(uint32_t pos, int nbits) const {
        const uint32_t i = pos >> 5, sh = pos & 31u;
        uint32_t w0, w1;
        if (i + 1 < (uint32_t)kFinCache) { w0 = cache[i]; w1 = cache[i + 1]; } else { w0 = words[i]; w1 = words[i + 1]; }
        return (uint32_t)(((((uint64_t)w0 << 32) | w1) << sh) >> (64 - nbits));
    }
    __device__ __forceinline__ uint32_t owned_byte(uint32_t r) const {
        if (r == 0 && lead) return (leadbits << (8 - lead)) | bits_at(0, 8 - lead);
        return bits_at(8u * r - (uint32_t)lead, 8);
    }
};

// Stage the wave's segment into LDS and fetch what the leading partial byte needs from the previous segment.
// Nothing here waits for a loaded value before issuing the next load: the first 128 words are requested
// unconditionally (the per-segment reservation is far larger), the rest only for unusually long segments.
__device__ __forceinline__ void fin_stage(const FinalizeArgs &a, int s, bool have, uint32_t my_bits, uint32_t *cache, int lane,
                                          uint32_t *prev_bits, uint32_t *prev_tail7) {
    const uint32_t *words = a.seg_words + (size_t)(have ? s : 0) * a.seg_stride;
    const uint32_t w0 = words[lane], w1 = words[64 + lane];
    uint32_t pb = 0, pt = 0;
    if (have && s > 0) {
        pb = a.seg_bits[s - 1];
        if (a.seg_tail) pt = a.seg_tail[s - 1];                    // written by the transform kernel
        else if (pb >= 7) pt = fin_bits_at(a.seg_words + (size_t)(s - 1) * a.seg_stride, pb - 7u, 7);
    }
    cache[lane] = w0;
    cache[64 + lane] = w1;
    const uint32_t nw = have ? min((my_bits + 31u) / 32u + 1u, (uint32_t)kFinCache) : 0u;
    for (uint32_t j = 128u + (uint32_t)lane; j < nw; j += 64) cache[j] = words[j];
    *prev_bits = pb;
    *prev_tail7 = pt;
}

__device__ __forceinline__ FinSeg fin_view(const FinalizeArgs &a, int s, bool have, unsigned long long b0, uint32_t my_bits,
                                           const uint32_t *cache, uint32_t prev_bits, uint32_t prev_tail7) {
    FinSeg v;
    v.words = a.seg_words + (size_t)(have ? s : 0) * a.seg_stride;
    v.cache = cache;
    v.b0 = b0;
    v.b1 = b0 + my_bits;
    v.nown = have ? (uint32_t)((v.b1 >> 3) - (b0 >> 3)) : 0u;
    v.lead = (int)(b0 & 7u);
    v.leadbits = 0;
    if (have && v.lead) v.leadbits = prev_bits >= 7 ? (prev_tail7 & ((1u << v.lead) - 1u)) : fin_tail_bits(a, s, v.lead);
    return v;
}

__global__ __launch_bounds__(64 * kFinWaves) void k_fin_count(const FinalizeArgs a) {
    __shared__ uint32_t s_seg[kFinWaves][kFinCache];
    __shared__ uint32_t s_cnt[kFinWaves];
    __shared__ unsigned long long s_part[kFinWaves];

    const int lane = lane_id(), wave = (int)(threadIdx.x >> 6);
    const int g = (int)blockIdx.x;
    const int s = g * kFinWaves + wave;
    const bool have = s < a.num_segs;

    const uint32_t my_bits = have ? a.seg_bits[s] : 0u;
    uint32_t prev_bits, prev_tail7;
    fin_stage(a, s, have, my_bits, s_seg[wave], lane, &prev_bits, &prev_tail7);

    // bit offset of the chunk: sum of all earlier segments (16 g values, g*16 % 4 == 0)
    unsigned long long part = 0;
    {
        const int n_before = g * kFinWaves, step = 64 * kFinWaves * 4;
        for (int i = (int)threadIdx.x * 4; i < n_before; i += 4 * step) {          // up to 4 loads in flight per trip
            uint4 q[4];
#pragma unroll
            for (int k = 0; k < 4; ++k)
                q[k] = (i + k * step < n_before) ? *reinterpret_cast<const uint4 *>(a.seg_bits + i + k * step) : make_uint4(0, 0, 0, 0);
#pragma unroll
            for (int k = 0; k < 4; ++k) part += (unsigned long long)q[k].x + q[k].y + q[k].z + q[k].w;
        }
    }
    if (lane == 0) s_cnt[wave] = my_bits;
    const unsigned long long chunk_b0 = block_sum_u64(part, s_part);     // (syncs: s_cnt is visible afterwards)
    uint32_t before = 0;
    for (int w = 0; w < wave; ++w) before += s_cnt[w];
    const FinSeg v = fin_view(a, s, have, chunk_b0 + before, my_bits, s_seg[wave], prev_bits, prev_tail7);

    int ffc = 0;
    for (uint32_t r = (uint32_t)lane; r < v.nown; r += 64) ffc += (v.owned_byte(r) == 0xFFu) ? 1 : 0;
    ffc = wave_sum_i32(ffc);
    __syncthreads();
    if (lane == 0) { s_cnt[wave] = (uint32_t)ffc; if (have) a.seg_ff[s] = (uint32_t)ffc; }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = 0;
        for (int w = 0; w < kFinWaves; ++w) t += s_cnt[w];
        a.chunk_ff[g] = t;
        a.chunk_b0[g] = chunk_b0;
    }
}

__global__ __launch_bounds__(64 * kFinWaves) void k_fin_write(const FinalizeArgs a) {
    __shared__ uint32_t s_seg[kFinWaves][kFinCache];
    __shared__ unsigned long long s_part[kFinWaves];

    const int lane = lane_id(), wave = (int)(threadIdx.x >> 6);
    const int g = (int)blockIdx.x;
    const int s = g * kFinWaves + wave;
    const bool have = s < a.num_segs;

    if (g == 0 && a.prefix_len > 0)                             // JFIF prefix (jpeg_handler.c:220-233)
        for (int i = (int)threadIdx.x; i < a.prefix_len; i += 64 * kFinWaves)
            if ((uint64_t)i < a.out_capacity) a.out[i] = a.prefix[i];

    const uint32_t my_bits = have ? a.seg_bits[s] : 0u;
    uint32_t prev_bits, prev_tail7;
    fin_stage(a, s, have, my_bits, s_seg[wave], lane, &prev_bits, &prev_tail7);
    // offsets inside the chunk: bits and 0xFF counts of the chunk's earlier segments
    uint32_t b_in, ff_in;                                       // every wave: lane w loads segment w of the chunk, DPP scan
    {
        const int sp = g * kFinWaves + lane;
        const bool in = lane < kFinWaves && sp < a.num_segs;
        const uint32_t vb = in ? a.seg_bits[sp] : 0u, vf = in ? a.seg_ff[sp] : 0u;
        const uint32_t ib = wave_incl_scan_u32(vb), iff = wave_incl_scan_u32(vf);
        b_in = (uint32_t)__builtin_amdgcn_readlane((int)(ib - vb), wave);
        ff_in = (uint32_t)__builtin_amdgcn_readlane((int)(iff - vf), wave);
    }
    // stuffed bytes in front of the chunk
    unsigned long long part = 0;
    for (int i = (int)threadIdx.x; i < g; i += 4 * 64 * kFinWaves) {
        uint32_t q[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) q[k] = (i + k * 64 * kFinWaves < g) ? a.chunk_ff[i + k * 64 * kFinWaves] : 0u;
        part += (unsigned long long)q[0] + q[1] + q[2] + q[3];
    }
    const unsigned long long chunk_ff0 = block_sum_u64(part, s_part);
    if (!have) return;
    const FinSeg v = fin_view(a, s, have, a.chunk_b0[g] + b_in, my_bits, s_seg[wave], prev_bits, prev_tail7);

    const uint64_t base = (uint64_t)a.prefix_len + (v.b0 >> 3) + chunk_ff0 + ff_in;
    uint32_t running = 0;
    bool overflow = false;
    for (uint32_t r0 = 0; r0 < v.nown; r0 += 64) {
        const uint32_t r = r0 + (uint32_t)lane;
        const bool valid = r < v.nown;
        const uint32_t byte = valid ? v.owned_byte(r) : 0u;
        const bool isff = valid && byte == 0xFFu;
        const unsigned long long m = __ballot(isff);
        const uint32_t before = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        const uint64_t pos = base + r + running + before;
        if (valid) {
            if (pos + (isff ? 2u : 1u) <= a.out_capacity) {
                a.out[pos] = (uint8_t)byte;
                if (isff) a.out[pos + 1] = 0x00;               // huffman.c:29-31
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
        a.stats->out_size = end;
        a.stats->total_bits = v.b1;
        a.stats->total_ff = chunk_ff0 + ff_in + running;
    }
}

// ------------------------------------------------------------------------------------------------------
// Second generation of the two kernels (default).  Same ownership rule and the same two passes, rebuilt
// around the instruction count (the whole pipeline is issue-bound; the first generation spent
// 7.5 M instructions per 8192^2 image on 3 MB of payload):
//   * a lane handles FOUR owned bytes at a time: the segment prefixed by its `lead` borrowed bits is a byte
//     string, its i-th dword is one funnel shift of two adjacent segment words (v_alignbit by `lead`),
//     0xFF bytes are found with a 7-instruction SWAR test -- instead of ~25 instructions per byte;
//   * no LDS staging of the segment (each word is read twice from L1/L2, in adjacent lanes);
//   * in-chunk offsets by one DPP scan of the chunk's 16 bit counts in every wave; the sum over all earlier
//     chunks is computed by wave 0 only (count kernel) or from the 16x smaller per-chunk arrays (write kernel).
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t fin_owned_dword(const uint32_t *__restrict__ words, uint32_t i, uint32_t lead, uint32_t leadbits) {
    const uint32_t cur = words[i];
    const uint32_t prev = i ? words[i - 1] : leadbits;
    return __builtin_amdgcn_alignbit(prev, cur, lead);              // ({prev, cur} >> lead): lead = 0 gives cur
}

// bit 8k set <=> byte k (little-endian numbering) of w is 0xFF
__device__ __forceinline__ uint32_t fin_ff_mask(uint32_t w) {
    uint32_t t = w & (w >> 4);
    t &= t >> 2;
    t &= t >> 1;
    return t & 0x01010101u;
}

struct FinPlace {
    const uint32_t *words;
    unsigned long long b0, b1;
    uint32_t nown, lead, leadbits;
};

// Offsets inside the chunk (one DPP scan of its 16 bit counts) and the segment's borrowed leading bits.
__device__ __forceinline__ FinPlace fin_place(const FinalizeArgs &a, int g, int wave, int lane, unsigned long long chunk_b0,
                                              uint32_t vb /*lane l < 16: bits of segment 16g + l*/, uint32_t excl_b) {
    const int s = g * kFinWaves + wave;
    const bool have = s < a.num_segs;
    FinPlace v;
    v.words = a.seg_words + (size_t)(have ? s : 0) * a.seg_stride;
    const uint32_t my_bits = (uint32_t)__builtin_amdgcn_readlane((int)vb, wave);
    v.b0 = chunk_b0 + (uint32_t)__builtin_amdgcn_readlane((int)excl_b, wave);
    v.b1 = v.b0 + my_bits;
    v.nown = have ? (uint32_t)((v.b1 >> 3) - (v.b0 >> 3)) : 0u;
    v.lead = (uint32_t)(v.b0 & 7u);
    v.leadbits = 0;
    if (have && v.lead && s > 0) {
        const uint32_t pb = a.seg_bits[s - 1];
        if (pb >= 7u) {
            const uint32_t t7 = a.seg_tail ? (uint32_t)a.seg_tail[s - 1]
                                           : fin_bits_at(a.seg_words + (size_t)(s - 1) * a.seg_stride, pb - 7u, 7);
            v.leadbits = t7 & ((1u << v.lead) - 1u);
        } else {
            v.leadbits = fin_tail_bits(a, s, (int)v.lead);       // predecessor shorter than a byte: walk further back
        }
    }
    (void)lane;
    return v;
}

__global__ __launch_bounds__(64 * kFinWaves) void k_fin_count2(const FinalizeArgs a) {
    __shared__ uint32_t s_cnt[kFinWaves];
    __shared__ unsigned long long s_b0;
    const int lane = lane_id(), wave = (int)(threadIdx.x >> 6);
    const int g = (int)blockIdx.x;
    const int s = g * kFinWaves + wave;
    const bool have = s < a.num_segs;

    const int sp = g * kFinWaves + (lane & 15);
    const uint32_t vb = (lane < kFinWaves && sp < a.num_segs) ? a.seg_bits[sp] : 0u;
    if (wave == 0) {                                              // bit offset of the chunk: all earlier segments (16 g of them)
        unsigned long long part = 0;
        const int n_before = g * kFinWaves;
        for (int i = lane * 4; i < n_before; i += 256) {
            const uint4 q = *reinterpret_cast<const uint4 *>(a.seg_bits + i);
            part += (unsigned long long)q.x + q.y + q.z + q.w;
        }
        const uint32_t plo = (uint32_t)wave_sum_i32((int)(uint32_t)(part & 0xFFFFFFu));
        const uint32_t phi = (uint32_t)wave_sum_i32((int)(uint32_t)(part >> 24));
        if (lane == 0) s_b0 = (unsigned long long)plo + ((unsigned long long)phi << 24);
    }
    const uint32_t ib = wave_incl_scan_u32(vb);
    __syncthreads();
    const unsigned long long chunk_b0 = s_b0;
    const FinPlace v = fin_place(a, g, wave, lane, chunk_b0, vb, ib - vb);

    const uint32_t ndw = (v.nown + 3u) >> 2;
    int ffc = 0;
    for (uint32_t i = (uint32_t)lane; i < ndw; i += 64) {
        const uint32_t w = fin_owned_dword(v.words, i, v.lead, v.leadbits);
        const uint32_t nv = min(4u, v.nown - 4u * i);                       // 1..4 owned bytes in this dword, from the top
        ffc += __popc(fin_ff_mask(w) & (0x01010101u << (8u * (4u - nv))));
    }
    ffc = wave_sum_i32(ffc);
    if (lane == 0) { s_cnt[wave] = (uint32_t)ffc; if (have) a.seg_ff[s] = (uint32_t)ffc; }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = 0;
        for (int w = 0; w < kFinWaves; ++w) t += s_cnt[w];
        a.chunk_ff[g] = t;
        a.chunk_b0[g] = chunk_b0;
    }
}

__global__ __launch_bounds__(64 * kFinWaves) void k_fin_write2(const FinalizeArgs a) {
    const int lane = lane_id(), wave = (int)(threadIdx.x >> 6);
    const int g = (int)blockIdx.x;
    const int s = g * kFinWaves + wave;
    const bool have = s < a.num_segs;

    if (g == 0 && a.prefix_len > 0)                             // JFIF prefix (jpeg_handler.c:220-233)
        for (int i = (int)threadIdx.x; i < a.prefix_len; i += 64 * kFinWaves)
            if ((uint64_t)i < a.out_capacity) a.out[i] = a.prefix[i];
    if (!have) return;                                          // no workgroup-wide synchronisation below

    const int sp = g * kFinWaves + (lane & 15);
    const bool in = lane < kFinWaves && sp < a.num_segs;
    const uint32_t vb = in ? a.seg_bits[sp] : 0u, vf = in ? a.seg_ff[sp] : 0u;
    const unsigned long long chunk_b0 = a.chunk_b0[g];
    uint32_t cf = 0;                                            // stuffed bytes in front of the chunk
    for (int c = lane; c < g; c += 64) cf += a.chunk_ff[c];
    const uint32_t ib = wave_incl_scan_u32(vb), iff = wave_incl_scan_u32(vf);
    const unsigned long long chunk_ff0 = (unsigned long long)(uint32_t)wave_sum_i32((int)cf);
    const uint32_t ff_in = (uint32_t)__builtin_amdgcn_readlane((int)(iff - vf), wave);
    const FinPlace v = fin_place(a, g, wave, lane, chunk_b0, vb, ib - vb);

    const uint64_t base = (uint64_t)a.prefix_len + (v.b0 >> 3) + chunk_ff0 + ff_in;
    const uint32_t ndw = (v.nown + 3u) >> 2;
    uint32_t running = 0;
    bool overflow = false;
    for (uint32_t i0 = 0; i0 < ndw; i0 += 64) {
        const uint32_t i = i0 + (uint32_t)lane;
        const bool valid = i < ndw;
        const uint32_t w = valid ? fin_owned_dword(v.words, i, v.lead, v.leadbits) : 0u;
        const uint32_t nv = valid ? min(4u, v.nown - 4u * i) : 0u;
        const uint32_t m = valid ? (fin_ff_mask(w) & (0x01010101u << (8u * (4u - nv)))) : 0u;
        const uint32_t c = (uint32_t)__popc(m);
        const uint32_t incl = wave_incl_scan_u32(c);
        const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        uint64_t pos = base + 4ull * i + running + (incl - c);
        if (valid) {
            if (pos + nv + c <= a.out_capacity) {
#pragma unroll
                for (uint32_t k = 0; k < 4; ++k) {                  // byte k from the top (stream order)
                    if (k < nv) {
                        a.out[pos++] = (uint8_t)(w >> (24u - 8u * k));
                        if (m & (1u << (24u - 8u * k))) a.out[pos++] = 0x00;       // huffman.c:29-31
                    }
                }
            } else {
                overflow = true;
            }
        }
        running += tot;
    }
    if (__any(overflow) && lane == 0) atomicOr(&a.stats->status, 1u);

    if (s == a.num_segs - 1 && lane == 0) {
        uint64_t end = base + v.nown + running;
        const int rem = (int)(v.b1 & 7u);
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
        a.stats->out_size = end;
        a.stats->total_bits = v.b1;
        a.stats->total_ff = chunk_ff0 + ff_in + running;
    }
}

int launch_finalize(const FinalizeArgs &a, void *stream) {
    static const bool first_gen = [] { const char *e = std::getenv("JPEGAMD_FINALIZE"); return e && std::strcmp(e, "v1") == 0; }();
    if (first_gen) {
        hipLaunchKernelGGL(k_fin_count, dim3(a.num_chunks), dim3(64 * kFinWaves), 0, (hipStream_t)stream, a);
        hipLaunchKernelGGL(k_fin_write, dim3(a.num_chunks), dim3(64 * kFinWaves), 0, (hipStream_t)stream, a);
    } else {
        hipLaunchKernelGGL(k_fin_count2, dim3(a.num_chunks), dim3(64 * kFinWaves), 0, (hipStream_t)stream, a);
        hipLaunchKernelGGL(k_fin_write2, dim3(a.num_chunks), dim3(64 * kFinWaves), 0, (hipStream_t)stream, a);
    }
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------
// Segment exchange (one image sharded over GPUs by block rows): used words of segments [s0, s1) packed densely +
// 8 metadata words per segment {bits, word offset, tail, symbols, exact-path count, 0, 0, 0}; and the inverse.
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_seg_offsets(const SegExchange x) {
    __shared__ uint32_t s_w[16];
    __shared__ uint32_t s_carry;
    const int n = x.s1 - x.s0, lane = lane_id(), wave = (int)(threadIdx.x >> 6);
    if (threadIdx.x == 0) s_carry = 0u;
    __syncthreads();
    for (int base = 0; base < n; base += 1024) {
        const int i = base + (int)threadIdx.x;
        const uint32_t bits = i < n ? x.seg_bits[x.s0 + i] : 0u;
        const uint32_t nw = (bits + 31u) >> 5;
        const uint32_t incl = wave_incl_scan_u32(nw);
        if (lane == 63) s_w[wave] = incl;
        __syncthreads();
        uint32_t before = s_carry;
        for (int w = 0; w < wave; ++w) before += s_w[w];
        if (i < n) {
            uint32_t *m = x.meta + (size_t)i * 8;
            m[0] = bits; m[1] = before + incl - nw; m[2] = x.seg_tail[x.s0 + i];
            m[3] = x.seg_syms[x.s0 + i]; m[4] = x.seg_exact[x.s0 + i]; m[5] = m[6] = m[7] = 0u;
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
    const uint32_t *m = x.meta + (size_t)i * 8;
    const uint32_t bits = m[0], off = m[1], nw = (bits + 31u) >> 5;
    uint32_t *strided = x.seg_words + (size_t)(x.s0 + i) * x.seg_stride;
    if (kExport) {
        if ((uint64_t)off + nw > x.dense_cap_words) return;           // flagged by k_seg_offsets
        for (uint32_t j = (uint32_t)lane; j < nw; j += 64) x.dense[off + j] = strided[j];
    } else {
        for (uint32_t j = (uint32_t)lane; j < nw; j += 64) strided[j] = x.dense[off + j];
        if (lane == 0) {
            x.seg_bits[x.s0 + i] = bits; x.seg_tail[x.s0 + i] = (uint8_t)m[2];
            x.seg_syms[x.s0 + i] = m[3]; x.seg_exact[x.s0 + i] = m[4];
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

}  // namespace jpegamd
