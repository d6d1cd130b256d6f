// jpegamd_tile_pipeline.hip -- k_tile_encode: pixels -> per-tile Huffman bit strings (first of the pipeline's three kernels).
//
// Persistent 512-thread workgroups; one wave-iteration per TILE of 32 blocks of one block row:
//   luma (converter.c:31-51,60-90)            v_dot4 per pixel, straight into binary16 MFMA operands
//   8x8 DCT (dct.c:63-96)                     the 64x64 LUT-product matrix on the matrix pipe, as two integer-valued binary16
//                                             terms in separate accumulator chains: exact in any summation order
//   quantisation (quantization.c:34-36)       guard-band quantiser; a coefficient within delta of a rounding tie is recomputed in
//                                             the reference's own float order (one coefficient per wave pass)
//   zigzag (zigzag.c:21-68)                   the matrix rows are stored in grouped zigzag order: no data movement
//   RLE (rle.c:51-127)                        DC DPCM inside the tile, non-zero compaction: every lane appends its block's
//                                             symbols-to-be -- (zigzag position, value) items: DC, non-zero ACs, EOB -- to the
//                                             tile's list in LDS
//   Huffman coding (huffman.c:121-193)        the SAME wave codes its list, two items per lane: size / amplitude / run from the
//                                             item and its predecessor, ONE table lookup per symbol, a wave prefix sum over the
//                                             bit counts, ds_or into a bit window in LDS -- the tile's bit string, which leaves the
//                                             kernel with its 8-word record as ONE 8-byte-per-lane store
// k_segment_merge (jpegamd_entropy.hip) joins the strings of a segment's 8 tiles (and codes the one symbol per tile that
// needs the tile before: the DC of its first block), k_finalize (jpegamd_finalize.hip) stitches the segments.  Round 2 wrote the
// item lists to HBM (24 MB per 8192^2 picture) and coded them in a second kernel that read them back (28 MB): 1.31 x the
// algorithmic traffic, a launch whose waves all start cold, and two phases that could not overlap -- this kernel's waves spent
// 48 % of their time in s_waitcnt while that one's were bound by VALU issue.
// Tiles are handed out dynamically through 64 ticket counters; the next tile's pixel rows are requested as soon as its ticket
// is in, and are in flight during the appends and the coding.  A launch covers one image, a block-row shard of one, or a batch
// of images of one geometry (ImageDesc::batch).  DESIGN.md 4.0-4.1 has the measurements behind each choice.
#include <hip/hip_ext.h>
#include "jpegamd_device.h"

// This file is compiled TWICE into the library: plain (the hot path: no stamp instruction executed) and with
// -DJPEGAMD_STAMPED_TU, where every phase of the tile loop is bracketed by s_memtime reads whose per-wave sums leave in
// TransformOutM::stamps -- launch_tile_transform_stamped, what convertToJpeg runs to fill the six stage counters of the reference's
// DTO (dsp_port/jpeg_compression/src/jpeg_compression.c:188-210) from ONE fused kernel.  The stamps cost ~10 % of the kernel.
#ifdef JPEGAMD_STAMPED_TU
#ifndef JPEGAMD_STAMPS
#define JPEGAMD_STAMPS 1
#endif
#define k_tile_encode k_tile_encode_stamped
#define launch_tile_transform launch_tile_transform_stamped
#endif

namespace jpegamd {

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#ifndef JPEGAMD_TILE_WG_WAVES
#define JPEGAMD_TILE_WG_WAVES 8
#endif
#ifndef JPEGAMD_TILE_MAX_WGS
#define JPEGAMD_TILE_MAX_WGS 512
#endif
constexpr int kWavesT = JPEGAMD_TILE_WG_WAVES;  // waves per workgroup; they share the 24 KiB matrix image in LDS

struct RawRow { uint32_t d[6]; };      // one block row: 8 pixels x 3 bytes

__device__ __forceinline__ RawRow load_raw_row(const uint32_t *__restrict__ src) {
    RawRow r;
#pragma unroll
    for (int i = 0; i < 6; ++i) r.d[i] = src[i];
    return r;
}

// Luma weights as the six dot-product operands of a pixel row (3 bytes per pixel: a pixel starts at byte 0, 3, 2 or 1 of a dword).
struct LumaWeights { uint32_t a, b0, b1, c0, c1, d; };
__device__ __forceinline__ LumaWeights luma_weights(uint32_t w /*weights of stored bytes 0, 1, 2*/) {
    const uint32_t c0 = w & 0xFFu, c1 = (w >> 8) & 0xFFu, c2 = (w >> 16) & 0xFFu;
    LumaWeights l;
    l.a = w; l.b0 = c0 << 24; l.b1 = c1 | (c2 << 8); l.c0 = (c0 << 16) | (c1 << 24); l.c1 = c2; l.d = w << 8;
    return l;
}

// 8 pixels (24 bytes) -> 8 luma values (converter.c:51,84-86) as the B fragment of one k-step half.
// Y = (w . rgb) >> 8 is byte 1 of the dot product, and the integer Y in a 16-bit half IS the binary16 subnormal Y 2^-24: the matrix pipe
// takes it at face value (tools/ubench/mfma_denorm.hip), so two dot products are packed by ONE v_perm -- 12 v_dot4 + 4 v_perm per 8
// pixels (rounds 2-4: 8 v_cvt_f16_i16 with SDWA byte selects on the CENTRED value; round 1: + 6 packs to bf16 + 4 permutes).  The
// operand is the UNCENTRED luma: the hi terms of every AC row of the matrix add up to zero and the DC row's surplus is a constant
// (quant_consts.cpp: zoff, dc_off).
__device__ __forceinline__ f16x8 luma_row8_f16(const RawRow &raw, const LumaWeights &w, uint32_t sel /*0x0C050C01: byte 1 of each source into the halves*/) {
    uint32_t o0, o1, o2, o3, t0, t1, t2, t3, t4, t5, t6, t7;
    asm("v_dot4_u32_u8 %[t0], %[d0], %[wa], 0\n\t"
        "v_dot4_u32_u8 %[t1], %[d0], %[wb0], 0\n\t"
        "v_dot4_u32_u8 %[t2], %[d1], %[wc0], 0\n\t"
        "v_dot4_u32_u8 %[t3], %[d2], %[wd], 0\n\t"
        "v_dot4_u32_u8 %[t4], %[d3], %[wa], 0\n\t"
        "v_dot4_u32_u8 %[t5], %[d3], %[wb0], 0\n\t"
        "v_dot4_u32_u8 %[t6], %[d4], %[wc0], 0\n\t"
        "v_dot4_u32_u8 %[t7], %[d5], %[wd], 0\n\t"
        "v_dot4_u32_u8 %[t1], %[d1], %[wb1], %[t1]\n\t"
        "v_dot4_u32_u8 %[t2], %[d2], %[wc1], %[t2]\n\t"
        "v_dot4_u32_u8 %[t5], %[d4], %[wb1], %[t5]\n\t"
        "v_dot4_u32_u8 %[t6], %[d5], %[wc1], %[t6]\n\t"
        "v_perm_b32 %[o0], %[t1], %[t0], %[sel]\n\t"
        "v_perm_b32 %[o1], %[t3], %[t2], %[sel]\n\t"
        "v_perm_b32 %[o2], %[t5], %[t4], %[sel]\n\t"
        "v_perm_b32 %[o3], %[t7], %[t6], %[sel]"
        : [o0] "=&v"(o0), [o1] "=&v"(o1), [o2] "=&v"(o2), [o3] "=&v"(o3), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3),
          [t4] "=&v"(t4), [t5] "=&v"(t5), [t6] "=&v"(t6), [t7] "=&v"(t7)
        : [d0] "v"(raw.d[0]), [d1] "v"(raw.d[1]), [d2] "v"(raw.d[2]), [d3] "v"(raw.d[3]), [d4] "v"(raw.d[4]), [d5] "v"(raw.d[5]),
          [wa] "s"(w.a), [wb0] "s"(w.b0), [wb1] "s"(w.b1), [wc0] "s"(w.c0), [wc1] "s"(w.c1), [wd] "s"(w.d), [sel] "v"(sel));
    typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
    const u32x4 packed = {o0, o1, o2, o3};
    return __builtin_bit_cast(f16x8, packed);
}


// In-kernel phase stamps (diagnostic builds only: -DJPEGAMD_STAMPS; the shipped kernel executes none).
// The stamps do NOT drain vmcnt, so the prefetch / store overlap stays as shipped;
// a phase is charged with whatever its own s_waitcnt instructions wait for.
#ifdef JPEGAMD_STAMPS
// (the phase sums live in LDS, added to by lane 0 alone: eleven scalar accumulators cost the kernel spilled registers, and a
//  scratch reload -- a vector memory operation behind the closing store -- distorted the very profile they were for)
#define TSTAMP(i)                                                                             \
    do {                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                    \
        unsigned long long t_, sv_;                                                           \
        unsigned va_, vd_;                                                                    \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");            \
        const unsigned d_ = (unsigned)t_ - st_last;                                           \
        st_last = (unsigned)t_;                                                               \
        asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, 1\n\tv_mov_b32 %1, %3\n\tv_mov_b32 %2, %4\n\tds_add_u32 %1, %2\n\ts_mov_b64 exec, %0" \
                     : "=&s"(sv_), "=&v"(va_), "=&v"(vd_) : "s"(st_lds + 4u * (i)), "s"(d_) : "memory");          \
        __builtin_amdgcn_sched_barrier(0);                                                    \
    } while (0)
#elif defined(JPEGAMD_MARKS)      // static instruction census: markers in the .s file (tools/isa_census.py)
#define TSTAMP(i) do { __builtin_amdgcn_sched_barrier(0); asm volatile("; MARK " #i ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define TSTAMP(i) do { } while (0)
#endif

#ifndef JPEGAMD_TILE_GROUPS
#define JPEGAMD_TILE_GROUPS 64
#endif
constexpr int kTileGroups = JPEGAMD_TILE_GROUPS;                 // ticket counters (one cache line each) of the dynamic tile hand-out
#ifndef JPEGAMD_TILE_WAVES
#define JPEGAMD_TILE_WAVES 4
#endif
constexpr int kWinWords = 256;                                   // bit window per wave in LDS: the tile's record (8 words) + 248 string words
constexpr int kWinStr = kWinWords - kTileRecWords;
constexpr int kPassItems = 128;                                  // items coded per pass: two per lane
constexpr int kQuadMinItems = 384;                               // lists at least this long try four items per lane first
static_assert(8 * 65 <= kStageItemCap, "a quarter of a tile's items (8 blocks) always fits the staging region");
static_assert(kPassItems * (27 + 3 * (int)kZrlBits) / 32 + 6 <= kWinStr && 2 * kPassItems * 27 / 32 + 6 <= kWinStr, "the window takes a whole pass, ZRLs included, and a four-item pass without");

struct TileSched {            // division-free launch geometry, filled by launch_tile_transform
    int32_t grp_shift;        // workgroups form 1 << grp_shift ticket groups (blockIdx & mask)
    int32_t tiles_per_group;  // contiguous tiles owned by a group
    uint32_t tpr_magic;       // floor(2^32 / tiles_per_row) + 1: tile / tiles_per_row by multiply-high (+ one correction)
};

// Appends of one group's sites to the staged list, branch-free: per site one compare that narrows EXEC to the lanes holding
// a non-zero value, the SDWA add that writes the zigzag position into the value's upper half (the item format of
// jpegamd_internal.h), the LDS write, the address increment (5 issue slots; the compiler's version costs 4 slots for a site
// no lane uses and ~10, with a taken branch, for the others).  `addr` is the byte address in LDS of the lane's next item,
// `zg` the zigzag position of the group's site 0; values are modified in place.
// (skipping a site no lane of the wave uses -- s_cbranch_execz behind the compare, a third of the sites of an active group -- was
// measured: no difference, 379.0 / 376.7 vs 378.4 / 377.7 us per launch of eight; profiles/r04_notes_experiments.txt)
// One site: EXEC is narrowed to the lanes whose value is non-zero (CMP: the compare), the SDWA add writes the site's zigzag position
// into the upper half of the value's register (the item format of jpegamd_internal.h), the LDS write, the address increment.
#define JPEGAMD_APPEND_ONE(CMP, V, J)                                                                                   \
    CMP " 0, %[" #V "]\n\t"                                                                                             \
    "v_add_u32_sdwa %[" #V "], %[zg], " #J " dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n\t" \
    "ds_write_b32 %[addr], %[" #V "]\n\t"                                                                               \
    "v_add_u32_e32 %[addr], 4, %[addr]\n\t"                                                                            \
    "s_mov_b64 exec, %[save]\n\t"
// A PAIR of sites lives in one register (low half: site J, high half: site J + 1): the high value moves to a scratch register first
// (a full-rate right shift), then both go out in site order; the low one is compared as 16 bits.
#define JPEGAMD_APPEND_PAIR(V, J0, J1)                                                                                  \
    "v_lshrrev_b32_e32 %[t], 16, %[" #V "]\n\t"                                                                         \
    JPEGAMD_APPEND_ONE("v_cmpx_ne_u16_e32", V, J0)                                                                      \
    JPEGAMD_APPEND_ONE("v_cmpx_ne_u32_e32", t, J1)
#define JPEGAMD_APPEND_HIGH(V, J1)                                                                                      \
    "v_lshrrev_b32_e32 %[t], 16, %[" #V "]\n\t"                                                                         \
    JPEGAMD_APPEND_ONE("v_cmpx_ne_u32_e32", t, J1)
// The four pairs of one group (kSkipFirst: site 0 of group 0 -- the DC item, or zigzag 8 -- was written by the caller).
template <bool kSkipFirst>
__device__ __forceinline__ void append_group_lds(uint32_t &addr, uint32_t (&v)[4], uint32_t zg) {
    uint64_t save;
    uint32_t t;
    if (kSkipFirst) {
        asm volatile("s_mov_b64 %[save], exec\n\t"
                     JPEGAMD_APPEND_HIGH(v0, 1) JPEGAMD_APPEND_PAIR(v1, 2, 3) JPEGAMD_APPEND_PAIR(v2, 4, 5) JPEGAMD_APPEND_PAIR(v3, 6, 7)
                     : [addr] "+v"(addr), [save] "=&s"(save), [t] "=&v"(t), [v0] "+v"(v[0]), [v1] "+v"(v[1]), [v2] "+v"(v[2]), [v3] "+v"(v[3])
                     : [zg] "v"(zg)
                     : "vcc", "memory");
    } else {
        asm volatile("s_mov_b64 %[save], exec\n\t"
                     JPEGAMD_APPEND_PAIR(v0, 0, 1) JPEGAMD_APPEND_PAIR(v1, 2, 3) JPEGAMD_APPEND_PAIR(v2, 4, 5) JPEGAMD_APPEND_PAIR(v3, 6, 7)
                     : [addr] "+v"(addr), [save] "=&s"(save), [t] "=&v"(t), [v0] "+v"(v[0]), [v1] "+v"(v[1]), [v2] "+v"(v[2]), [v3] "+v"(v[3])
                     : [zg] "v"(zg)
                     : "vcc", "memory");
    }
}
#undef JPEGAMD_APPEND_PAIR
#undef JPEGAMD_APPEND_HIGH
#undef JPEGAMD_APPEND_ONE

// ---- the coder ---------------------------------------------------------------------------------------------------------
// item -> code table entry + left-aligned bit string (Huffman code, then amplitude bits: huffman.c:145-153,176-186).
// `prev` is the item in front of it in the list; only its position field is used, and only by a non-zero AC item, whose
// predecessor is always an item of the same block (its DC item, position 0, or the non-zero coefficient before it).
//   amplitude code  w = v + (v >> 31)                     rle.c:24-35: v, or v - 1 when negative
//   size            31 - v_ffbh_i32(2 w), -1 -> 0         rle.c:9-22 without the abs / zero special cases
//   row             gap to the predecessor = run + 1; class D items (position 0) take row 0 whatever precedes them:
//                   min(gap, own position) -- a gap is never larger than the position, and a class D item's is 0
__device__ __forceinline__ uint32_t item_row(uint32_t it, uint32_t prev) {      // table row of an item: run + 1, or 0 (class D); rows above 16 carry ZRLs
    const uint32_t pos = it >> 16;
    return min(pos - (prev >> 16), pos);
}
__device__ __forceinline__ void code_item_row(uint32_t it, uint32_t row, const uint32_t *tab /*LDS: the code table*/, uint32_t &e, uint32_t &bits) {
    const int v = (int)(short)(it & 0xFFFFu);
    const int x2 = (v + (v >> 31)) << 1;
    int fb;
    asm("v_ffbh_i32 %0, %1" : "=v"(fb) : "v"(x2));
    const uint32_t al = (uint32_t)x2 << (fb & 31);                            // amplitude bits, left-aligned
    e = tab[kCodeLead + (int)__umul24(row, (uint32_t)kCodeRowStride) + fb];
    bits = (e & 0xFFFF0000u) | (al >> (e & 31u));
}
__device__ __forceinline__ void code_item(uint32_t it, uint32_t prev, const uint32_t *tab, uint32_t &e, uint32_t &bits) {
    code_item_row(it, item_row(it, prev), tab, e, bits);
}

// z ZRL codes (huffman.c:158-188 codes them as ordinary symbols: 0xF0 is 11111111001, 11 bits) in front of a left-aligned
// string of <= 27 bits: (bits : 0) >> 11 z under the constant prefix.  z = 3 shifts by 33: the string lands in the low word.
__device__ __forceinline__ void zrl_prefix(uint32_t bits, uint32_t z /*0..3*/, uint32_t &hi, uint32_t &lo) {
    static_assert(kZrlBits == 11 && kZrlCode == 0x7F9, "the prefix constants below are three copies of this code");
    const uint32_t sh = z * 11u;
    const uint32_t hs = __builtin_amdgcn_alignbit(0u, bits, sh), ls = __builtin_amdgcn_alignbit(bits, 0u, sh);     // (the funnel shifts use sh mod 32)
    // three copies of the code are 0xFF3FE7FC : 0x80000000; z of them = its top 11 z bits (branch-free: the compiler makes a switch of a constant table)
    const bool three = z == 3u;
    const uint32_t chi = 0xFF3FE7FCu & ~(three ? 0u : 0xFFFFFFFFu >> sh);
    hi = chi | (three ? 0u : hs);
    lo = three ? (0x80000000u | hs) : ls;
}
// sum over the wave of a small per-lane count (0..7): three ballots and scalar population counts instead of a DPP prefix sum
__device__ __forceinline__ uint32_t wave_sum_3bit(uint32_t v) {
    return (uint32_t)__popcll(__ballot(v & 1u)) + 2u * (uint32_t)__popcll(__ballot(v & 2u)) + 4u * (uint32_t)__popcll(__ballot(v & 4u));
}

// OR a left-aligned string (hi:lo, <= 64 bits) into the window at bit `rel`.
__device__ __forceinline__ void window_or(uint32_t *win, uint32_t rel, uint32_t hi, uint32_t lo, bool third) {
    const uint32_t w = rel >> 5, sh = rel & 31u;
    atomicOr(&win[w], __builtin_amdgcn_alignbit(0u, hi, sh));
    atomicOr(&win[w + 1], __builtin_amdgcn_alignbit(hi, lo, sh));
    if (third) atomicOr(&win[w + 2], __builtin_amdgcn_alignbit(lo, 0u, sh));
}

template <bool kTaps>
__global__ __launch_bounds__(64 * kWavesT) __attribute__((amdgpu_waves_per_eu(JPEGAMD_TILE_WAVES, JPEGAMD_TILE_WAVES)))
void k_tile_encode(const ImageDesc im, const TransformOutM out, const TileSched sch) {
    __shared__ __attribute__((aligned(16))) uint32_t s_afrag[kAFragWords];
    // s_qt[0..127]: (multiplier, additive constant: bias + zoff) by zigzag position; [136 + 16 h]: the DC row's surplus (h == 0; 0 for h == 1); [160 + 16 h + 32 G + j]: flag threshold (2 bias - 1) of zigzag 16 G + 8 h + j; [128 + 16 h + G]: zero threshold of group G for lane half h
    // (|acc| below it => every site of the group quantises to an unflagged 0), [132 + 16 h + G]: the largest tie threshold of the
    // group's sites (fract(zc) above it => no site is flagged).  One layout with 64 bytes per lane half: one address register.
    __shared__ __attribute__((aligned(16))) float s_qt[128 + 32 + 128];
    __shared__ float s_qstep[64];
    __shared__ float s_cos[64];
    __shared__ uint32_t s_zz[64];               // zigzag position -> raster index (exact-order path)
    // The tile's centred luma (binary16, exact), kept for the exact-order path: row r of block b at word r * 132 + b * 4
    // (528-byte rows: the four 1 KiB stores of a wave and the 64 two-byte reads of one block are conflict-free).
    // Reloading the pixels from HBM instead made every exact-order event wait for vmcnt(0), i.e. for the
    // prefetched rows of the NEXT tile as well: ~40 % of a tile's time per event (tools/stamp_profile_tile.py).
    // After the exact-order phase the same words hold the tile's item list.
    __shared__ __attribute__((aligned(16))) uint32_t s_pix[kWavesT][kStageWords];
    __shared__ __attribute__((aligned(16))) uint32_t s_win[kWavesT][kWinWords];      // per wave: the tile's record + bit window; all zero between tiles
    static_assert(kWinWords == 4 * 64, "the exact-order path parks the 64 terms of four coefficients in the (idle, zero) bit window and zeroes it again");
    __shared__ __attribute__((aligned(16))) uint32_t s_code[kCodeWords];
#ifdef JPEGAMD_STAMPS
    unsigned long long st_rt0;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_rt0)::"memory");
#endif

// The workgroup's tables go to LDS BEHIND the first tile's row requests: the two HBM latencies of the prologue overlap
// (-11 us per launch of eight pictures, -1.5 us per single one).
#define JPEGAMD_LOAD_TABLES() \
        { \
            const int t = (int)threadIdx.x; \
            const uint4 *src = reinterpret_cast<const uint4 *>(out.tables->afrag); \
            uint4 *dst = reinterpret_cast<uint4 *>(s_afrag); \
            for (int i = t; i < kAFragWords / 4; i += 64 * kWavesT) dst[i] = src[i]; \
            const uint4 *csrc = reinterpret_cast<const uint4 *>(out.code_tab); \
            for (int i = t; i < kCodeWords / 4; i += 64 * kWavesT) reinterpret_cast<uint4 *>(s_code)[i] = csrc[i]; \
            for (int i = t; i < kWavesT * kWinWords / 4; i += 64 * kWavesT) reinterpret_cast<uint4 *>(&s_win[0][0])[i] = make_uint4(0u, 0u, 0u, 0u); \
            if (t < 64) { \
                s_qt[2 * t] = out.tables->qmul[t]; s_qt[2 * t + 1] = out.tables->qadd[t]; \
                s_qt[160 + 16 * ((t >> 3) & 1) + 32 * (t >> 4) + (t & 7)] = out.tables->qthr[t]; \
                s_qstep[t] = out.tables->qstep[t]; \
                s_cos[t] = kCosFM[t]; \
                s_zz[t] = kZZ[t]; \
                if (t < 8) { s_qt[128 + 16 * (t & 1) + (t >> 1)] = out.tables->grp_thr[t]; s_qt[132 + 16 * (t & 1) + (t >> 1)] = out.tables->flag_thr[t]; } \
                if (t < 2) s_qt[136 + 16 * t] = t ? 0.0f : out.tables->dc_off; \
            } \
        } \
        __syncthreads(); \

    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // provably uniform: tile indices, list pointers and the buffer descriptor stay on the scalar unit
    const int h = lane >> 5, b = lane & 31;
    const LumaWeights lw = luma_weights(im.weights);
    const uint32_t luma_sel = 0x0C050C01u;
    const float *q_lane = &s_qt[16 * h];

    // Persistent waves: tile = first, first + stride, ...  The matrix image is loaded once per workgroup and the
    // NEXT tile's pixel rows are requested as soon as the current ones are converted, so their HBM latency
    // hides behind the MFMA / quantise / append phases of the current tile.
    // Tile hand-out.  Per-tile time varies a lot with content (symbols per tile: mean 127, std 91 on the bench
    // image; exact-order events), so a static split of 8 tiles per wave leaves SIMDs idle behind the slowest wave
    // (per-wave totals: max/mean 1.5, tools/stamp_profile_tile.py).  One global ticket per tile is no answer
    // either: same-address device atomics retire at ~12 ns each (measured: 6x slower).  So the workgroups form
    // kTileGroups groups (blockIdx % groups: the members of a group sit on one XCD), each group owns a share of the
    // tiles and hands them to its waves through its OWN ticket counter (own cache line).
    const int groups = 1 << sch.grp_shift;
    const int bid = (int)blockIdx.x;
    const int grp = bid & (groups - 1);
    const int gmask = groups - 1;
    const auto waves_of = [&](int g) { return ((((int)gridDim.x - 1 - g) >> sch.grp_shift) + 1) * kWavesT; };   // waves of group g
    // A group owns one CHUNK (kWavesT consecutive tiles, what its 8 waves work on side by side) of every STEP of `groups`
    // consecutive chunks, and walks its chunks in order: content density varies slowly over the picture, so contiguous
    // ranges per group left the densest group 10-20 % behind the mean (profiles/r02_stamps_r01_kernel.txt).  Within step k
    // the group takes slot (g + k) mod groups: with a fixed slot a group kept to ONE column band of the picture (8192
    // wide: 4 chunks per row, 64 % 4 == 0), and the bands differ (workgroup means 77 k .. 93 k cycles,
    // profiles/r02_stamps_interleaved.txt).
    // Tile indices of the loop (li, nxt, cur_hi) are GROUP-LOCAL; to_tile() maps them to picture tiles.
    const int ntl = im.tile_end - im.tile_begin;
    const int nchunks = (ntl + kWavesT - 1) / kWavesT;                       // >= 1 (launch_tile_transform)
    const int full_steps = nchunks >> sch.grp_shift, rem = nchunks & gmask;
    const int last_owner = (((nchunks - 1) & gmask) - ((nchunks - 1) >> sch.grp_shift)) & gmask;   // group of the last (maybe short) chunk
    const auto tiles_of = [&](int g) {                                       // group-local tile count of group g
        const int chunks = full_steps + ((rem > 0 && ((g + full_steps) & gmask) < rem) ? 1 : 0);
        return chunks * kWavesT - (g == last_owner ? nchunks * kWavesT - ntl : 0);
    };
    // (Drawing from partner groups on other XCDs once the own group is dry was tried in rounds 2 and 3, 1-3 levels deep: no gain --
    // the spread INSIDE a group, one tile-time, dominates.  profiles/r02_notes_experiments.txt, r03_notes_experiments.txt)
    int cur_grp = grp, cur_hi = tiles_of(grp), cur_waves = waves_of(grp);
    const auto to_tile = [&](int li) {
        const int k = li / kWavesT;
        return im.tile_begin + (((k << sch.grp_shift) + ((cur_grp + k) & gmask)) * kWavesT) + (li % kWavesT);
    };
    uint32_t *ctr = out.tile_ctr + grp * 32;                                 // word 0: tickets handed out
    // This launch's counters were zeroed by the previous launch on this context; zero the next launch's (the other set).
    if (bid == 0 && threadIdx.x < kTileGroups) out.tile_ctr_next[threadIdx.x * 32] = 0u;   // all of them: the next launch may form more groups
    const int first = (bid >> sch.grp_shift) * kWavesT + wave;
    // One lane draws the ticket: EXEC = 1 around the atomic (no lane mask to keep in two scalar registers across the loop).  The result is
    // collected behind an explicit s_waitcnt vmcnt(0) -- the compiler does not count this operation (it then waits for one more than it
    // thinks wherever it waits: safe).
    const auto ticket = [&]() -> uint32_t {
        uint32_t r = 0u, one = 1u, zero = 0u;
        uint64_t save;
        asm volatile("s_mov_b64 %[sv], exec\n\ts_mov_b64 exec, 1\n\tglobal_atomic_add %[r], %[z], %[one], %[p] sc0\n\ts_mov_b64 exec, %[sv]"
                     : [r] "+v"(r), [sv] "=&s"(save) : [z] "v"(zero), [one] "v"(one), [p] "s"(ctr) : "memory");
        return r;
    };
    struct TileGeo { int img, by, tbx0, nblk, interior; };      // (no padding bytes: a bool at the end made the copy of the struct carry three of them through scratch)
    const auto geo = [&](int tile) {
        TileGeo g;
        g.img = 0;
        if (im.batch > 1) {                                     // a batch: image = tile / num_tiles, then the tile inside it
            int i = (int)__umulhi((uint32_t)tile, im.tpi_magic), tl = tile - i * im.num_tiles;
            if (tl >= im.num_tiles) { ++i; tl -= im.num_tiles; }
            g.img = i;
            tile = tl;
        }
        int q = (int)__umulhi((uint32_t)tile, sch.tpr_magic), r = tile - q * im.tiles_per_row;   // tile / tiles_per_row
        if (r >= im.tiles_per_row) { ++q; r -= im.tiles_per_row; }
        g.by = q;
        g.tbx0 = r * kTileBlocks;
        g.nblk = min(kTileBlocks, im.blocks_w - g.tbx0);
        // (integer arithmetic: as booleans the three conditions meet in lane masks and come back through a v_cndmask and a v_readfirstlane)
        const int over_x = (im.width - (g.tbx0 + g.nblk) * 8) >> 31, over_y = (im.height - (g.by * 8 + 8)) >> 31;      // -1: beyond the picture
        g.interior = (im.fast_ok != 0 ? 1 : 0) & ~(over_x | over_y);
        return g;
    };
    // Interior tiles: a scalar base (lowest-address row of the tile's 8, first block) plus 32-bit lane offsets --
    // one multiply-add and three adds per tile instead of eight 64-bit multiply-adds (quarter-rate instructions).
    // Lane (h, b) reads picture rows by*8 + 2s + h; in a bottom-up BMP those lie at DEscending addresses (bmp_handler.c:109).
    // (the lane's row term, (bottom_up ? 7 - h : h) * row_stride, is recomputed per tile from an opaque copy of h: as a loop
    //  invariant it was spilled, and its reload sat, behind a vmcnt(0), right in front of the row requests)
    const int32_t row_step = im.bottom_up ? -2 * im.row_stride : 2 * im.row_stride;
    const auto request_rows = [&](const TileGeo &g, RawRow (&raw)[4]) {
        const int row_low = im.bottom_up ? im.height - 8 - g.by * 8 : g.by * 8;
        const uint8_t *tb = im.batch_pixels[g.img] + (size_t)row_low * (size_t)im.row_stride + 24 * (size_t)g.tbx0;
        uint32_t hh = (uint32_t)h;
        asm volatile("" : "+v"(hh));
        uint32_t off = __umul24((uint32_t)min(b, g.nblk - 1), 24u) + __umul24(im.bottom_up ? 7u - hh : hh, (uint32_t)im.row_stride);   // row_stride < 2^24
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            raw[s] = load_raw_row(reinterpret_cast<const uint32_t *>(tb + off));
            off += (uint32_t)row_step;
        }
    };
    RawRow raw[4];
    int tile = first < cur_hi ? to_tile(first) : im.tile_begin;     // picture tile of the iteration (carried: to_tile() once per tile)
    TileGeo tg = geo(tile);
    if (first < cur_hi && tg.interior) request_rows(tg, raw);
    JPEGAMD_LOAD_TABLES()
#undef JPEGAMD_LOAD_TABLES
#ifdef JPEGAMD_STAMPS
    __shared__ unsigned s_st[kWavesT][16];
    if (lane < 16) s_st[wave][lane] = 0u;
    const unsigned st_lds = (unsigned)(uintptr_t)&s_st[wave][0];           // LDS byte address of this wave's phase sums
    unsigned st_last;
    unsigned long long st_rt1, st_c1;
    asm volatile("s_memrealtime %0\n\ts_memtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_rt1), "=s"(st_c1)::"memory");
    st_last = (unsigned)st_c1;
#endif

#pragma unroll 1
    for (int li = first; li < cur_hi;) {
        const int by = tg.by, nblk = tg.nblk, bx = tg.tbx0 + min(b, tg.nblk - 1);     // idle columns shadow the last block
        const int py0 = by * 8, px0 = bx * 8;
        const bool active = b < nblk, interior = tg.interior;
        int nexact = 0;
        TSTAMP(0);   // loop overhead / geometry
        // ---- 1. pixels -> B fragments ----------------------------------------------------------
        f16x8 bfrag[4];
        if (interior) {                        // rows requested one iteration ago, behind the ticket (below)
#pragma unroll
            for (int s = 0; s < 4; ++s) bfrag[s] = luma_row8_f16(raw[s], lw, luma_sel);
        } else {
            // edge tile (right/bottom replication, converter.c:31,36) or unaligned source: clamped byte gather
            typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                u32x4 pk;
#pragma unroll
                for (int j = 0; j < 8; j += 2)                  // two values per register: Y in each 16-bit half (= Y 2^-24 as binary16, as above)
                    pk[j >> 1] = (uint32_t)luma_clamped(im, im.batch_pixels[tg.img], px0 + j, py0 + 2 * s + h) |
                                 ((uint32_t)luma_clamped(im, im.batch_pixels[tg.img], px0 + j + 1, py0 + 2 * s + h) << 16);
                bfrag[s] = __builtin_bit_cast(f16x8, pk);
            }
        }
        TSTAMP(1);   // wait for the prefetched rows + luma
        {
        uint32_t sl;                           // (an opaque lane id: the stash addresses are not worth four registers across the whole loop)
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(sl));
        // word (sl >> 5) * 132 + (sl & 31) * 4, as bytes: 16 sl + 16 (sl >> 5) -- two shifts, an and, an add (the product with 132 compiled to a v_mul_lo_u32)
        uint32_t *const sp0 = reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(&s_pix[wave][0]) + ((sl << 4) + ((sl >> 1) & 16u)));
#pragma unroll
        for (int s = 0; s < 4; ++s) *reinterpret_cast<f16x8 *>(&sp0[2 * s * 132]) = bfrag[s];     // (one address, four immediate offsets)
        }
        TSTAMP(2);   // luma -> LDS
        if (kTaps && active && out.tap_y) {
            int8_t *ty = out.tap_y + ((size_t)by * im.blocks_w + bx) * 64;
            typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const u32x4 pk = __builtin_bit_cast(u32x4, bfrag[s]);
#pragma unroll
                for (int j = 0; j < 8; ++j) ty[(2 * s + h) * 8 + j] = (int8_t)((int)((pk[j >> 1] >> (16 * (j & 1))) & 0xFFFFu) - 128);     // converter.c:51: the centred value
            }
        }

        // ---- 2. the 64x64 transform on the matrix pipe ------------------------------------------
        // The LUT products are split into two INTEGER-valued binary16 terms (hi = round(2^11 K), lo = round(2^22 (K - hi 2^-11)),
        // stored as lo 2^-11), and each term has its own accumulator chain: with |pixel| <= 128 every product and every partial
        // sum of a chain is a multiple of its unit below 2^24 units, so the float32 accumulation is EXACT in any order the
        // matrix pipe may use.  One float add joins the chains (its single rounding is in the guard band).  Round 2's first
        // kernel ran both terms through one accumulator and had to budget 2 x 16 roundings per MFMA: 2.3 x the reference's
        // own evaluation error, i.e. three times as many exact-order events (quant_consts.cpp).
        f32x16 acc[2][2];                      // [term][chain]
        // One chain = the four k-steps of one term (t) for one half of the matrix rows (H): four A fragments in one batch of
        // LDS reads, then four MFMAs back to back behind ONE wait.
        const auto load_afrag = [&](const int t, const int H, f16x8 (&afr)[4]) {
            const uint32_t *at = &s_afrag[((t * 2 + H) * 4 * 64 + lane) * 4];
#pragma unroll
            for (int i = 0; i < 4; ++i) afr[i] = *reinterpret_cast<const f16x8 *>(&at[(i * 64) * 4]);
            __builtin_amdgcn_sched_barrier(0);     // keep the LDS reads where they are (hipcc otherwise sinks each read next to its use)
        };
        const auto mfma_chain = [&](const int t, const int H, const f16x8 (&afr)[4]) {
            __builtin_amdgcn_sched_barrier(0);
            const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            acc[t][H] = __builtin_amdgcn_mfma_f32_32x32x16_f16(afr[0], bfrag[0], zero, 0, 0, 0);     // C = 0 as an inline constant: no accumulator clearing
#pragma unroll
            for (int i = 1; i < 4; ++i) acc[t][H] = __builtin_amdgcn_mfma_f32_32x32x16_f16(afr[i], bfrag[i], acc[t][H], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        };
        // Is any site of group G alive?  |sum| below the group's zero threshold in every lane => every site quantises to an
        // unflagged 0.  Tested on the hi chain alone (the threshold has the lo chain's bound taken off).
        const auto group_alive = [&](const int G) -> int {
            float m = fmaxf(fabsf(acc[1][(8 * G) >> 4][(8 * G) & 15]), fabsf(acc[1][(8 * G + 1) >> 4][(8 * G + 1) & 15]));
#pragma unroll
            for (int j = 2; j < 8; ++j) m = fmaxf(m, fabsf(acc[1][(8 * G + j) >> 4][(8 * G + j) & 15]));
            int alive = (int)__popcll(__ballot(m >= q_lane[128 + G]));      // (a scalar count, not a lane mask: tested by s_cmp where it is used)
            asm volatile("" : "+s"(alive));
            return alive;
        };
        // The hi chains first, the upper matrix rows (zigzag 32..63: groups 2 and 3) ahead of the lower ones: in photo-like content those
        // groups are dead in two tiles of three, the test on their hi sums runs beside the lower half's MFMAs, and a dead upper half
        // never runs its lo chain (4 of the 16 MFMAs; an MFMA costs the SIMD's vector side 10-20 cycles beside three other waves,
        // profiles/r04_ubench_mfma_overlap.txt).
        // ---- 3. quantise with the guard band, one GROUP of 8 sites at a time -----------------------
        // Site s = 16H + r of lane (h, b) holds zigzag position 16 * (s >> 3) + 8 * h + (s & 7): group G = s >> 3
        // covers zigzag 16G .. 16G + 15 across the two lanes of a block.  In photo-like content most tiles have NO non-zero
        // coefficient in the higher groups: one max|acc| test (5 instructions) skips the quantiser, the counts
        // and the appends of such a group (11+ instructions per site).
        // The quantised values live as PACKED int16 pairs -- n2[p]: site 2p in the low half, site 2p + 1 in the high half (16 registers
        // instead of 32: what a fifth wave per SIMD needs).  A value is the low half of fma(sum, multiplier, 1.5 * 2^23): round-to-
        // nearest-even of z, which differs from floor(z + 0.5 + delta) only inside the flagged band, where the exact-order path
        // overwrites it anyway; the fraction that decides the flag still comes from z + 0.5 + delta_z.
#define JPEGAMD_ACC(site) (acc[1][(site) >> 4][(site) & 15] + acc[0][(site) >> 4][(site) & 15])     /* hi chain + lo chain */
        uint32_t n2[16];
        uint32_t flagbits = 0;                                      // bit s: site s of this lane is within delta of a rounding tie
        int gact[4];
        gact[0] = 1;
        // (read through a scalar the compiler cannot trace at every use: else each flag becomes a lane mask AND its negation, made by
        //  two vector instructions, held in two register pairs)
        const auto alive = [&](const int G) -> bool { int g = gact[G]; asm volatile("" : "+s"(g)); return g != 0; };
        const auto quantise_group = [&](const int G) {
            if (G == 0 || alive(G)) {
                float a8[8];                                        // the group's LUT sums (times kMfmaScale)
#pragma unroll
                for (int j = 0; j < 8; ++j) a8[j] = JPEGAMD_ACC(8 * G + j);
                if (G == 0) a8[0] -= q_lane[136];                   // the DC row holds the UNCENTRED pixel sum: 64 x 128 (1.0 in accumulator units) too much; exact.  (lanes h == 1: 0)
                float fr[8];
                uint32_t tv[8];
                // DC (zigzag 0: site 0 of lanes h == 0).  Its LUT sum S is an exact integer and the reference's value has a closed form,
                // sign(S) floor((|S| + 4 q) / 8 q): the scale 0.25 * 0.707107^2 lies 6.2e-7 ABOVE 1/8, which pushes the ties S = 4 q (2 m + 1)
                // away from zero by ten float32 steps and nothing else across a tie (checked for every S and every q in 1 .. 255 by
                // tests/test_host.py::test_dc_closed_form).  floor(|z| + 0.5 + delta) is that value, so the DC lanes quantise |S|,
                // take the sign afterwards and never flag.
                // (the mask of the DC lanes, 0x00000000FFFFFFFF, is made where it is used: one scalar move instead of a register pair held
                //  across the whole loop)
                unsigned long long dc_lanes;
                float a0 = a8[0];
                if (G == 0) asm("s_mov_b64 %1, 0xffffffff\n\tv_cndmask_b32_e64 %0, %2, |%2|, %1" : "=v"(a0), "=&s"(dc_lanes) : "v"(a8[0]));
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float2 q = reinterpret_cast<const float2 *>(q_lane)[16 * G + j];
                    const float aj = j == 0 ? a0 : a8[j];
                    const float zc = fmaf(aj, q.x, q.y);            // z + 0.5 + delta_z
                    fr[j] = __builtin_amdgcn_fractf(zc);
                    if (G == 0 && j == 0) {                         // (site 0 keeps the floor: the DC's closed form is stated for it)
                        const int sg = (int)(__builtin_bit_cast(uint32_t, a0) ^ __builtin_bit_cast(uint32_t, a8[0])) >> 31;    // -1 in the DC lanes with S < 0
                        tv[0] = (uint32_t)((floor_to_int(zc) ^ sg) - sg);
                    } else {
                        tv[j] = __builtin_bit_cast(uint32_t, fmaf(aj, q.x, 12582912.0f));      // 0x4B400000 + n: the value is the low half
                    }
                }
                if (G == 0) asm("s_mov_b64 %1, 0xffffffff\n\tv_cndmask_b32_e64 %0, %0, 1.0, %1" : "+v"(fr[0]), "=&s"(dc_lanes));           // a DC is never flagged
#pragma unroll
                for (int p2 = 0; p2 < 4; ++p2) n2[4 * G + p2] = __builtin_amdgcn_perm(tv[2 * p2 + 1], tv[2 * p2], 0x05040100u);
                // Flags are rare (0.4 per tile): one min tree over the fractions against the group's largest threshold decides
                // for the whole wave whether the per-site compares (16 instructions) are needed at all.
                const float fmin8 = fminf(fminf(__builtin_fminf(fr[0], fminf(fr[1], fr[2])), fminf(fr[3], fminf(fr[4], fr[5]))), fminf(fr[6], fr[7]));
                if (__ballot(fmin8 <= q_lane[132 + G]) != 0ull) {
                    float th[8];                                             // (read here, two 16-byte reads, not carried from above)
#pragma unroll
                    for (int j = 0; j < 8; ++j) th[j] = q_lane[160 + 32 * G + j];
                    flagbits |= shift_in_le8(0u, fr, th) << (8 * G);        // bit j: site 8G + j is within delta of a tie
                }
            } else if (kTaps) {
#pragma unroll
                for (int p2 = 0; p2 < 4; ++p2) n2[4 * G + p2] = 0u;
            }
        };
        // The transform in two halves of 32 matrix rows, the UPPER rows (zigzag 32..63: groups 2 and 3) first: one half's two
        // accumulator chains are 32 registers.  In photo-like content the upper groups are dead in two tiles of three: the test on their
        // hi sums decides whether the half's lo chain runs at all (4 of the 16 MFMAs; an MFMA costs the SIMD's vector side 10-20 cycles
        // beside the other waves, profiles/r04_ubench_mfma_overlap.txt).  A chain's A fragments are read from LDS one chain AHEAD (their
        // latency runs beside the matrix pipe and the zero tests: -1.5 % per launch of eight), and the B fragments stay in registers for
        // both halves (re-reading them from the luma stash, as the five-waves-per-SIMD attempt needed, cost four more LDS reads).
        f16x8 afrA[4], afrB[4];
        load_afrag(1, 1, afrA);
        load_afrag(1, 0, afrB);                      // (the lower half's hi terms: on their way while the upper half's chain runs and is tested)
        mfma_chain(1, 1, afrA);
        gact[2] = group_alive(2);
        gact[3] = group_alive(3);
        if (kTaps || (gact[2] | gact[3]) != 0) { load_afrag(0, 1, afrA); mfma_chain(0, 1, afrA); }
        quantise_group(2);
        quantise_group(3);
        load_afrag(0, 0, afrA);
        mfma_chain(1, 0, afrB);
        mfma_chain(0, 0, afrA);
        TSTAMP(3);   // MFMA (+ the upper half's quantiser)
        const uint32_t ticket_v = ticket();          // (requested here, collected behind the counts: at the top of the iteration or behind the luma
                                                     //  conversion measured 1.2 % slower -- a wave then sits longer on a reserved, unstarted tile at the end)
        gact[1] = group_alive(1);
        quantise_group(0);
        quantise_group(1);
        if (!active) flagbits = 0u;
        TSTAMP(4);   // quantise

        // The ticket for this wave's NEXT tile is requested here and collected before the appends (its latency hides
        // behind the exact-order and count phases).  Round 1 asked one whole iteration earlier, for the tile after next:
        // every wave then sat on a reserved, unstarted tile when its group ran dry, and the last waves of a group
        // finished two tile-times (11 us of 58) after the first (profiles/r02_stamps_interleaved.txt).
        // ---- 4. exact-order recomputation of flagged coefficients ------------------------------
        uint64_t exact_mask = 0;
        unsigned long long fm = __ballot(flagbits != 0u);
        if (__builtin_expect(fm != 0ull, 0)) {
            // Everything this rare path derives from the lane number is computed HERE, from a lane id the compiler cannot trace
            // to the kernel's own: as loop invariants of the tile loop those values were spilled to scratch and reloaded for
            // every event (and the zigzag table was a global load): three memory round trips per event behind a vmcnt(0).
            int el;
            asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(el));
            const uint32_t *pix_lane = &s_pix[wave][(el >> 3) * 132 + ((el & 7) >> 1)];
            float *const terms = reinterpret_cast<float *>(&s_win[wave][0]);      // (the window is idle -- and all zero -- until the coder: zeroed again below)
            const float *const my_terms = terms + (el >> 4) * 64;       // lanes 16 e .. 16 e + 15 add up event e of a batch
            // An event = one coefficient recomputed in the reference's own order (dct.c:72-93): lane j forms term j = x * 8 + y, the 64
            // terms go to LDS, and the ordered sum is 64 dependent adds that EVERY lane executes.  Up to four events share those
            // adds: each sixteen lanes read another event's terms (Q=90 has 1-2 events per tile, the adds were 60 % of an event).
            // The values land in ONE register (the flagged lane's), and go into n[] behind the loop in straight-line code: a loop
            // that updates n[st] itself makes the register allocator keep a second copy of all 32 values and move it back and forth
            // per event (90 of the 230 instructions an event cost in round 2).  A round handles the LOWEST flagged site of every
            // lane; a lane with two flagged sites (rare) makes a second round.
            do {
                const uint32_t low = flagbits & (0u - flagbits);
                int fixval = 0;
                unsigned long long todo = fm;
                while (todo) {
                    float sc = 0.0f, qs = 1.0f;
                    unsigned long long me[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        me[e] = 0ull;
                        if (todo) {                                 // (uniform)
                            const int fl = __ffsll((long long)todo) - 1;
                            todo &= todo - 1;
                            const int st = __ffs(__builtin_amdgcn_readlane((int)low, fl)) - 1;
                            const int z = 16 * (st >> 3) + 8 * (fl >> 5) + (st & 7);
                            const int k = __builtin_amdgcn_readfirstlane((int)s_zz[z]), u = k >> 3, v = k & 7;
                            const uint32_t pw = pix_lane[(fl & 31) * 4];
                            const float pix = (float)((int)((el & 1) ? pw >> 16 : pw & 0xFFFFu) - 128);      // the stash holds Y; the reference sums Y - 128
                            const float cx = s_cos[u * 8 + (el >> 3)];     // COS_LUT[x][u]
                            const float cy = s_cos[v * 8 + (el & 7)];      // COS_LUT[y][v]
                            terms[e * 64 + el] = __fmul_rn(__fmul_rn(pix, cx), cy);             // dct.c:84
                            const bool mine = (el >> 4) == e;
                            sc = mine ? ref_scale(u, v) : sc;
                            qs = mine ? s_qstep[z] : qs;
                            me[e] = 1ull << fl;
                            ++nexact;
                            if (kTaps && el == fl) exact_mask |= 1ull << k;
                        }
                    }
                    float sum = 0.0f;                               // dct.c:68
#pragma unroll
                    for (int c = 0; c < 8; ++c) {
                        const float4 a = *reinterpret_cast<const float4 *>(&my_terms[8 * c]);
                        const float4 b4 = *reinterpret_cast<const float4 *>(&my_terms[8 * c + 4]);
                        sum = __fadd_rn(sum, a.x); sum = __fadd_rn(sum, a.y); sum = __fadd_rn(sum, a.z); sum = __fadd_rn(sum, a.w);
                        sum = __fadd_rn(sum, b4.x); sum = __fadd_rn(sum, b4.y); sum = __fadd_rn(sum, b4.z); sum = __fadd_rn(sum, b4.w);
                    }
                    const int val = ref_quantise(__fmul_rn(sc, sum), qs);          // dct.c:93, quantization.c:34-36 (lanes of an unused quarter: anything)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        int vtmp;
                        const int ve = __builtin_amdgcn_readlane(val, 16 * e);
                        asm volatile("v_mov_b32 %1, %2\n\tv_cndmask_b32_e64 %0, %0, %1, %3" : "+v"(fixval), "=&v"(vtmp) : "s"(ve), "s"(me[e]));
                    }
                }
                const uint32_t fix2 = __builtin_amdgcn_perm((uint32_t)fixval, (uint32_t)fixval, 0x05040100u);      // the value in both halves
#pragma unroll
                for (int G = 0; G < 4; ++G) {
                    if (__ballot((low >> (8 * G)) & 0xFFu) == 0ull) continue;
#pragma unroll
                    for (int p2 = 0; p2 < 4; ++p2) {            // (asm: the compiler's version is and + compare + wait state + select)
                        int m0, m1;
                        asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(m0) : "v"(low), "n"(8 * G + 2 * p2));
                        asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(m1) : "v"(low), "n"(8 * G + 2 * p2 + 1));
                        const uint32_t m = __builtin_amdgcn_perm((uint32_t)m1, (uint32_t)m0, 0x05040100u);
                        asm("v_bfi_b32 %0, %1, %2, %0" : "+v"(n2[4 * G + p2]) : "v"(m), "v"(fix2));
                    }
                }
                flagbits ^= low;
                fm = __ballot(flagbits != 0u);
            } while (__builtin_expect(fm != 0ull, 0));
#pragma unroll
            for (int i = 0; i < kWinWords / 64; ++i) s_win[wave][i * 64 + el] = 0u;      // the window as the coder expects it
        }
        if (kTaps && active) {
            const size_t blk = (size_t)by * im.blocks_w + bx;
            if (out.tap_zz) {
#pragma unroll
                for (int st = 0; st < 32; ++st) out.tap_zz[blk * 64 + 16 * (st >> 3) + 8 * h + (st & 7)] = (int16_t)(n2[st >> 1] >> (16 * (st & 1)));
            }
            if (out.tap_mask) atomicOr((unsigned long long *)&out.tap_mask[blk], (unsigned long long)exact_mask);
        }
        TSTAMP(5);   // exact fallback

        // ---- 5. symbol counts per (lane, group), packed one byte per group -> list positions ---------
        // A block's list is ordered by zigzag position: group 0 of lane h=0, group 0 of lane h=1, group 1 of h=0, ...
        const bool eob = (h == 1) && (alive(3) ? ((n2[15] >> 16) == 0u) : true);    // rle.c:121-123 (zigzag 63)
        uint32_t cnt = (h == 0) ? 1u : 0u;                                          // the DC item
        {
            uint32_t hh = (uint32_t)h;
            asm volatile("" : "+v"(hh));
            const uint32_t dc_off = hh ? 0xFFFFFFFFu : 0xFFFF0000u;                  // the DC (site 0 of the lanes h == 0) is not an AC symbol
            const uint32_t ones = 0x00010001u;
#pragma unroll
            for (int G = 0; G < 4; ++G) {
                if (G && !alive(G)) continue;
                // non-zero values of the group: min(v as unsigned, 1) on both halves of a pair at once, summed by plain (full-rate) adds
                uint32_t f[4];
#pragma unroll
                for (int p2 = 0; p2 < 4; ++p2) asm("v_pk_min_u16 %0, %1, %2" : "=v"(f[p2]) : "v"(n2[4 * G + p2]), "s"(ones));
                if (G == 0) f[0] &= dc_off;
                const uint32_t c2 = (f[0] + f[1]) + (f[2] + f[3]);                  // (low half: even sites, high half: odd sites; <= 4 each)
                cnt += __builtin_amdgcn_sad_u8(c2, 0u, 0u) << (8 * G);
            }
        }
        cnt += eob ? (1u << 24) : 0u;
        if (!active) cnt = 0u;
        const uint32_t partner = other_half(cnt, lane);
        const uint32_t tg4 = cnt + partner;                                         // per-group symbols of block b (bytes)
        const uint32_t tb = __builtin_amdgcn_sad_u8(tg4, 0u, 0u);                   // ... and their sum
        const uint32_t incl = half_incl_scan_dpp(tb);
        const uint32_t t_all = (uint32_t)__builtin_amdgcn_readlane((int)incl, 31);
        uint32_t pre = tg4 + (tg4 << 8);
        pre += pre << 16;                                                           // inclusive prefix over the groups, per byte
        const uint32_t starts = (pre << 8) + (h ? partner : 0u);                    // byte G: items of the block ahead of my run G
        const uint32_t blk_base = incl - tb;

        // DC prediction inside the tile (rle.c:59-70).  The DC of the tile's FIRST block needs the last block of the tile before:
        // a padding item holds its place in the list, k_segment_merge codes the symbol from the two tiles' records.
        int n0;                                                                     // site 0: the DC in the lanes h == 0
        asm("v_bfe_i32 %0, %1, 0, 16" : "=v"(n0) : "v"(n2[0]));
        int pred = lane_shift_up1(n0);
        asm volatile("" : "+v"(pred));             // (kept apart from the subtraction: fused into one v_subrev_u32_dpp ... wave_shr:1 the difference came out WRONG on the GPU)
        uint32_t dc_item = (uint32_t)((n0 - pred) & 0xFFFF);
        asm("s_nop 1\n\tv_writelane_b32 %0, %1, 0" : "+v"(dc_item) : "s"(kItPadValue));      // lane 0 (the tile's first block) takes the padding item
        const int first_dc = __builtin_amdgcn_readlane(n0, 0), last_dc = __builtin_amdgcn_readlane(n0, nblk - 1);
        TSTAMP(6);   // counts + scans

        // The ticket requested behind the MFMAs is collected here, BEFORE any younger memory operation is issued
        // (built with the atomic optimizer off -- its expansion reads the result back, and waits for vmcnt(0), right
        // behind the atomic).
        uint32_t ticket_in = ticket_v;
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(ticket_in) :: "memory");
        int nxt = cur_waves + (int)__builtin_amdgcn_readfirstlane(ticket_in);
        TileGeo tg_next = tg;
        int tile_next = tile;
        if (nxt < cur_hi) { tile_next = to_tile(nxt); tg_next = geo(tile_next); }
        // The next tile's pixel rows: 8 loads, in flight behind the appends and the coding.  vmcnt retires in issue order, so
        // the wait at the top of the loop must not have to count a VARYING number of younger stores: whatever path the tile
        // takes, exactly ONE store closes the iteration (the compiler then waits for vmcnt(1), not 0).
        if (nxt < cur_hi && tg_next.interior) {
            request_rows(tg_next, raw);
        } else {                               // (defined on every path -- by an empty asm, i.e. no instruction: else the old rows stay
#pragma unroll                                 //  live through the whole iteration, and zeroing them was hoisted in front of the branch)
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < 6; ++i) asm volatile("" : "=v"(raw[s].d[i]));
        }
        TSTAMP(7);   // ticket, geometry of the next tile, row requests

        // ---- 6. the tile's item list, in LDS (the tile's luma there is dead by now) ----
        // Per item: one SDWA add writes 8 x the zigzag position into the upper half of the value's own register, one LDS
        // write, one address increment.  A list longer than the staging region (noise, very high qualities: up to 65 items
        // per block) is built and coded in two halves of 16 blocks each; the values are modified in place, each lane's in
        // the half its block belongs to.
        uint32_t *const head = out.tile_head + (size_t)tile * kTileHeadWords;
        const auto str_word = [&](uint32_t g) -> uint32_t * {                 // word g of the tile's string: head, then the sparse reservation
            return g < (uint32_t)kTileHeadStr ? head + kTileRecWords + g : out.tile_over + (size_t)tile * kTileOverCap + (g - (uint32_t)kTileHeadStr);
        };
        uint32_t *const stage = &s_pix[wave][0];
        uint32_t *const win = &s_win[wave][0];
        // A list longer than the staging region (noise, very high qualities: up to 65 items per block) is built and coded in PARTS of 16
        // or 8 blocks each.
        const uint32_t items_h0 = (uint32_t)__builtin_amdgcn_readlane((int)incl, 15);
        const int nparts = t_all <= (uint32_t)kStageItemCap ? 1 : (max(items_h0, t_all - items_h0) <= (uint32_t)kStageItemCap ? 2 : 4);
        const int part_shift = nparts == 1 ? 5 : (nparts == 2 ? 4 : 3);                 // blocks per part = 1 << part_shift
        uint32_t cur_bits = 0, wbase = 0, nzrl = 0;          // bits of the tile's string so far; string words already in HBM; ZRL symbols
        uint32_t carry_item = 0;                             // the last item of the pass before
#pragma unroll 1
        for (int part = 0; part < nparts; ++part) {              // (one copy of the code: inlined copies cost more in instruction fetch than they save)
            const uint32_t list_base = part ? (uint32_t)__builtin_amdgcn_readlane((int)incl, (part << part_shift) - 1) : 0u;
            const uint32_t list_end = part == nparts - 1 ? t_all : (uint32_t)__builtin_amdgcn_readlane((int)incl, ((part + 1) << part_shift) - 1);
            const uint32_t nitems = list_end - list_base;
            if (active && (b >> part_shift) == part) {
                const uint32_t stage_addr = (uint32_t)(uintptr_t)stage - list_base * 4u;      // LDS byte address (the low 32 bits of the flat one)
                uint32_t addr = 0;
                uint32_t hh8 = (uint32_t)h << 3;               // (opaque: else the four groups' positions are four more loop invariants, spilled)
                asm volatile("" : "+v"(hh8));
#pragma unroll
                for (int G = 0; G < 4; ++G) {
                    if (G && !alive(G)) continue;
                    addr = stage_addr + (blk_base + ((starts >> (8 * G)) & 0xFFu)) * 4u;
                    const uint32_t zg = (uint32_t)(16 * G) + hh8;
                    if (G == 0) {
                        // site 0: the DC item (always stored) in lanes h == 0, zigzag 8 in lanes h == 1
                        uint32_t first_item = dc_item;
                        if (h) {
                            first_item = (uint32_t)n0 & 0xFFFFu;
                            asm("v_add_u32_sdwa %0, %1, 0 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD"
                                : "+v"(first_item) : "v"(zg));
                        }
                        if (h == 0 || n0 != 0) {
                            asm volatile("ds_write_b32 %0, %1" :: "v"(addr), "v"(first_item) : "memory");
                            addr += 4u;
                        }
                        append_group_lds<true>(addr, reinterpret_cast<uint32_t (&)[4]>(n2[0]), zg);
                    } else {
                        append_group_lds<false>(addr, reinterpret_cast<uint32_t (&)[4]>(n2[4 * G]), zg);
                    }
                }
                if (eob) {
                    if (!alive(3)) addr = stage_addr + (blk_base + (starts >> 24)) * 4u;
                    asm volatile("ds_write_b32 %0, %1" :: "v"(addr), "v"(kItEobValue) : "memory");
                }
            }
            // (everything the coder derives from the lane number comes from a lane id the compiler cannot trace to the kernel's own:
            //  as loop invariants of the tile loop those values were spilled, and reloaded from scratch -- a vector memory
            //  operation -- in the middle of the phases that are supposed to hide the next tile's row loads)
            uint32_t cl;
            asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(cl));
            {   // the list is padded to whole passes with padding items (no bits): the coder then needs no tail masks
                uint32_t padv = kItPadValue;
                asm volatile("" : "+v"(padv));                 // (not a register held across the whole tile loop)
                const uint32_t pi0 = nitems + cl, pend = (nitems + (uint32_t)kPassItems - 1u) & ~(uint32_t)(kPassItems - 1);
                if (pi0 < pend) stage[pi0] = padv;
                if (pi0 + 64u < pend) stage[pi0 + 64u] = padv;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");               // the asm writes are invisible to the compiler's counters
            TSTAMP(8);   // appends

            // The window is written out in the middle of a tile only when the next pass might not fit (very dense tiles).
            const auto make_room = [&](uint32_t add_bits) {
                if (__builtin_expect(((cur_bits + add_bits) >> 5) - wbase + 5u > (uint32_t)kWinStr, 0)) {
                    const uint32_t done = (cur_bits >> 5) - wbase;              // complete words in the window
                    if (done) {
                        const uint32_t part = win[kTileRecWords + done];
                        for (uint32_t j = cl; j < done; j += 64) *str_word(wbase + j) = win[kTileRecWords + j];
#pragma unroll
                        for (int i = 0; i < kWinWords / 64; ++i) win[i * 64 + cl] = 0u;
                        if (cl == 0) win[kTileRecWords] = part;
                        wbase += done;
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // leave no store of a varying count pending
                    }
                }
            };

            // ---- 7. Huffman coding of the list (rle.c:83-123, huffman.c:145-188): two items per lane and pass ----
#pragma unroll 1
            for (uint32_t base = 0; base < nitems; base += (uint32_t)kPassItems) {
                if (nitems - base <= 64u) {                       // the list's tail: one item per lane
                    const uint32_t it1 = stage[base + cl];
                    const uint32_t p1 = (uint32_t)__builtin_amdgcn_update_dpp((int)carry_item, (int)it1, 0x138 /*wave_shr:1*/, 0xF, 0xF, false);
                    uint32_t e1, s1;
                    code_item(it1, p1, s_code, e1, s1);
                    uint32_t l1 = (e1 >> 8) & 0xFFu, h1 = s1, lo1 = 0u;
                    const bool zrl1 = __ballot((e1 & 0x60u) != 0u) != 0ull;
                    if (__builtin_expect(zrl1, 0)) {              // runs >= 16 (rle.c:99-103): ZRL symbols in front
                        const uint32_t z1 = (e1 >> 5) & 3u;
                        l1 += z1 * kZrlBits;
                        nzrl += wave_sum_3bit(z1);
                        zrl_prefix(s1, z1, h1, lo1);
                    }
                    const uint32_t incl1 = wave_incl_scan_u32(l1);
                    const uint32_t bits1 = (uint32_t)__builtin_amdgcn_readlane((int)incl1, 63);
                    // (64 items: <= 64 x 60 bits = 120 words -- make room as the long pass does)
                    make_room(bits1);
                    const uint32_t rel1 = (uint32_t)(kTileRecWords * 32) + cur_bits + incl1 - l1 - wbase * 32u;
                    window_or(win, rel1, h1, lo1, zrl1);
                    cur_bits += bits1;
                    break;
                }
                if (nitems >= (uint32_t)kQuadMinItems && nitems - base > (uint32_t)kPassItems) {
                    // A dense list (noise, high qualities: 12 items and more per block) with more than a pass's worth left: FOUR items per
                    // lane, 256 per pass -- one prefix sum, one window check and one trip round the loop for twice the items: 125
                    // instructions against 2 x 80.  A pass that holds a ZRL falls through to the two-item pass below; sparse lists, where
                    // two passes of three hold one, do not try (Q=50: +0.8 % with the attempt, noise -6 %).
                    const uint4 quad = *reinterpret_cast<const uint4 *>(&stage[base + 4u * cl]);
                    const uint32_t p0 = (uint32_t)__builtin_amdgcn_update_dpp((int)carry_item, (int)quad.w, 0x138 /*wave_shr:1*/, 0xF, 0xF, false);
                    // (a run of 16 and more shows in the positions alone -- row = run + 1 -- before any table lookup: a pass that has one
                    //  costs 14 instructions here; with the test behind the lookups it cost 70, and photo-like tiles have one in two of three)
                    const uint32_t r0 = item_row(quad.x, p0), r1 = item_row(quad.y, quad.x), r2 = item_row(quad.z, quad.y), r3 = item_row(quad.w, quad.z);
                    if (__ballot(max(max(r0, r1), max(r2, r3)) > 16u) == 0ull) {
                        uint32_t e0, e1, e2, e3, s0, s1, s2, s3;
                        code_item_row(quad.x, r0, s_code, e0, s0);
                        code_item_row(quad.y, r1, s_code, e1, s1);
                        code_item_row(quad.z, r2, s_code, e2, s2);
                        code_item_row(quad.w, r3, s_code, e3, s3);
                        carry_item = (uint32_t)__builtin_amdgcn_readlane((int)quad.w, 63);
                        const uint32_t l0 = (e0 >> 8) & 0xFFu, l2 = (e2 >> 8) & 0xFFu;
                        const uint32_t l01 = l0 + ((e1 >> 8) & 0xFFu), l23 = l2 + ((e3 >> 8) & 0xFFu), lsum = l01 + l23;      // <= 54 + 54 bits
                        const uint32_t incl_q = wave_incl_scan_u32(lsum);
                        const uint32_t pass_bits = (uint32_t)__builtin_amdgcn_readlane((int)incl_q, 63);                       // <= 256 x 27 bits = 216 words
                        make_room(pass_bits);
                        // the lane's four strings, joined pairwise (<= 54 bits each), then the second pair behind the first: 128 bits, left-aligned
                        const uint32_t h01 = s0 | __builtin_amdgcn_alignbit(0u, s1, l0), g01 = __builtin_amdgcn_alignbit(s1, 0u, l0);
                        const uint32_t h23 = s2 | __builtin_amdgcn_alignbit(0u, s3, l2), g23 = __builtin_amdgcn_alignbit(s3, 0u, l2);
                        const uint32_t a0 = __builtin_amdgcn_alignbit(0u, h23, l01), a1 = __builtin_amdgcn_alignbit(h23, g23, l01), a2 = __builtin_amdgcn_alignbit(g23, 0u, l01);
                        const bool big = l01 >= 32u;                              // (the funnel shifts use l01 mod 32)
                        const uint32_t w0 = h01 | (big ? 0u : a0), w1 = g01 | (big ? a0 : a1), w2 = big ? a1 : a2, w3 = big ? a2 : 0u;
                        const uint32_t rel = (uint32_t)(kTileRecWords * 32) + cur_bits + incl_q - lsum - wbase * 32u;
                        const uint32_t w = rel >> 5, sh = rel & 31u;
                        atomicOr(&win[w], __builtin_amdgcn_alignbit(0u, w0, sh));
                        atomicOr(&win[w + 1], __builtin_amdgcn_alignbit(w0, w1, sh));
                        atomicOr(&win[w + 2], __builtin_amdgcn_alignbit(w1, w2, sh));
                        atomicOr(&win[w + 3], __builtin_amdgcn_alignbit(w2, w3, sh));
                        atomicOr(&win[w + 4], __builtin_amdgcn_alignbit(w3, 0u, sh));     // (zero for all but the pass's longest strings: an OR of 0 is harmless wherever it lands)
                        cur_bits += pass_bits;
                        base += (uint32_t)kPassItems;
                        continue;
                    }
                }
                const uint2 pair = *reinterpret_cast<const uint2 *>(&stage[base + 2u * cl]);
                const uint32_t ia = pair.x, ib = pair.y;
                // the item in front of a lane's first item: the second item of the lane before (lane 0: the pass before)
                const uint32_t pa = (uint32_t)__builtin_amdgcn_update_dpp((int)carry_item, (int)ib, 0x138 /*wave_shr:1*/, 0xF, 0xF, false);
                carry_item = (uint32_t)__builtin_amdgcn_readlane((int)ib, 63);
                uint32_t ea, eb, sa, sb;
                code_item(ia, pa, s_code, ea, sa);
                code_item(ib, ia, s_code, eb, sb);
                uint32_t la = (ea >> 8) & 0xFFu, lb = (eb >> 8) & 0xFFu;
                const bool any_zrl = __ballot(((ea | eb) & 0x60u) != 0u) != 0ull;
                if (__builtin_expect(any_zrl, 0)) {               // runs >= 16 (rle.c:99-103): ZRL symbols in front, coded below
                    const uint32_t za = (ea >> 5) & 3u, zb = (eb >> 5) & 3u;
                    la += za * kZrlBits;
                    lb += zb * kZrlBits;
                    nzrl += wave_sum_3bit(za + zb);
                }
                const uint32_t lab = la + lb;
                const uint32_t incl_b = wave_incl_scan_u32(lab);
                const uint32_t pass_bits = (uint32_t)__builtin_amdgcn_readlane((int)incl_b, 63);
                make_room(pass_bits);
                const uint32_t rel = (uint32_t)(kTileRecWords * 32) + cur_bits + incl_b - lab - wbase * 32u;
                if (__builtin_expect(!any_zrl, 1)) {
                    // join the lane's two strings: (bits_a : bits_b >> len_a), <= 54 bits
                    const uint32_t hi = sa | __builtin_amdgcn_alignbit(0u, sb, la);
                    const uint32_t lo = __builtin_amdgcn_alignbit(sb, 0u, la);
                    window_or(win, rel, hi, lo, true);
                } else {
                    // symbol by symbol, each with its ZRLs in front (huffman.c:158-188 codes them as ordinary symbols)
                    uint32_t hi, lo;
                    zrl_prefix(sa, (ea >> 5) & 3u, hi, lo);
                    window_or(win, rel, hi, lo, true);
                    zrl_prefix(sb, (eb >> 5) & 3u, hi, lo);
                    window_or(win, rel + la, hi, lo, true);
                }
                cur_bits += pass_bits;
            }
            TSTAMP(9);   // coding
        }

        // ---- 8. record + string leave the kernel ----
        {
            uint32_t cl;
            asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(cl));
            const uint32_t nw = ((cur_bits + 31u) >> 5) - wbase;                // string words still in the window
            const bool whole = wbase == 0u && nw <= (uint32_t)kTileHeadStr;
            if (__builtin_expect(!whole, 0)) {                                  // a long string: its last words go out one by one
                for (uint32_t j = cl; j < nw; j += 64) *str_word(wbase + j) = win[kTileRecWords + j];
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            {   // the record: lane i < 5 writes word i (one 4-byte LDS write; 16-byte writes cost the LDS write path 13 cycles each)
                uint32_t rv = cur_bits;
                asm("v_writelane_b32 %0, %1, 1\n\tv_writelane_b32 %0, %2, 2\n\tv_writelane_b32 %0, %3, 3\n\tv_writelane_b32 %0, %4, 4"
                    : "+v"(rv) : "s"(first_dc), "s"(last_dc), "s"(nexact), "s"(t_all + nzrl));
                if (cl < 5) win[cl] = rv;
            }
            // the closing store: the tile's head -- record + the first 120 string words -- 8 bytes per lane; lanes beyond the string fall to the
            // descriptor's range check (whole 8-byte pieces: the window is zero behind the string)
            typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
            const uint32_t head_bytes = whole ? (((uint32_t)kTileRecWords + nw + 1u) & ~1u) * 4u : (uint32_t)kTileRecWords * 4u;
            const __amdgpu_buffer_rsrc_t crsrc = __builtin_amdgcn_make_buffer_rsrc(head, 0, head_bytes, 0x00020000);
            const u32x2 piece = *reinterpret_cast<const u32x2 *>(&win[cl * 2]);
            __builtin_amdgcn_raw_buffer_store_b64(piece, crsrc, cl * 8u, 0, 0);
            // the window is all zero again: its first 64 words with one 4-byte write, the rest only when the string reached there
            win[cl] = 0u;
            if (__builtin_expect(!(wbase == 0u && nw + (uint32_t)kTileRecWords <= 64u), 0)) {
#pragma unroll
                for (int i = 1; i < kWinWords / 64; ++i) win[i * 64 + cl] = 0u;
            }
        }
#undef JPEGAMD_ACC
        TSTAMP(10);  // record, copy-out
        li = nxt;
        tile = tile_next;
        tg = tg_next;
    }
#ifdef JPEGAMD_STAMPS
    {   // [0..10] phase sums; [11] kernel entry, [12] loop start, [13] loop end in 100 MHz ticks; [14] shader cycles of the loop
        unsigned long long st_rt2, st_c2;
        asm volatile("s_memrealtime %0\n\ts_memtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_rt2), "=s"(st_c2)::"memory");
        if (lane == 0 && out.stamps) {
            unsigned long long *o = out.stamps + (size_t)(blockIdx.x * kWavesT + wave) * 16;
            for (int i = 0; i < 11; ++i) o[i] = s_st[wave][i];
            o[11] = st_rt0; o[12] = st_rt1; o[13] = st_rt2; o[14] = st_c2 - st_c1;
        }
    }
#endif
}

int launch_tile_transform(const ImageDesc &im, const TransformOutM &out, bool taps, void *stream, void *const *ev) {
    // persistent: at most 2 workgroups per CU (16 waves/CU at 4 waves/SIMD), fewer for small images
    const int ntiles = im.tile_end - im.tile_begin;
    if (ntiles <= 0) return 0;
    const int wgs_all = (ntiles + kWavesT - 1) / kWavesT;
    const int wgs = wgs_all < JPEGAMD_TILE_MAX_WGS ? wgs_all : JPEGAMD_TILE_MAX_WGS;
    TileSched sch;
    sch.grp_shift = 0;
    while ((2 << sch.grp_shift) <= wgs && (2 << sch.grp_shift) <= kTileGroups) ++sch.grp_shift;
    const int groups = 1 << sch.grp_shift;
    sch.tiles_per_group = (ntiles + groups - 1) / groups;
    const uint64_t magic = 0x100000000ull / (uint64_t)im.tiles_per_row + 1ull;
    sch.tpr_magic = magic > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)magic;   // tiles_per_row == 1: the correction step makes up for it
    const dim3 grid(wgs), block(64 * kWavesT);
    if (taps) hipLaunchKernelGGL(k_tile_encode<true>, grid, block, 0, (hipStream_t)stream, im, out, sch);
    else if (ev) hipExtLaunchKernelGGL(k_tile_encode<false>, grid, block, 0, (hipStream_t)stream, (hipEvent_t)ev[0], (hipEvent_t)ev[1], 0, im, out, sch);
    else hipLaunchKernelGGL(k_tile_encode<false>, grid, block, 0, (hipStream_t)stream, im, out, sch);
    return (int)hipGetLastError();
}

}  // namespace jpegamd
