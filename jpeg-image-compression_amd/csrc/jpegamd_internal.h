// jpegamd_internal.h -- shared between the HIP kernels and the C-ABI host layer.
// Not installed; the public surface is include/jpeg_compression.h.
#pragma once

#include <stddef.h>
#include <stdint.h>

struct JpegAmdEncoder;
struct JpegAmdImage;

namespace jpegamd {

// ---- geometry of the device pipeline ---------------------------------------------------
// A "segment" is the unit of bitstream ownership: up to 64 consecutive 8x8 blocks of one
// block row, processed by ONE wavefront (one block per lane).
constexpr int kSegBlocks = 64;
constexpr int kWavesPerGroup = 4;                    // 256-thread workgroups
// Worst case bits per block: DC 9+11, 63 x (16+11) AC (quality 100 -> 11-bit amplitudes).
constexpr int kMaxBlockBits = 20 + 63 * 27;          // 1721
constexpr int kMaxBlockWords = (kMaxBlockBits + 31) / 32 + 1;   // 55 (+1: window reads)
constexpr int kPrivWords = 15;                       // per-lane bit words kept in LDS
constexpr int kOvfWords = kMaxBlockWords - kPrivWords + 1;      // rest spills to HBM
constexpr int kSegCapWords = ((kSegBlocks * kMaxBlockBits + 31) / 32 + 1 + 63) / 64 * 64;   // 3456

// Per-coefficient constants of the fast path, raster order k = u*8+v (see quant_consts.cpp).
struct QuantConsts {
    float mult[64];    // M_k = K_k / (q_k * G_k): AAN output -> z = coef / q
    float bias[64];    // delta_k + 0.5
    float thr[64];     // 2 * delta_k
    float qstep[64];   // (float) q_k, for the exact path (quantization.c:35)
};

// Constants of the kernel specialised for the reference's table: one bias for every
// coefficient (so it can live in a VGPR while mult/thr are instruction literals).
struct StdConsts {
    float mult[64];
    float thr[64];     // (bias - 0.5) + delta_k
    float bias;        // 0.5 + max_k delta_k
};

// ---- matrix-pipe transform (jpegamd_transform_mfma.hip) ---------------------------------
// A "tile" is 32 consecutive blocks of one block row, one v_mfma_f32_32x32x16_bf16 column
// each; lane l = (h = l >> 5, b = l & 31) ends up with 32 coefficients of block b: zigzag
// positions 32h + 16H + r for chain H in {0,1}, accumulator register r in [0,16).
constexpr int kTileBlocks = 32;
constexpr int kSegTiles = 8;                         // tiles per segment (= per wavefront)
constexpr int kSegBlocksM = kTileBlocks * kSegTiles; // 256
constexpr int kSegCapWordsM = ((kSegBlocksM * kMaxBlockBits + 31) / 32 + 1 + 63) / 64 * 64;
constexpr int kAFragWords = 3 * 2 * 4 * 64 * 4;      // [term][chain][kstep][lane] x 8 bf16 = 24 KiB
// split pipeline (jpegamd_tile_pipeline.hip): per-tile symbol lists in HBM; slot 0 is a sentinel, 64 words of read slack
constexpr int kTileItemCap = (1 + kTileBlocks * 65 + 64 + 63) / 64 * 64;   // 2176
constexpr int kTileRecord = kTileItemCap - 4;   // the list's last 4 words: {items, DC of the last block, exact-path count, 0}, one 16-byte store
static_assert(1 + kTileBlocks * 65 + 64 <= kTileRecord, "the per-tile record must lie behind the longest list and its read-ahead");

struct MfmaTables {
    uint32_t afrag[kAFragWords];   // LUT-product matrix, 3-way bf16 split (lo, mid, hi), MFMA A-operand order
    float qmul[64];                // by zigzag position z: M_z = K/(q)  (the MFMA output is the plain LUT sum)
    float qthr[64];                // (bias - 0.5) + delta_z
    float qstep[64];               // (float) q, by zigzag position
    float bias;                    // 0.5 + max_z delta_z
    float pad[3];
    float grp_thr[8];              // grouped layout: [group G][lane half h], |LUT sum| below it => zigzag 16G+8h .. +7 all quantise to an unflagged 0
};

struct ScanStats {                   // device-side per-call record (scan kernels + k_pack)
    uint64_t total_bits;
    uint64_t total_syms;
    uint64_t total_exact;
    uint64_t total_ff;
    uint64_t out_size;               // copy of *out_size (k_pack)
    uint32_t status;                 // bit0 = output capacity overflow; zeroed by the bit scan
    uint32_t pad;
};

// What a transform kernel clears for the finalize kernels that follow it on the stream.
struct FinReset {
    ScanStats *stats;           // status word
};

// Post-processing (jpegamd_finalize.hip): bit / stuffing offsets + stitch + stuffing + container, 2 launches.
struct FinalizeArgs {
    const uint32_t *seg_words;
    uint32_t seg_stride;
    const uint32_t *seg_bits;
    const uint8_t *seg_tail;        // [num_segs] last 7 bits of each segment (null: read them from seg_words)
    int32_t num_segs;
    int32_t num_chunks;             // workgroups = ceil(num_segs / 16)
    uint8_t *out;
    uint64_t out_capacity;
    uint64_t *out_size;             // device
    ScanStats *stats;               // device
    const uint8_t *prefix;
    int32_t prefix_len;
    int32_t write_eoi;
    uint32_t *seg_ff;               // [num_segs]  owned 0xFF bytes per segment   (k_fin_count -> k_fin_write)
    uint32_t *chunk_ff;             // [num_chunks] 0xFF bytes per chunk
    unsigned long long *chunk_b0;   // [num_chunks] bit offset of the chunk
};
int launch_finalize(const FinalizeArgs &a, void *stream);
// Segment exchange for one image sharded over GPUs by block rows (jpegamd_finalize.hip): dense copy of the used words of
// segments [s0, s1) + 8 words of metadata per segment (bits, word offset, tail, symbols, exact-path count), and back.
struct SegExchange {
    uint32_t *seg_words; uint32_t seg_stride;
    uint32_t *seg_bits, *seg_syms, *seg_exact; uint8_t *seg_tail;
    int32_t s0, s1;
    uint32_t *dense; uint64_t dense_cap_words;
    uint32_t *meta;                 // [s1 - s0][8]
    uint32_t *total_words;          // [1] (export: written; import: unused)
    uint32_t *status;               // ScanStats::status word: bit 0 set when dense_cap_words was too small
};
int launch_seg_export(const SegExchange &x, void *stream);
int launch_seg_import(const SegExchange &x, void *stream);
int launch_sum_stats(const uint32_t *seg_syms, const uint32_t *seg_exact, int n, ScanStats *stats, void *stream);
int finalize_chunks(int num_segs);

struct ImageDesc {
    const uint8_t *pixels;
    int32_t width, height, row_stride, bottom_up;
    uint32_t weights;          // luma weights for stored bytes 0,1,2 (byte 3 = 0)
    int32_t blocks_w, blocks_h, segs_per_row, num_segs;
    int32_t tiles_per_row, num_tiles;   // 32-block tiles (matrix-pipe kernels)
    int32_t tile_begin, tile_end;       // tiles this launch transforms (whole image: 0, num_tiles; a block-row shard otherwise)
    int32_t seg_begin, seg_end;         // segments this launch codes
    int32_t fast_ok;           // pixels % 4 == 0 && row_stride % 4 == 0
};

struct TransformOut {
    uint32_t *seg_words;       // [num_segs][kSegCapWords] MSB-first bit words
    uint32_t *seg_bits;        // [num_segs]
    uint32_t *seg_syms;        // [num_segs] run/size symbols coded (DTO rle_count)
    uint32_t *seg_exact;       // [num_segs] coefficients recomputed in exact order
    uint32_t *ovf_words;       // [num_segs][kOvfWords][64] private-word overflow
    const uint32_t *huff;      // [256] AC (len<<16|code) + [16] DC
    // stage taps (debug variant only)
    int8_t *tap_y;
    int16_t *tap_zz;
    uint64_t *tap_mask;
    FinReset reset;            // cleared by workgroup 0
};

struct PackArgs {
    const uint32_t *seg_words;
    uint32_t seg_stride;            // words reserved per segment (kSegCapWords or kSegCapWordsM)
    const uint32_t *seg_bits;
    const uint64_t *seg_bitstart;   // [num_segs+1]
    uint32_t *seg_ff;               // [num_segs] (count kernel output)
    const uint64_t *seg_ffstart;    // [num_segs+1]
    int32_t num_segs;
    uint8_t *out;
    uint64_t out_capacity;
    uint64_t *out_size;             // device
    ScanStats *stats;               // device: status bit0 = capacity overflow; out_size copy
    const uint8_t *prefix;          // 328-byte JFIF prefix template (device) or null
    int32_t prefix_len;             // 0 or 328
    int32_t write_eoi;
};


// ---- launchers (jpegamd_kernels.hip) ---------------------------------------------------
// All take a hipStream_t as void* and return a hipError_t as int.
int launch_transform(const ImageDesc &im, const QuantConsts &qc, const TransformOut &out,
                     bool taps, bool std_table, int entropy_backend, void *stream);
int launch_scan_bits(const uint32_t *seg_bits, const uint32_t *seg_syms, const uint32_t *seg_exact,
                     uint64_t *seg_bitstart, int num_segs, ScanStats *stats, void *stream);
int launch_count_ff(const PackArgs &a, void *stream);
int launch_scan_ff(const uint32_t *seg_ff, uint64_t *seg_ffstart, int num_segs, ScanStats *stats,
                   void *stream);
int launch_pack(const PackArgs &a, void *stream);
struct TransformOutM {          // like TransformOut, for the matrix-pipe kernel (segments of 128 blocks)
    uint32_t *seg_words;        // [num_segs][kSegCapWordsM]
    uint32_t *seg_bits, *seg_syms, *seg_exact;
    uint8_t *seg_tail;          // [num_segs] last 7 bits of the segment's bit string (for the next segment's first byte)
    const uint32_t *huff;       // [272]
    const MfmaTables *tables;   // device copy
    unsigned long long *stamps; // [num_segs][16] per-phase cycle sums (diagnostic builds with -DJPEGAMD_STAMPS only)
    // split pipeline only: per-tile outputs of k_tile_transform
    uint32_t *tile_items;       // [num_tiles][kTileItemCap]: word 0 sentinel, items from word 1, per-tile record at kTileRecord
    uint32_t *tile_ctr;         // [64 groups][32 words]: word 0 ticket counter of the dynamic tile hand-out, word 1 waves finished; zero between launches
    FinReset reset;             // cleared by workgroup 0
    int8_t *tap_y;
    int16_t *tap_zz;
    uint64_t *tap_mask;
};
int launch_transform_mfma(const ImageDesc &im, const TransformOutM &out, bool taps, void *stream);
int launch_tile_transform(const ImageDesc &im, const TransformOutM &out, bool taps, void *stream);
struct EntropyArgs {            // k_entropy: per-tile symbol lists -> per-segment bit strings
    const uint32_t *tile_items;     // [num_tiles][kTileItemCap]: sentinel, items, ..., record at kTileRecord
    const uint32_t *huff;
    int32_t num_segs, segs_per_row, tiles_per_row;
    int32_t seg_begin, seg_end;     // segments this launch codes (whole image: 0, num_segs)
    uint32_t *seg_words, *seg_bits, *seg_syms, *seg_exact;
    uint8_t *seg_tail;
};
int launch_entropy(const EntropyArgs &a, void *stream);
int launch_dct_exact(const int8_t *blocks, float *coeffs, int64_t nblocks, void *stream);

// ---- host-side constant derivation (quant_consts.cpp) ----------------------------------
void quant_table_for_quality(int quality, uint8_t table[64]);
void derive_quant_consts(const uint8_t table[64], QuantConsts *qc, double delta_out[64]);
void derive_std_consts(const uint8_t table[64], StdConsts *sc);
void derive_mfma_tables(const uint8_t table[64], MfmaTables *mt, double delta_out[64] /*by raster k, may be null*/,
                        bool grouped = false /*A-row order of the split pipeline: lane (h), site s <-> zigzag 16(s>>3)+8h+(s&7)*/);
bool std_consts_match_baked(const uint8_t table[64]);   // table is the reference's AND baked == derived
void build_huffman_words(uint32_t words[272]);
void cos_lut_copy(float out[64]);              // COS_LUT[x][u] as the kernels use it (natural_c/src/core/dct.c:9-18)
size_t build_jfif_prefix(int width, int height, const uint8_t table[64], uint8_t out[328]);

extern const uint8_t kZigzagHost[64];

// Process-wide context behind JpegCompression_Init / convertToJpeg / saveJPEGGrayscale
// (jpegamd_api.cpp); grows to fit w x h.  Returns nullptr without a usable HIP device.
::JpegAmdEncoder *shared_context(int w, int h, uint64_t **size_dev);
// Block (0,0) stage taps (host outputs): centred luma, exact-order DCT, quantised zigzag.
int32_t first_block_taps(::JpegAmdEncoder *e, const struct ::JpegAmdImage *img, int8_t y[64], float dct[64],
                         int16_t zz[64]);

}  // namespace jpegamd
