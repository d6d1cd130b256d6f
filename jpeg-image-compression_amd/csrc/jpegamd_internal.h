// jpegamd_internal.h -- shared between the HIP kernels and the C-ABI host layer.
// Not installed; the public surface is include/jpeg_compression.h.
#pragma once

#include <stddef.h>
#include <stdint.h>

struct JpegAmdEncoder;
struct JpegAmdImage;

namespace jpegamd {

// ---- geometry of the device pipeline ---------------------------------------------------
// block   8x8 pixels
// tile    32 consecutive blocks of one block row: one wave-iteration of k_tile_encode, one 32-column MFMA operand, one bit string
// segment 8 consecutive tiles of one block row (<= 256 blocks): one wave of k_segment_merge, the unit of bitstream ownership
constexpr int kTileBlocks = 32;
#ifndef JPEGAMD_SEG_TILES
#define JPEGAMD_SEG_TILES 8
#endif
constexpr int kSegTiles = JPEGAMD_SEG_TILES;
constexpr int kSegBlocks = kTileBlocks * kSegTiles;                  // 256
// A launch that codes four or more pictures (jpegamd_encode_batch_async) has waves to spare and uses segments of 16 tiles:
// the per-segment costs of k_segment_merge and k_finalize -- both mostly per-segment bookkeeping -- are paid half as often.
// (A single picture keeps 8: with 16 its two small kernels have too few waves: measured +20 % each.)
constexpr int kSegTilesBatch = 16;
constexpr int kSegTilesMax = kSegTilesBatch;
// Worst case bits per block: DC 9+11, 63 x (16+11) AC (quality 100 -> 11-bit amplitudes).
constexpr int kMaxBlockBits = 20 + 63 * 27;                          // 1721
constexpr int seg_cap_words(int seg_tiles) { return ((kTileBlocks * seg_tiles * kMaxBlockBits + 31) / 32 + 1 + 63) / 64 * 64; }   // words reserved per segment
constexpr int kSegCapWords = seg_cap_words(kSegTiles);
constexpr int kAFragWords = 2 * 2 * 4 * 64 * 4;                      // [term][chain][kstep][lane] x 8 binary16 = 16 KiB
constexpr float kMfmaScale = 1.0f / 8192.0f;                         // the accumulator chains hold kMfmaScale * LUT sum (hi chain + lo chain): the A terms are 2048 K, the B operand is y 2^-24

// Per-tile symbol lists (private to k_tile_encode: built in LDS, coded by the wave that built them, never written out).
// An item is one symbol-to-be, 4 bytes: bits 15..0 the value (int16), bits 21..16 the zigzag position (the producer writes
// the upper half with one SDWA add of an inline constant, and the coder's run is one 16-bit subtraction).  A block's run is: DC item, non-zero AC items, EOB item unless zigzag 63 is non-zero.  DC, EOB and padding
// items have position 0 ("class D": coded from row 0 of the code table, whatever precedes them); a non-zero AC item always
// has position >= 1.
// The padding item: size 12 -- a size the table has no code for -- with all-zero amplitude bits (value -4095: amplitude code
// -4096 = ...1 0000 0000 0000): it yields no bit.  It stands in for the DC of a tile's FIRST block, whose predictor is the last
// block of the tile before (rle.c:59-70): that one symbol is coded by k_segment_merge, which knows both.
constexpr uint32_t kItPadValue = 0xF001u;
// The EOB item (rle.c:121-123) in the same style: size 13, amplitude bits all zero (value -8191); row 0 of the code table
// holds the EOB code under "size 13".
constexpr uint32_t kItEobValue = 0xE001u;
constexpr int kStageWords = 8 * 132;            // a wave's LDS region: the tile's luma stash (8 rows of 132 words), then -- the stash dead -- its item list
constexpr int kStageItemCap = kStageWords - 128;   // items of one PART of a tile: the list is padded to whole passes of 128 items inside the region (928 >= 8 blocks x 65)

// Code table of k_tile_encode (LDS), built once on the host (quant_consts.cpp: build_code_table), independent of the quality.
// Entry (row r, fb) at word kCodeLead + 33 r + fb (odd row stride: symbols of one size and different runs sit in different LDS banks):
//   r   0: class D (fb selects the DC size, 12 = no code, 13 = EOB); r >= 1: AC symbol with run r - 1 (runs >= 16: the entry
//       is that of run & 15 and carries the number of ZRL symbols in front of it, rle.c:99-103)
//   fb  v_ffbh_i32 of twice the amplitude code: 31 - size, or -1 for a zero value (size 0)
//   entry: bits 31..16 the Huffman code LEFT-ALIGNED, 15..8 code length + size, 6..5 ZRLs, 4..0 code length
constexpr int kCodeLead = 32;
constexpr int kCodeRows = 64;
constexpr int kCodeRowStride = 33;
constexpr int kCodeWords = kCodeLead + kCodeRows * kCodeRowStride;   // 2144
constexpr uint32_t kZrlBits = 11, kZrlCode = 0x7F9u;                  // symbol 0xF0: '11111111001' (jpeg_tables.c:36-48)

// Per-tile output of k_tile_encode in HBM: an 8-word record and the tile's bit string (MSB-first): every symbol of the tile
// except the DC of its first block.
//   record: {bits of the string, DC of the first block (absolute), DC of the last block, exact-order fallbacks,
//            symbols (run/size symbols incl. ZRLs and the first DC), 0, 0, 0}
// The record and the first 120 words of the string -- all of it for photo-like content -- form the tile's HEAD, 512 bytes in a
// DENSE array (16 MB per 8192^2 picture: it stays in L2 / Infinity Cache between the two kernels); what lies beyond goes to a
// sparse array reserved for the worst case (only touched bytes cost anything).  Round 2 kept everything in worst-case strides
// of 8.7 KB per tile: every tile a DRAM row and a TLB entry of its own for ~100 useful bytes.
constexpr int kTileRecWords = 8;
constexpr int kTileHeadWords = 128;             // record + first string words: ONE 8-byte-per-lane store
constexpr int kTileHeadStr = kTileHeadWords - kTileRecWords;   // 120
constexpr int kTileOverCap = ((kTileBlocks * kMaxBlockBits + 31) / 32 + 2 + 63) / 64 * 64;   // words reserved per tile behind the head (1728)
// (A dense array for the segments' strings as well was measured: k_finalize 38 -> 51 us per launch of eight pictures.)

struct MfmaTables {
    uint32_t afrag[kAFragWords];   // 2048 x the LUT-product matrix as two integer-valued binary16 terms (lo 2^-11, hi), MFMA A-operand order (the B operand brings 2^-24)
    float qmul[64];                // by zigzag position z: M_z = K / (q * kMfmaScale)  (the MFMA output is kMfmaScale * LUT sum)
    float qthr[64];                // flag threshold 2 (bias_z - 0.5), exactly (the kernel derives it as fma(2, bias_z, -1))
    float qstep[64];               // (float) q, by zigzag position
    float bias[64];                // by zigzag position: 0.5 + delta_z (with margin) -- a band of delta_z on EITHER side of a rounding tie
    float grp_thr[8];              // [group G][lane half h]: |hi-chain output| below it in every site => zigzag 16G+8h .. +7 all quantise to an unflagged 0
                                   // (the lo chain's largest possible contribution, lo_bound, is taken off: the test runs ahead of the add that joins the chains)
    float lo_bound[8];             // [group G][lane half h]: max over the sites of |lo-chain output|, in accumulator units
    float flag_thr[8];             // [group G][lane half h]: max qthr over zigzag 16G+8h .. +7
    float zoff[64];                // by zigzag position: what the quantiser adds besides the bias -- minus (K/q) x the constant the UNCENTRED B operand leaves in the row (five positions; 0 elsewhere)
    float qadd[64];                // bias + zoff: the additive constant of the quantiser's fma, by zigzag position
    float dc_off;                  // what the DC row's accumulator holds beyond the centred pixel sum: kMfmaScale * 64 * 128 (= 1.0)
    float pad[3];
};

struct ScanStats {                   // device-side per-call record
    uint64_t total_bits;
    uint64_t total_syms;
    uint64_t total_exact;
    uint64_t total_ff;
    uint64_t out_size;               // copy of *out_size
    uint32_t status;                 // bit0 = output capacity overflow (sticky until jpegamd_encoder_finish reads it)
    uint32_t pad;
};

constexpr int kMaxBatch = 32;           // images of one geometry coded by ONE launch of each kernel (jpegamd_encode_batch_async)

struct ImageDesc {
    const uint8_t *pixels;              // image 0 (== batch_pixels[0])
    const uint8_t *batch_pixels[kMaxBatch];
    int32_t batch;                      // images in this launch; tiles / segments of image i are [i * num_tiles, ..) / [i * num_segs, ..)
    uint32_t tpi_magic;                 // min(floor(2^32 / num_tiles), 2^32 - 1): global tile / num_tiles by multiply-high, at most one too small
    int32_t width, height, row_stride, bottom_up;
    uint32_t weights;          // luma weights for stored bytes 0,1,2 (byte 3 = 0)
    int32_t blocks_w, blocks_h, segs_per_row, num_segs;
    int32_t seg_tiles;                  // tiles per segment of this launch: kSegTiles, or kSegTilesBatch
    int32_t tiles_per_row, num_tiles;
    int32_t tile_begin, tile_end;       // tiles this launch transforms (whole images: 0, batch * num_tiles; a block-row shard otherwise)
    int32_t seg_begin, seg_end;         // segments this launch codes (whole images: 0, batch * num_segs)
    int32_t fast_ok;           // pixels % 4 == 0 && row_stride % 4 == 0
};

// What k_segment_merge leaves per segment (and what one image sharded over GPUs exchanges, besides the bit strings).
struct SegArrays {
    uint32_t *words;            // [num_segs][words_stride] MSB-first bit string, unstuffed
    uint32_t words_stride;      // seg_cap_words(tiles per segment of the launch)
    uint32_t *bits;             // [num_segs] bit count
    uint32_t *syms;             // [num_segs] run/size symbols coded (DTO rle_count)
    uint32_t *exact;            // [num_segs] coefficients recomputed in exact order
    uint32_t *edge;             // [num_segs] (first 8 bits << 8) | last 7 bits: what the byte straddling two segments is made of
    uint16_t *ffin;             // [num_segs][8] 0xFF bytes lying wholly inside the segment when its first bit sits at byte phase p
    // Per GROUP of kSegGroup consecutive segments (one workgroup of k_segment_merge), so that k_finalize's scan over everything in
    // front of a chunk reads a quarter of the entries: the group's bits, and the 0xFF bytes its segments own when the group's
    // first bit sits at byte phase p -- without the byte straddling the group's start (that needs the segment in front).
    uint32_t *grp_bits;         // [ceil(num_segs / kSegGroup)]
    uint32_t *grp_ff;           // [ceil(num_segs / kSegGroup)][8] (32-bit: sixteen worst-case segments hold more than 65535 bytes)
};
constexpr int kSegGroup = 4;      // (16 -- one k_finalize chunk -- was measured: k_segment_merge's 1024-thread workgroups cost it 60 %)

struct TransformOutM {
    const MfmaTables *tables;   // device copy
    uint32_t *tile_head;        // [num_tiles][kTileHeadWords]: record + first string words of every tile
    uint32_t *tile_over;        // [num_tiles][kTileOverCap]: string words from kTileHeadStr on (index w - kTileHeadStr)
    const uint32_t *code_tab;   // [kCodeWords] device copy of the code table
    uint32_t *tile_ctr;         // [64 groups][32 words]: word 0 = ticket counter of the group's tile hand-out; zero at launch
    uint32_t *tile_ctr_next;    // the set the NEXT launch on this context uses: zeroed by this launch
    unsigned long long *stamps; // per-wave phase cycle sums (launch_tile_transform_stamped only)
    int8_t *tap_y;              // stage taps (debug variant only)
    int16_t *tap_zz;
    uint64_t *tap_mask;
};
// `ev` (optional): two hipEvent_t that receive the kernel's OWN begin / end timestamps (hipExtLaunchKernelGGL), i.e. what a
// kernel trace reports as its duration -- an event recorded in front of a launch also sees the dispatch latency.
int launch_tile_transform(const ImageDesc &im, const TransformOutM &out, bool taps, void *stream, void *const *ev = nullptr);
// the same kernel with its phases stamped (out.stamps must point at 16 words per wave): jpegamd_tile_pipeline.hip, -DJPEGAMD_STAMPED_TU
int launch_tile_transform_stamped(const ImageDesc &im, const TransformOutM &out, bool taps, void *stream, void *const *ev = nullptr);

struct MergeArgs {              // k_segment_merge: the tile strings of a segment -> ONE bit string per segment + its numbers
    const uint32_t *tile_head, *tile_over;
    const uint32_t *huff;           // [272] (len << 16) | code: AC by run/size symbol, then 16 DC sizes
    int32_t num_segs, segs_per_row, tiles_per_row;     // per image
    int32_t seg_tiles;              // tiles per segment: kSegTiles or kSegTilesBatch
    int32_t seg_begin, seg_end;     // segments this launch codes (whole images: 0, batch * num_segs)
    int32_t tiles_per_image;        // a batch: segment s belongs to image s / num_segs, whose tiles start at image * tiles_per_image
    SegArrays seg;
    uint32_t *status;               // ScanStats::status: bit 1 = a tile record did not fit its reservation (JPEGAMD_ERR_RLE_CAPACITY)
};
int launch_segment_merge(const MergeArgs &a, void *stream, void *const *ev = nullptr);

// Post-processing (jpegamd_finalize.hip): global bit / stuffing offsets, stitch, stuffing, container -- ONE launch.
struct FinalizeArgs {
    SegArrays seg;
    int32_t num_segs;               // per image
    int32_t num_chunks;             // per image: workgroups = batch * ceil(num_segs / 16)
    int32_t batch;
    int32_t use_groups;             // the group aggregates are valid for these segments (whole images merged by k_segment_merge, num_segs % kSegGroup == 0)
    uint8_t *out[kMaxBatch];
    uint64_t out_capacity;          // of every output
    uint64_t *out_size[kMaxBatch];  // device
    ScanStats *stats;               // device
    const uint8_t *prefix;
    int32_t prefix_len;
    int32_t write_eoi;
};
int launch_finalize(const FinalizeArgs &a, void *stream, void *const *ev = nullptr);
int finalize_chunks(int num_segs);

// k_stitch (jpegamd_stitch.hip): whole images -- the tiles' strings -> the finished entropy-coded segment in ONE pass (what
// k_segment_merge + k_finalize do in two for the block-row shards of one image over several GPUs).
// Workgroups hand their numbers on in 16-byte granules tagged with the launch's 14-bit epoch (layout: jpegamd_stitch.hip); the
// array is zeroed when the context is created and whenever the epoch wraps, so a granule of an older launch never matches.
struct StitchArgs {
    const uint32_t *tile_head, *tile_over;
    const uint32_t *huff;           // [272] (len << 16) | code: AC by run/size symbol, then 16 DC sizes
    int32_t num_segs, segs_per_row, tiles_per_row;     // per image
    int32_t seg_tiles;              // tiles per segment: kSegTiles or kSegTilesBatch
    int32_t tiles_per_image;
    int32_t batch, wgs_per_image;   // grid = batch * wgs_per_image
    uint32_t epoch;                 // 1 .. 16383
    uint32_t *desc;                 // [batch * wgs_per_image][4] granules
    uint32_t *desc_ffx;             // [batch * wgs_per_image][8] a workgroup's 0xFF counts by byte phase in full (read when a granule's 8-bit counts saturated)
    uint32_t *seg_syms, *seg_exact; // [batch * num_segs] per-segment counters (summed on request)
    uint8_t *out[kMaxBatch];
    uint64_t out_capacity;          // of every output
    uint64_t *out_size[kMaxBatch];  // device
    ScanStats *stats;               // device
    uint32_t *status;               // &stats->status: bit 0 output capacity, bit 1 corrupt tile record, bit 2 a look-back gave up
    const uint8_t *prefix;
    int32_t prefix_len;
    int32_t write_eoi;
};
int launch_stitch(const StitchArgs &a, void *stream, void *const *ev = nullptr);
int stitch_workgroups(int num_segs);

// Segment exchange for one image sharded over GPUs by block rows (jpegamd_finalize.hip): dense copy of the used words of
// segments [s0, s1) + kSegMetaWords words of metadata per segment, and back.
constexpr int kSegMetaWords = 12;   // {bits, word offset, edge, symbols, exact-path count, 0, 0, 0, ffin[0..7] as 4 words}
struct SegExchange {
    SegArrays seg;
    int32_t s0, s1;
    uint32_t *dense; uint64_t dense_cap_words;
    uint32_t *meta;                 // [s1 - s0][kSegMetaWords]
    uint32_t *total_words;          // [1] (export: written; import: unused)
    uint32_t *status;               // ScanStats::status word: bit 0 set when dense_cap_words was too small
};
int launch_seg_export(const SegExchange &x, void *stream);
int launch_seg_import(const SegExchange &x, void *stream);
int launch_sum_stats(const uint32_t *seg_syms, const uint32_t *seg_exact, int n, ScanStats *stats, void *stream);
int launch_dct_exact(const int8_t *blocks, float *coeffs, int64_t nblocks, void *stream);

// ---- host-side constant derivation (quant_consts.cpp) ----------------------------------
void quant_table_for_quality(int quality, uint8_t table[64]);
// A-row order of the pipeline: lane half h, site s <-> zigzag 16(s>>3) + 8h + (s&7)
void derive_mfma_tables(const uint8_t table[64], MfmaTables *mt, double delta_out[64] /*by raster k, may be null*/);
void build_huffman_words(uint32_t words[272]);
void build_code_table(uint32_t words[kCodeWords]);
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
