/*
 * jpeg_compression.h -- C-ABI of the MI355X-native BMP -> grayscale baseline-JPEG encoder.
 *
 * This is the drop-in boundary for the reference's encode path
 * (strbac-damjan/jpeg-image-compression).  Every entry point cites the reference interface
 * it replaces; file:line are relative to the reference root.  Plain C types only: a
 * maintainer binds these from C (natural_c/src/main.c links unchanged), from ctypes, or
 * from any FFI.  See INTEGRATION.md for the reference-side stubs.
 *
 * Three levels, lowest first:
 *   1. jpegamd_*            device-resident, stream-ordered hot path (what bench.py times)
 *   2. JpegCompression_Init / convertToJpeg(JPEG_COMPRESSION_DTO*)
 *                           the reference's accelerator boundary
 *                           (dsp_port/jpeg_compression/include/jpeg_compression.h:32-77,111)
 *   3. loadBMPImage / saveJPEGGrayscale / stage functions
 *                           the natural_c library surface
 *                           (natural_c/include/bmp_handler.h:37-47, jpeg_handler.h:100-109)
 *
 * There is NO CPU fallback anywhere behind this header: without a HIP device every
 * compute entry point fails with JPEGAMD_ERR_NO_DEVICE (and the natural_c-shaped ones
 * return NULL / false after printing the reason).
 *
 * Threading: one in-flight call per encoder context; JpegCompression_Init once per
 * process (the reference is single-threaded and non-re-entrant as well:
 * natural_c/src/core/huffman.c:9-11,106-117).
 */
#ifndef JPEGAMD_JPEG_COMPRESSION_H
#define JPEGAMD_JPEG_COMPRESSION_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------------------
 * Status codes.  0 / -6 / -8 keep the reference's meaning
 * (dsp_port/jpeg_compression/src/jpeg_compression.c:181,206,214).
 * ---------------------------------------------------------------------------------- */
#define JPEGAMD_OK                0
#define JPEGAMD_ERR_ARG          (-1)  /* NULL / non-positive dims / bad stride */
#define JPEGAMD_ERR_NO_DEVICE    (-2)  /* no HIP device or kernel image not loadable */
#define JPEGAMD_ERR_HIP          (-3)  /* a HIP runtime call failed */
#define JPEGAMD_ERR_NOT_INIT     (-4)  /* JpegCompression_Init() not called */
#define JPEGAMD_ERR_TOO_LARGE    (-5)  /* image exceeds the context's max dims */
#define JPEGAMD_ERR_RLE_CAPACITY (-6)  /* reference: RLE capacity exhausted */
#define JPEGAMD_ERR_BMP          (-7)  /* malformed BMP (magic / bit count / compression / short) */
#define JPEGAMD_ERR_HUFF_CAPACITY (-8) /* reference: Huffman buffer too small */

#define JPEGAMD_JFIF_PREFIX_BYTES 328  /* APP0+DQT+SOF0+DHT+DHT+SOS, natural_c/src/io/jpeg_handler.c:220-233 */

/* ------------------------------------------------------------------------------------
 * Level 1: device-resident hot path
 * ---------------------------------------------------------------------------------- */

/* Pixel source description.  Covers both things the reference feeds its codec:
 *  - BMP file rows as they lie in the file: BGR, bottom-up, stride (3W+3)&~3
 *    (natural_c/src/io/bmp_handler.c:68-75,109-122)      -> JPEGAMD_ORDER_BGR, bottom_up=1
 *  - a loaded BMPImage: RGB, top-down, tightly packed
 *    (natural_c/include/bmp_handler.h:37-41)              -> JPEGAMD_ORDER_RGB, bottom_up=0
 */
#define JPEGAMD_ORDER_BGR 0
#define JPEGAMD_ORDER_RGB 1

typedef struct JpegAmdImage {
    const void *pixels;    /* DEVICE pointer to the first stored row */
    int32_t width;         /* original (unpadded) width,  1..65535 */
    int32_t height;        /* original (unpadded) height, 1..65535 */
    int32_t row_stride;    /* bytes between stored rows (>= 3*width) */
    int32_t bottom_up;     /* 1: stored row 0 is the LAST image row (BMP default) */
    int32_t channel_order; /* JPEGAMD_ORDER_BGR or JPEGAMD_ORDER_RGB */
    int32_t quality;       /* 0 or 50: the reference's only table
                              (natural_c/src/core/jpeg_tables.c:3-12); 1..100 otherwise =
                              libjpeg scaling of that table (extension) */
} JpegAmdImage;

/* Per-call statistics, filled on request (host memory).  The *_ns fields are hipEvent
 * times of the device phases; they play the role of the reference DTO's cycles_* fields
 * (dsp_port/jpeg_compression/include/jpeg_compression.h:55-62). */
typedef struct JpegAmdStats {
    uint64_t jfif_bytes;        /* total bytes written (prefix + segment + EOI) */
    uint64_t entropy_bits;      /* unstuffed entropy-coded bits */
    uint64_t stuffed_bytes;     /* number of 0x00 bytes inserted after 0xFF */
    uint64_t exact_fallbacks;   /* coefficients recomputed in the reference's float order */
    /* With profiling on (jpegamd_encoder_set_profiling) every kernel is launched with its own begin / end events
       (hipExtLaunchKernelGGL): the three figures are the kernels' OWN durations, what a kernel trace shows. */
    uint64_t ns_transform;      /* k_tile_encode: luma + DCT + quantisation + zigzag + run/size symbols + Huffman coding, per tile */
    uint64_t ns_entropy;        /* k_segment_merge: the tiles' bit strings -> one per segment (the field keeps its round-1 name) */
    uint64_t ns_pack;           /* k_finalize: bit / stuffing offsets, stitch, 0xFF stuffing, container */
    uint64_t ns_total;          /* begin of the first kernel .. end of the last: the three durations plus the launch gaps between them */
} JpegAmdStats;

typedef struct JpegAmdEncoder JpegAmdEncoder;   /* opaque; owns device scratch */

/* Create a context able to encode images up to max_width x max_height on the current
 * HIP device.  Scratch is allocated once here (nothing is allocated per call). */
int32_t jpegamd_encoder_create(JpegAmdEncoder **enc, int32_t max_width, int32_t max_height);
int32_t jpegamd_encoder_destroy(JpegAmdEncoder *enc);

/* Upper bound on the JFIF bytes an image of this size can produce (every block at the
 * 1658-bit worst case and every byte stuffed): size `out` with this. */
uint64_t jpegamd_max_jfif_bytes(int32_t width, int32_t height);

/* Enqueue one encode on `stream` (a hipStream_t, or NULL for the default stream).
 *   out_dev        DEVICE buffer receiving the JFIF file bytes
 *   out_capacity   its size in bytes (>= jpegamd_max_jfif_bytes for a guaranteed fit;
 *                  smaller is allowed, overflow is reported through out_size_dev = 0 and
 *                  JPEGAMD_ERR_HUFF_CAPACITY from jpegamd_encoder_finish)
 *   out_size_dev   DEVICE uint64_t receiving the byte count
 *   with_container 1: JFIF prefix + entropy segment + EOI (what saveJPEGGrayscale writes,
 *                  natural_c/src/io/jpeg_handler.c:220-262); 0: entropy segment only
 *                  (what the reference's accelerator hands back,
 *                  dsp_port/jpeg_compression/src/jpeg_compression.c:198-203)
 * Asynchronous: returns once the kernels are enqueued. */
int32_t jpegamd_encode_async(JpegAmdEncoder *enc, const JpegAmdImage *img, void *out_dev,
                             uint64_t out_capacity, uint64_t *out_size_dev,
                             int32_t with_container, void *stream);

/* Enqueue the encode of `count` images (1 .. JPEGAMD_MAX_BATCH) of ONE geometry -- same width, height, row_stride,
 * bottom_up, channel_order and quality, different pixels -- as ONE launch of each kernel: the reference codes one file per
 * process (natural_c/src/main.c:21-24), a server codes many, and small images leave a launch per image mostly idle (a 4096^2
 * image gives every wave of the transform two tiles).  outs_dev[i] / out_sizes_dev[i] receive image i's bytes and byte
 * count; every output has out_capacity bytes.  The context must have been created for at least count x the tiles and
 * segments of one image (e.g. jpegamd_encoder_create(W, count * H) for count W x H images -- max_height may go up to
 * JPEGAMD_MAX_BATCH x 65535 for that purpose); JPEGAMD_ERR_TOO_LARGE otherwise.
 * jpegamd_encoder_finish then reports the LAST image's size and the batch's summed counters. */
#define JPEGAMD_MAX_BATCH 32
int32_t jpegamd_encode_batch_async(JpegAmdEncoder *enc, const JpegAmdImage *imgs, int32_t count, void *const *outs_dev,
                                   uint64_t out_capacity, uint64_t *const *out_sizes_dev, int32_t with_container,
                                   void *stream);

/* Which kernels follow k_tile_encode for whole pictures (no reference counterpart: a tuning knob, results are byte-identical).
 *   PAIR    k_segment_merge + k_finalize: the tiles' bit strings joined per segment, then stitched behind a kernel boundary;
 *   STITCH  k_stitch: one pass, the offsets handed from workgroup to workgroup inside the launch (decoupled look-back);
 *   AUTO    (default) PAIR, and STITCH for pictures of 16 384 segments and more (16384^2 and up), where k_finalize's scan over
 *           every predecessor of every workgroup would grow quadratically.
 * Takes effect with the next encode on the context. */
#define JPEGAMD_PIPELINE_AUTO   0
#define JPEGAMD_PIPELINE_PAIR   1
#define JPEGAMD_PIPELINE_STITCH 2
int32_t jpegamd_encoder_set_pipeline(JpegAmdEncoder *enc, int32_t pipeline);

/* Block until the last enqueued encode on this context finished; optionally fetch stats
 * (stats may be NULL).  Returns JPEGAMD_ERR_HUFF_CAPACITY if the output did not fit. */
int32_t jpegamd_encoder_finish(JpegAmdEncoder *enc, JpegAmdStats *stats);

/* Per-phase hipEvent timing.  slots = 0 turns it off (default).  slots = n keeps a ring of n
 * event sets: encode call i records into slot i % n on ITS stream, so a caller can enqueue
 * many encodes, synchronise once, and read every call's kernel times with
 * jpegamd_encoder_profile (only the *_ns fields are filled). */
int32_t jpegamd_encoder_set_profiling(JpegAmdEncoder *enc, int32_t slots);
int32_t jpegamd_encoder_profile(JpegAmdEncoder *enc, int32_t slot, JpegAmdStats *stats);

/* Stage taps for parity tests and the DTO's debug pointers.  Runs the same device code as
 * the hot path over the whole image and writes, for every 8x8 block in raster block order,
 * into DEVICE buffers (any of them may be NULL):
 *   y_centered   int8  [NB][64]  luma - 128, row-major inside the block  (converter.c:51,84-86)
 *   quant_zigzag int16 [NB][64]  quantised coefficients in zigzag order  (quantization.c:34-36, zigzag.c:51-61)
 *   exact_mask   u64   [NB]      bit k set = zigzag-raster coefficient k took the exact path
 * Synchronous. */
int32_t jpegamd_debug_stages(JpegAmdEncoder *enc, const JpegAmdImage *img, int8_t *y_centered,
                             int16_t *quant_zigzag, uint64_t *exact_mask);

/* Exact-order DCT of arbitrary centred blocks on the device (dct.c:63-96), for the float
 * parity test: in int8 [n][64] (row-major), out float [n][64] (out[u*8+v]).  DEVICE ptrs. */
int32_t jpegamd_debug_dct_exact(JpegAmdEncoder *enc, const int8_t *blocks, float *coeffs, int64_t nblocks);

/* Deterministic integer-only synthetic BMP generator (host code, no device needed); the
 * same generator feeds tests, goldens and bench.py on every machine.
 *   kind: 0 photo-like (smooth + textured regions), 1 uniform noise, 2 flat grey (seed&255),
 *         3 horizontal+vertical gradient
 *   flags bit0: top-down (negative biHeight); bit1: 138-byte offset to pixel data (V5-style)
 * Returns the BMP file size, or 0 if cap is too small (call with out=NULL to query). */
uint64_t jpegamd_synth_bmp(int32_t width, int32_t height, uint32_t seed, int32_t kind,
                           uint32_t flags, uint8_t *out, uint64_t cap);

/* Host-only introspection for tests.  The quantisation table for `quality` (raster order; 50 = the reference's,
 * natural_c/src/core/jpeg_tables.c:3-12; other values = its libjpeg scaling, an extension). */
int32_t jpegamd_debug_quant_table(int32_t quality, uint8_t *table);

/* Constants of the fast quantiser: qmul, qthr, bias float[64] by ZIGZAG position; the rigorous guard band delta, double[64], by raster k. */
int32_t jpegamd_debug_mfma_consts(int32_t quality, float *qmul, float *qthr, float *bias, double *delta);
/* Zero thresholds of the coefficient groups ([group 0..3][lane half 0..1], group G of half h = zigzag 16G+8h .. +7): a tile
 * whose hi-chain LUT sums all stay below grp_thr skips that group's quantiser entirely; lo_bound (may be NULL) is the largest
 * magnitude the lo chain can add to a site of the group -- grp_thr has it taken off. */
int32_t jpegamd_debug_group_thresholds(int32_t quality, float *grp_thr, float *lo_bound);
/* ... and what the uncentred matrix operand (luma 0 .. 255 as binary16 subnormals) adds: zoff[64] / qadd[64] = bias + zoff by zigzag
 * position, the DC row's surplus in accumulator units, the accumulator scale.  Any pointer may be NULL. */
int32_t jpegamd_debug_mfma_offsets(int32_t quality, float *zoff, float *qadd, float *dc_off, float *scale);
/* The six-decimal cosine table the kernels multiply with, [x][u] (natural_c/src/core/dct.c:9-18), for host-side emulations. */
int32_t jpegamd_debug_cos_lut(float *lut);

const char *jpegamd_version(void);

/* ------------------------------------------------------------------------------------
 * Level 2: the reference's accelerator boundary
 * ---------------------------------------------------------------------------------- */

/* Same role and field order as dsp_port/jpeg_compression/include/jpeg_compression.h:32-64.
 * Differences forced by the hardware: the TI planar r / gb "physical" pointers become ONE
 * interleaved device pointer (r_phy_ptr) plus layout fields appended at the end; the
 * cycle counters count nanoseconds.  As in the reference the CALLER owns every buffer and
 * the callee fills rle_count, huff_size and the counters. */
typedef struct JPEG_COMPRESSION_DTO {
    int32_t width;
    int32_t height;

    uint64_t r_phy_ptr;       /* DEVICE address of the interleaved pixel rows */
    uint64_t gb_phy_ptr;      /* unused (kept for layout compatibility), must be 0 */

    /* First-block debug taps, HOST addresses, each may be 0
     * (dsp_port/jpeg_compression/src/jpeg_compression.c:150-169). */
    uint64_t y_phy_ptr;       /* int8  [64] */
    uint64_t dct_phy_ptr;     /* float [64] exact-order DCT of block 0 */
    uint64_t quant_phy_ptr;   /* int16 [64] raster order */
    uint64_t zigzag_phy_ptr;  /* int16 [64] zigzag order */

    uint64_t rle_phy_ptr;     /* unused: symbols never leave the chip; must be 0 */
    uint32_t rle_count;       /* OUT: number of run/size symbols coded */

    uint64_t huff_phy_ptr;    /* DEVICE address receiving the entropy-coded segment */
    uint32_t huff_size;       /* IN: capacity in bytes; OUT: bytes written */

    uint64_t cycles_color_conversion; /* OUT, ns: colour, DCT, quantisation and zigzag run fused in one  */
    uint64_t cycles_dct;              /*   kernel: cycles_dct carries its time, the other three are 0    */
    uint64_t cycles_quantization;
    uint64_t cycles_zigzag;
    uint64_t cycles_rle;              /* OUT, ns: symbol kernel (run/size + Huffman codes per segment)   */
    uint64_t cycles_huffman;          /* OUT, ns: offsets + stitch + stuffing kernels */
    uint64_t cycles_total;            /* OUT, ns */

    /* MI355X additions */
    int32_t row_stride;
    int32_t bottom_up;
    int32_t channel_order;
    int32_t quality;
} JPEG_COMPRESSION_DTO;

/* dsp_port/jpeg_compression/include/jpeg_compression.h:77 (registration of the remote
 * service becomes: pick the HIP device, load kernels, allocate the shared context).
 * Idempotent; returns 0 on success like the reference. */
int32_t JpegCompression_Init(void);
int32_t JpegCompression_DeInit(void);
/* Optional: pre-size the shared context (Init sizes it for 2048x2048, or JPEGAMD_INIT_DIM;
 * convertToJpeg / saveJPEGGrayscale grow it on demand). */
int32_t JpegCompression_Reserve(int32_t max_width, int32_t max_height);

/* dsp_port/jpeg_compression/include/jpeg_compression.h:111.  Synchronous.
 * Returns 0, or -6 / -8 with the reference's meaning, or a JPEGAMD_ERR_* code. */
int32_t convertToJpeg(JPEG_COMPRESSION_DTO *dto);

/* dsp_port/jpeg_compression/include/jpeg_compression.h:72-73 (handler signature). */
int32_t JpegCompression_RemoteServiceHandler(char *service_name, uint32_t cmd, void *prm,
                                             uint32_t prm_size, uint32_t flags);

/* ------------------------------------------------------------------------------------
 * Level 3: the natural_c library surface (same names and signatures, so
 * natural_c/src/main.c compiles and links against this library unchanged)
 * ---------------------------------------------------------------------------------- */

/* natural_c/include/bmp_handler.h:37-41 */
typedef struct BMPImage {
    int32_t width;
    int32_t height;
    uint8_t *data;   /* RGB, top-down, tightly packed; malloc'd */
} BMPImage;

/* natural_c/include/bmp_handler.h:43-45 (src/io/bmp_handler.c:5-129). Host code. */
BMPImage *loadBMPImage(const char *filename);
void freeBMPImage(BMPImage *image);

/* natural_c/include/jpeg_handler.h:107 (src/io/jpeg_handler.c:119-282): runs the whole
 * pipeline on the GPU and writes the file.  Opens the file first, prints the reference's
 * progress lines, returns false on any failure. */
bool saveJPEGGrayscale(const char *filename, const BMPImage *img);

/* In-memory variants used by tests and the CLI (no reference counterpart; they are what
 * loadBMPImage + saveJPEGGrayscale do without the file system).
 * jpegamd_encode_bmp_memory: BMP file bytes (host) -> JFIF file bytes (host).
 * Returns the JFIF size, or a negative JPEGAMD_ERR_* code. */
int64_t jpegamd_encode_bmp_memory(const uint8_t *bmp, uint64_t bmp_len, int32_t quality,
                                  uint8_t *out, uint64_t out_cap);

/* ---- One image sharded over several GPUs by block rows (no reference counterpart) --------
 * Blocks are independent up to the entropy stage, which couples them only through the previous
 * block's DC (rle.c:59-70) and the running bit offset (huffman.c:35-62).  Each rank calls
 * jpegamd_encode_rows_async for its block rows [begin, end) of the SAME image description
 * (it needs the pixel rows of its range and of the one block row above); the unstuffed
 * per-segment bit strings stay in its context.  jpegamd_export_segments packs them densely:
 * `dense_words` (used 32-bit words of the range's segments, back to back), `meta`
 * (jpegamd_segment_meta_words() = 12 uint32 per segment: bits, word offset, first 8 / last 7 bits, symbols,
 * exact-path count, 0, 0, 0, and the counts of 0xFF bytes inside the segment for the 8 byte phases) and the word total.
 * After moving both buffers to the root (RCCL), jpegamd_import_segments places them at their
 * global segment indices in the root's context, and jpegamd_finalize_async stitches all
 * segments: bit offsets, 0xFF stuffing and the zero-padded flush happen once, there.
 * All four are stream-ordered; errors as jpegamd_encode_async. */
int32_t jpegamd_segment_meta_words(void);
int32_t jpegamd_encode_rows_async(JpegAmdEncoder *enc, const JpegAmdImage *img, int32_t block_row_begin,
                                  int32_t block_row_end, void *stream);
int32_t jpegamd_export_segments(JpegAmdEncoder *enc, const JpegAmdImage *img, int32_t block_row_begin,
                                int32_t block_row_end, uint32_t *dense_words_dev,
                                uint64_t dense_capacity_words, uint32_t *meta_dev,
                                uint32_t *total_words_dev, void *stream);
int32_t jpegamd_import_segments(JpegAmdEncoder *enc, const JpegAmdImage *img, int32_t block_row_begin,
                                int32_t block_row_end, const uint32_t *dense_words_dev,
                                const uint32_t *meta_dev, void *stream);
int32_t jpegamd_finalize_async(JpegAmdEncoder *enc, const JpegAmdImage *img, void *out_dev,
                               uint64_t out_capacity, uint64_t *out_size_dev, int32_t with_container,
                               void *stream);

/* ---- Independent images sharded over the GPUs of a node: the exchange step as a C entry (no reference counterpart: its
 * accelerator is one DSP core; the role is that of the host loop in dsp_port/jpeg_client/main.c:397-530) ------------------------
 * One process per GPU encodes its own images with jpegamd_encode_async / _batch_async, pointing image k's output at a staging
 * RECORD: `slot_bytes` long, the stream at offset 0 (capacity slot_bytes - 8), its byte count -- out_size_dev -- in the record's
 * last 8 bytes.  jpegamd_gather_streams then moves the `slots` records of every rank to `root` over RCCL as a gather-v: one
 * all-gather of the size tables, one host wait, one grouped launch of exact-size sends / receives (rounded up to 8 bytes) on
 * `stream`; synchronous.  `rccl_comm` is the caller's ncclComm_t (RCCL is dlopen'ed here, the library does not link it).
 *   sizes_host  HOST, [world][slots], filled on every rank.  A count above slot_bytes - 8 marks a stream the encoder had to
 *               cut: it does not travel; its owner encodes it again into a buffer of that size and sends it by itself.
 *   recv        DEVICE, root only: rank r's streams densely from recv + r * recv_stride on, in record order, each rounded up to
 *               8 bytes (the root's own included).  JPEGAMD_ERR_HUFF_CAPACITY when a rank's total exceeds recv_stride. */
int32_t jpegamd_gather_streams(void *rccl_comm, int32_t rank, int32_t world, int32_t root, const void *records_dev,
                               uint64_t slot_bytes, int32_t slots, uint64_t *sizes_host, void *recv_dev, uint64_t recv_stride,
                               void *stream);

/* File-to-file batch encoding with the host I/O and the PCIe transfers overlapped (no
 * reference counterpart: natural_c/src/main.c:21-24 handles one file, synchronously).
 * Files are processed in order with a few in flight: pinned staging buffers, one HIP stream
 * and one encoder context per slot; a file's read/upload overlaps the encode and the
 * download/write of its neighbours.  status[i] (optional) receives 0 or the JPEGAMD_ERR_*
 * code of file i; a failing file does not stop the batch.  Returns 0 when every file was
 * written, JPEGAMD_ERR_BMP when some failed, another code when nothing could run. */
typedef struct JpegAmdBatchStats {
    int32_t files_ok, files_failed;
    uint64_t bytes_in, bytes_out;       /* BMP file bytes read / JFIF bytes written (files_ok only) */
    double seconds_total;               /* wall time of the call */
    double seconds_read, seconds_write; /* host time spent inside fread / fwrite */
} JpegAmdBatchStats;
int32_t jpegamd_encode_files(const char *const *in_paths, const char *const *out_paths,
                             int32_t count, int32_t quality, int32_t *status,
                             JpegAmdBatchStats *stats);

/* Parse a BMP header the way loadBMPImage does (bmp_handler.c:22-88); fills a JpegAmdImage
 * whose `pixels` is an OFFSET into the file (cast to pointer), for callers that upload the
 * file themselves.  Returns 0 or JPEGAMD_ERR_BMP. */
int32_t jpegamd_parse_bmp(const uint8_t *bmp, uint64_t bmp_len, JpegAmdImage *view,
                          uint64_t *pixel_offset);

#ifdef __cplusplus
}
#endif
#endif /* JPEGAMD_JPEG_COMPRESSION_H */
