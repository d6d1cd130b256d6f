// jpegamd_api.cpp -- C-ABI host layer over the HIP kernels (include/jpeg_compression.h).
//
// Level 1 (jpegamd_*): device-resident, stream-ordered encode: k_tile_encode -> k_segment_merge -> k_finalize, or -- very large
//          pictures, where k_finalize's scan over every predecessor would grow quadratically -- k_tile_encode -> k_stitch.
// Level 2 (JpegCompression_Init / convertToJpeg): the reference's accelerator boundary
//          (dsp_port/jpeg_compression/src/jpeg_compression.c:6-33,35-216).
// There is no CPU fallback: without a HIP device every compute entry fails.
#include <hip/hip_runtime_api.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include "jpeg_compression.h"
#include "jpegamd_internal.h"

using namespace jpegamd;

#define HIP_TRY(expr)                                                                           \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess) {                                                                 \
            std::fprintf(stderr, "jpegamd: %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(e_), \
                         __FILE__, __LINE__);                                                   \
            return JPEGAMD_ERR_HIP;                                                             \
        }                                                                                       \
    } while (0)

struct JpegAmdEncoder {
    int device = -1;
    int max_w = 0, max_h = 0, max_segs = 0, max_tiles = 0;
    size_t words_cap = 0;               // words of seg.words
    // device scratch, sized once for max_w x max_h
    SegArrays seg = {};
    uint32_t *huff = nullptr;
    uint8_t *prefix = nullptr;
    ScanStats *stats_dev = nullptr;
    ScanStats mirror;            // host copy of stats_dev, fetched by finish()
    MfmaTables *tables_dev = nullptr;
    MfmaTables *tables_host = nullptr;          // this context's own staging copy (contexts may be driven from different threads)
    uint32_t *tile_head = nullptr, *tile_over = nullptr, *tile_ctr = nullptr, *code_tab = nullptr;
    int ctr_set = 0;                    // which half of tile_ctr the next k_tile_encode launch uses
    uint32_t *desc = nullptr;           // k_stitch's hand-off granules: max_wgs x 16 bytes, then max_wgs x 8 counts in full
    int max_wgs = 0;
    uint32_t epoch = 0;                 // 1 .. 16383: tag of the last k_stitch launch's granules
    int pipeline = JPEGAMD_PIPELINE_AUTO;
    int poison_tile = -1;               // jpegamd_debug_poison_tile_record: the next encode overwrites this tile's record word 0 ...
    uint32_t poison_value = 0;          // ... with this value, between k_tile_encode and k_segment_merge
    unsigned long long *stamps_dev = nullptr;   // per-wave phase cycle sums of the stamped kernel (allocated by the first convertToJpeg, or with JPEGAMD_STAMPS=1)
    bool stamp_next = false;                    // the next k_tile_encode launch is the stamped variant (convertToJpeg)
    // cached constants
    int cur_quality = -1;
    uint8_t qtable[64];
    int prefix_w = -1, prefix_h = -1, prefix_q = -1;
    // profiling: a ring of event quadruples so callers can time many async encodes and read
    // the per-kernel durations after ONE synchronisation
    struct EventSet { hipEvent_t ev[6]; bool merged; };   // begin / end of k_tile_encode, [k_segment_merge,] k_stitch or k_finalize (the kernels' own timestamps)
    std::vector<EventSet> ring;
    uint64_t calls = 0;          // encodes enqueued since profiling was (re)enabled
    int last_slot = -1;
    // last call
    int last_segs = 0;
    hipStream_t last_stream = nullptr;
    bool pending = false;
    bool timed = false;
};

static int segs_for(int w, int h, int *bw, int *bh, int *spr, int seg_tiles = kSegTiles) {
    const int blocks_w = (w + 7) / 8, blocks_h = (h + 7) / 8;
    const int per_row = (blocks_w + kTileBlocks * seg_tiles - 1) / (kTileBlocks * seg_tiles);
    if (bw) *bw = blocks_w;
    if (bh) *bh = blocks_h;
    if (spr) *spr = per_row;
    return blocks_h * per_row;
}

extern "C" const char *jpegamd_version(void) { return "jpegamd 0.4 (gfx950)"; }

extern "C" int32_t jpegamd_segment_meta_words(void) { return kSegMetaWords; }

extern "C" int32_t jpegamd_debug_quant_table(int32_t quality, uint8_t *table) {
    if (!table) return JPEGAMD_ERR_ARG;
    quant_table_for_quality(quality, table);
    return JPEGAMD_OK;
}

// Host-only: the constants of the fast path for `quality` (tests pin them against the oracle's arithmetic).  Reentrant.
extern "C" int32_t jpegamd_debug_mfma_consts(int32_t quality, float *qmul, float *qthr, float *bias, double *delta) {
    uint8_t t[64];
    MfmaTables *mt = new (std::nothrow) MfmaTables;
    if (!mt) return JPEGAMD_ERR_HIP;
    double d[64];
    quant_table_for_quality(quality, t);
    derive_mfma_tables(t, mt, d);
    if (qmul) std::memcpy(qmul, mt->qmul, sizeof(mt->qmul));
    if (qthr) std::memcpy(qthr, mt->qthr, sizeof(mt->qthr));
    if (bias) std::memcpy(bias, mt->bias, sizeof(mt->bias));
    if (delta) std::memcpy(delta, d, sizeof(d));
    delete mt;
    return JPEGAMD_OK;
}

// Host-only: what the UNCENTRED matrix operand adds to the quantiser's constants: qadd = bias + zoff by zigzag position, the DC row's
// surplus in accumulator units, the accumulator scale (kMfmaScale).
extern "C" int32_t jpegamd_debug_mfma_offsets(int32_t quality, float *zoff, float *qadd, float *dc_off, float *scale) {
    uint8_t t[64];
    MfmaTables *mt = new (std::nothrow) MfmaTables;
    if (!mt) return JPEGAMD_ERR_HIP;
    quant_table_for_quality(quality, t);
    derive_mfma_tables(t, mt, nullptr);
    if (zoff) std::memcpy(zoff, mt->zoff, sizeof(mt->zoff));
    if (qadd) std::memcpy(qadd, mt->qadd, sizeof(mt->qadd));
    if (dc_off) *dc_off = mt->dc_off;
    if (scale) *scale = kMfmaScale;
    delete mt;
    return JPEGAMD_OK;
}

extern "C" int32_t jpegamd_debug_group_thresholds(int32_t quality, float *grp_thr /*[4 groups][2 lane halves]*/, float *lo_bound /*same shape, may be NULL*/) {
    uint8_t t[64];
    if (!grp_thr) return JPEGAMD_ERR_ARG;
    MfmaTables *mt = new (std::nothrow) MfmaTables;
    if (!mt) return JPEGAMD_ERR_HIP;
    quant_table_for_quality(quality, t);
    derive_mfma_tables(t, mt, nullptr);
    std::memcpy(grp_thr, mt->grp_thr, sizeof(mt->grp_thr));
    if (lo_bound) std::memcpy(lo_bound, mt->lo_bound, sizeof(mt->lo_bound));
    delete mt;
    return JPEGAMD_OK;
}

extern "C" int32_t jpegamd_debug_cos_lut(float *lut /*[8][8]: COS_LUT[x][u]*/) {
    if (!lut) return JPEGAMD_ERR_ARG;
    cos_lut_copy(lut);
    return JPEGAMD_OK;
}

extern "C" uint64_t jpegamd_max_jfif_bytes(int32_t width, int32_t height) {
    if (width <= 0 || height <= 0) return 0;
    const uint64_t nb = (uint64_t)((width + 7) / 8) * (uint64_t)((height + 7) / 8);
    // every block at the worst-case bit count, every byte stuffed, + container
    return JPEGAMD_JFIF_PREFIX_BYTES + 2 + 2 * ((nb * kMaxBlockBits + 7) / 8 + 1) + 16;
}

// Allocation failures free everything allocated so far (jpegamd_encoder_destroy tolerates a half-built context).
#define HIP_TRY_CREATE(expr)                                                                    \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess) {                                                                 \
            std::fprintf(stderr, "jpegamd: %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(e_), \
                         __FILE__, __LINE__);                                                   \
            jpegamd_encoder_destroy(e);                                                         \
            return JPEGAMD_ERR_HIP;                                                             \
        }                                                                                       \
    } while (0)

extern "C" int32_t jpegamd_encoder_create(JpegAmdEncoder **out, int32_t max_width, int32_t max_height) {
    // (an image is at most 65535 rows -- describe() checks that; a context may reserve room for a batch of them)
    if (!out || max_width <= 0 || max_height <= 0 || max_width > 65535 || max_height > 65535 * kMaxBatch) return JPEGAMD_ERR_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        std::fprintf(stderr, "jpegamd: no HIP device available (this library has no CPU path)\n");
        return JPEGAMD_ERR_NO_DEVICE;
    }
    JpegAmdEncoder *e = new (std::nothrow) JpegAmdEncoder();
    if (!e) return JPEGAMD_ERR_HIP;
    std::memset(&e->mirror, 0, sizeof(e->mirror));
    e->tables_host = new (std::nothrow) MfmaTables;
    if (!e->tables_host) { delete e; return JPEGAMD_ERR_HIP; }
    HIP_TRY_CREATE(hipGetDevice(&e->device));
    e->max_w = max_width; e->max_h = max_height;
    e->max_segs = segs_for(max_width, max_height, nullptr, nullptr, nullptr);
    e->max_tiles = ((max_height + 7) / 8) * (((max_width + 7) / 8 + kTileBlocks - 1) / kTileBlocks);
    const size_t segs = (size_t)e->max_segs + 16;                 // k_finalize reads the per-segment arrays four at a time
    // (room for either segment length: the same blocks as fewer, longer segments need a little more than as many short ones)
    {
        const size_t w8 = (size_t)e->max_segs * kSegCapWords;
        const size_t w16 = (size_t)segs_for(max_width, max_height, nullptr, nullptr, nullptr, kSegTilesBatch) * seg_cap_words(kSegTilesBatch);
        e->words_cap = (w8 > w16 ? w8 : w16) + seg_cap_words(kSegTilesBatch);
    }
    HIP_TRY_CREATE(hipMalloc((void **)&e->seg.words, e->words_cap * sizeof(uint32_t)));
    e->seg.words_stride = kSegCapWords;
    HIP_TRY_CREATE(hipMalloc((void **)&e->seg.bits, segs * sizeof(uint32_t)));
    HIP_TRY_CREATE(hipMalloc((void **)&e->seg.syms, segs * sizeof(uint32_t)));
    HIP_TRY_CREATE(hipMalloc((void **)&e->seg.exact, segs * sizeof(uint32_t)));
    HIP_TRY_CREATE(hipMalloc((void **)&e->seg.edge, segs * sizeof(uint32_t)));
    HIP_TRY_CREATE(hipMalloc((void **)&e->seg.ffin, segs * 8 * sizeof(uint16_t)));
    HIP_TRY_CREATE(hipMalloc((void **)&e->seg.grp_bits, (segs / kSegGroup + 2) * sizeof(uint32_t)));
    HIP_TRY_CREATE(hipMalloc((void **)&e->seg.grp_ff, (segs / kSegGroup + 2) * 8 * sizeof(uint32_t)));
    HIP_TRY_CREATE(hipMalloc((void **)&e->huff, 272 * sizeof(uint32_t)));
    HIP_TRY_CREATE(hipMalloc((void **)&e->prefix, 512));
    HIP_TRY_CREATE(hipMalloc((void **)&e->stats_dev, sizeof(ScanStats)));
    HIP_TRY_CREATE(hipMemset(e->stats_dev, 0, sizeof(ScanStats)));
    HIP_TRY_CREATE(hipMalloc((void **)&e->tables_dev, sizeof(MfmaTables)));
    HIP_TRY_CREATE(hipMalloc((void **)&e->tile_head, ((size_t)e->max_tiles + 1) * kTileHeadWords * sizeof(uint32_t)));
    HIP_TRY_CREATE(hipMalloc((void **)&e->tile_over, (size_t)e->max_tiles * kTileOverCap * sizeof(uint32_t)));
    HIP_TRY_CREATE(hipMalloc((void **)&e->code_tab, kCodeWords * sizeof(uint32_t)));
    HIP_TRY_CREATE(hipMalloc((void **)&e->tile_ctr, 2 * 64 * 128));      // two sets of ticket-group cache lines, used alternately
    HIP_TRY_CREATE(hipMemset(e->tile_ctr, 0, 2 * 64 * 128));
    e->max_wgs = e->max_segs + 2;                                          // (a picture never has more workgroups than segments)
    HIP_TRY_CREATE(hipMalloc((void **)&e->desc, 12 * (size_t)e->max_wgs * sizeof(uint32_t)));
    HIP_TRY_CREATE(hipMemset(e->desc, 0, 12 * (size_t)e->max_wgs * sizeof(uint32_t)));
    if (std::getenv("JPEGAMD_STAMPS")) {
        const size_t n = (size_t)(e->max_segs > 4096 ? e->max_segs : 4096) * 16 * sizeof(unsigned long long);
        HIP_TRY_CREATE(hipMalloc((void **)&e->stamps_dev, n));
        HIP_TRY_CREATE(hipMemset(e->stamps_dev, 0, n));
    }
    uint32_t words[272];
    build_huffman_words(words);
    HIP_TRY_CREATE(hipMemcpy(e->huff, words, sizeof(words), hipMemcpyHostToDevice));
    {
        std::vector<uint32_t> ct(kCodeWords);
        build_code_table(ct.data());
        HIP_TRY_CREATE(hipMemcpy(e->code_tab, ct.data(), kCodeWords * sizeof(uint32_t), hipMemcpyHostToDevice));
    }
    *out = e;
    return JPEGAMD_OK;
}

extern "C" int32_t jpegamd_encoder_destroy(JpegAmdEncoder *e) {
    if (!e) return JPEGAMD_OK;
    if (e->pending) hipStreamSynchronize(e->last_stream);
    hipFree(e->seg.words); hipFree(e->seg.bits); hipFree(e->seg.syms); hipFree(e->seg.exact); hipFree(e->seg.edge); hipFree(e->seg.ffin); hipFree(e->seg.grp_bits); hipFree(e->seg.grp_ff);
    hipFree(e->huff); hipFree(e->prefix); hipFree(e->stats_dev); hipFree(e->tables_dev);
    hipFree(e->tile_head); hipFree(e->tile_over); hipFree(e->code_tab); hipFree(e->tile_ctr); hipFree(e->desc); hipFree(e->stamps_dev);
    for (auto &set : e->ring) for (auto &ev : set.ev) if (ev) hipEventDestroy(ev);
    delete e->tables_host;
    delete e;
    return JPEGAMD_OK;
}

extern "C" int32_t jpegamd_encoder_set_pipeline(JpegAmdEncoder *e, int32_t pipeline) {
    if (!e || pipeline < JPEGAMD_PIPELINE_AUTO || pipeline > JPEGAMD_PIPELINE_STITCH) return JPEGAMD_ERR_ARG;
    e->pipeline = pipeline;
    return JPEGAMD_OK;
}

extern "C" int32_t jpegamd_encoder_set_profiling(JpegAmdEncoder *e, int32_t slots) {
    if (!e || slots < 0 || slots > 65536) return JPEGAMD_ERR_ARG;
    if (e->pending) HIP_TRY(hipStreamSynchronize(e->last_stream));
    for (auto &set : e->ring) for (auto &ev : set.ev) if (ev) hipEventDestroy(ev);
    e->ring.clear();
    e->ring.resize((size_t)slots);
    for (auto &set : e->ring) { set.merged = false; for (auto &ev : set.ev) HIP_TRY(hipEventCreate(&ev)); }
    e->calls = 0;
    e->last_slot = -1;
    return JPEGAMD_OK;
}

static int32_t read_slot(JpegAmdEncoder *e, int slot, JpegAmdStats *stats) {
    if (slot < 0 || (size_t)slot >= e->ring.size()) return JPEGAMD_ERR_ARG;
    hipEvent_t *ev = e->ring[(size_t)slot].ev;
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, ev[0], ev[1])); stats->ns_transform = (uint64_t)((double)ms * 1e6);
    stats->ns_entropy = 0;                                     // (whole images: there is no separate merge kernel)
    if (e->ring[(size_t)slot].merged) { HIP_TRY(hipEventElapsedTime(&ms, ev[2], ev[3])); stats->ns_entropy = (uint64_t)((double)ms * 1e6); }
    HIP_TRY(hipEventElapsedTime(&ms, ev[4], ev[5])); stats->ns_pack = (uint64_t)((double)ms * 1e6);
    HIP_TRY(hipEventElapsedTime(&ms, ev[0], ev[5])); stats->ns_total = (uint64_t)((double)ms * 1e6);     // first begin .. last end: includes the launch gaps
    return JPEGAMD_OK;
}

extern "C" int32_t jpegamd_encoder_profile(JpegAmdEncoder *e, int32_t slot, JpegAmdStats *stats) {
    if (!e || !stats) return JPEGAMD_ERR_ARG;
    std::memset(stats, 0, sizeof(*stats));
    return read_slot(e, slot, stats);
}

static int32_t prepare_constants(JpegAmdEncoder *e, const JpegAmdImage *img, bool need_prefix) {
    const int q = (img->quality <= 0) ? 50 : (img->quality > 100 ? 100 : img->quality);
    if (q != e->cur_quality) {
        quant_table_for_quality(q, e->qtable);
        derive_mfma_tables(e->qtable, e->tables_host, nullptr);
        if (e->pending) HIP_TRY(hipStreamSynchronize(e->last_stream));
        HIP_TRY(hipMemcpy(e->tables_dev, e->tables_host, sizeof(MfmaTables), hipMemcpyHostToDevice));
        e->cur_quality = q;
    }
    if (need_prefix && (e->prefix_w != img->width || e->prefix_h != img->height || e->prefix_q != q)) {
        uint8_t hdr[JPEGAMD_JFIF_PREFIX_BYTES];
        build_jfif_prefix(img->width, img->height, e->qtable, hdr);
        if (e->pending) HIP_TRY(hipStreamSynchronize(e->last_stream));
        HIP_TRY(hipMemcpy(e->prefix, hdr, sizeof(hdr), hipMemcpyHostToDevice));
        e->prefix_w = img->width; e->prefix_h = img->height; e->prefix_q = q;
    }
    return JPEGAMD_OK;
}

// Does a w x h image fit the scratch of `e`?  Every derived count is checked on its own: an image wider than max_w with
// fewer rows can need MORE segments or tiles than max_w x max_h (per-row rounding).
static bool context_fits(const JpegAmdEncoder *e, int w, int h) {
    if (!e || w <= 0 || h <= 0) return false;
    const int bw = (w + 7) / 8, bh = (h + 7) / 8;
    const int tiles = bh * ((bw + kTileBlocks - 1) / kTileBlocks);
    return segs_for(w, h, nullptr, nullptr, nullptr) <= e->max_segs && tiles <= e->max_tiles;
}

static int32_t describe(const JpegAmdEncoder *e, const JpegAmdImage *img, ImageDesc *d, int seg_tiles = kSegTiles) {
    if (!img || !img->pixels || img->width <= 0 || img->height <= 0 || img->width > 65535 || img->height > 65535)
        return JPEGAMD_ERR_ARG;
    if (img->row_stride < 3 * img->width) return JPEGAMD_ERR_ARG;
    if (img->channel_order != JPEGAMD_ORDER_BGR && img->channel_order != JPEGAMD_ORDER_RGB) return JPEGAMD_ERR_ARG;
    d->pixels = (const uint8_t *)img->pixels;
    d->width = img->width; d->height = img->height; d->row_stride = img->row_stride;
    d->bottom_up = img->bottom_up ? 1 : 0;
    // Y = (77 R + 150 G + 29 B) >> 8 (natural_c/src/core/converter.c:51); weights follow the STORED byte order.
    d->weights = img->channel_order == JPEGAMD_ORDER_BGR ? (29u | (150u << 8) | (77u << 16))
                                                         : (77u | (150u << 8) | (29u << 16));
    d->seg_tiles = seg_tiles;
    d->num_segs = segs_for(img->width, img->height, &d->blocks_w, &d->blocks_h, &d->segs_per_row, seg_tiles);
    d->tiles_per_row = (d->blocks_w + kTileBlocks - 1) / kTileBlocks;
    d->num_tiles = d->tiles_per_row * d->blocks_h;
    d->tile_begin = 0; d->tile_end = d->num_tiles;
    d->seg_begin = 0; d->seg_end = d->num_segs;
    d->batch = 1;
    for (int i = 0; i < kMaxBatch; ++i) d->batch_pixels[i] = d->pixels;
    const uint64_t tpi = 0x100000000ull / (uint64_t)d->num_tiles;
    d->tpi_magic = tpi > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)tpi;
    // the dword loader multiplies the row stride by a 24-bit multiply (and keeps a tile's eight row offsets in 32 bits): wider
    // strides -- a region of interest inside a huge buffer -- take the clamped byte loader, whose addresses are 64-bit
    d->fast_ok = ((((uintptr_t)img->pixels) & 3u) == 0 && (img->row_stride & 3) == 0 && img->row_stride < (1 << 24)) ? 1 : 0;
    if (e && !context_fits(e, img->width, img->height)) return JPEGAMD_ERR_TOO_LARGE;
    return JPEGAMD_OK;
}

// k_tile_encode (+ the fault injection of the tests).
static int launch_transform(JpegAmdEncoder *e, const ImageDesc &im, bool taps, int8_t *ty, int16_t *tzz, uint64_t *tmask,
                            void *stream, hipEvent_t *ev = nullptr /*2: begin/end*/) {
    TransformOutM to;
    std::memset(&to, 0, sizeof(to));
    to.tables = e->tables_dev; to.stamps = e->stamps_dev;
    to.tap_y = ty; to.tap_zz = tzz; to.tap_mask = tmask;
    to.tile_head = e->tile_head; to.tile_over = e->tile_over; to.code_tab = e->code_tab;
    // Launches on one context are stream-ordered by contract (they share the scratch): launch i draws tickets from set
    // i % 2 and zeroes the other one for launch i + 1.
    to.tile_ctr = e->tile_ctr + (e->ctr_set ? 64 * 32 : 0);
    to.tile_ctr_next = e->tile_ctr + (e->ctr_set ? 0 : 64 * 32);
    const bool stamped = e->stamp_next && e->stamps_dev && !taps;
    e->stamp_next = false;
    if (int err = stamped ? launch_tile_transform_stamped(im, to, taps, stream, ev ? (void *const *)ev : nullptr)
                          : launch_tile_transform(im, to, taps, stream, (ev && !taps) ? (void *const *)ev : nullptr)) return err;
    if (im.tile_end > im.tile_begin) e->ctr_set ^= 1;      // (an empty range launches nothing)
    if (e->poison_tile >= 0) {                             // fault injection for the tests: a corrupt record must end in a status code
        if (e->poison_tile < e->max_tiles &&
            hipMemsetD32Async((hipDeviceptr_t)(e->tile_head + (size_t)e->poison_tile * kTileHeadWords), (int)e->poison_value, 1, (hipStream_t)stream) != hipSuccess)
            return (int)hipErrorUnknown;
        e->poison_tile = -1;
    }
    return 0;
}

// k_tile_encode, then k_segment_merge (block-row shards, stage taps).
static int launch_transform_and_entropy(JpegAmdEncoder *e, const ImageDesc &im, bool taps, int8_t *ty, int16_t *tzz, uint64_t *tmask,
                                        void *stream, hipEvent_t *ev = nullptr /*4: begin/end of the two kernels*/) {
    if (int err = launch_transform(e, im, taps, ty, tzz, tmask, stream, ev)) return err;
    MergeArgs ea;
    std::memset(&ea, 0, sizeof(ea));
    ea.tile_head = e->tile_head; ea.tile_over = e->tile_over;
    ea.huff = e->huff; ea.num_segs = im.num_segs; ea.segs_per_row = im.segs_per_row; ea.tiles_per_row = im.tiles_per_row;
    ea.seg_tiles = im.seg_tiles;
    ea.seg_begin = im.seg_begin; ea.seg_end = im.seg_end;
    ea.tiles_per_image = im.batch > 1 ? im.num_tiles : 0;
    ea.seg = e->seg;
    ea.seg.words_stride = (uint32_t)seg_cap_words(im.seg_tiles);       // (the stride follows the launch's segment length: carried in the arguments, the context keeps none)
    ea.status = &e->stats_dev->status;
    return launch_segment_merge(ea, stream, ev ? (void *const *)(ev + 2) : nullptr);
}

static int run_finalize_batch(JpegAmdEncoder *e, const ImageDesc &im, void *const *outs_dev, uint64_t out_capacity,
                              uint64_t *const *out_sizes_dev, int32_t with_container, hipStream_t stream, hipEvent_t *ev = nullptr) {
    FinalizeArgs fa;
    std::memset(&fa, 0, sizeof(fa));
    fa.seg = e->seg;
    fa.seg.words_stride = (uint32_t)seg_cap_words(im.seg_tiles);
    fa.num_segs = im.num_segs; fa.num_chunks = finalize_chunks(im.num_segs);
    fa.batch = im.batch;
    fa.use_groups = (im.num_segs % kSegGroup == 0) ? 1 : 0;      // every image then starts on a group boundary
    for (int i = 0; i < im.batch; ++i) { fa.out[i] = (uint8_t *)outs_dev[i]; fa.out_size[i] = out_sizes_dev[i]; }
    fa.out_capacity = out_capacity; fa.stats = e->stats_dev;
    fa.prefix = e->prefix; fa.prefix_len = with_container ? JPEGAMD_JFIF_PREFIX_BYTES : 0;
    fa.write_eoi = with_container ? 1 : 0;
    return launch_finalize(fa, stream, (void *const *)ev);
}

// Which pipeline codes a launch of whole pictures (jpegamd_encoder_set_pipeline): the pair k_segment_merge + k_finalize -- faster
// up to 8192^2-class pictures and on dense content (profiles/r04_notes_experiments.txt) -- or the single-pass k_stitch, whose
// look-back reads at most 256 predecessors per round where k_finalize's scan reads every predecessor of every workgroup.
constexpr int kStitchAutoSegs = 16384;      // 8-tile segments of ONE picture from which AUTO takes k_stitch (a 16384^2 picture)
static bool use_stitch(const JpegAmdEncoder *e, int w, int h) {
    if (e->pipeline == JPEGAMD_PIPELINE_PAIR) return false;
    if (e->pipeline == JPEGAMD_PIPELINE_STITCH) return true;
    return segs_for(w, h, nullptr, nullptr, nullptr) >= kStitchAutoSegs;
}

// k_stitch over the tiles k_tile_encode left: whole images, one or a batch.
static int run_stitch(JpegAmdEncoder *e, const ImageDesc &im, void *const *outs_dev, uint64_t out_capacity,
                      uint64_t *const *out_sizes_dev, int32_t with_container, hipStream_t stream, hipEvent_t *ev = nullptr) {
    StitchArgs sa;
    std::memset(&sa, 0, sizeof(sa));
    sa.tile_head = e->tile_head; sa.tile_over = e->tile_over; sa.huff = e->huff;
    sa.num_segs = im.num_segs; sa.segs_per_row = im.segs_per_row; sa.tiles_per_row = im.tiles_per_row;
    sa.seg_tiles = im.seg_tiles; sa.tiles_per_image = im.num_tiles;
    sa.batch = im.batch; sa.wgs_per_image = stitch_workgroups(im.num_segs);
    if ((int64_t)sa.batch * sa.wgs_per_image > e->max_wgs) return (int)hipErrorInvalidValue;
    // a fresh epoch per launch: granules of older launches never match (no zeroing between launches); the arrays are cleared
    // when the 14-bit tag wraps
    if (e->epoch >= 16383u) {
        if (hipMemsetAsync(e->desc, 0, 12 * (size_t)e->max_wgs * sizeof(uint32_t), stream) != hipSuccess) return (int)hipErrorUnknown;
        e->epoch = 0;
    }
    sa.epoch = ++e->epoch;
    sa.desc = e->desc; sa.desc_ffx = e->desc + 4 * (size_t)e->max_wgs;
    sa.seg_syms = e->seg.syms; sa.seg_exact = e->seg.exact;
    for (int i = 0; i < im.batch; ++i) { sa.out[i] = (uint8_t *)outs_dev[i]; sa.out_size[i] = out_sizes_dev[i]; }
    sa.out_capacity = out_capacity; sa.stats = e->stats_dev; sa.status = &e->stats_dev->status;
    sa.prefix = e->prefix; sa.prefix_len = with_container ? JPEGAMD_JFIF_PREFIX_BYTES : 0;
    sa.write_eoi = with_container ? 1 : 0;
    return launch_stitch(sa, stream, (void *const *)ev);
}

static int run_finalize(JpegAmdEncoder *e, const ImageDesc &im, void *out_dev, uint64_t out_capacity, uint64_t *out_size_dev,
                        int32_t with_container, hipStream_t stream, hipEvent_t *ev = nullptr, bool groups_valid = false) {
    FinalizeArgs fa;
    std::memset(&fa, 0, sizeof(fa));
    fa.seg = e->seg;
    fa.seg.words_stride = (uint32_t)seg_cap_words(im.seg_tiles);
    fa.num_segs = im.num_segs; fa.num_chunks = finalize_chunks(im.num_segs);
    fa.batch = 1;
    fa.use_groups = groups_valid ? 1 : 0;       // (segments imported from other ranks have no group aggregates)
    fa.out[0] = (uint8_t *)out_dev; fa.out_capacity = out_capacity; fa.out_size[0] = out_size_dev; fa.stats = e->stats_dev;
    fa.prefix = e->prefix; fa.prefix_len = with_container ? JPEGAMD_JFIF_PREFIX_BYTES : 0;
    fa.write_eoi = with_container ? 1 : 0;
    return launch_finalize(fa, stream, (void *const *)ev);
}

// ---------------------------------------------------------------------------------------------------------
// One image sharded over several GPUs by block rows.  Each rank transforms and entropy-codes its rows
// (jpegamd_encode_rows_async: the unstuffed per-segment bit strings stay in its scratch), exports them densely
// (jpegamd_export_segments), the root imports every rank's segments at their global indices
// (jpegamd_import_segments) and runs the ordinary finalize over all of them (jpegamd_finalize_async): bit offsets,
// 0xFF stuffing and the zero-padded flush depend on the global byte phase and so happen once, at the root.
// ---------------------------------------------------------------------------------------------------------
static int32_t shard_desc(JpegAmdEncoder *e, const JpegAmdImage *img, int32_t by0, int32_t by1, ImageDesc *im) {
    if (!e) return JPEGAMD_ERR_ARG;
    int32_t rc = describe(e, img, im);
    if (rc) return rc;
    if (by0 < 0 || by1 < by0 || by1 > im->blocks_h) return JPEGAMD_ERR_ARG;
    return JPEGAMD_OK;
}

extern "C" int32_t jpegamd_encode_rows_async(JpegAmdEncoder *e, const JpegAmdImage *img, int32_t block_row_begin,
                                             int32_t block_row_end, void *stream_) {
    ImageDesc im;
    int32_t rc = shard_desc(e, img, block_row_begin, block_row_end, &im);
    if (rc) return rc;
    rc = prepare_constants(e, img, false);
    if (rc) return rc;
    // the DC predictor of the shard's first block is the last block of the row above (rle.c:59-70 chains across rows):
    // that ONE tile is transformed here too (its quantised DC does not depend on anything before it), not coded
    im.tile_begin = block_row_begin * im.tiles_per_row - (block_row_begin > 0 ? 1 : 0);
    im.tile_end = block_row_end * im.tiles_per_row;
    im.seg_begin = block_row_begin * im.segs_per_row;
    im.seg_end = block_row_end * im.segs_per_row;
    hipStream_t stream = (hipStream_t)stream_;
    if (launch_transform_and_entropy(e, im, false, nullptr, nullptr, nullptr, stream, nullptr)) return JPEGAMD_ERR_HIP;
    e->last_segs = im.num_segs;
    e->last_stream = stream;
    e->pending = true;
    e->timed = false;
    return JPEGAMD_OK;
}

static int32_t exchange_args(JpegAmdEncoder *e, const JpegAmdImage *img, int32_t by0, int32_t by1, uint32_t *dense, uint64_t cap,
                             uint32_t *meta, uint32_t *total, SegExchange *x) {
    ImageDesc im;
    int32_t rc = shard_desc(e, img, by0, by1, &im);
    if (rc) return rc;
    if (!dense || !meta) return JPEGAMD_ERR_ARG;
    std::memset(x, 0, sizeof(*x));
    x->seg = e->seg;
    x->seg.words_stride = (uint32_t)kSegCapWords;           // (the sharded path uses the standard segment length)
    x->s0 = by0 * im.segs_per_row; x->s1 = by1 * im.segs_per_row;
    x->dense = dense; x->dense_cap_words = cap; x->meta = meta; x->total_words = total;
    x->status = &e->stats_dev->status;
    return JPEGAMD_OK;
}

extern "C" int32_t jpegamd_export_segments(JpegAmdEncoder *e, const JpegAmdImage *img, int32_t block_row_begin, int32_t block_row_end,
                                           uint32_t *dense_words_dev, uint64_t dense_capacity_words, uint32_t *meta_dev,
                                           uint32_t *total_words_dev, void *stream) {
    SegExchange x;
    if (!total_words_dev) return JPEGAMD_ERR_ARG;
    int32_t rc = exchange_args(e, img, block_row_begin, block_row_end, dense_words_dev, dense_capacity_words, meta_dev, total_words_dev, &x);
    if (rc) return rc;
    if (launch_seg_export(x, stream)) return JPEGAMD_ERR_HIP;
    e->last_stream = (hipStream_t)stream; e->pending = true; e->timed = false;
    return JPEGAMD_OK;
}

extern "C" int32_t jpegamd_import_segments(JpegAmdEncoder *e, const JpegAmdImage *img, int32_t block_row_begin, int32_t block_row_end,
                                           const uint32_t *dense_words_dev, const uint32_t *meta_dev, void *stream) {
    SegExchange x;
    int32_t rc = exchange_args(e, img, block_row_begin, block_row_end, const_cast<uint32_t *>(dense_words_dev), ~0ull,
                               const_cast<uint32_t *>(meta_dev), nullptr, &x);
    if (rc) return rc;
    if (launch_seg_import(x, stream)) return JPEGAMD_ERR_HIP;
    e->last_stream = (hipStream_t)stream; e->pending = true; e->timed = false;
    return JPEGAMD_OK;
}

extern "C" int32_t jpegamd_finalize_async(JpegAmdEncoder *e, const JpegAmdImage *img, void *out_dev, uint64_t out_capacity,
                                          uint64_t *out_size_dev, int32_t with_container, void *stream_) {
    if (!out_dev || !out_size_dev) return JPEGAMD_ERR_ARG;
    ImageDesc im;
    int32_t rc = shard_desc(e, img, 0, 0, &im);
    if (rc) return rc;
    rc = prepare_constants(e, img, with_container != 0);
    if (rc) return rc;
    hipStream_t stream = (hipStream_t)stream_;
    if (run_finalize(e, im, out_dev, out_capacity, out_size_dev, with_container, stream)) return JPEGAMD_ERR_HIP;
    e->last_segs = im.num_segs;
    e->last_stream = stream;
    e->pending = true;
    e->timed = false;
    return JPEGAMD_OK;
}

extern "C" int32_t jpegamd_encode_async(JpegAmdEncoder *e, const JpegAmdImage *img, void *out_dev,
                                        uint64_t out_capacity, uint64_t *out_size_dev, int32_t with_container,
                                        void *stream_) {
    if (!e || !out_dev || !out_size_dev) return JPEGAMD_ERR_ARG;
    ImageDesc im;
    const bool stitch = use_stitch(e, img ? img->width : 0, img ? img->height : 0);
    int32_t rc = describe(e, img, &im, stitch ? kSegTilesBatch : kSegTiles);   // (k_stitch works on segments of 16 tiles)
    if (rc) return rc;
    rc = prepare_constants(e, img, with_container != 0);
    if (rc) return rc;
    hipStream_t stream = (hipStream_t)stream_;

    const bool timed = !e->ring.empty();
    hipEvent_t *ev = nullptr;
    if (timed) {
        e->last_slot = (int)(e->calls % e->ring.size());
        ev = e->ring[(size_t)e->last_slot].ev;
        ++e->calls;
    }
    if (timed) e->ring[(size_t)e->last_slot].merged = !stitch;
    if (stitch) {
        if (launch_transform(e, im, false, nullptr, nullptr, nullptr, stream, ev)) return JPEGAMD_ERR_HIP;
        void *const outs[1] = {out_dev};
        uint64_t *const sizes[1] = {out_size_dev};
        if (run_stitch(e, im, outs, out_capacity, sizes, with_container, stream, ev ? ev + 4 : nullptr)) return JPEGAMD_ERR_HIP;
    } else {
        if (launch_transform_and_entropy(e, im, false, nullptr, nullptr, nullptr, stream, ev)) return JPEGAMD_ERR_HIP;
        if (run_finalize(e, im, out_dev, out_capacity, out_size_dev, with_container, stream, ev ? ev + 4 : nullptr,
                         im.num_segs % kSegGroup == 0)) return JPEGAMD_ERR_HIP;
    }
    e->last_segs = im.num_segs;
    e->last_stream = stream;
    e->pending = true;
    e->timed = timed;
    return JPEGAMD_OK;
}

// `count` images of one geometry through ONE launch of each kernel (see the header): image i's tiles are
// [i * num_tiles, (i + 1) * num_tiles), its segments [i * num_segs, ..); DC prediction, bit offsets and stuffing restart per image.
extern "C" int32_t jpegamd_encode_batch_async(JpegAmdEncoder *e, const JpegAmdImage *imgs, int32_t count, void *const *outs_dev,
                                              uint64_t out_capacity, uint64_t *const *out_sizes_dev, int32_t with_container,
                                              void *stream_) {
    if (!e || !imgs || !outs_dev || !out_sizes_dev || count < 1 || count > kMaxBatch) return JPEGAMD_ERR_ARG;
    ImageDesc im;
    const bool stitch = use_stitch(e, imgs[0].width, imgs[0].height);
    int seg_tiles = (stitch || count >= 4) ? kSegTilesBatch : kSegTiles;  // many pictures: longer segments (jpegamd_internal.h); k_stitch: always
    int32_t rc = describe(e, &imgs[0], &im, seg_tiles);
    if (rc) return rc;
    if (!stitch && seg_tiles != kSegTiles && (size_t)count * im.num_segs * seg_cap_words(seg_tiles) > e->words_cap) {   // (a geometry other than the context's own)
        seg_tiles = kSegTiles;
        rc = describe(e, &imgs[0], &im, seg_tiles);
        if (rc) return rc;
    }
    for (int i = 0; i < count; ++i) {
        const JpegAmdImage &g = imgs[i];
        if (!outs_dev[i] || !out_sizes_dev[i] || !g.pixels) return JPEGAMD_ERR_ARG;
        if (g.width != imgs[0].width || g.height != imgs[0].height || g.row_stride != imgs[0].row_stride ||
            (g.bottom_up != 0) != (imgs[0].bottom_up != 0) || g.channel_order != imgs[0].channel_order || g.quality != imgs[0].quality)
            return JPEGAMD_ERR_ARG;
        im.batch_pixels[i] = (const uint8_t *)g.pixels;
        if ((((uintptr_t)g.pixels) & 3u) != 0) im.fast_ok = 0;
    }
    if ((int64_t)count * im.num_tiles > e->max_tiles || (int64_t)count * im.num_segs > e->max_segs ||
        (!stitch && (size_t)count * im.num_segs * seg_cap_words(seg_tiles) > e->words_cap)) return JPEGAMD_ERR_TOO_LARGE;
    im.batch = count;
    im.tile_end = count * im.num_tiles;
    im.seg_end = count * im.num_segs;
    rc = prepare_constants(e, &imgs[0], with_container != 0);
    if (rc) return rc;
    hipStream_t stream = (hipStream_t)stream_;

    const bool timed = !e->ring.empty();
    hipEvent_t *ev = nullptr;
    if (timed) {
        e->last_slot = (int)(e->calls % e->ring.size());
        ev = e->ring[(size_t)e->last_slot].ev;
        ++e->calls;
    }
    if (timed) e->ring[(size_t)e->last_slot].merged = !stitch;
    if (stitch) {
        if (launch_transform(e, im, false, nullptr, nullptr, nullptr, stream, ev)) return JPEGAMD_ERR_HIP;
        if (run_stitch(e, im, outs_dev, out_capacity, out_sizes_dev, with_container, stream, ev ? ev + 4 : nullptr)) return JPEGAMD_ERR_HIP;
    } else {
        if (launch_transform_and_entropy(e, im, false, nullptr, nullptr, nullptr, stream, ev)) return JPEGAMD_ERR_HIP;
        if (run_finalize_batch(e, im, outs_dev, out_capacity, out_sizes_dev, with_container, stream, ev ? ev + 4 : nullptr)) return JPEGAMD_ERR_HIP;
    }
    e->last_segs = count * im.num_segs;
    e->last_stream = stream;
    e->pending = true;
    e->timed = timed;
    return JPEGAMD_OK;
}

// The capacity status is STICKY on the device: every kernel only ORs into it, and it is cleared here, after it was read.
// A pipelined caller that keeps several encodes in flight on one context therefore learns about an overflow in ANY of
// them (JPEGAMD_ERR_HUFF_CAPACITY, jpeg_compression.c:205-206) at its next finish; which one it was follows from the sizes
// (*out_size_dev holds the would-be size of each call even when it did not fit).
extern "C" int32_t jpegamd_encoder_finish(JpegAmdEncoder *e, JpegAmdStats *stats) {
    if (!e) return JPEGAMD_ERR_ARG;
    if (!e->pending) return JPEGAMD_ERR_ARG;
    if (stats)                          // symbol / exact-path totals are only summed when somebody asks
        if (launch_sum_stats(e->seg.syms, e->seg.exact, e->last_segs, e->stats_dev, e->last_stream)) return JPEGAMD_ERR_HIP;
    HIP_TRY(hipStreamSynchronize(e->last_stream));
    e->pending = false;
    HIP_TRY(hipMemcpy(&e->mirror, e->stats_dev, sizeof(ScanStats), hipMemcpyDeviceToHost));
    if (e->mirror.status) HIP_TRY(hipMemset(&e->stats_dev->status, 0, sizeof(uint32_t)));
    if (stats) {
        std::memset(stats, 0, sizeof(*stats));
        stats->jfif_bytes = e->mirror.out_size;
        stats->entropy_bits = e->mirror.total_bits;
        stats->stuffed_bytes = e->mirror.total_ff;
        stats->exact_fallbacks = e->mirror.total_exact;
        if (e->timed) {
            int32_t rc = read_slot(e, e->last_slot, stats);
            if (rc) return rc;
        }
    }
    if (e->mirror.status & 4u) return JPEGAMD_ERR_HIP;              // a look-back of k_stitch gave up waiting (never seen; the spins are bounded so that it cannot hang)
    if (e->mirror.status & 2u) return JPEGAMD_ERR_RLE_CAPACITY;     // a tile record outside its reservation (corrupt scratch)
    return (e->mirror.status & 1u) ? JPEGAMD_ERR_HUFF_CAPACITY : JPEGAMD_OK;
}

// Symbols coded by the last finished call (DTO rle_count).
static uint64_t last_symbol_count(const JpegAmdEncoder *e) { return e->mirror.total_syms; }

extern "C" int32_t jpegamd_debug_stages(JpegAmdEncoder *e, const JpegAmdImage *img, int8_t *y_centered,
                                        int16_t *quant_zigzag, uint64_t *exact_mask) {
    if (!e) return JPEGAMD_ERR_ARG;
    ImageDesc im;
    int32_t rc = describe(e, img, &im);
    if (rc) return rc;
    rc = prepare_constants(e, img, false);
    if (rc) return rc;
    if (launch_transform_and_entropy(e, im, true, y_centered, quant_zigzag, exact_mask, nullptr)) return JPEGAMD_ERR_HIP;
    HIP_TRY(hipStreamSynchronize(nullptr));
    return JPEGAMD_OK;
}

// Fault injection (tests): the next encode on this context finds `value` in word 0 (the string's bit count) of tile `tile`'s
// record when k_segment_merge reads it.  Not part of the public header.
extern "C" int32_t jpegamd_debug_poison_tile_record(JpegAmdEncoder *e, int32_t tile, uint32_t value) {
    if (!e || tile < 0 || tile >= e->max_tiles) return JPEGAMD_ERR_ARG;
    e->poison_tile = tile;
    e->poison_value = value;
    return JPEGAMD_OK;
}

// Diagnostic: copy the per-wave phase cycle sums of the last launch (null unless JPEGAMD_STAMPS is set
// in the environment AND the library was built with -DJPEGAMD_STAMPS).  Not part of the public header.
extern "C" int32_t jpegamd_debug_read_stamps(JpegAmdEncoder *e, unsigned long long *host, int64_t nwaves) {
    if (!e || !e->stamps_dev || !host || nwaves > (e->max_segs > 4096 ? e->max_segs : 4096)) return JPEGAMD_ERR_ARG;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(host, e->stamps_dev, (size_t)nwaves * 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return JPEGAMD_OK;
}

extern "C" int32_t jpegamd_debug_dct_exact(JpegAmdEncoder *e, const int8_t *blocks, float *coeffs, int64_t nblocks) {
    if (!e || !blocks || !coeffs || nblocks < 0) return JPEGAMD_ERR_ARG;
    if (launch_dct_exact(blocks, coeffs, nblocks, nullptr)) return JPEGAMD_ERR_HIP;
    HIP_TRY(hipStreamSynchronize(nullptr));
    return JPEGAMD_OK;
}

// ---------------------------------------------------------------------------------------
// Level 2: JpegCompression_Init / convertToJpeg
// ---------------------------------------------------------------------------------------
static std::mutex g_mu;
static JpegAmdEncoder *g_ctx = nullptr;
static uint64_t *g_size_dev = nullptr;

static int32_t ensure_ctx(int w, int h) {
    if (g_ctx && context_fits(g_ctx, w, h)) return JPEGAMD_OK;
    int mw = w, mh = h;
    if (g_ctx) {
        if (g_ctx->max_w > mw) mw = g_ctx->max_w;
        if (g_ctx->max_h > mh) mh = g_ctx->max_h;
        jpegamd_encoder_destroy(g_ctx);
        g_ctx = nullptr;
    }
    int32_t rc = jpegamd_encoder_create(&g_ctx, mw, mh);
    if (rc) return rc;
    if (!g_size_dev) HIP_TRY(hipMalloc((void **)&g_size_dev, 16));
    return JPEGAMD_OK;
}

extern "C" int32_t JpegCompression_Init(void) {
    // dsp_port/jpeg_compression/src/jpeg_compression.c:18-33: 0 on success.
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_ctx) return 0;
    const char *env = std::getenv("JPEGAMD_INIT_DIM");
    int dim = env ? std::atoi(env) : 2048;
    if (dim <= 0) dim = 2048;
    return ensure_ctx(dim, dim);
}

extern "C" int32_t JpegCompression_Reserve(int32_t max_width, int32_t max_height) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (max_width <= 0 || max_height <= 0) return JPEGAMD_ERR_ARG;
    return ensure_ctx(max_width, max_height);
}

extern "C" int32_t JpegCompression_DeInit(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_ctx) { jpegamd_encoder_destroy(g_ctx); g_ctx = nullptr; }
    if (g_size_dev) { hipFree(g_size_dev); g_size_dev = nullptr; }
    return 0;
}

// Shared by convertToJpeg and the natural_c-shaped host functions (host_compat.cpp).
namespace jpegamd {
JpegAmdEncoder *shared_context(int w, int h, uint64_t **size_dev) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (ensure_ctx(w, h) != JPEGAMD_OK) return nullptr;
    if (size_dev) *size_dev = g_size_dev;
    return g_ctx;
}
}  // namespace jpegamd

namespace jpegamd {
// Block (0,0) through the stage taps: centred luma, exact-order DCT, quantised zigzag.
int32_t first_block_taps(JpegAmdEncoder *e, const JpegAmdImage *img, int8_t y[64], float dct[64], int16_t zz[64]) {
    ImageDesc im;
    int32_t rc = prepare_constants(e, img, false);
    if (rc) return rc;
    rc = describe(e, img, &im);
    if (rc) return rc;
    im.blocks_w = 1; im.blocks_h = 1; im.segs_per_row = 1; im.num_segs = 1;   // block (0,0) only
    im.tiles_per_row = 1; im.num_tiles = 1; im.tile_begin = 0; im.tile_end = 1; im.seg_begin = 0; im.seg_end = 1;
    int8_t *y_dev = nullptr; int16_t *zz_dev = nullptr; float *dct_dev = nullptr;
    int err = (int)hipMalloc((void **)&y_dev, 64);              // (every exit path below frees what was allocated)
    if (!err) err = (int)hipMalloc((void **)&zz_dev, 128);
    if (!err) err = (int)hipMalloc((void **)&dct_dev, 256);
    if (!err) err = launch_transform_and_entropy(e, im, true, y_dev, zz_dev, nullptr, nullptr);
    if (!err) err = launch_dct_exact(y_dev, dct_dev, 1, nullptr);
    if (!err) err = (int)hipMemcpy(y, y_dev, 64, hipMemcpyDeviceToHost);
    if (!err) err = (int)hipMemcpy(zz, zz_dev, 128, hipMemcpyDeviceToHost);
    if (!err) err = (int)hipMemcpy(dct, dct_dev, 256, hipMemcpyDeviceToHost);
    hipFree(y_dev); hipFree(zz_dev); hipFree(dct_dev);
    return err ? JPEGAMD_ERR_HIP : JPEGAMD_OK;
}
}  // namespace jpegamd

extern "C" int32_t convertToJpeg(JPEG_COMPRESSION_DTO *dto) {
    if (!dto) return JPEGAMD_ERR_ARG;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        if (!g_ctx) return JPEGAMD_ERR_NOT_INIT;
    }
    if (dto->gb_phy_ptr != 0 || dto->rle_phy_ptr != 0) return JPEGAMD_ERR_ARG;
    uint64_t *size_dev = nullptr;
    JpegAmdEncoder *e = shared_context(dto->width > 0 ? dto->width : 1, dto->height > 0 ? dto->height : 1, &size_dev);
    if (!e) return JPEGAMD_ERR_NO_DEVICE;

    JpegAmdImage img;
    img.pixels = (const void *)(uintptr_t)dto->r_phy_ptr;
    img.width = dto->width; img.height = dto->height; img.row_stride = dto->row_stride;
    img.bottom_up = dto->bottom_up; img.channel_order = dto->channel_order; img.quality = dto->quality;

    if (e->ring.empty()) {
        int32_t prc = jpegamd_encoder_set_profiling(e, 1);     // the DTO always reports stage times
        if (prc) return prc;
    }
    // the six stage counters (jpeg_compression.c:188-210) come from the STAMPED variant of the fused kernel: the same code with its
    // phases bracketed by cycle-counter reads (~10 % slower; the asynchronous entry points never run it)
    const size_t stamp_words = (size_t)(e->max_segs > 4096 ? e->max_segs : 4096) * 16;
    if (!e->stamps_dev) {
        HIP_TRY(hipMalloc((void **)&e->stamps_dev, stamp_words * sizeof(unsigned long long)));
    }
    HIP_TRY(hipMemsetAsync(e->stamps_dev, 0, stamp_words * sizeof(unsigned long long), nullptr));
    e->stamp_next = true;
    int32_t rc = jpegamd_encode_async(e, &img, (void *)(uintptr_t)dto->huff_phy_ptr, dto->huff_size, size_dev, 0, nullptr);
    JpegAmdStats st;
    if (rc == JPEGAMD_OK) rc = jpegamd_encoder_finish(e, &st);
    if (rc != JPEGAMD_OK) return rc;   // -8 when huff_size was too small (jpeg_compression.c:205-206)

    dto->huff_size = (uint32_t)st.jfif_bytes;
    dto->rle_count = (uint32_t)last_symbol_count(e);
    // Stage counters (jpeg_compression.c:188-210 fills six), in nanoseconds.  Colour conversion, DCT, quantisation, run/size symbols
    // and Huffman coding are phases of ONE kernel here; this call ran its STAMPED variant (above), whose in-kernel cycle-counter
    // reads split the kernel's duration by the phases' shares of the waves' time.  Zigzag is the row order of the matrix operand
    // (no instruction, no phase): its counter stays 0.  cycles_rle / cycles_huffman also carry the stitching kernels.
    dto->cycles_color_conversion = 0;
    dto->cycles_dct = st.ns_transform;
    dto->cycles_quantization = 0;
    dto->cycles_zigzag = 0;
    dto->cycles_rle = st.ns_entropy;
    dto->cycles_huffman = st.ns_pack;
    dto->cycles_total = st.ns_total;
    if (e->stamps_dev) {
        const int tiles = ((dto->height + 7) / 8) * (((dto->width + 7) / 8 + kTileBlocks - 1) / kTileBlocks);
        const int wgs = (tiles + 7) / 8 < 512 ? (tiles + 7) / 8 : 512;
        std::vector<unsigned long long> stamps((size_t)wgs * 8 * 16);
        if (hipMemcpy(stamps.data(), e->stamps_dev, stamps.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost) == hipSuccess) {
            double ph[11] = {0};
            for (size_t w = 0; w < (size_t)wgs * 8; ++w)
                for (int i = 0; i < 11; ++i) ph[i] += (double)stamps[w * 16 + i];
            double all = 0;
            for (double v : ph) all += v;
            if (all > 0) {
                const double k = (double)st.ns_transform / all;
                // phases: 0 loop, 1 wait rows + luma, 2 luma -> LDS, 3 MFMA, 4 quantise, 5 exact order, 6 counts, 7 ticket / row
                // requests, 8 appends, 9 coding, 10 record / copy-out; the loop's own overhead (0, 7) goes with the colour stage
                dto->cycles_color_conversion = (uint64_t)(k * (ph[0] + ph[1] + ph[2] + ph[7]));
                dto->cycles_dct = (uint64_t)(k * ph[3]);
                dto->cycles_quantization = (uint64_t)(k * (ph[4] + ph[5]));
                dto->cycles_rle = (uint64_t)(k * (ph[6] + ph[8])) + st.ns_entropy;
                dto->cycles_huffman = (uint64_t)(k * (ph[9] + ph[10])) + st.ns_pack;
            }
        }
    }

    // First-block debug taps (jpeg_compression.c:150-169), host pointers.
    if (dto->y_phy_ptr || dto->dct_phy_ptr || dto->quant_phy_ptr || dto->zigzag_phy_ptr) {
        int8_t y[64]; int16_t zz[64]; float dct[64];
        rc = first_block_taps(e, &img, y, dct, zz);
        if (rc) return rc;
        if (dto->y_phy_ptr) std::memcpy((void *)(uintptr_t)dto->y_phy_ptr, y, 64);
        if (dto->dct_phy_ptr) std::memcpy((void *)(uintptr_t)dto->dct_phy_ptr, dct, 256);
        if (dto->zigzag_phy_ptr) std::memcpy((void *)(uintptr_t)dto->zigzag_phy_ptr, zz, 128);
        if (dto->quant_phy_ptr) {
            int16_t raster[64];
            for (int i = 0; i < 64; ++i) raster[kZigzagHost[i]] = zz[i];
            std::memcpy((void *)(uintptr_t)dto->quant_phy_ptr, raster, 128);
        }
    }
    return 0;
}

extern "C" int32_t JpegCompression_RemoteServiceHandler(char *service_name, uint32_t cmd, void *prm, uint32_t prm_size,
                                                        uint32_t flags) {
    // dsp_port/jpeg_compression/src/jpeg_compression.c:6-14: cast and forward.
    (void)service_name; (void)cmd; (void)flags;
    if (!prm || prm_size < sizeof(JPEG_COMPRESSION_DTO)) return JPEGAMD_ERR_ARG;
    return convertToJpeg((JPEG_COMPRESSION_DTO *)prm);
}
