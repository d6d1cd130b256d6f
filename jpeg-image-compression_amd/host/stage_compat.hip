// stage_compat.hip -- natural_c's stage functions (include/natural_c_stages.h) on the GPU.
//
// One small kernel (or a few) per stage, host structs in and out exactly like the reference.  This is the
// verification / debugging surface (each stage can be compared with the reference's or the oracle's output of the
// same stage); it is deliberately simple and is NOT the production path (jpegamd_encode_async fuses all of it).
// No CPU implementation: without a HIP device every function returns NULL.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <cstdlib>
#include <cstring>
#include <vector>

#include "natural_c_stages.h"
#include "jpegamd_device.h"

namespace jpegamd {
int finalize_chunks(int num_segs);
}
using namespace jpegamd;

// The reference rounds every product and every sum separately (dct.c:84); hipcc's default fuses fl(sum + fl(t * c))
// into one fma even through __fmul_rn / __fadd_rn (observed: 1-400 ulp differences in every column but v = 0), so
// this file is built with -ffp-contract=off (Makefile).

namespace {

template <typename T>
struct Dev {                                   // RAII device buffer
    T *p = nullptr;
    size_t n = 0;
    bool alloc(size_t count) { n = count; return hipMalloc((void **)&p, (count ? count : 1) * sizeof(T)) == hipSuccess; }
    bool up(const T *h, size_t count) { return alloc(count) && hipMemcpy(p, h, count * sizeof(T), hipMemcpyHostToDevice) == hipSuccess; }
    bool down(T *h, size_t count) const { return hipMemcpy(h, p, count * sizeof(T), hipMemcpyDeviceToHost) == hipSuccess; }
    ~Dev() { if (p) hipFree(p); }
};

bool have_device() {
    int n = 0;
    return hipGetDeviceCount(&n) == hipSuccess && n > 0;
}

// ---- converter.c:4-58: edge-replicated luma, padded to multiples of 8 ------------------------------------
__global__ void k_st_luma(const uint8_t *__restrict__ rgb, int w, int h, int pw, int ph, uint8_t *__restrict__ y) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)pw * ph) return;
    const int px = (int)(i % pw), py = (int)(i / pw);
    const int sx = min(px, w - 1), sy = min(py, h - 1);                       // converter.c:31,36
    const uint8_t *p = rgb + ((size_t)sy * w + sx) * 3;                        // BMPImage: RGB, top-down, tight
    y[i] = (uint8_t)((77u * p[0] + 150u * p[1] + 29u * p[2]) >> 8);            // converter.c:51
}

// ---- converter.c:60-90 -------------------------------------------------------------------------------------
__global__ void k_st_center(const uint8_t *__restrict__ y, size_t n, int8_t *__restrict__ c) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) c[i] = (int8_t)((int)y[i] - 128);
}

// ---- dct.c:63-151: one wave per block, lane = coefficient (u, v); the 64 terms are added in the reference's order
__global__ __launch_bounds__(64) void k_st_dct(const int8_t *__restrict__ img, int pw, int ph, float *__restrict__ out) {
    __shared__ float s_p[64];
    const int lane = (int)threadIdx.x, u = lane >> 3, v = lane & 7;
    const int bw = pw / 8;
    float cx[8], cy[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { cx[k] = kCosFM[u * 8 + k]; cy[k] = kCosFM[v * 8 + k]; }      // COS_LUT[k][u], COS_LUT[k][v]
    const long long nblk = (long long)bw * (ph / 8);
    for (long long blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        const int by = (int)(blk / bw), bx = (int)(blk % bw);
        __syncthreads();
        s_p[lane] = (float)img[((size_t)by * 8 + u) * pw + (size_t)bx * 8 + v];   // lane (x = u, y = v) loads p[x][y]
        __syncthreads();
        float sum = 0.0f;                                                          // dct.c:68
#pragma unroll
        for (int x = 0; x < 8; ++x)
#pragma unroll
            for (int y = 0; y < 8; ++y) sum = __fadd_rn(sum, __fmul_rn(__fmul_rn(s_p[x * 8 + y], cx[x]), cy[y]));   // dct.c:84
        out[((size_t)by * 8 + u) * pw + (size_t)bx * 8 + v] = __fmul_rn(ref_scale(u, v), sum);                      // dct.c:93
    }
}

// ---- quantization.c:3-43 -------------------------------------------------------------------------------------
__global__ void k_st_quant(const float *__restrict__ f, int pw, size_t n, const float *__restrict__ qstep /*[64] raster*/, int16_t *__restrict__ q) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int x = (int)(i % pw), y = (int)(i / pw);
    q[i] = (int16_t)ref_quantise(f[i], qstep[(y & 7) * 8 + (x & 7)]);
}

// ---- zigzag.c:21-68 -------------------------------------------------------------------------------------------
__global__ void k_st_zigzag(const int16_t *__restrict__ q, int pw, int bw, size_t nblk, int16_t *__restrict__ z) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nblk * 64) return;
    const size_t b = i >> 6;
    const int k = kZZ[i & 63];
    const size_t by = b / bw, bx = b % bw;
    z[i] = q[(by * 8 + (k >> 3)) * pw + bx * 8 + (k & 7)];
}

// ---- rle.c:51-127: thread per block, two passes around a prefix sum ------------------------------------------
__device__ __forceinline__ int st_size(int v) { v = v < 0 ? -v : v; return v ? 32 - __clz(v) : 0; }     // rle.c:9-22
__device__ __forceinline__ uint16_t st_amp(int v, int) { return (uint16_t)(v > 0 ? v : v - 1); }   // rle.c:24-35: NOT masked there, putBits masks (huffman.c:39)

template <bool kWrite>
__global__ void k_st_rle(const int16_t *__restrict__ z, size_t nblk, unsigned long long *__restrict__ count_or_offset, RLESymbol *__restrict__ out) {
    const size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nblk) return;
    const int16_t *c = z + b * 64;
    unsigned long long n = 0;
    RLESymbol *o = kWrite ? out + count_or_offset[b] : nullptr;
    const auto put = [&](uint8_t sym, uint16_t code, uint8_t bits) {
        if (kWrite) { o[n].symbol = sym; o[n].code = code; o[n].codeBits = bits; }
        ++n;
    };
    const int diff = (int)c[0] - (b ? (int)z[(b - 1) * 64] : 0);               // rle.c:59-70 (lastDC chains over all blocks)
    const int ds = st_size(diff);
    put((uint8_t)ds, st_amp(diff, ds), (uint8_t)ds);
    int last = 0;
    for (int k = 63; k >= 1; --k) if (c[k] != 0) { last = k; break; }           // rle.c:83-89
    int run = 0;
    for (int k = 1; k <= last; ++k) {
        if (c[k] == 0) { ++run; continue; }
        while (run >= 16) { put(0xF0, 0, 0); run -= 16; }                       // rle.c:99-103
        const int s = st_size(c[k]);
        put((uint8_t)((run << 4) | s), st_amp(c[k], s), (uint8_t)s);            // rle.c:110
        run = 0;
    }
    if (last < 63) put(0x00, 0, 0);                                             // rle.c:121-123
    if (!kWrite) count_or_offset[b] = n;
}

// ---- huffman.c:121-193 -----------------------------------------------------------------------------------------
// Which table a symbol uses depends on a walk over the stream (first symbol of a block = DC; a block ends at EOB
// or when 63 AC coefficients are accounted for).  One wave walks it: 64 symbols per coalesced load, a scalar loop
// over the lanes with v_readlane, the DC flags of the chunk collected in a 64-bit mask.
__global__ __launch_bounds__(64) void k_st_walk(const RLESymbol *__restrict__ sym, unsigned long long n, int total_blocks,
                                                uint8_t *__restrict__ kind /*1 DC, 2 AC, 0 ignored*/) {
    const int lane = (int)threadIdx.x;
    int block = 0, coeffs = 64;                    // coeffs == 64: the next symbol opens a block (is DC)
    bool done = total_blocks <= 0;
    for (unsigned long long base = 0; base < n; base += 64) {
        const unsigned long long i = base + (unsigned long long)lane;
        const int s = i < n ? (int)sym[i].symbol : 0;
        unsigned long long dc = 0, used = 0;
        const int m = (int)((n - base) < 64ull ? (n - base) : 64ull);
        for (int j = 0; j < m && !done; ++j) {
            const int sj = __builtin_amdgcn_readlane(s, j);
            used |= 1ull << j;
            if (coeffs >= 64) {                    // DC of a new block
                dc |= 1ull << j;
                coeffs = 1;
                ++block;
            } else if (sj == 0x00) {
                coeffs = 64;                       // EOB
            } else if (sj == 0xF0) {
                coeffs += 16;
            } else {
                coeffs += ((sj >> 4) & 15) + 1;
            }
            if (coeffs >= 64 && block >= total_blocks) done = true;            // the reference's outer loop ends here
        }
        if (i < n) kind[i] = ((used >> lane) & 1ull) ? (((dc >> lane) & 1ull) ? 1 : 2) : 0;
    }
}

__global__ void k_st_bits(const RLESymbol *__restrict__ sym, const uint8_t *__restrict__ kind, unsigned long long n,
                          const uint32_t *__restrict__ huff /*[272] len<<16|code: AC then DC*/, unsigned long long *__restrict__ len) {
    const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int k = kind[i];
    const uint32_t hc = k == 1 ? huff[256 + (sym[i].symbol & 15)] : huff[sym[i].symbol];
    len[i] = k ? (hc >> 16) + sym[i].codeBits : 0;
}

__global__ void k_st_pack(const RLESymbol *__restrict__ sym, const uint8_t *__restrict__ kind, unsigned long long n, const uint32_t *__restrict__ huff,
                          const unsigned long long *__restrict__ off, uint32_t *__restrict__ words) {
    const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || kind[i] == 0) return;
    const uint32_t hc = kind[i] == 1 ? huff[256 + (sym[i].symbol & 15)] : huff[sym[i].symbol];
    const int bits = (int)(hc >> 16) + sym[i].codeBits;
    if (bits == 0) return;
    const unsigned long long amp = (unsigned long long)(sym[i].code & ((1u << sym[i].codeBits) - 1u));      // huffman.c:39
    const unsigned long long val = ((unsigned long long)(hc & 0xFFFFu) << sym[i].codeBits) | amp;            // huffman.c:145-176
    const unsigned long long pos = off[i];
    const unsigned long long w = pos >> 5;
    const int sh = (int)(pos & 31);
    const unsigned long long placed = (val << (64 - bits)) >> sh;              // MSB-first, bits <= 27
    atomicOr(&words[w], (uint32_t)(placed >> 32));
    if ((uint32_t)placed) atomicOr(&words[w + 1], (uint32_t)placed);
}


template <typename S>
S *make_struct() { return (S *)std::calloc(1, sizeof(S)); }

inline unsigned grid_for(size_t n, unsigned block = 256) { return (unsigned)((n + block - 1) / block); }

}  // namespace

#define ST_FAIL(obj, freefn) do { freefn(obj); return nullptr; } while (0)

extern "C" void freeYImage(YImage *img) { if (img) { std::free(img->data); std::free(img); } }
extern "C" void freeCenteredYImage(CenteredYImage *img) { if (img) { std::free(img->data); std::free(img); } }
extern "C" void freeDCTImage(DCTImage *img) { if (img) { std::free(img->coefficients); std::free(img); } }
extern "C" void freeQuantizedImage(QuantizedImage *img) { if (img) { std::free(img->data); std::free(img); } }
extern "C" void freeZigZagData(ZigZagData *z) { if (z) { std::free(z->data); std::free(z); } }
extern "C" void freeRLEData(RLEData *r) { if (r) { std::free(r->data); std::free(r); } }
extern "C" void freeJpegEncoderBuffer(JpegEncoderBuffer *b) { if (b) { std::free(b->data); std::free(b); } }

extern "C" YImage *convertBMPToJPEGGrayscale(const BMPImage *image) {
    if (!image || !image->data || image->width <= 0 || image->height <= 0 || !have_device()) return nullptr;
    const int w = image->width, h = image->height, pw = (w + 7) & ~7, ph = (h + 7) & ~7;
    const size_t n = (size_t)pw * ph;
    YImage *y = make_struct<YImage>();
    if (!y) return nullptr;
    y->width = pw; y->height = ph;
    y->data = (uint8_t *)std::malloc(n);
    Dev<uint8_t> src, dst;
    if (!y->data || !src.up(image->data, (size_t)w * h * 3) || !dst.alloc(n)) ST_FAIL(y, freeYImage);
    hipLaunchKernelGGL(k_st_luma, dim3(grid_for(n)), dim3(256), 0, nullptr, src.p, w, h, pw, ph, dst.p);
    if (hipGetLastError() != hipSuccess || !dst.down(y->data, n)) ST_FAIL(y, freeYImage);
    return y;
}

extern "C" CenteredYImage *centerYImage(const YImage *source) {
    if (!source || !source->data || source->width <= 0 || source->height <= 0 || !have_device()) return nullptr;
    const size_t n = (size_t)source->width * source->height;
    CenteredYImage *c = make_struct<CenteredYImage>();
    if (!c) return nullptr;
    c->width = source->width; c->height = source->height;
    c->data = (int8_t *)std::malloc(n);
    Dev<uint8_t> src; Dev<int8_t> dst;
    if (!c->data || !src.up(source->data, n) || !dst.alloc(n)) ST_FAIL(c, freeCenteredYImage);
    hipLaunchKernelGGL(k_st_center, dim3(grid_for(n)), dim3(256), 0, nullptr, src.p, n, dst.p);
    if (hipGetLastError() != hipSuccess || !dst.down(c->data, n)) ST_FAIL(c, freeCenteredYImage);
    return c;
}

extern "C" DCTImage *performDCT(const CenteredYImage *image) {
    if (!image || !image->data || image->width <= 0 || image->height <= 0 || (image->width & 7) || (image->height & 7) || !have_device()) return nullptr;
    const size_t n = (size_t)image->width * image->height;
    DCTImage *d = make_struct<DCTImage>();
    if (!d) return nullptr;
    d->width = image->width; d->height = image->height;
    d->coefficients = (float *)std::malloc(n * sizeof(float));
    Dev<int8_t> src; Dev<float> dst;
    if (!d->coefficients || !src.up(image->data, n) || !dst.alloc(n)) ST_FAIL(d, freeDCTImage);
    const long long nblk = (long long)(n / 64);
    hipLaunchKernelGGL(k_st_dct, dim3((unsigned)(nblk < 16384 ? nblk : 16384)), dim3(64), 0, nullptr, src.p, image->width, image->height, dst.p);
    if (hipGetLastError() != hipSuccess || !dst.down(d->coefficients, n)) ST_FAIL(d, freeDCTImage);
    return d;
}

extern "C" void computeDCTBlock(const int8_t inputBlock[8][8], float outputBlock[8][8]) {
    if (!inputBlock || !outputBlock) return;
    CenteredYImage in;
    in.width = 8; in.height = 8; in.data = const_cast<int8_t *>(&inputBlock[0][0]);
    DCTImage *d = performDCT(&in);
    if (!d) return;
    std::memcpy(&outputBlock[0][0], d->coefficients, 64 * sizeof(float));
    freeDCTImage(d);
}

extern "C" QuantizedImage *quantizeImage(const DCTImage *dctImg) {
    if (!dctImg || !dctImg->coefficients || dctImg->width <= 0 || dctImg->height <= 0 || !have_device()) return nullptr;
    const size_t n = (size_t)dctImg->width * dctImg->height;
    QuantizedImage *q = make_struct<QuantizedImage>();
    if (!q) return nullptr;
    q->width = dctImg->width; q->height = dctImg->height;
    q->data = (int16_t *)std::malloc(n * sizeof(int16_t));
    uint8_t table[64];
    float qf[64];
    quant_table_for_quality(50, table);                     // the reference's only table (jpeg_tables.c:3-12)
    for (int i = 0; i < 64; ++i) qf[i] = (float)table[i];
    Dev<float> src, dq; Dev<int16_t> dst;
    if (!q->data || !src.up(dctImg->coefficients, n) || !dq.up(qf, 64) || !dst.alloc(n)) ST_FAIL(q, freeQuantizedImage);
    hipLaunchKernelGGL(k_st_quant, dim3(grid_for(n)), dim3(256), 0, nullptr, src.p, dctImg->width, n, dq.p, dst.p);
    if (hipGetLastError() != hipSuccess || !dst.down(q->data, n)) ST_FAIL(q, freeQuantizedImage);
    return q;
}

extern "C" ZigZagData *performZigZag(const QuantizedImage *qImg) {
    if (!qImg || !qImg->data || qImg->width <= 0 || qImg->height <= 0 || (qImg->width & 7) || (qImg->height & 7) || !have_device()) return nullptr;
    const size_t n = (size_t)qImg->width * qImg->height;
    ZigZagData *z = make_struct<ZigZagData>();
    if (!z) return nullptr;
    z->numBlocksW = qImg->width / 8; z->numBlocksH = qImg->height / 8; z->totalBlocks = z->numBlocksW * z->numBlocksH;   // zigzag.c:30-32
    z->data = (int16_t *)std::malloc(n * sizeof(int16_t));
    Dev<int16_t> src, dst;
    if (!z->data || !src.up(qImg->data, n) || !dst.alloc(n)) ST_FAIL(z, freeZigZagData);
    hipLaunchKernelGGL(k_st_zigzag, dim3(grid_for(n)), dim3(256), 0, nullptr, src.p, qImg->width, z->numBlocksW, (size_t)z->totalBlocks, dst.p);
    if (hipGetLastError() != hipSuccess || !dst.down(z->data, n)) ST_FAIL(z, freeZigZagData);
    return z;
}

// Byte stuffing of an MSB-first bit string held in 32-bit words: a piece is 64 output-independent bytes.
// kWrite == false: ffoff[piece] = its 0xFF count;  kWrite == true: ffoff holds the exclusive sums, bytes go out.
template <bool kWrite>
__global__ void k_st_stuff(const uint32_t *__restrict__ words, unsigned long long nbytes, size_t npieces,
                           unsigned long long *__restrict__ ffoff, uint8_t *__restrict__ out) {
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npieces) return;
    const unsigned long long b0 = (unsigned long long)p * 64ull, b1 = b0 + 64ull < nbytes ? b0 + 64ull : nbytes;
    unsigned long long pos = kWrite ? b0 + ffoff[p] : 0ull, cnt = 0;
    for (unsigned long long i = b0; i < b1; ++i) {
        const uint8_t v = (uint8_t)(words[i >> 2] >> (24u - 8u * (uint32_t)(i & 3ull)));
        if (kWrite) { out[pos++] = v; if (v == 0xFF) out[pos++] = 0x00; }      // huffman.c:29-31
        else cnt += v == 0xFF;
    }
    if (!kWrite) ffoff[p] = cnt;
}

static bool exclusive_sum(unsigned long long *d, size_t n) {     // in place; d has n + 1 slots, d[n] receives the total
    size_t tmp_bytes = 0;
    if (hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, d, d, (int)(n + 1)) != hipSuccess) return false;
    Dev<uint8_t> tmp;
    if (!tmp.alloc(tmp_bytes)) return false;
    return hipcub::DeviceScan::ExclusiveSum(tmp.p, tmp_bytes, d, d, (int)(n + 1)) == hipSuccess;
}

extern "C" RLEData *performRLE(const ZigZagData *zz) {
    if (!zz || !zz->data || zz->totalBlocks <= 0 || !have_device()) return nullptr;
    const size_t nblk = (size_t)zz->totalBlocks;
    RLEData *r = make_struct<RLEData>();
    if (!r) return nullptr;
    Dev<int16_t> src; Dev<unsigned long long> cnt;
    if (!src.up(zz->data, nblk * 64) || !cnt.alloc(nblk + 1) || hipMemset(cnt.p, 0, (nblk + 1) * sizeof(unsigned long long)) != hipSuccess) ST_FAIL(r, freeRLEData);
    hipLaunchKernelGGL((k_st_rle<false>), dim3(grid_for(nblk, 128)), dim3(128), 0, nullptr, src.p, nblk, cnt.p, (RLESymbol *)nullptr);
    unsigned long long total = 0;
    if (hipGetLastError() != hipSuccess || !exclusive_sum(cnt.p, nblk) ||
        hipMemcpy(&total, cnt.p + nblk, sizeof(total), hipMemcpyDeviceToHost) != hipSuccess) ST_FAIL(r, freeRLEData);
    r->count = r->capacity = (size_t)total;
    r->data = (RLESymbol *)std::calloc(total ? total : 1, sizeof(RLESymbol));
    Dev<RLESymbol> out;
    if (!r->data || !out.alloc((size_t)total) || hipMemset(out.p, 0, (total ? total : 1) * sizeof(RLESymbol)) != hipSuccess) ST_FAIL(r, freeRLEData);
    hipLaunchKernelGGL((k_st_rle<true>), dim3(grid_for(nblk, 128)), dim3(128), 0, nullptr, src.p, nblk, cnt.p, out.p);
    if (hipGetLastError() != hipSuccess || (total && !out.down(r->data, (size_t)total))) ST_FAIL(r, freeRLEData);
    return r;
}

extern "C" JpegEncoderBuffer *encodeHuffman(const RLEData *rle, int totalBlocks) {
    if (!rle || (rle->count && !rle->data) || !have_device()) return nullptr;
    JpegEncoderBuffer *b = make_struct<JpegEncoderBuffer>();
    if (!b) return nullptr;
    const unsigned long long n = rle->count;
    uint32_t hw[272];
    build_huffman_words(hw);
    Dev<RLESymbol> sym; Dev<uint8_t> kind; Dev<uint32_t> huff; Dev<unsigned long long> off;
    if (!sym.up(rle->data, (size_t)n) || !kind.alloc((size_t)n) || !huff.up(hw, 272) || !off.alloc((size_t)n + 1) ||
        hipMemset(off.p, 0, ((size_t)n + 1) * sizeof(unsigned long long)) != hipSuccess) ST_FAIL(b, freeJpegEncoderBuffer);
    if (n) {
        hipLaunchKernelGGL(k_st_walk, dim3(1), dim3(64), 0, nullptr, sym.p, n, totalBlocks, kind.p);
        hipLaunchKernelGGL(k_st_bits, dim3(grid_for((size_t)n)), dim3(256), 0, nullptr, sym.p, kind.p, n, huff.p, off.p);
    }
    unsigned long long total_bits = 0;
    if (hipGetLastError() != hipSuccess || !exclusive_sum(off.p, (size_t)n) ||
        hipMemcpy(&total_bits, off.p + n, sizeof(total_bits), hipMemcpyDeviceToHost) != hipSuccess) ST_FAIL(b, freeJpegEncoderBuffer);
    // byte stuffing of the packed string (huffman.c:26-32): 0xFF count per 64-byte piece, exclusive sum, write
    const unsigned long long nbytes = (total_bits + 7ull) / 8ull;           // the last byte is zero-padded (huffman.c:65-81)
    const size_t npieces = (size_t)((nbytes + 63ull) / 64ull);
    const size_t nwords = (size_t)((nbytes + 3ull) / 4ull) + 2;
    const uint64_t out_cap = 2 * nbytes + 16;
    Dev<uint32_t> words; Dev<unsigned long long> ffoff; Dev<uint8_t> out;
    if (!words.alloc(nwords) || hipMemset(words.p, 0, nwords * sizeof(uint32_t)) != hipSuccess || !ffoff.alloc(npieces + 1) ||
        hipMemset(ffoff.p, 0, (npieces + 1) * sizeof(unsigned long long)) != hipSuccess || !out.alloc((size_t)out_cap)) ST_FAIL(b, freeJpegEncoderBuffer);
    uint64_t size = 0;
    if (nbytes > 0) {
        hipLaunchKernelGGL(k_st_pack, dim3(grid_for((size_t)n)), dim3(256), 0, nullptr, sym.p, kind.p, n, huff.p, off.p, words.p);
        hipLaunchKernelGGL((k_st_stuff<false>), dim3(grid_for(npieces)), dim3(256), 0, nullptr, words.p, nbytes, npieces, ffoff.p, out.p);
        unsigned long long total_ff = 0;
        if (hipGetLastError() != hipSuccess || !exclusive_sum(ffoff.p, npieces) ||
            hipMemcpy(&total_ff, ffoff.p + npieces, sizeof(total_ff), hipMemcpyDeviceToHost) != hipSuccess) ST_FAIL(b, freeJpegEncoderBuffer);
        hipLaunchKernelGGL((k_st_stuff<true>), dim3(grid_for(npieces)), dim3(256), 0, nullptr, words.p, nbytes, npieces, ffoff.p, out.p);
        size = nbytes + total_ff;
        if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess || size > out_cap) ST_FAIL(b, freeJpegEncoderBuffer);
    }
    b->size = b->capacity = (size_t)size;
    b->data = (uint8_t *)std::malloc(size ? (size_t)size : 1);
    if (!b->data || (size && !out.down(b->data, (size_t)size))) ST_FAIL(b, freeJpegEncoderBuffer);
    return b;
}
