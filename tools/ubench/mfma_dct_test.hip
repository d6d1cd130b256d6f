// mfma_dct_test.hip -- validates the matrix-pipe form of the 8x8 DCT before it goes into the codec.
//
// out[c][blk] = sum_p K[c][p] * pix[p][blk]  with K[c][p] = COS_LUT[x][u] * COS_LUT[y][v]  (the
// reference's LUT products, natural_c/src/core/dct.c:9-18,84), computed as 24
// v_mfma_f32_32x32x16_bf16 per 32 blocks: K is split into three bf16 terms (lo, mid, hi; 24 bits),
// the pixels (int8) are exact in bf16, every product is exact in f32, only the accumulation rounds.
// Prints the worst observed |mfma - exact| per coefficient class against the bound the codec would use.
//
// build: hipcc --offload-arch=gfx950 -O3 mfma_dct_test.hip -o mfma_dct_test
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

static const float kCosLut[8][8] = {
    {1.000000f, 0.980785f, 0.923880f, 0.831470f, 0.707107f, 0.555570f, 0.382683f, 0.195090f},
    {1.000000f, 0.831470f, 0.382683f, -0.195090f, -0.707107f, -0.980785f, -0.923880f, -0.555570f},
    {1.000000f, 0.555570f, -0.382683f, -0.980785f, -0.707107f, 0.195090f, 0.923880f, 0.831470f},
    {1.000000f, 0.195090f, -0.923880f, -0.555570f, 0.707107f, 0.831470f, -0.382683f, -0.980785f},
    {1.000000f, -0.195090f, -0.923880f, 0.555570f, 0.707107f, -0.831470f, -0.382684f, 0.980785f},
    {1.000000f, -0.555570f, -0.382684f, 0.980785f, -0.707107f, -0.195090f, 0.923880f, -0.831470f},
    {1.000000f, -0.831470f, 0.382684f, 0.195091f, -0.707107f, 0.980785f, -0.923879f, 0.555570f},
    {1.000000f, -0.980785f, 0.923880f, -0.831470f, 0.707107f, -0.555570f, 0.382684f, -0.195090f}};
static const uint8_t kZZ[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

static uint16_t f2bf(double x) {            // round-to-nearest-even to bf16
    float f = (float)x;
    uint32_t u; memcpy(&u, &f, 4);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static double bf2d(uint16_t b) { uint32_t u = (uint32_t)b << 16; float f; memcpy(&f, &u, 4); return f; }

// A fragments: [term 3][chain 2][kstep 4][lane 64][8] bf16
__global__ __launch_bounds__(64) void k_dct(const int8_t* __restrict__ pix, const uint16_t* __restrict__ afrag,
                                            float* __restrict__ out, int ntiles) {
    const int lane = threadIdx.x, h = lane >> 5, b = lane & 31;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int8_t* blk = pix + ((size_t)tile * 32 + b) * 64;
        bf16x8 bfrag[4];
        for (int s = 0; s < 4; ++s) {
            for (int j = 0; j < 8; ++j) bfrag[s][j] = (__bf16)(float)blk[16 * s + 8 * h + j];
        }
        f32x16 acc[2] = {{0}, {0}};
        for (int t = 0; t < 3; ++t)            // lo, mid, hi: the small terms first keeps the running sum small
            for (int H = 0; H < 2; ++H)
                for (int s = 0; s < 4; ++s) {
                    const bf16x8 a = *reinterpret_cast<const bf16x8*>(afrag + ((((size_t)t * 2 + H) * 4 + s) * 64 + lane) * 8);
                    acc[H] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bfrag[s], acc[H], 0, 0, 0);
                }
        float* o = out + ((size_t)tile * 32 + b) * 64;
        for (int H = 0; H < 2; ++H)
            for (int r = 0; r < 16; ++r) o[32 * h + 16 * H + r] = acc[H][r];     // zigzag position
    }
}

int main() {
    // exact LUT-product matrix, raster coefficient index c = u*8+v, pixel p = x*8+y
    static double K[64][64];
    for (int u = 0; u < 8; ++u) for (int v = 0; v < 8; ++v) for (int x = 0; x < 8; ++x) for (int y = 0; y < 8; ++y)
        K[u * 8 + v][x * 8 + y] = (double)kCosLut[x][u] * (double)kCosLut[y][v];
    std::vector<uint16_t> afrag(3 * 2 * 4 * 64 * 8);
    static double Ksplit[64][64];   // what the three bf16 terms actually sum to
    for (int H = 0; H < 2; ++H) for (int s = 0; s < 4; ++s) for (int l = 0; l < 64; ++l) for (int j = 0; j < 8; ++j) {
        const int hk = l >> 5, R = l & 31;
        const int hh = (R >> 2) & 1, r = (R & 3) + 4 * (R >> 3);
        const int z = 32 * hh + 16 * H + r, c = kZZ[z], p = 16 * s + 8 * hk + j;
        const double k = K[c][p];
        const uint16_t hi = f2bf(k); const double r1 = k - bf2d(hi);
        const uint16_t mid = f2bf(r1); const double r2 = r1 - bf2d(mid);
        const uint16_t lo = f2bf(r2);
        const uint16_t terms[3] = {lo, mid, hi};
        for (int t = 0; t < 3; ++t) afrag[((((size_t)t * 2 + H) * 4 + s) * 64 + l) * 8 + j] = terms[t];
        Ksplit[c][p] = bf2d(hi) + bf2d(mid) + bf2d(lo);
    }
    const int ntiles = 4096, nblk = ntiles * 32;
    std::vector<int8_t> pix((size_t)nblk * 64);
    std::mt19937 rng(7);
    for (int b = 0; b < nblk; ++b) {
        const int mode = b % 8;
        for (int p = 0; p < 64; ++p) {
            int v;
            if (mode < 4) v = (int)(rng() % 256) - 128;                               // uniform
            else if (mode == 4) v = (int)(rng() % 9) - 4 + 100;                        // bright, flat-ish
            else if (mode == 5) v = (rng() & 1) ? 127 : -128;                          // extreme
            else { const int c = kZZ[(b / 8) % 64]; v = K[c][p] >= 0 ? 127 : -128; }   // adversarial: aligned with one basis
            pix[(size_t)b * 64 + p] = (int8_t)v;
        }
    }
    int8_t* dpix; uint16_t* da; float* dout;
    hipMalloc(&dpix, pix.size()); hipMalloc(&da, afrag.size() * 2); hipMalloc(&dout, (size_t)nblk * 64 * 4);
    hipMemcpy(dpix, pix.data(), pix.size(), hipMemcpyHostToDevice);
    hipMemcpy(da, afrag.data(), afrag.size() * 2, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_dct, dim3(1024), dim3(64), 0, 0, dpix, da, dout, ntiles);
    std::vector<float> out((size_t)nblk * 64);
    hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost);

    const double U = ldexp(1.0, -24);
    double worst_ratio = 0, worst_abs = 0, worst_ratio_split = 0;
    long dc_bad = 0;
    for (int c = 0; c < 64; ++c) {
        int z = 0; while (kZZ[z] != c) ++z;
        double sabs = 0; for (int p = 0; p < 64; ++p) sabs += fabs(K[c][p]) * 128.0;
        // bound: 16 adds per MFMA, 12 MFMAs; running sum <= S*2^-16 (lo), S*2^-8 (mid), then S*{.25,.5,.75,1}; x2 safety
        const double bound = 2.0 * 16.0 * U * sabs * (4 * 2e-5 + 4 * 4e-3 + 0.25 + 0.5 + 0.75 + 1.0 + 1.0) + sabs * ldexp(1.0, -24);
        for (int b = 0; b < nblk; ++b) {
            double ex = 0, exs = 0;
            for (int p = 0; p < 64; ++p) { ex += K[c][p] * pix[(size_t)b * 64 + p]; exs += Ksplit[c][p] * pix[(size_t)b * 64 + p]; }
            const double got = out[(size_t)b * 64 + z];
            const double e = fabs(got - ex);
            if (c == 0 && got != ex) ++dc_bad;
            if (e > worst_abs) worst_abs = e;
            if (e / bound > worst_ratio) worst_ratio = e / bound;
            if (fabs(got - exs) / bound > worst_ratio_split) worst_ratio_split = fabs(got - exs) / bound;
        }
    }
    printf("blocks %d: worst |mfma - exact| = %.6f (s units), worst error/bound = %.4f (vs split matrix %.4f), DC inexact in %ld blocks\n",
           nblk, worst_abs, worst_ratio, worst_ratio_split, dc_bad);
    // timing
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k_dct, dim3(1024), dim3(64), 0, 0, dpix, da, dout, ntiles);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("naive test kernel: %.1f us per %d blocks\n", ms * 100, nblk);
    return worst_ratio < 1.0 && dc_bad == 0 ? 0 : 1;
}
