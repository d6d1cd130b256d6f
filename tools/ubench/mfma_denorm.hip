// mfma_denorm.hip -- does v_mfma_f32_32x32x16_f16 take binary16 SUBNORMAL inputs at face value?  (round 4)
// B operand: the raw integers 0..255 as binary16 bit patterns (= n * 2^-24, subnormal); A operand: integers (normal binary16 values,
// up to 2048, and lo-style multiples of 2^-11).  Expected: out[i][j] = 2^-24 * sum_k A[i][k] * n[k][j], exact in float32.
// Prints the number of mismatching outputs of a random test and a few samples.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
// A: 32 x 16 (row-major halves as bits), B: 16 x 32 (bits), out: 32 x 32 floats.  One wave.
__global__ void k(const uint16_t *A, const uint16_t *B, float *out) {
    const int lane = threadIdx.x & 63;
    // A fragment: lane (hk = lane >> 5, row = lane & 31) holds A[row][8 hk + j]; B fragment: lane (hk, col) holds B[8 hk + j][col]
    union { f16x8 v; uint16_t u[8]; } a, b;
    for (int j = 0; j < 8; ++j) { a.u[j] = A[(lane & 31) * 16 + 8 * (lane >> 5) + j]; b.u[j] = B[(8 * (lane >> 5) + j) * 32 + (lane & 31)]; }
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const f32x16 c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.v, b.v, zero, 0, 0, 0);
    // C layout: lane (h = lane >> 5, col = lane & 31), element e: row = (e & 3) + 8 (e >> 2) + 4 h
    for (int e = 0; e < 16; ++e) out[((e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)) * 32 + (lane & 31)] = c[e];
}
static uint16_t f16_bits_of_int(int v) {      // exact binary16 of an integer |v| <= 2048
    if (v == 0) return 0;
    const uint16_t s = v < 0 ? 0x8000 : 0; int a = abs(v), e = 0;
    while ((a >> (e + 1)) != 0) ++e;            // leading bit
    const int mant = (a << (10 - e)) & 0x3FF;   // (e <= 11: a = 2048 -> mant 0, e 11)
    return (uint16_t)(s | ((e + 15) << 10) | mant);
}
int main() {
    uint16_t hA[32 * 16], hB[16 * 32]; int iA[32 * 16], iB[16 * 32];
    srand(7);
    int bad_total = 0;
    for (int trial = 0; trial < 3; ++trial) {
        for (int i = 0; i < 32 * 16; ++i) {
            iA[i] = (rand() % 4097) - 2048;
            if (trial == 2) iA[i] = (rand() % 2049) - 1024;                           // lo-style: value 2^-11 * integer
            hA[i] = f16_bits_of_int(iA[i]);
            if (trial == 2 && iA[i] != 0) {                                           // scale by 2^-11: exponent field minus 11 (stays normal for |v| >= 1 -> 2^-11 >= 2^-14)
                hA[i] = (uint16_t)(hA[i] - (11 << 10));
            }
        }
        for (int i = 0; i < 16 * 32; ++i) { iB[i] = trial == 0 ? (rand() & 255) : (i % 3 == 0 ? 255 : (rand() & 255)); hB[i] = (uint16_t)iB[i]; }   // the raw integer IS the subnormal bit pattern
        uint16_t *dA, *dB; float *dO;
        (void)hipMalloc(&dA, sizeof(hA)); (void)hipMalloc(&dB, sizeof(hB)); (void)hipMalloc(&dO, 32 * 32 * 4);
        (void)hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); (void)hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dO);
        std::vector<float> o(32 * 32);
        (void)hipMemcpy(o.data(), dO, 32 * 32 * 4, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int r = 0; r < 32; ++r)
            for (int c = 0; c < 32; ++c) {
                long long s = 0;
                for (int kk = 0; kk < 16; ++kk) s += (long long)iA[r * 16 + kk] * iB[kk * 32 + c];
                const double want = std::ldexp((double)s, trial == 2 ? -35 : -24);
                if ((double)o[r * 32 + c] != want) { if (bad < 4) printf("  trial %d [%d][%d]: got %.9g want %.9g\n", trial, r, c, (double)o[r * 32 + c], want); ++bad; }
            }
        printf("trial %d (%s): %d of 1024 outputs differ\n", trial, trial == 2 ? "A = integers * 2^-11, B subnormal" : "A = integers <= 2048, B subnormal 0..255", bad);
        bad_total += bad;
    }
    printf(bad_total ? "SUBNORMAL INPUTS ARE NOT TAKEN AT FACE VALUE\n" : "subnormal binary16 inputs are exact\n");
    return 0;
}
