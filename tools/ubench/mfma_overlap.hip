// mfma_overlap.hip -- does VALU work overlap the matrix pipe, inside one wave and across the waves of a SIMD?  (round 4)
// Per loop iteration a wave runs  A: 16 v_mfma_f32_32x32x16_f16 (4 accumulators in rotation)   B: 256 v_xor (2-cycle class)
// C: A then B (B independent of A)   D: 16 x (1 mfma + 16 xor) interleaved   E: 256 v_bfe (4-cycle class)   F: A then E
// G: 16 x (mfma + 16 bfe).  One workgroup per CU; 1 / 2 / 4 waves per SIMD.
// Output: shader cycles per iteration of the slowest wave, wall ns per iteration.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CLOBV "v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33"
#define X16 "v_xor_b32 v10, v10, v26\n v_xor_b32 v11, v11, v26\n v_xor_b32 v12, v12, v26\n v_xor_b32 v13, v13, v26\n v_xor_b32 v14, v14, v26\n v_xor_b32 v15, v15, v26\n v_xor_b32 v16, v16, v26\n v_xor_b32 v17, v17, v26\n v_xor_b32 v18, v18, v26\n v_xor_b32 v19, v19, v26\n v_xor_b32 v20, v20, v26\n v_xor_b32 v21, v21, v26\n v_xor_b32 v22, v22, v26\n v_xor_b32 v23, v23, v26\n v_xor_b32 v24, v24, v26\n v_xor_b32 v25, v25, v26\n"
#define B16 "v_bfe_u32 v10, v10, 1, 8\n v_bfe_u32 v11, v11, 1, 8\n v_bfe_u32 v12, v12, 1, 8\n v_bfe_u32 v13, v13, 1, 8\n v_bfe_u32 v14, v14, 1, 8\n v_bfe_u32 v15, v15, 1, 8\n v_bfe_u32 v16, v16, 1, 8\n v_bfe_u32 v17, v17, 1, 8\n v_bfe_u32 v18, v18, 1, 8\n v_bfe_u32 v19, v19, 1, 8\n v_bfe_u32 v20, v20, 1, 8\n v_bfe_u32 v21, v21, 1, 8\n v_bfe_u32 v22, v22, 1, 8\n v_bfe_u32 v23, v23, 1, 8\n v_bfe_u32 v24, v24, 1, 8\n v_bfe_u32 v25, v25, 1, 8\n"
#define M1(A) "v_mfma_f32_32x32x16_f16 a[" A "], v[28:31], v[28:31], a[" A "]\n"
#define M4 M1("0:15") M1("16:31") M1("32:47") M1("48:63")
#define CLOBA "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15","a16","a17","a18","a19","a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31","a32","a33","a34","a35","a36","a37","a38","a39","a40","a41","a42","a43","a44","a45","a46","a47","a48","a49","a50","a51","a52","a53","a54","a55","a56","a57","a58","a59","a60","a61","a62","a63"
template <int S> __global__ __launch_bounds__(1024) void k(unsigned long long *out, int iters) {
    asm volatile("v_mov_b32 v26, 3\n v_mov_b32 v28, 0\n v_mov_b32 v29, 0\n v_mov_b32 v30, 0\n v_mov_b32 v31, 0\n" ::: CLOBV);
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        if (S == 0) asm volatile(M4 M4 M4 M4 ::: CLOBV, CLOBA);
        if (S == 1) asm volatile(".rept 16\n" X16 ".endr\n" ::: CLOBV);
        if (S == 2) asm volatile(M4 M4 M4 M4 ".rept 16\n" X16 ".endr\n" ::: CLOBV, CLOBA);
        if (S == 3) asm volatile(".rept 4\n" M1("0:15") X16 M1("16:31") X16 M1("32:47") X16 M1("48:63") X16 ".endr\n" ::: CLOBV, CLOBA);
        if (S == 4) asm volatile(".rept 16\n" B16 ".endr\n" ::: CLOBV);
        if (S == 5) asm volatile(M4 M4 M4 M4 ".rept 16\n" B16 ".endr\n" ::: CLOBV, CLOBA);
        if (S == 6) asm volatile(".rept 4\n" M1("0:15") B16 M1("16:31") B16 M1("32:47") B16 M1("48:63") B16 ".endr\n" ::: CLOBV, CLOBA);
    }
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}
template <int S> static void go(int b, int t, unsigned long long *o, int it) { hipLaunchKernelGGL(k<S>, dim3(b), dim3(t), 0, 0, o, it); }
int main() {
    unsigned long long *d; (void)hipMalloc(&d, 8192 * 8);
    const char *names[7] = {"A 16 mfma", "B 256 xor (2c)", "C 16 mfma then 256 xor", "D 16 x (mfma + 16 xor)", "E 256 bfe (4c)", "F 16 mfma then 256 bfe", "G 16 x (mfma + 16 bfe)"};
    void (*fn[7])(int, int, unsigned long long *, int) = {go<0>, go<1>, go<2>, go<3>, go<4>, go<5>, go<6>};
    printf("# shader cycles per loop iteration of the slowest wave (s_memtime) | wall ns per iteration\n%-28s %24s %24s %24s\n", "stream", "1 w/SIMD", "2 w/SIMD", "4 w/SIMD");
    for (int s = 0; s < 7; ++s) {
        printf("%-28s", names[s]);
        for (int c = 0; c < 3; ++c) {
            const int threads = 256 << c, blocks = 256, iters = 4096;
            hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
            fn[s](blocks, threads, d, iters / 8);
            (void)hipEventRecord(e0); fn[s](blocks, threads, d, iters); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            const int waves = blocks * threads / 64;
            std::vector<unsigned long long> h(waves);
            (void)hipMemcpy(h.data(), d, waves * 8, hipMemcpyDeviceToHost);
            unsigned long long mx = 0; for (auto v : h) mx = v > mx ? v : mx;
            printf("   %8.0f c %8.1f ns", (double)mx / iters, (double)ms * 1e6 / iters);
        }
        printf("\n"); fflush(stdout);
    }
    return 0;
}
