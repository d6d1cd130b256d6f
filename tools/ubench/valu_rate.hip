// Micro-benchmark: per-SIMD issue rate of the VALU ops the fused kernel leans on, as a
// function of waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define ITER 2048
#define UNROLL 16

template <int OP>
__global__ void k(uint32_t* out, uint32_t seed) {
    uint32_t a[UNROLL];
    uint64_t b64 = seed * 0x9E3779B97F4A7C15ull + threadIdx.x;
    float f[UNROLL];
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) { a[i] = seed + threadIdx.x * (i + 1); f[i] = (float)a[i] * 1e-3f; }
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < UNROLL; ++i) {
            if (OP == 0) a[i] = a[i] + seed;                                   // v_add_u32
            if (OP == 1) f[i] = __builtin_fmaf(f[i], 1.0001f, 0.5f);             // v_fma_f32
            if (OP == 2) { b64 = (b64 << (a[i] & 7)) | 1ull; }                    // v_lshlrev_b64 (dependent)
            if (OP == 3) a[i] = __builtin_amdgcn_udot4(a[i], seed, a[i], false);  // v_dot4_u32_u8
            if (OP == 4) a[i] = (a[i] > seed) ? a[i] - seed : a[i] + 3;           // cmp+cndmask style
            if (OP == 5) f[i] = __builtin_amdgcn_fractf(f[i] + 0.37f);            // add + fract
            if (OP == 6) a[i] = (a[i] << 3) | seed;                               // v_lshl_or_b32
            if (OP == 7) f[i] = f[i] + 1.25f;                                      // v_add_f32
        }
    }
    uint32_t r = (uint32_t)b64;
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) r += a[i] + (uint32_t)f[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int OP>
void run(const char* name, uint32_t* d) {
    for (int wpb : {64, 128, 256, 512, 1024}) {       // threads per block; 1 block per CU => waves/SIMD = wpb/256
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k<OP>, dim3(256), dim3(wpb), 0, 0, d, 3u);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<OP>, dim3(256), dim3(wpb), 0, 0, d, 3u);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double waves_per_simd = wpb / 256.0;
        double insts = (double)ITER * UNROLL * (wpb / 64.0) / 4.0;    // wave-instructions per SIMD (>=1 wave/SIMD)
        if (wpb < 256) insts = (double)ITER * UNROLL;                  // some SIMDs idle: per busy SIMD
        printf("%-14s threads/CU %4d (%.2f waves/SIMD): %8.3f ms  -> %.2f ns per wave-instr per SIMD (%.2f cyc @2.4GHz)\n",
               name, wpb, waves_per_simd, ms, ms * 1e6 / insts, ms * 1e6 / insts * 2.4);
    }
}

int main() {
    uint32_t* d; hipMalloc(&d, 256 * 1024 * 4);
    run<0>("v_add_u32", d); run<1>("v_fma_f32", d); run<7>("v_add_f32", d); run<2>("v_lshl_b64 dep", d);
    run<3>("v_dot4_u32_u8", d); run<4>("cmp+cndmask+2", d); run<5>("add+fract", d); run<6>("v_lshl_or_b32", d);
    return 0;
}
