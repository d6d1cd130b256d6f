// Micro-benchmark: do packed-f32 VALU ops (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32) issue at the
// same per-instruction rate as their scalar forms on gfx950?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define ITER 2048
#define UNROLL 8
typedef float float2v __attribute__((ext_vector_type(2)));

template <int OP>
__global__ void k(float* out, float seed) {
    float2v a[UNROLL];
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) { a[i].x = seed + threadIdx.x * (i + 1) * 1e-3f; a[i].y = a[i].x * 0.5f; }
    float2v c = {1.0001f, 0.9999f}, b = {0.5f, 0.25f};
    asm volatile("" : "+v"(c), "+v"(b));
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < UNROLL; ++i) {
            if (OP == 0) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 1) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            if (OP == 2) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(b));
            if (OP == 3) { asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i].x) : "v"(b.x)); asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i].y) : "v"(b.y)); }
            if (OP == 4) { asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i].x) : "v"(c.x), "v"(b.x)); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i].y) : "v"(c.y), "v"(b.y)); }
        }
    }
    float r = 0;
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) r += a[i].x + a[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int OP>
void run(const char* name, float* d, int per_iter_insts) {
    for (int wpb : {256, 512, 1024}) {
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(k<OP>, dim3(256), dim3(wpb), 0, 0, d, 3.f);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<OP>, dim3(256), dim3(wpb), 0, 0, d, 3.f);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        double insts = (double)ITER * UNROLL * per_iter_insts * (wpb / 256.0);
        printf("%-22s %.0f waves/SIMD: %7.3f ms -> %.2f ns per instr per SIMD, %.2f ns per float-op pair\n", name, wpb / 256.0, ms,
               ms * 1e6 / insts, ms * 1e6 / (ITER * UNROLL * (wpb / 256.0)));
    }
}
int main() {
    float* d; (void)hipMalloc(&d, 256 * 1024 * 4);
    run<0>("v_pk_add_f32", d, 1); run<3>("2x v_add_f32", d, 2);
    run<1>("v_pk_mul_f32", d, 1);
    run<2>("v_pk_fma_f32", d, 1); run<4>("2x v_fma_f32", d, 2);
    return 0;
}
