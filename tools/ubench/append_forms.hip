// append_forms.hip -- what one site of the item appends costs a wave / a SIMD in two forms (round 4)
//   A: EXEC form (shipped): v_cmpx_ne_u32 0, v ; v_add_u32_sdwa (tag) ; ds_write_b32 ; v_add_u32 addr ; s_mov_b64 exec, save
//   B: redirect form:       v_and nz ; v_mad_u32_u24 (address or the lane's dump slot) ; v_add_u32_sdwa (tag) ; ds_write_b32 ; v_lshl_add_u32
// 32 sites per loop iteration over 8 value registers, ~15 % of the lanes non-zero per site; 1 / 2 / 4 waves per SIMD, one workgroup per CU.
// Output: shader cycles per site of the slowest wave; wall ns per site per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define SITE_A(V) "v_cmpx_ne_u32_e32 0, " V "\n v_add_u32_sdwa " V ", v20, 3 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n ds_write_b32 v21, " V "\n v_add_u32_e32 v21, 4, v21\n s_mov_b64 exec, s[10:11]\n"
#define SITE_B(V) "v_and_b32_e32 v24, 1, " V "\n v_mad_u32_u24 v25, v24, v22, v23\n v_add_u32_sdwa " V ", v20, 3 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n ds_write_b32 v25, " V "\n v_lshl_add_u32 v22, v24, 2, v22\n"
#define ALL8(S) S("v10") S("v11") S("v12") S("v13") S("v14") S("v15") S("v16") S("v17")
template <int F> __global__ __launch_bounds__(1024) void k(unsigned long long *out, int iters) {
    __shared__ unsigned buf[16][64 * 40];
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned base = (unsigned)(size_t)&buf[wave][0];
    unsigned long long t0, t1;
    // v10..v17: values (low bit set in ~15 % of the lanes), v20: zigzag tag, v21 / v22: running address, v23: dump slot
    asm volatile("v_mov_b32 v20, 5\n v_mov_b32 v21, %0\n v_mov_b32 v22, 0\n v_mov_b32 v23, %1\n s_mov_b64 s[10:11], exec\n"
                 "v_mov_b32 v10, %2\n v_mov_b32 v11, %3\n v_mov_b32 v12, %2\n v_mov_b32 v13, 0\n v_mov_b32 v14, %3\n v_mov_b32 v15, 0\n v_mov_b32 v16, %2\n v_mov_b32 v17, 0\n"
                 :: "v"(base + lane * 128), "v"(base + 64 * 128 + lane * 4), "v"((lane % 7) == 0 ? 1u : 0u), "v"((lane % 5) == 0 ? 1u : 0u)
                 : "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v20", "v21", "v22", "v23", "v24", "v25", "s10", "s11", "memory");
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        if (F == 0) asm volatile(ALL8(SITE_A) ALL8(SITE_A) ALL8(SITE_A) ALL8(SITE_A) "v_mov_b32 v21, %0\n s_waitcnt lgkmcnt(0)\n" :: "v"(base + lane * 128) : "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v21", "vcc", "memory");
        if (F == 1) asm volatile(ALL8(SITE_B) ALL8(SITE_B) ALL8(SITE_B) ALL8(SITE_B) "v_mov_b32 v22, 0\n s_waitcnt lgkmcnt(0)\n" ::: "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v22", "v24", "v25", "memory");
    }
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (lane == 0) out[blockIdx.x * (blockDim.x >> 6) + wave] = t1 - t0;
}
template <int F> static void go(int b, int t, unsigned long long *o, int it) { hipLaunchKernelGGL(k<F>, dim3(b), dim3(t), 0, 0, o, it); }
int main() {
    unsigned long long *d; (void)hipMalloc(&d, 8192 * 8);
    const char *names[2] = {"A exec form (5 instr / site)", "B redirect form (5 instr / site)"};
    void (*fn[2])(int, int, unsigned long long *, int) = {go<0>, go<1>};
    printf("# shader cycles per site of the slowest wave (s_memtime x clock ratio not applied: 100 MHz ticks x 21) | wall ns per site per SIMD\n%-36s %26s %26s %26s\n", "form", "1 w/SIMD", "2 w/SIMD", "4 w/SIMD");
    for (int s = 0; s < 2; ++s) {
        printf("%-36s", names[s]);
        for (int c = 0; c < 3; ++c) {
            const int threads = 256 << c, blocks = 256, iters = 2048;
            hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
            fn[s](blocks, threads, d, iters / 8);
            (void)hipEventRecord(e0); fn[s](blocks, threads, d, iters); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            const int waves = blocks * threads / 64;
            std::vector<unsigned long long> h(waves);
            (void)hipMemcpy(h.data(), d, waves * 8, hipMemcpyDeviceToHost);
            unsigned long long mx = 0; for (auto v : h) mx = v > mx ? v : mx;
            const double sites = (double)iters * 32.0;
            // wall ns per site per SIMD: the launch's time x SIMDs busy / (sites x waves): each SIMD runs (threads / 256) waves
            printf("   %8.2f ticks %8.2f ns", (double)mx / sites, (double)ms * 1e6 / (sites * (threads / 256)));
        }
        printf("\n"); fflush(stdout);
    }
    return 0;
}
