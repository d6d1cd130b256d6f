// issue_model.hip -- what does one instruction cost a SIMD?  
//
// Every stream below is hand-written asm (`.rept` blocks, checked in the disassembly: nothing for the compiler to fold),
// with INDEPENDENT instructions (16 destination registers in rotation) unless the name says "dep".  One workgroup per CU,
// 1/2/4 waves per SIMD (256/512/1024 threads); 8 waves per SIMD = two 1024-thread workgroups per CU.
// Reported per stream and occupancy: shader cycles per instruction per SIMD (s_memtime delta of the slowest wave x
// 1 / (instructions per wave x waves per SIMD)) and ns per instruction per SIMD from the wall clock of a long launch,
// plus the clock the chip held (s_memtime / s_memrealtime).
//
// Build: hipcc --offload-arch=gfx950 -O3 issue_model.hip -o issue_model
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#define CLOB_V "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27"
#define CLOB_S "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55", "vcc", "scc"

// 16 instructions per group; GROUPS_PER_BODY groups per loop iteration
#define V16(OP, SRC)                                                                                          \
    OP " v10, v10, " SRC "\n" OP " v11, v11, " SRC "\n" OP " v12, v12, " SRC "\n" OP " v13, v13, " SRC "\n"     \
    OP " v14, v14, " SRC "\n" OP " v15, v15, " SRC "\n" OP " v16, v16, " SRC "\n" OP " v17, v17, " SRC "\n"     \
    OP " v18, v18, " SRC "\n" OP " v19, v19, " SRC "\n" OP " v20, v20, " SRC "\n" OP " v21, v21, " SRC "\n"     \
    OP " v22, v22, " SRC "\n" OP " v23, v23, " SRC "\n" OP " v24, v24, " SRC "\n" OP " v25, v25, " SRC "\n"
#define S16(OP, SRC)                                                                                          \
    OP " s40, s40, " SRC "\n" OP " s41, s41, " SRC "\n" OP " s42, s42, " SRC "\n" OP " s43, s43, " SRC "\n"     \
    OP " s44, s44, " SRC "\n" OP " s45, s45, " SRC "\n" OP " s46, s46, " SRC "\n" OP " s47, s47, " SRC "\n"     \
    OP " s48, s48, " SRC "\n" OP " s49, s49, " SRC "\n" OP " s50, s50, " SRC "\n" OP " s51, s51, " SRC "\n"     \
    OP " s52, s52, " SRC "\n" OP " s53, s53, " SRC "\n" OP " s54, s54, " SRC "\n" OP " s55, s55, " SRC "\n"
// 8 vector + 8 scalar, strictly alternating
#define VS16(VOP, SOP)                                                                                        \
    VOP " v10, v10, v26\n" SOP " s40, s40, 1\n" VOP " v11, v11, v26\n" SOP " s41, s41, 1\n"                     \
    VOP " v12, v12, v26\n" SOP " s42, s42, 1\n" VOP " v13, v13, v26\n" SOP " s43, s43, 1\n"                     \
    VOP " v14, v14, v26\n" SOP " s44, s44, 1\n" VOP " v15, v15, v26\n" SOP " s45, s45, 1\n"                     \
    VOP " v16, v16, v26\n" SOP " s46, s46, 1\n" VOP " v17, v17, v26\n" SOP " s47, s47, 1\n"
// 3 vector : 1 scalar (the transform kernel's mix)
#define V3S1_16(VOP, SOP)                                                                                     \
    VOP " v10, v10, v26\n" VOP " v11, v11, v26\n" VOP " v12, v12, v26\n" SOP " s40, s40, 1\n"                   \
    VOP " v13, v13, v26\n" VOP " v14, v14, v26\n" VOP " v15, v15, v26\n" SOP " s41, s41, 1\n"                   \
    VOP " v16, v16, v26\n" VOP " v17, v17, v26\n" VOP " v18, v18, v26\n" SOP " s42, s42, 1\n"                   \
    VOP " v19, v19, v26\n" VOP " v20, v20, v26\n" VOP " v21, v21, v26\n" SOP " s43, s43, 1\n"
// 12 vector + 4 LDS reads (addresses in v27, results discarded into v22..v25), one wait per group
#define VL16(VOP)                                                                                             \
    VOP " v10, v10, v26\n" VOP " v11, v11, v26\n" VOP " v12, v12, v26\n" "ds_read_b32 v22, v27\n"               \
    VOP " v13, v13, v26\n" VOP " v14, v14, v26\n" VOP " v15, v15, v26\n" "ds_read_b32 v23, v27 offset:256\n"    \
    VOP " v16, v16, v26\n" VOP " v17, v17, v26\n" VOP " v18, v18, v26\n" "ds_read_b32 v24, v27 offset:512\n"    \
    VOP " v19, v19, v26\n" VOP " v20, v20, v26\n" VOP " v21, v21, v26\n" "ds_read_b32 v25, v27 offset:768\n"

enum Stream {
    kValuXor, kValuFma, kValuXorDep, kValuVop3, kSaluAdd, kSaluMov, kSNop, kValuSalu11, kValuSalu31, kValuLds, kDot4,
    kDot4Valu, kDpp, kSdwa, kCmpAddc, kCvtFlr, kFract, kPerm, kPkFma, kReadlane, kBigBody, kNumStreams
};
static const char *kNames[kNumStreams] = {
    "valu v_xor_b32 (VOP2, 4 B)", "valu v_fma_f32 (VOP3, 8 B)", "valu v_xor_b32 DEPENDENT chain", "valu v_add3_u32 (VOP3, 8 B)",
    "salu s_add_u32", "salu s_mov_b32", "s_nop 0", "valu:salu 1:1 alternating", "valu:salu 3:1", "valu:lds 3:1 (ds_read_b32)",
    "v_dot4_u32_u8", "v_dot4 : v_xor 1:1", "v_add_u32 dpp row_shr:1", "v_cvt_f32_i32 sdwa BYTE_1", "v_cmp -> v_addc (sgpr mask)",
    "v_cvt_flr_i32_f32", "v_fract_f32", "v_perm_b32", "v_pk_fma_f32", "v_readlane_b32 -> sgpr", "valu v_xor_b32, 4096-instruction body (16 KiB of code)"};

constexpr int kGroups = 16;          // 16 groups x 16 = 256 instructions per loop iteration

template <int S>
__global__ __launch_bounds__(1024) void k_stream(unsigned long long *out, int iters) {
    __shared__ uint32_t lds[1024];
    lds[threadIdx.x & 1023] = threadIdx.x;
    __syncthreads();
    asm volatile("v_mov_b32 v26, 0x55\n v_mov_b32 v27, 0\n"
                 "v_mov_b32 v10, 1\n v_mov_b32 v11, 2\n v_mov_b32 v12, 3\n v_mov_b32 v13, 4\n v_mov_b32 v14, 5\n v_mov_b32 v15, 6\n"
                 "v_mov_b32 v16, 7\n v_mov_b32 v17, 8\n v_mov_b32 v18, 9\n v_mov_b32 v19, 10\n v_mov_b32 v20, 11\n v_mov_b32 v21, 12\n"
                 "v_mov_b32 v22, 13\n v_mov_b32 v23, 14\n v_mov_b32 v24, 15\n v_mov_b32 v25, 16\n" ::: CLOB_V);
    asm volatile("s_mov_b32 s40, 0\n s_mov_b32 s41, 0\n s_mov_b32 s42, 0\n s_mov_b32 s43, 0\n s_mov_b32 s44, 0\n s_mov_b32 s45, 0\n"
                 "s_mov_b32 s46, 0\n s_mov_b32 s47, 0\n s_mov_b32 s48, 0\n s_mov_b32 s49, 0\n s_mov_b32 s50, 0\n s_mov_b32 s51, 0\n"
                 "s_mov_b32 s52, 0\n s_mov_b32 s53, 0\n s_mov_b32 s54, 0\n s_mov_b32 s55, 0\n" ::: CLOB_S);
    unsigned long long t0, t1, r0, r1;
    asm volatile("s_memrealtime %0\n s_memtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(r0), "=s"(t0)::"memory");
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        if (S == kValuXor) asm volatile(".rept 16\n" V16("v_xor_b32", "v26") ".endr\n" ::: CLOB_V);
        if (S == kValuFma) asm volatile(".rept 16\n" V16("v_fma_f32", "v26, v26") ".endr\n" ::: CLOB_V);
        if (S == kValuXorDep) asm volatile(".rept 256\n v_xor_b32 v10, v10, v26\n .endr\n" ::: CLOB_V);
        if (S == kValuVop3) asm volatile(".rept 16\n" V16("v_add3_u32", "v26, v26") ".endr\n" ::: CLOB_V);
        if (S == kSaluAdd) asm volatile(".rept 16\n" S16("s_add_u32", "1") ".endr\n" ::: CLOB_S);
        if (S == kSaluMov) asm volatile(".rept 256\n s_mov_b32 s40, 0x1234\n .endr\n" ::: CLOB_S);
        if (S == kSNop) asm volatile(".rept 256\n s_nop 0\n .endr\n" :::);
        if (S == kValuSalu11) asm volatile(".rept 16\n" VS16("v_xor_b32", "s_add_u32") ".endr\n" ::: CLOB_V, CLOB_S);
        if (S == kValuSalu31) asm volatile(".rept 16\n" V3S1_16("v_xor_b32", "s_add_u32") ".endr\n" ::: CLOB_V, CLOB_S);
        if (S == kValuLds) asm volatile(".rept 16\n" VL16("v_xor_b32") "s_waitcnt lgkmcnt(0)\n .endr\n" ::: CLOB_V, "memory");
        if (S == kDot4) asm volatile(".rept 16\n" V16("v_dot4_u32_u8", "v26, 0") ".endr\n" ::: CLOB_V);
        if (S == kDot4Valu)
            asm volatile(".rept 32\n"
                         "v_dot4_u32_u8 v10, v10, v26, 0\n v_xor_b32 v14, v14, v26\n v_dot4_u32_u8 v11, v11, v26, 0\n v_xor_b32 v15, v15, v26\n"
                         "v_dot4_u32_u8 v12, v12, v26, 0\n v_xor_b32 v16, v16, v26\n v_dot4_u32_u8 v13, v13, v26, 0\n v_xor_b32 v17, v17, v26\n"
                         ".endr\n" ::: CLOB_V);
        if (S == kDpp)
            asm volatile(".rept 16\n" V16("v_add_u32_dpp", "v26 row_shr:1 row_mask:0xf bank_mask:0xf") ".endr\n" ::: CLOB_V);
        if (S == kSdwa)
            asm volatile(".rept 16\n"
                         "v_cvt_f32_i32_sdwa v10, sext(v26) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1\n"
                         "v_cvt_f32_i32_sdwa v11, sext(v26) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1\n"
                         "v_cvt_f32_i32_sdwa v12, sext(v26) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1\n"
                         "v_cvt_f32_i32_sdwa v13, sext(v26) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1\n"
                         "v_cvt_f32_i32_sdwa v14, sext(v26) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1\n"
                         "v_cvt_f32_i32_sdwa v15, sext(v26) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1\n"
                         "v_cvt_f32_i32_sdwa v16, sext(v26) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1\n"
                         "v_cvt_f32_i32_sdwa v17, sext(v26) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1\n"
                         "v_cvt_f32_i32_sdwa v18, sext(v26) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1\n"
                         "v_cvt_f32_i32_sdwa v19, sext(v26) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1\n"
                         "v_cvt_f32_i32_sdwa v20, sext(v26) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1\n"
                         "v_cvt_f32_i32_sdwa v21, sext(v26) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1\n"
                         "v_cvt_f32_i32_sdwa v22, sext(v26) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1\n"
                         "v_cvt_f32_i32_sdwa v23, sext(v26) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1\n"
                         "v_cvt_f32_i32_sdwa v24, sext(v26) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1\n"
                         "v_cvt_f32_i32_sdwa v25, sext(v26) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1\n"
                         ".endr\n" ::: CLOB_V);
        if (S == kCmpAddc)     // the fold-8 pattern of the quantiser: compare into an SGPR pair, add-with-carry two instructions later
            asm volatile(".rept 32\n"
                         "v_cmp_le_f32 s[40:41], v10, v26\n v_cmp_le_f32 s[42:43], v11, v26\n v_cmp_le_f32 vcc, v12, v26\n"
                         "v_addc_co_u32 v13, s[40:41], v13, v13, s[40:41]\n v_cmp_le_f32 s[40:41], v14, v26\n"
                         "v_addc_co_u32 v13, s[42:43], v13, v13, s[42:43]\n v_addc_co_u32 v13, vcc, v13, v13, vcc\n"
                         "v_addc_co_u32 v13, s[40:41], v13, v13, s[40:41]\n"
                         ".endr\n" ::: CLOB_V, CLOB_S);
        if (S == kCvtFlr) asm volatile(".rept 256\n v_cvt_flr_i32_f32 v10, v26\n .endr\n" ::: CLOB_V);
        if (S == kFract) asm volatile(".rept 256\n v_fract_f32 v10, v26\n .endr\n" ::: CLOB_V);
        if (S == kPerm) asm volatile(".rept 16\n" V16("v_perm_b32", "v26, v26") ".endr\n" ::: CLOB_V);
        if (S == kPkFma)
            asm volatile(".rept 32\n"
                         "v_pk_fma_f32 v[10:11], v[10:11], v[26:27], v[26:27]\n v_pk_fma_f32 v[12:13], v[12:13], v[26:27], v[26:27]\n"
                         "v_pk_fma_f32 v[14:15], v[14:15], v[26:27], v[26:27]\n v_pk_fma_f32 v[16:17], v[16:17], v[26:27], v[26:27]\n"
                         "v_pk_fma_f32 v[18:19], v[18:19], v[26:27], v[26:27]\n v_pk_fma_f32 v[20:21], v[20:21], v[26:27], v[26:27]\n"
                         "v_pk_fma_f32 v[22:23], v[22:23], v[26:27], v[26:27]\n v_pk_fma_f32 v[24:25], v[24:25], v[26:27], v[26:27]\n"
                         ".endr\n" ::: CLOB_V);
        if (S == kReadlane)
            asm volatile(".rept 32\n"
                         "v_readlane_b32 s40, v10, 3\n v_readlane_b32 s41, v11, 3\n v_readlane_b32 s42, v12, 3\n v_readlane_b32 s43, v13, 3\n"
                         "v_readlane_b32 s44, v14, 3\n v_readlane_b32 s45, v15, 3\n v_readlane_b32 s46, v16, 3\n v_readlane_b32 s47, v17, 3\n"
                         ".endr\n" ::: CLOB_S);
        if (S == kBigBody) asm volatile(".rept 256\n" V16("v_xor_b32", "v26") ".endr\n" ::: CLOB_V);
    }
    asm volatile("s_memtime %0\n s_memrealtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1)::"memory");
    if ((threadIdx.x & 63) == 0) {
        const size_t w = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        out[2 * w] = t1 - t0;
        out[2 * w + 1] = r1 - r0;
    }
    if (iters < 0) out[0] = lds[threadIdx.x];      // keep the LDS alive
}

template <int S>
static void s_readlane_fix() {}

static double run_one(int s, int threads, int blocks, int iters, unsigned long long *d_out, double *cyc_per_instr, double *clock_ghz) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    auto launch = [&](int it) {
        switch (s) {
#define CASE(X) case X: hipLaunchKernelGGL(k_stream<X>, dim3(blocks), dim3(threads), 0, 0, d_out, it); break;
            CASE(kValuXor) CASE(kValuFma) CASE(kValuXorDep) CASE(kValuVop3) CASE(kSaluAdd) CASE(kSaluMov) CASE(kSNop)
            CASE(kValuSalu11) CASE(kValuSalu31) CASE(kValuLds) CASE(kDot4) CASE(kDot4Valu) CASE(kDpp) CASE(kSdwa) CASE(kCmpAddc)
            CASE(kCvtFlr) CASE(kFract) CASE(kPerm) CASE(kPkFma) CASE(kReadlane) CASE(kBigBody)
#undef CASE
        }
    };
    launch(iters / 8 + 1);                       // warm-up (clock, instruction cache)
    hipEventRecord(e0);
    launch(iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const int waves = blocks * threads / 64;
    std::vector<unsigned long long> h(2 * waves);
    hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost);
    unsigned long long tmax = 0, rsum = 0, tsum = 0;
    for (int w = 0; w < waves; ++w) { tmax = h[2 * w] > tmax ? h[2 * w] : tmax; tsum += h[2 * w]; rsum += h[2 * w + 1]; }
    const double per_wave = (double)iters * (s == kBigBody ? 4096.0 : 256.0);
    const double waves_per_simd = (double)waves / 1024.0;
    *cyc_per_instr = (double)tmax / (per_wave * waves_per_simd);
    *clock_ghz = (double)tsum / (double)rsum * 0.1;       // s_memrealtime ticks at 100 MHz
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    return (double)ms * 1e6 / (per_wave * waves_per_simd);   // ns per instruction per SIMD (wall clock)
}

int main(int argc, char **argv) {
    int only = argc > 1 ? atoi(argv[1]) : -1;
    unsigned long long *d_out;
    hipMalloc(&d_out, 2 * 8192 * 8);
    printf("# cycles (s_memtime) and wall ns per instruction per SIMD; 256 CUs x 4 SIMDs; stream body = 256 instructions + 3 loop instructions\n");
    printf("%-52s %14s %14s %14s %14s\n", "stream", "1 wave/SIMD", "2 waves/SIMD", "4 waves/SIMD", "8 waves/SIMD");
    for (int s = 0; s < kNumStreams; ++s) {
        if (only >= 0 && s != only) continue;
        printf("%-52s", kNames[s]);
        const int cfg[4][2] = {{256, 256}, {512, 256}, {1024, 256}, {1024, 512}};
        double clk = 0;
        for (int c = 0; c < 4; ++c) {
            const int iters = (s == kBigBody ? 256 : 4096) / (c == 3 ? 2 : 1);
            double cyc, ghz;
            const double ns = run_one(s, cfg[c][0], cfg[c][1], iters, d_out, &cyc, &ghz);
            printf("  %5.2f c %5.2f ns", cyc, ns);
            clk = ghz;
        }
        printf("   clock %.2f GHz\n", clk);
        fflush(stdout);
    }
    return 0;
}
