// issue_model2.hip -- cost table of the instruction FORMS the pipeline kernels use (follow-up of issue_model.hip).
// Same method: hand-written asm streams (.rept), independent instructions (rotating destinations v10..v25 / s40..s55),
// wall-clock ns per instruction per SIMD at 1, 2, 4 and 8 waves per SIMD.  Each stream is a group of 16 instructions x 16.
// Build: hipcc --offload-arch=gfx950 -O3 issue_model2.hip -o issue_model2
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#define CLOB "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", \
             "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "vcc", "scc", "memory"

// R16(pre, post): 16 instructions "pre vN post" with N = 10..25
#define R16(PRE, POST)                                                                                                  \
    PRE "v10" POST "\n" PRE "v11" POST "\n" PRE "v12" POST "\n" PRE "v13" POST "\n" PRE "v14" POST "\n" PRE "v15" POST "\n" \
    PRE "v16" POST "\n" PRE "v17" POST "\n" PRE "v18" POST "\n" PRE "v19" POST "\n" PRE "v20" POST "\n" PRE "v21" POST "\n" \
    PRE "v22" POST "\n" PRE "v23" POST "\n" PRE "v24" POST "\n" PRE "v25" POST "\n"
// RR16: destination and first source are the same rotating register
#define RR16(OP, POST)                                                                                                  \
    OP " v10, v10" POST "\n" OP " v11, v11" POST "\n" OP " v12, v12" POST "\n" OP " v13, v13" POST "\n"                     \
    OP " v14, v14" POST "\n" OP " v15, v15" POST "\n" OP " v16, v16" POST "\n" OP " v17, v17" POST "\n"                     \
    OP " v18, v18" POST "\n" OP " v19, v19" POST "\n" OP " v20, v20" POST "\n" OP " v21, v21" POST "\n"                     \
    OP " v22, v22" POST "\n" OP " v23, v23" POST "\n" OP " v24, v24" POST "\n" OP " v25, v25" POST "\n"

#define STREAMS(X)                                                                                                      \
    X(0, "v_xor_b32 v,v,v (VOP2)", RR16("v_xor_b32", ", v26"))                                                          \
    X(1, "v_add_u32 v,v,v (VOP2)", RR16("v_add_u32", ", v26"))                                                          \
    X(2, "v_lshlrev_b32 v,3,v (VOP2)", R16("v_lshlrev_b32 ", ", 3, v26"))                                               \
    X(3, "v_mov_b32 v,v (VOP1)", R16("v_mov_b32 ", ", v26"))                                                            \
    X(4, "v_add_f32 v,v,v (VOP2)", RR16("v_add_f32", ", v26"))                                                          \
    X(5, "v_mul_f32 v,v,v (VOP2)", RR16("v_mul_f32", ", v26"))                                                          \
    X(6, "v_fmac_f32 v,v,v (VOP2)", R16("v_fmac_f32 ", ", v26, v27"))                                                   \
    X(7, "v_fma_f32 v,v,s,s (VOP3, 1 vgpr src)", RR16("v_fma_f32", ", s40, s40"))                                       \
    X(8, "v_fma_f32 v,v,v,s (VOP3, 2 vgpr src)", RR16("v_fma_f32", ", v26, s40"))                                       \
    X(9, "v_fma_f32 v,v,v,v (VOP3, 3 vgpr src)", RR16("v_fma_f32", ", v26, v27"))                                       \
    X(10, "v_fract_f32 v,v (VOP1)", R16("v_fract_f32 ", ", v26"))                                                       \
    X(11, "v_cvt_flr_i32_f32 v,v (VOP1)", R16("v_cvt_flr_i32_f32 ", ", v26"))                                           \
    X(12, "v_cvt_f32_i32 v,v (VOP1)", R16("v_cvt_f32_i32 ", ", v26"))                                                   \
    X(13, "v_cvt_f32_ubyte1 v,v (VOP1)", R16("v_cvt_f32_ubyte1 ", ", v26"))                                             \
    X(14, "v_max_f32 v,v,v (VOP2)", RR16("v_max_f32", ", v26"))                                                         \
    X(15, "v_max3_f32 v,v,v,v (VOP3)", RR16("v_max3_f32", ", v26, v27"))                                                \
    X(16, "v_cndmask_b32 v,v,v,vcc (VOP2)", RR16("v_cndmask_b32", ", v26, vcc"))                                        \
    X(17, "v_cmp_ne_u32 vcc,v,v (VOPC)", R16("v_cmp_ne_u32 vcc, ", ", v26"))                                            \
    X(18, "v_cmp_ne_u32 s[..],v,v (VOP3)", R16("v_cmp_ne_u32 s[40:41], ", ", v26"))                                     \
    X(19, "v_addc_co_u32 v,vcc,v,v,vcc (VOP2)", R16("v_addc_co_u32 ", ", vcc, v26, v27, vcc"))                         \
    X(20, "v_lshl_add_u32 v,v,3,v (VOP3, 2 vgpr)", RR16("v_lshl_add_u32", ", 3, v26"))                                  \
    X(21, "v_lshl_or_b32 v,v,3,s (VOP3, 1 vgpr)", RR16("v_lshl_or_b32", ", 3, s40"))                                    \
    X(22, "v_alignbit_b32 v,v,v,v (VOP3, 3 vgpr)", RR16("v_alignbit_b32", ", v26, v27"))                                \
    X(23, "v_alignbit_b32 v,v,v,7 (VOP3, 2 vgpr)", RR16("v_alignbit_b32", ", v26, 7"))                                  \
    X(24, "v_bfe_u32 v,v,4,8 (VOP3, 1 vgpr)", RR16("v_bfe_u32", ", 4, 8"))                                              \
    X(25, "v_ffbh_u32 v,v (VOP1)", R16("v_ffbh_u32 ", ", v26"))                                                         \
    X(26, "v_bcnt_u32_b32 v,v,v (VOP3)", RR16("v_bcnt_u32_b32", ", v26"))                                               \
    X(27, "v_mbcnt_lo_u32_b32 v,s,v", R16("v_mbcnt_lo_u32_b32 ", ", s40, v26"))                                         \
    X(28, "v_and_or_b32 v,v,v,v (VOP3)", RR16("v_and_or_b32", ", v26, v27"))                                            \
    X(29, "v_sad_u8 v,v,v,v (VOP3)", RR16("v_sad_u8", ", v26, v27"))                                                    \
    X(30, "v_cvt_pk_bf16_f32 v,v,v (VOP3)", RR16("v_cvt_pk_bf16_f32", ", v26"))                                         \
    X(31, "v_mul_u32_u24 v,v,v (VOP2)", RR16("v_mul_u32_u24", ", v26"))                                                 \
    X(32, "v_mad_u32_u24 v,v,v,v (VOP3)", RR16("v_mad_u32_u24", ", v26, v27"))                                          \
    X(33, "v_pk_mul_f32 v[2],v[2],v[2]", ".rept 2\n v_pk_mul_f32 v[10:11], v[10:11], v[26:27]\n v_pk_mul_f32 v[12:13], v[12:13], v[26:27]\n v_pk_mul_f32 v[14:15], v[14:15], v[26:27]\n v_pk_mul_f32 v[16:17], v[16:17], v[26:27]\n v_pk_mul_f32 v[18:19], v[18:19], v[26:27]\n v_pk_mul_f32 v[20:21], v[20:21], v[26:27]\n v_pk_mul_f32 v[22:23], v[22:23], v[26:27]\n v_pk_mul_f32 v[24:25], v[24:25], v[26:27]\n .endr\n") \
    X(34, "ds_write_b32 (lane-linear)", ".rept 16\n ds_write_b32 v28, v26\n .endr\n s_waitcnt lgkmcnt(0)\n")             \
    X(35, "ds_read_b128 (lane-linear)", ".rept 4\n ds_read_b128 v[10:13], v29\n ds_read_b128 v[14:17], v29 offset:1024\n ds_read_b128 v[18:21], v29 offset:2048\n ds_read_b128 v[22:25], v29 offset:3072\n .endr\n s_waitcnt lgkmcnt(0)\n") \
    X(36, "ds_read_b64 (lane-linear)", ".rept 4\n ds_read_b64 v[10:11], v29\n ds_read_b64 v[14:15], v29 offset:1024\n ds_read_b64 v[18:19], v29 offset:2048\n ds_read_b64 v[22:23], v29 offset:3072\n .endr\n s_waitcnt lgkmcnt(0)\n") \
    X(37, "8 v_xor : 1 mfma_32x32x16_bf16 (2 per group)", "v_mfma_f32_32x32x16_bf16 a[0:15], v[10:13], v[14:17], a[0:15]\n .rept 7\n v_xor_b32 v20, v20, v26\n .endr\n v_mfma_f32_32x32x16_bf16 a[16:31], v[10:13], v[14:17], a[16:31]\n .rept 7\n v_xor_b32 v21, v21, v26\n .endr\n") \
    X(38, "mfma_32x32x16_bf16 only (2 accumulators)", ".rept 8\n v_mfma_f32_32x32x16_bf16 a[0:15], v[10:13], v[14:17], a[0:15]\n v_mfma_f32_32x32x16_bf16 a[16:31], v[10:13], v[14:17], a[16:31]\n .endr\n") \
    X(39, "mfma_i32_32x32x32_i8 only (2 accumulators)", ".rept 8\n v_mfma_i32_32x32x32_i8 a[0:15], v[10:13], v[14:17], a[0:15]\n v_mfma_i32_32x32x32_i8 a[16:31], v[10:13], v[14:17], a[16:31]\n .endr\n") \
    X(40, "s_and_saveexec_b64 + s_mov exec (pairs)", ".rept 8\n s_and_saveexec_b64 s[42:43], s[44:45]\n s_mov_b64 exec, s[42:43]\n .endr\n") \
    X(41, "s_cbranch_scc0 not taken", ".rept 16\n s_cbranch_scc0 1f\n .endr\n 1:\n")                                      \
    X(42, "s_branch taken (to next instruction)", ".rept 16\n s_branch 2f\n 2:\n .endr\n")                              \
    X(43, "v_readfirstlane_b32 s,v", ".rept 2\n v_readfirstlane_b32 s40, v10\n v_readfirstlane_b32 s41, v11\n v_readfirstlane_b32 s42, v12\n v_readfirstlane_b32 s43, v13\n v_readfirstlane_b32 s44, v14\n v_readfirstlane_b32 s45, v15\n v_readfirstlane_b32 s46, v16\n v_readfirstlane_b32 s47, v17\n .endr\n") \
    X(44, "v_xor VOP2 : s_add 1:1 : 4 waves typical mix", ".rept 8\n v_xor_b32 v10, v10, v26\n s_add_u32 s40, s40, 1\n .endr\n") \
    X(45, "v_fma VOP3 3-vgpr : v_xor VOP2 1:1", ".rept 8\n v_fma_f32 v10, v10, v26, v27\n v_xor_b32 v11, v11, v26\n .endr\n") \
    X(46, "v_add_u32_sdwa dst WORD_1 (item tag)", RR16("v_add_u32_sdwa", ", v26 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD")) \
    X(47, "v_dot4_i32_i8 v,v,v,v", RR16("v_dot4_i32_i8", ", v26, v27"))

#define KERNEL(ID, NAME, BODY)                                                                                   \
    __global__ __launch_bounds__(1024) void k_s##ID(unsigned long long *out, int iters) {                      \
        __shared__ uint32_t lds[4096];                                                                         \
        for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = i;                                       \
        __syncthreads();                                                                                       \
        asm volatile("v_mov_b32 v26, 0x3f900000\n v_mov_b32 v27, 0x3f800000\n v_lshlrev_b32 v28, 2, %0\n v_lshlrev_b32 v29, 4, %0\n" \
                     "v_mov_b32 v10, 1\n v_mov_b32 v11, 2\n v_mov_b32 v12, 3\n v_mov_b32 v13, 4\n v_mov_b32 v14, 5\n v_mov_b32 v15, 6\n" \
                     "v_mov_b32 v16, 7\n v_mov_b32 v17, 8\n v_mov_b32 v18, 9\n v_mov_b32 v19, 10\n v_mov_b32 v20, 11\n v_mov_b32 v21, 12\n" \
                     "v_mov_b32 v22, 13\n v_mov_b32 v23, 14\n v_mov_b32 v24, 15\n v_mov_b32 v25, 16\n"        \
                     "s_mov_b32 s40, 0x3f800000\n s_mov_b32 s41, 0\n s_mov_b64 s[44:45], exec\n" ::"v"(threadIdx.x & 63) : CLOB); \
        _Pragma("unroll 1") for (int it = 0; it < iters; ++it) asm volatile(".rept 16\n" BODY ".endr\n" ::: CLOB, "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31"); \
        if (iters < 0) out[0] = lds[threadIdx.x];                                                              \
    }
STREAMS(KERNEL)

struct Entry { const char *name; void (*fn)(unsigned long long *, int); };
#define ENTRY(ID, NAME, BODY) {NAME, k_s##ID},
static Entry kTable[] = {STREAMS(ENTRY)};

int main(int argc, char **argv) {
    const int only = argc > 1 ? atoi(argv[1]) : -1;
    unsigned long long *d_out;
    (void)hipMalloc(&d_out, 1 << 20);
    printf("# wall ns per instruction per SIMD (256 instructions per loop iteration + 3 of loop control); clock ~2.4 GHz idle: 0.85 ns = 2 cycles, 1.7 ns = 4 cycles\n");
    printf("%-52s %9s %9s %9s %9s\n", "stream", "1 w/SIMD", "2 w/SIMD", "4 w/SIMD", "8 w/SIMD");
    const int n = (int)(sizeof(kTable) / sizeof(kTable[0]));
    for (int s = 0; s < n; ++s) {
        if (only >= 0 && s != only) continue;
        printf("%-52s", kTable[s].name);
        const int cfg[4][2] = {{256, 256}, {512, 256}, {1024, 256}, {1024, 512}};
        for (int c = 0; c < 4; ++c) {
            const int iters = 2048 / (c == 3 ? 2 : 1);
            hipEvent_t e0, e1;
            (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
            hipLaunchKernelGGL(kTable[s].fn, dim3(cfg[c][1]), dim3(cfg[c][0]), 0, 0, d_out, iters / 8);
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(kTable[s].fn, dim3(cfg[c][1]), dim3(cfg[c][0]), 0, 0, d_out, iters);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float ms = 0;
            (void)hipEventElapsedTime(&ms, e0, e1);
            const double waves_per_simd = (double)cfg[c][0] * cfg[c][1] / 64.0 / 1024.0;
            printf(" %9.2f", (double)ms * 1e6 / ((double)iters * 256.0 * waves_per_simd));
            (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
        }
        printf("\n");
        fflush(stdout);
    }
    return 0;
}
