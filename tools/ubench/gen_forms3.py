#!/usr/bin/env python3
"""gen_forms3.py > forms3.hip -- round-4 issue-cost census of VALU instruction FORMS on gfx950.

Every stream is `.rept 16` x 16 independent instructions (destinations v10..v25 in rotation); sources: v26, v27, v28 hold
small VGPR constants, s40 / s[42:43] scalar ones.  Reported: wall ns per instruction per SIMD at 2 and 4 waves per SIMD
(0.85 ns = 2 cycles, 1.7 ns = 4 cycles at the ~2.4 GHz these streams hold)."""
FORMS = [
    ("v_add_u32 v,v,v", "v_add_u32 {d}, {d}, v26"),
    ("v_add_u32 v,4,v (inline const)", "v_add_u32 {d}, 4, {d}"),
    ("v_add_u32 v,s,v", "v_add_u32 {d}, s40, {d}"),
    ("v_add_u32 v,0x1234,v (literal)", "v_add_u32 {d}, 0x1234, {d}"),
    ("v_sub_u32 v,v,v", "v_sub_u32 {d}, {d}, v26"),
    ("v_subrev_u32 v,v,v", "v_subrev_u32 {d}, v26, {d}"),
    ("v_and_b32 v,v,v", "v_and_b32 {d}, {d}, v26"),
    ("v_and_b32 v,15,v", "v_and_b32 {d}, 15, {d}"),
    ("v_and_b32 v,0xff00,v (literal)", "v_and_b32 {d}, 0xff00, {d}"),
    ("v_or_b32 v,v,v", "v_or_b32 {d}, {d}, v26"),
    ("v_xor_b32 v,1,v", "v_xor_b32 {d}, 1, {d}"),
    ("v_lshlrev_b32 v,v,v", "v_lshlrev_b32 {d}, v26, {d}"),
    ("v_lshrrev_b32 v,v,v", "v_lshrrev_b32 {d}, v26, {d}"),
    ("v_ashrrev_i32 v,v,v", "v_ashrrev_i32 {d}, v26, {d}"),
    ("v_ashrrev_i32 v,31,v", "v_ashrrev_i32 {d}, 31, {d}"),
    ("v_min_u32 v,v,v", "v_min_u32 {d}, {d}, v26"),
    ("v_max_i32 v,v,v", "v_max_i32 {d}, {d}, v26"),
    ("v_min_f32 v,v,v", "v_min_f32 {d}, {d}, v26"),
    ("v_sub_f32 v,v,v", "v_sub_f32 {d}, {d}, v26"),
    ("v_add_f32 v,1.0,v", "v_add_f32 {d}, 1.0, {d}"),
    ("v_mul_f32 v,s,v", "v_mul_f32 {d}, s40, {d}"),
    ("v_mov_b32 v,v", "v_mov_b32 {d}, v26"),
    ("v_mov_b32 v,s", "v_mov_b32 {d}, s40"),
    ("v_mov_b32 v,1", "v_mov_b32 {d}, 1"),
    ("v_mov_b32 v,literal", "v_mov_b32 {d}, 0x12345"),
    ("v_fmac_f32 v,v,v", "v_fmac_f32 {d}, v26, v27"),
    ("v_fmac_f32 v,s,v", "v_fmac_f32 {d}, s40, v27"),
    ("v_fmaak_f32 v,v,v,K", "v_fmaak_f32 {d}, {d}, v27, 0x3f000000"),
    ("v_fma_f32 v,|v|,v,v (abs mod)", "v_fma_f32 {d}, |{d}|, v26, v27"),
    ("v_add_f32_e64 v,|v|,v", "v_add_f32_e64 {d}, |{d}|, v26"),
    ("v_cndmask_b32 v,v,v,vcc", "v_cndmask_b32 {d}, {d}, v26, vcc"),
    ("v_cndmask_b32_e64 v,v,v,s[42:43]", "v_cndmask_b32_e64 {d}, {d}, v26, s[42:43]"),
    ("v_cmp_ne_u32 vcc,v,v", "v_cmp_ne_u32 vcc, v26, {d}"),
    ("v_cmp_gt_f32 vcc,v,v", "v_cmp_gt_f32 vcc, v26, {d}"),
    ("v_cmp_gt_f32 s[44:45],|v|,s", "v_cmp_gt_f32_e64 s[44:45], |{d}|, s40"),
    ("v_pk_add_u16", "v_pk_add_u16 {d}, {d}, v26"),
    ("v_pk_sub_i16", "v_pk_sub_i16 {d}, {d}, v26"),
    ("v_pk_min_u16", "v_pk_min_u16 {d}, {d}, v26"),
    ("v_pk_max_i16", "v_pk_max_i16 {d}, {d}, v26"),
    ("v_pk_lshlrev_b16", "v_pk_lshlrev_b16 {d}, v26, {d}"),
    ("v_pk_ashrrev_i16", "v_pk_ashrrev_i16 {d}, v26, {d}"),
    ("v_pk_mad_i16", "v_pk_mad_i16 {d}, {d}, v26, v27"),
    ("v_pk_add_f16", "v_pk_add_f16 {d}, {d}, v26"),
    ("v_pk_mul_f16", "v_pk_mul_f16 {d}, {d}, v26"),
    ("v_pk_fma_f16", "v_pk_fma_f16 {d}, {d}, v26, v27"),
    ("v_pk_max_f16", "v_pk_max_f16 {d}, {d}, v26"),
    ("v_add_f16", "v_add_f16 {d}, {d}, v26"),
    ("v_fma_f16", "v_fma_f16 {d}, {d}, v26, v27"),
    ("v_add3_u32", "v_add3_u32 {d}, {d}, v26, v27"),
    ("v_or3_b32", "v_or3_b32 {d}, {d}, v26, v27"),
    ("v_xad_u32", "v_xad_u32 {d}, {d}, v26, v27"),
    ("v_bfi_b32", "v_bfi_b32 {d}, {d}, v26, v27"),
    ("v_bfe_i32 v,v,0,16", "v_bfe_i32 {d}, {d}, 0, 16"),
    ("v_med3_i32", "v_med3_i32 {d}, {d}, v26, v27"),
    ("v_min3_f32", "v_min3_f32 {d}, {d}, v26, v27"),
    ("v_cvt_f16_f32", "v_cvt_f16_f32 {d}, {d}"),
    ("v_cvt_f32_f16", "v_cvt_f32_f16 {d}, {d}"),
    ("v_cvt_pkrtz_f16_f32", "v_cvt_pkrtz_f16_f32 {d}, {d}, v26"),
    ("v_cvt_i32_f32", "v_cvt_i32_f32 {d}, {d}"),
    ("v_cvt_pk_i16_i32", "v_cvt_pk_i16_i32 {d}, {d}, v26"),
    ("v_rndne_f32", "v_rndne_f32 {d}, {d}"),
    ("v_floor_f32", "v_floor_f32 {d}, {d}"),
    ("v_mul_i32_i24", "v_mul_i32_i24 {d}, {d}, v26"),
    ("v_mul_lo_u32", "v_mul_lo_u32 {d}, {d}, v26"),
    ("v_ffbl_b32", "v_ffbl_b32 {d}, {d}"),
    ("v_bcnt_u32_b32", "v_bcnt_u32_b32 {d}, {d}, v26"),
    ("v_mov_b32 dpp quad_perm", "v_mov_b32_dpp {d}, v26 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"),
    ("v_mov_b32 dpp row_shr:1", "v_mov_b32_dpp {d}, v26 row_shr:1 row_mask:0xf bank_mask:0xf"),
    ("v_add_u32 sdwa WORD_1", "v_add_u32_sdwa {d}, v26, v27 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD"),
    ("v_mov_b32 sdwa BYTE_1->BYTE_0", "v_mov_b32_sdwa {d}, v26 dst_sel:BYTE_0 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_1"),
    ("v_perm_b32", "v_perm_b32 {d}, {d}, v26, v27"),
    ("v_alignbit_b32 v,v,v,v", "v_alignbit_b32 {d}, {d}, v26, v27"),
    ("v_lshl_add_u32 v,v,2,v", "v_lshl_add_u32 {d}, {d}, 2, v26"),
    ("v_lshl_or_b32 v,v,16,v", "v_lshl_or_b32 {d}, {d}, 16, v26"),
    ("v_writelane_b32 v,s,3", "v_writelane_b32 {d}, s40, 3"),
    ("v_dot4_i32_i8", "v_dot4_i32_i8 {d}, {d}, v26, v27"),
    ("v_dot2_i32_i16", "v_dot2_i32_i16 {d}, {d}, v26, v27"),
    ("v_pk_fma_f32", None),
    ("v_pk_add_f32", None),
]
print('// generated by gen_forms3.py -- do not edit')
print('#include <hip/hip_runtime.h>\n#include <cstdio>\n#include <cstdlib>\n#include <vector>')
print('#define CLOB "v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","s40","s41","s42","s43","s44","s45","vcc","scc"')
regs = [f"v{10+i}" for i in range(16)]
bodies = []
for name, tmpl in FORMS:
    if tmpl is None:
        op = name
        lines = [f"{op} v[{10+2*i}:{11+2*i}], v[{10+2*i}:{11+2*i}], v[26:27]" + (", v[28:29]" if "fma" in op else "") for i in range(8)] * 2
    else:
        lines = [tmpl.format(d=r) for r in regs]
    bodies.append("\\n".join(lines))
print('template <int S> __global__ __launch_bounds__(1024) void k_form(unsigned long long *out, int iters) {')
print('  asm volatile("v_mov_b32 v26, 3\\n v_mov_b32 v27, 5\\n v_mov_b32 v28, 7\\n v_mov_b32 v29, 9\\n s_mov_b32 s40, 3\\n s_mov_b64 s[42:43], 0x5555\\n s_mov_b64 vcc, 0x3333\\n"')
print('    ' + ' '.join(f'"v_mov_b32 v{10+i}, {i+1}\\n"' for i in range(16)) + ' ::: CLOB);')
print('  unsigned long long t0, t1;\n  asm volatile("s_memtime %0\\n s_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");')
print('#pragma unroll 1\n  for (int it = 0; it < iters; ++it) {')
for i, b in enumerate(bodies):
    print(f'    if (S == {i}) asm volatile(".rept 16\\n{b}\\n.endr\\n" ::: CLOB);')
print('  }\n  asm volatile("s_memtime %0\\n s_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");')
print('  if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;\n}')
print(f'static const char *kNames[{len(FORMS)}] = {{' + ', '.join('"' + n + '"' for n, _ in FORMS) + '};')
print('template <int S> static void launch(int blocks, int threads, unsigned long long *o, int it) { hipLaunchKernelGGL(k_form<S>, dim3(blocks), dim3(threads), 0, 0, o, it); }')
print('typedef void (*LaunchFn)(int, int, unsigned long long *, int);')
print(f'static LaunchFn kLaunch[{len(FORMS)}] = {{' + ', '.join(f'launch<{i}>' for i in range(len(FORMS))) + '};')
print(r'''
int main() {
    unsigned long long *d; hipMalloc(&d, 8192 * 8);
    printf("# wall ns per instruction per SIMD (256 per loop iteration); 0.85 ns ~ 2 cycles, 1.7 ns ~ 4 cycles\n%-40s %10s %10s\n", "form", "2 w/SIMD", "4 w/SIMD");
    for (int s = 0; s < (int)(sizeof(kLaunch) / sizeof(kLaunch[0])); ++s) {
        printf("%-40s", kNames[s]);
        for (int c = 0; c < 2; ++c) {
            const int threads = c ? 1024 : 512, blocks = 256, iters = 2048;
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            kLaunch[s](blocks, threads, d, iters / 8);
            hipEventRecord(e0); kLaunch[s](blocks, threads, d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double waves_per_simd = blocks * threads / 64 / 1024.0;
            printf(" %10.2f", (double)ms * 1e6 / (256.0 * iters * waves_per_simd));
            hipEventDestroy(e0); hipEventDestroy(e1);
        }
        printf("\n"); fflush(stdout);
    }
    return 0;
}''')
