// launch_cost.hip -- what a kernel's traced duration holds besides its waves' work: empty grids of the pipeline's shapes, and
// grids that write 25 MB (the item lists' volume) with plain / nt / sc0 sc1 stores, to price the end-of-kernel write-back.
// Run under rocprofv3 --kernel-trace --stats.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_empty_512x512() {}
__global__ void k_empty_4096x64() {}
__global__ void k_empty_256x1024() {}
template <int kAux> __global__ __launch_bounds__(512) void k_write(uint32_t *dst, int n16 /*uint4 per thread*/) {
    // every wave writes n16 x 1 KiB, contiguous per wave
    const size_t wave = (size_t)blockIdx.x * 8 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // uniform: no waterfall loop around the stores
    uint32_t *p = dst + wave * (size_t)n16 * 256;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(p, 0, 0x7FFFFFFF, 0x00020000);
    typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
    const u32x4 v = {1u, 2u, 3u, (uint32_t)wave};
    const int l16 = (threadIdx.x & 63) * 16;
    for (int i = 0; i < n16; ++i) __builtin_amdgcn_raw_buffer_store_b128(v, r, l16 + i * 1024, 0, kAux);
}
template <int kAux> __global__ __launch_bounds__(512) void k_write_scatter(uint32_t *dst, int n) {
    // dword stores: lane l writes words l*n .. l*n+n-1 of its wave's 64*n-word range (runs per lane, like the item appends)
    const size_t wave = (size_t)blockIdx.x * 8 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint32_t *p = dst + wave * (size_t)n * 64;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(p, 0, 0x7FFFFFFF, 0x00020000);
    const int l = threadIdx.x & 63;
    for (int i = 0; i < n; ++i) __builtin_amdgcn_raw_buffer_store_b32((uint32_t)i, r, (l * n + i) * 4, 0, kAux);
}
__global__ __launch_bounds__(512) void k_rewrite(uint32_t *dst, int n16) {   // 25 MB of stores into the wave's own 1 KiB: little dirty data at the end
    const size_t wave = (size_t)blockIdx.x * 8 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint32_t *p = dst + wave * 256;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(p, 0, 0x7FFFFFFF, 0x00020000);
    typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
    for (int i = 0; i < n16; ++i) { const u32x4 v = {1u, 2u, 3u, (uint32_t)i}; __builtin_amdgcn_raw_buffer_store_b128(v, r, (threadIdx.x & 63) * 16, 0, 0); }
}
__global__ __launch_bounds__(512) void k_read(const uint4 *src, uint32_t *sink, int n16) {   // every wave reads n16 x 1 KiB
    const size_t wave = (size_t)blockIdx.x * 8 + (threadIdx.x >> 6);
    const uint4 *p = src + wave * (size_t)n16 * 64 + (threadIdx.x & 63);
    uint32_t a = 0;
    for (int i = 0; i < n16; ++i) { const uint4 v = p[i * 64]; a += v.x ^ v.y ^ v.z ^ v.w; }
    if (a == 0x12345u) sink[0] = a;
}
template <int kAux> __global__ __launch_bounds__(512) void k_write_sparse(uint32_t *dst, int n) {
    // n store instructions per wave, lane l active in instruction i iff ((l + i) & 3) == 0: the appends' shape (few lanes per store)
    const size_t wave = (size_t)blockIdx.x * 8 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint32_t *p = dst + wave * (size_t)n * 16;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(p, 0, 0x7FFFFFFF, 0x00020000);
    const int l = threadIdx.x & 63;
    int off = (l >> 2) * n;            // 16 lane quads, each lane of a quad contributes n/4 words to the quad's run of n words
    for (int i = 0; i < n; ++i)
        if (((l + i) & 3) == 0) { __builtin_amdgcn_raw_buffer_store_b32((uint32_t)i, r, off * 4, 0, kAux); ++off; }
}
int main() {
    uint32_t *d; hipMalloc(&d, 256u << 20); hipMemset(d, 0, 256u << 20);
    for (int rep = 0; rep < 20; ++rep) {
        k_empty_512x512<<<512, 512>>>();
        k_empty_4096x64<<<4096, 64>>>();
        k_empty_256x1024<<<256, 1024>>>();
        // 4096 waves x 6 KiB = 25 MB
        k_write<0><<<512, 512>>>(d, 6);
        k_write<2><<<512, 512>>>(d, 6);      // nt
        k_write<17><<<512, 512>>>(d, 6);     // sc0 sc1
        k_write<3><<<512, 512>>>(d, 6);      // sc0 nt
        k_write_scatter<0><<<512, 512>>>(d, 24);   // 4096 waves x 64 x 24 words = 25 MB
        k_write_scatter<2><<<512, 512>>>(d, 24);
        k_write_sparse<0><<<512, 512>>>(d, 96);   // 4096 waves x 96 stores x 16 lanes = 25 MB
        k_write_sparse<0><<<512, 512>>>(d, 48);   // 12.5 MB
        k_write<0><<<512, 512>>>(d, 1);      // 4 MB
        k_write<0><<<512, 512>>>(d, 24);     // 100 MB
        k_write<0><<<512, 512>>>(d, 48);     // 200 MB
        k_rewrite<<<512, 512>>>(d, 6);
        k_read<<<512, 512>>>((const uint4 *)d, d, 48);   // 200 MB
        hipDeviceSynchronize();
    }
    printf("done\n");
    return 0;
}
