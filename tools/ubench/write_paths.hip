// write_paths.hip -- why does a grid of 4096 waves storing 25 MB take 45 us?  Variants of the store path, same volume.
// Run under rocprofv3 --kernel-trace (tools/gpu_launch_cost.sh write_paths).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
// A: plain global stores, uint4 per lane, wave-contiguous, 4096 waves x 6
__global__ __launch_bounds__(512) void k_A_global_x4(uint4 *dst, int n16) {
    const size_t wave = (size_t)blockIdx.x * 8 + (threadIdx.x >> 6);
    uint4 *p = dst + wave * (size_t)n16 * 64 + (threadIdx.x & 63);
    for (int i = 0; i < n16; ++i) p[i * 64] = make_uint4(1, 2, 3, (uint32_t)i);
}
// B: grid-stride, one uint4 per thread per step, the whole grid contiguous (like a fill kernel)
__global__ __launch_bounds__(512) void k_B_gridstride(uint4 *dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 512 + threadIdx.x; i < n; i += (size_t)gridDim.x * 512) dst[i] = make_uint4(1, 2, 3, (uint32_t)i);
}
// C: as B with a big grid: one uint4 per thread
__global__ __launch_bounds__(256) void k_C_onepass(uint4 *dst, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = make_uint4(1, 2, 3, (uint32_t)i);
}
// D: dword per lane, grid contiguous
__global__ __launch_bounds__(256) void k_D_dword(uint32_t *dst, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = (uint32_t)i;
}
// E: A, but the waves wait for their stores before ending (s_waitcnt vmcnt(0))
__global__ __launch_bounds__(512) void k_E_wait(uint4 *dst, int n16) {
    const size_t wave = (size_t)blockIdx.x * 8 + (threadIdx.x >> 6);
    uint4 *p = dst + wave * (size_t)n16 * 64 + (threadIdx.x & 63);
    for (int i = 0; i < n16; ++i) p[i * 64] = make_uint4(1, 2, 3, (uint32_t)i);
    __builtin_amdgcn_s_waitcnt(0);
}
// F: A with data that differs per store and per lane (rule out any same-value effect)
__global__ __launch_bounds__(512) void k_F_data(uint4 *dst, int n16) {
    const size_t wave = (size_t)blockIdx.x * 8 + (threadIdx.x >> 6);
    uint4 *p = dst + wave * (size_t)n16 * 64 + (threadIdx.x & 63);
    for (int i = 0; i < n16; ++i) p[i * 64] = make_uint4(threadIdx.x * 77u + i, blockIdx.x, 3u * i, (uint32_t)wave);
}
int main() {
    uint4 *d; (void)hipMalloc(&d, 256u << 20); (void)hipMemset(d, 0, 256u << 20);
    const size_t n = (25u << 20) / 16;
    for (int rep = 0; rep < 20; ++rep) {
        k_A_global_x4<<<512, 512>>>(d, 6);
        k_B_gridstride<<<512, 512>>>(d, n);
        k_C_onepass<<<(unsigned)((n + 255) / 256), 256>>>(d, n);
        k_D_dword<<<(unsigned)((n * 4 + 255) / 256), 256>>>((uint32_t *)d, n * 4);
        k_E_wait<<<512, 512>>>(d, 6);
        k_F_data<<<512, 512>>>(d, 6);
        (void)hipMemsetAsync(d, 0, 25u << 20, 0);
        (void)hipDeviceSynchronize();
    }
    printf("done\n");
    return 0;
}
