// lds_dma_test.hip -- semantics check of the LDS-DMA form k_tile_transform uses: global_load_lds_dwordx4 with a scalar base,
// a per-lane 32-bit offset and M0 = LDS byte address of the wave's 1 KiB piece (lane l lands at M0 + 16 l).
// Build: hipcc --offload-arch=gfx950 -O3 lds_dma_test.hip -o lds_dma_test
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(256) void k(const uint8_t *src, uint32_t *out, int row_stride) {
    __shared__ __attribute__((aligned(16))) uint32_t s_raw[4][8 * 192];        // per wave: 8 rows x 768 bytes
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint8_t *tb = src + (size_t)wave * 768;                               // tile `wave` of the row group
    const uint32_t lds0 = (uint32_t)(uintptr_t)&s_raw[wave][0];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const uint32_t o = 1024u * i + 16u * lane;                              // byte offset in the row-major [8][768] image
        const uint32_t row = o / 768u, col = o - row * 768u;
        const uint32_t voff = row * (uint32_t)row_stride + col;
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(lds0 + 1024u * i), "s"(tb) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int j = lane; j < 8 * 192; j += 64) out[wave * 8 * 192 + j] = s_raw[wave][j];
}

int main() {
    const int stride = 8192 * 3, rows = 8;
    std::vector<uint8_t> h(stride * rows);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (uint8_t)((i * 2654435761u) >> 13);
    uint8_t *d; uint32_t *o;
    hipMalloc(&d, h.size()); hipMalloc(&o, 4 * 8 * 192 * 4);
    hipMemcpy(d, h.data(), h.size(), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, d, o, stride);
    std::vector<uint32_t> r(4 * 8 * 192);
    hipMemcpy(r.data(), o, r.size() * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int w = 0; w < 4; ++w)
        for (int row = 0; row < 8; ++row)
            for (int c = 0; c < 768; ++c) {
                const uint8_t got = ((const uint8_t *)r.data())[(w * 8 + row) * 768 + c];
                const uint8_t exp = h[(size_t)row * stride + w * 768 + c];
                if (got != exp && bad++ < 5) printf("mismatch wave %d row %d col %d: %02x vs %02x\n", w, row, c, got, exp);
            }
    printf(bad ? "LDS-DMA layout: %d mismatches\n" : "LDS-DMA layout OK (row-major [8][768] per wave)\n", bad);
    return bad != 0;
}
