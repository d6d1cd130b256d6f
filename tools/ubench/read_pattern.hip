// read_pattern.hip -- how fast can 4096 persistent waves read a 8192 x 8192 x 3-byte picture in k_tile_transform's pattern?
// A tile = 32 blocks of one block row: 8 rows x 768 contiguous bytes, rows 24 576 bytes apart.
//   k_lane24   : lane (h, b) reads 24 bytes (dwordx4 + dwordx2) of rows 2s + h, s = 0..3   (the kernel's loads)
//   k_lane16   : the same 6 KiB per tile, but 16 contiguous bytes per lane (48 lanes per row, 8 rows)
//   k_prefetch : k_lane24 with the next tile's loads issued before the current tile's data are consumed
// Three distinct pictures are read in rotation (603 MB > the 256 MB Infinity Cache).  Run under rocprofv3 --kernel-trace.
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int kW = 8192, kH = 8192, kStride = kW * 3, kTilesPerRow = 32, kTiles = 32768;
__device__ __forceinline__ int tile_of(int wave, int k) { return k * 4096 + wave; }      // 8 tiles per wave, consecutive waves side by side
__global__ __launch_bounds__(512) void k_lane24(const uint8_t *pix, uint32_t *sink) {
    const int lane = threadIdx.x & 63, wave = blockIdx.x * 8 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5, b = lane & 31;
    uint32_t acc = 0;
    for (int k = 0; k < 8; ++k) {
        const int tile = tile_of(wave, k), by = tile / kTilesPerRow, tx = tile % kTilesPerRow;
        const uint8_t *base = pix + (size_t)(by * 8 + h) * kStride + (size_t)tx * 768 + b * 24;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const uint4 a = *reinterpret_cast<const uint4 *>(base + (size_t)(2 * s) * kStride);
            const uint2 c = *reinterpret_cast<const uint2 *>(base + (size_t)(2 * s) * kStride + 16);
            acc ^= a.x ^ a.y ^ a.z ^ a.w ^ c.x ^ c.y;
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
__global__ __launch_bounds__(512) void k_lane16(const uint8_t *pix, uint32_t *sink) {
    const int lane = threadIdx.x & 63, wave = blockIdx.x * 8 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint32_t acc = 0;
    for (int k = 0; k < 8; ++k) {
        const int tile = tile_of(wave, k), by = tile / kTilesPerRow, tx = tile % kTilesPerRow;
        const uint8_t *base = pix + (size_t)(by * 8) * kStride + (size_t)tx * 768;
        // 8 rows x 48 pieces of 16 bytes = 384 pieces = 6 instructions of 64 lanes
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int piece = i * 64 + lane, row = piece / 48, col = piece % 48;
            const uint4 a = *reinterpret_cast<const uint4 *>(base + (size_t)row * kStride + col * 16);
            acc ^= a.x ^ a.y ^ a.z ^ a.w;
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
__global__ __launch_bounds__(512) void k_prefetch(const uint8_t *pix, uint32_t *sink) {
    const int lane = threadIdx.x & 63, wave = blockIdx.x * 8 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5, b = lane & 31;
    uint32_t acc = 0;
    uint4 a[4]; uint2 c[4];
    auto req = [&](int k) {
        const int tile = tile_of(wave, k), by = tile / kTilesPerRow, tx = tile % kTilesPerRow;
        const uint8_t *base = pix + (size_t)(by * 8 + h) * kStride + (size_t)tx * 768 + b * 24;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            a[s] = *reinterpret_cast<const uint4 *>(base + (size_t)(2 * s) * kStride);
            c[s] = *reinterpret_cast<const uint2 *>(base + (size_t)(2 * s) * kStride + 16);
        }
    };
    req(0);
    for (int k = 0; k < 8; ++k) {
        uint4 a0[4]; uint2 c0[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) { a0[s] = a[s]; c0[s] = c[s]; }
        if (k + 1 < 8) req(k + 1);
#pragma unroll
        for (int s = 0; s < 4; ++s) acc ^= a0[s].x ^ a0[s].y ^ a0[s].z ^ a0[s].w ^ c0[s].x ^ c0[s].y;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
int main() {
    uint8_t *p[3]; uint32_t *sink;
    for (auto &q : p) { (void)hipMalloc(&q, (size_t)kStride * kH); (void)hipMemset(q, 1, (size_t)kStride * kH); }
    (void)hipMalloc(&sink, 64);
    for (int rep = 0; rep < 30; ++rep) {
        k_lane24<<<512, 512>>>(p[rep % 3], sink);
        k_lane16<<<512, 512>>>(p[(rep + 1) % 3], sink);
        k_prefetch<<<512, 512>>>(p[(rep + 2) % 3], sink);
        (void)hipDeviceSynchronize();
    }
    printf("done\n");
    return 0;
}
