// host_multi.cpp -- the multi-GPU exchange as a C entry point: gather-v of finished bitstreams over RCCL.
//
// The reference's host side is C (dsp_port/jpeg_client/main.c:397-530 prepares buffers, calls the accelerator, collects the
// result); its accelerator is one DSP core, so it has no exchange step.  Here the path shards by independent images over the
// GPUs of a node (one process per GPU), and the one exchange -- every rank's finished JFIF streams to a root -- is
// jpegamd_gather_streams: what python/jpegamd/sharding.py::ExactStreamGather does over torch.distributed, for a C host that
// owns an RCCL communicator.  RCCL is loaded at the first call (dlopen): the library itself does not link against it, and a
// single-GPU user never loads it.
//
//   records     DEVICE, `slots` staging records of `slot_bytes` each: the stream at offset 0, its byte count (uint64) in the
//               record's last 8 bytes -- what jpegamd_encode_async leaves when pointed at (record, slot_bytes - 8, record +
//               slot_bytes - 8)
//   sizes_host  HOST, [world][slots] uint64, filled on EVERY rank (a stream longer than slot_bytes - 8 was cut by the encoder:
//               the caller sees that here, encodes it again into a buffer of that size and sends it by itself)
//   recv        DEVICE, root only: rank r's streams land densely at recv + r * recv_stride, each rounded up to 8 bytes, in
//               record order (the root's own records are copied there too, device to device)
// One size all-gather (device), one host wait for it, then one grouped launch of exact-size sends / receives on `stream`.
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>

#include <cstdio>
#include <cstring>
#include <mutex>
#include <vector>

#include "jpeg_compression.h"

namespace {

// the few RCCL entry points used, resolved once (types as in rccl.h: ncclComm_t is a pointer, ncclDataType_t an int enum)
typedef int (*fn_allgather)(const void *, void *, size_t, int, void *, hipStream_t);
typedef int (*fn_sendrecv)(void *, size_t, int, int, void *, hipStream_t);
typedef int (*fn_group)(void);
struct Rccl {
    void *lib = nullptr;
    fn_allgather all_gather = nullptr;
    fn_sendrecv send = nullptr, recv = nullptr;
    fn_group group_start = nullptr, group_end = nullptr;
    bool ok = false;
};
constexpr int kNcclUint8 = 1, kNcclUint64 = 5;             // rccl.h: ncclUint8 = 1, ncclUint64 = 5

Rccl &rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char *name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
            r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (r.lib) break;
        }
        if (!r.lib) return;
        r.all_gather = (fn_allgather)dlsym(r.lib, "ncclAllGather");
        r.send = (fn_sendrecv)dlsym(r.lib, "ncclSend");
        r.recv = (fn_sendrecv)dlsym(r.lib, "ncclRecv");
        r.group_start = (fn_group)dlsym(r.lib, "ncclGroupStart");
        r.group_end = (fn_group)dlsym(r.lib, "ncclGroupEnd");
        r.ok = r.all_gather && r.send && r.recv && r.group_start && r.group_end;
    });
    return r;
}

}  // namespace

extern "C" int32_t jpegamd_gather_streams(void *rccl_comm, int32_t rank, int32_t world, int32_t root, const void *records,
                                          uint64_t slot_bytes, int32_t slots, uint64_t *sizes_host, void *recv, uint64_t recv_stride,
                                          void *stream_) {
    if (!rccl_comm || world < 1 || rank < 0 || rank >= world || root < 0 || root >= world || !records || slots < 1 ||
        slot_bytes < 16 || (slot_bytes & 7u) || !sizes_host || (rank == root && !recv))
        return JPEGAMD_ERR_ARG;
    Rccl &r = rccl();
    if (!r.ok) {
        std::fprintf(stderr, "jpegamd: librccl.so could not be loaded (%s)\n", dlerror() ? dlerror() : "symbols missing");
        return JPEGAMD_ERR_NO_DEVICE;
    }
    hipStream_t stream = (hipStream_t)stream_;
    const uint64_t cap = slot_bytes - 8;
    uint64_t *table_dev = nullptr;                           // [world + 1][slots]: every rank's sizes, then this rank's own
    if (hipMalloc((void **)&table_dev, (size_t)(world + 1) * slots * sizeof(uint64_t)) != hipSuccess) return JPEGAMD_ERR_HIP;
    uint64_t *mine_dev = table_dev + (size_t)world * slots;
    int32_t rc = JPEGAMD_OK;
    // the size column of the records (strided) -> dense, then to every rank
    if (hipMemcpy2DAsync(mine_dev, 8, (const uint8_t *)records + cap, slot_bytes, 8, (size_t)slots, hipMemcpyDeviceToDevice, stream) != hipSuccess ||
        r.all_gather(mine_dev, table_dev, (size_t)slots, kNcclUint64, rccl_comm, stream) != 0 ||
        hipMemcpyAsync(sizes_host, table_dev, (size_t)world * slots * sizeof(uint64_t), hipMemcpyDeviceToHost, stream) != hipSuccess ||
        hipStreamSynchronize(stream) != hipSuccess)
        rc = JPEGAMD_ERR_HIP;
    if (rc == JPEGAMD_OK) {
        const auto wire = [&](uint64_t n) -> uint64_t { return n > cap ? 0 : (n + 7u) & ~7ull; };     // a cut stream does not travel (see above)
        bool failed = false;
        if (rank == root) {
            for (int rr = 0; rr < world && !failed; ++rr) {
                uint64_t total = 0;
                for (int k = 0; k < slots; ++k) total += wire(sizes_host[(size_t)rr * slots + k]);
                if (total > recv_stride) { failed = true; rc = JPEGAMD_ERR_HUFF_CAPACITY; }
            }
        }
        if (!failed) {
            failed = r.group_start() != 0;
            if (rank != root) {
                for (int k = 0; k < slots && !failed; ++k) {
                    const uint64_t n8 = wire(sizes_host[(size_t)rank * slots + k]);
                    if (n8) failed = r.send((uint8_t *)records + (size_t)k * slot_bytes, (size_t)n8, kNcclUint8, root, rccl_comm, stream) != 0;
                }
            } else {
                for (int rr = 0; rr < world && !failed; ++rr) {
                    uint64_t off = 0;
                    for (int k = 0; k < slots && !failed; ++k) {
                        const uint64_t n8 = wire(sizes_host[(size_t)rr * slots + k]);
                        if (!n8) continue;
                        uint8_t *dst = (uint8_t *)recv + (size_t)rr * recv_stride + off;
                        if (rr == rank)                       // the root's own streams: a local copy
                            failed = hipMemcpyAsync(dst, (const uint8_t *)records + (size_t)k * slot_bytes, (size_t)n8, hipMemcpyDeviceToDevice, stream) != hipSuccess;
                        else
                            failed = r.recv(dst, (size_t)n8, kNcclUint8, rr, rccl_comm, stream) != 0;
                        off += n8;
                    }
                }
            }
            if (r.group_end() != 0) failed = true;
            if (failed && rc == JPEGAMD_OK) rc = JPEGAMD_ERR_HIP;
        }
    }
    if (hipStreamSynchronize(stream) != hipSuccess && rc == JPEGAMD_OK) rc = JPEGAMD_ERR_HIP;    // (table_dev must outlive the all-gather)
    hipFree(table_dev);
    return rc;
}
