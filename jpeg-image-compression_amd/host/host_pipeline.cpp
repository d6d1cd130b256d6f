// host_pipeline.cpp -- file-to-file batch encoding with the transfers overlapped.
//
// What natural_c's main.c does for ONE file (loadBMPImage -> saveJPEGGrayscale, natural_c/src/main.c:21-24) is
// dominated on a GPU by everything around the kernels: reading 201 MB, moving it over PCIe, writing the result.
// jpegamd_encode_files keeps kSlots files in flight.  Each slot owns pinned host buffers, device buffers, a HIP
// stream and an encoder context; for file i the host thread
//     reads the BMP into the slot's pinned buffer        (header rules of bmp_handler.c:22-88 via jpegamd_parse_bmp;
//                                                         no flip, no channel swap: those are addressing modes of the kernel)
//     enqueues H2D of the pixel rows, the encode, and the D2H of the 8-byte size
// and then drains the slot of file i - (kSlots - 1): waits for its size, enqueues the D2H of exactly that many bytes,
// waits, writes the .jpg.  While the host sits in fread / fwrite of one file the GPU works on the others.
// Host code only; every byte of the stream is produced by the same device path as jpegamd_encode_async.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>

#include "jpeg_compression.h"

namespace {

constexpr int kSlots = 3;

struct Slot {
    JpegAmdEncoder *enc = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t size_ready = nullptr;
    uint8_t *h_in = nullptr, *d_in = nullptr, *d_out = nullptr, *h_out = nullptr;
    uint64_t *d_size = nullptr, *h_size = nullptr;
    size_t in_cap = 0, out_cap = 0;
    int32_t enc_w = 0, enc_h = 0;
    int file = -1;                  // index of the file in flight, -1 = idle
    int32_t status = 0;
    uint64_t in_bytes = 0;

    void release() {
        if (enc) jpegamd_encoder_destroy(enc);
        if (stream) hipStreamDestroy(stream);
        if (size_ready) hipEventDestroy(size_ready);
        if (h_in) hipHostFree(h_in);
        if (h_out) hipHostFree(h_out);
        if (h_size) hipHostFree(h_size);
        if (d_in) hipFree(d_in);
        if (d_out) hipFree(d_out);
        if (d_size) hipFree(d_size);
        *this = Slot();
    }
    bool grow_in(size_t n) {
        if (n <= in_cap) return true;
        if (h_in) hipHostFree(h_in);
        if (d_in) hipFree(d_in);
        h_in = d_in = nullptr; in_cap = 0;
        if (hipHostMalloc((void **)&h_in, n, hipHostMallocDefault) != hipSuccess) return false;
        if (hipMalloc((void **)&d_in, n) != hipSuccess) return false;
        in_cap = n;
        return true;
    }
    bool grow_out(size_t n) {
        if (n <= out_cap) return true;
        if (h_out) hipHostFree(h_out);
        if (d_out) hipFree(d_out);
        h_out = d_out = nullptr; out_cap = 0;
        if (hipHostMalloc((void **)&h_out, n, hipHostMallocDefault) != hipSuccess) return false;
        if (hipMalloc((void **)&d_out, n) != hipSuccess) return false;
        out_cap = n;
        return true;
    }
};

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// read, parse and enqueue one file on `s`; returns 0 or a JPEGAMD_ERR_* code (the slot stays idle on error)
int32_t submit(Slot &s, const char *path, int32_t quality, double *t_read) {
    const double t0 = now_s();
    FILE *fp = path ? std::fopen(path, "rb") : nullptr;
    if (!fp) return JPEGAMD_ERR_BMP;
    std::fseek(fp, 0, SEEK_END);
    const long len = std::ftell(fp);
    std::fseek(fp, 0, SEEK_SET);
    if (len <= 0 || !s.grow_in((size_t)len)) { std::fclose(fp); return len <= 0 ? JPEGAMD_ERR_BMP : JPEGAMD_ERR_HIP; }
    const size_t got = std::fread(s.h_in, 1, (size_t)len, fp);
    std::fclose(fp);
    *t_read += now_s() - t0;

    JpegAmdImage img;
    uint64_t off = 0;
    int32_t rc = jpegamd_parse_bmp(s.h_in, got, &img, &off);
    if (rc) return rc;
    if (!s.enc || img.width > s.enc_w || img.height > s.enc_h) {        // contexts are sized for the largest image seen
        if (s.enc) jpegamd_encoder_destroy(s.enc);
        s.enc = nullptr;
        const int32_t w = img.width > s.enc_w ? img.width : s.enc_w, h = img.height > s.enc_h ? img.height : s.enc_h;
        rc = jpegamd_encoder_create(&s.enc, w, h);
        if (rc) return rc;
        s.enc_w = w; s.enc_h = h;
    }
    const size_t bytes = (size_t)img.row_stride * (size_t)img.height;
    const uint64_t cap = jpegamd_max_jfif_bytes(img.width, img.height);
    // the output buffer is sized for typical content (1 byte per pixel + container); the encoder reports -8 beyond it
    const size_t out_cap = (size_t)img.width * (size_t)img.height + 4096 < cap ? (size_t)img.width * (size_t)img.height + 4096 : (size_t)cap;
    if (!s.grow_out(out_cap)) return JPEGAMD_ERR_HIP;
    if (hipMemcpyAsync(s.d_in, s.h_in + off, bytes, hipMemcpyHostToDevice, s.stream) != hipSuccess) return JPEGAMD_ERR_HIP;
    img.pixels = s.d_in;
    img.quality = quality;
    rc = jpegamd_encode_async(s.enc, &img, s.d_out, s.out_cap, s.d_size, 1, (void *)s.stream);
    if (rc) return rc;
    if (hipMemcpyAsync(s.h_size, s.d_size, sizeof(uint64_t), hipMemcpyDeviceToHost, s.stream) != hipSuccess) return JPEGAMD_ERR_HIP;
    if (hipEventRecord(s.size_ready, s.stream) != hipSuccess) return JPEGAMD_ERR_HIP;
    s.in_bytes = (uint64_t)len;
    return JPEGAMD_OK;
}

// wait for the slot's stream, fetch exactly the produced bytes, write the file
int32_t drain(Slot &s, const char *path, uint64_t *out_bytes, double *t_write) {
    if (hipEventSynchronize(s.size_ready) != hipSuccess) return JPEGAMD_ERR_HIP;
    int32_t rc = jpegamd_encoder_finish(s.enc, nullptr);      // status of the call (-8 when the stream outgrew the buffer)
    if (rc) return rc;
    const uint64_t n = *s.h_size;
    if (n == 0 || n > s.out_cap) return JPEGAMD_ERR_HUFF_CAPACITY;
    if (hipMemcpyAsync(s.h_out, s.d_out, (size_t)n, hipMemcpyDeviceToHost, s.stream) != hipSuccess) return JPEGAMD_ERR_HIP;
    if (hipStreamSynchronize(s.stream) != hipSuccess) return JPEGAMD_ERR_HIP;
    const double t0 = now_s();
    FILE *fp = path ? std::fopen(path, "wb") : nullptr;
    if (!fp) return JPEGAMD_ERR_ARG;
    const size_t put = std::fwrite(s.h_out, 1, (size_t)n, fp);
    const bool ok = std::fclose(fp) == 0 && put == (size_t)n;
    *t_write += now_s() - t0;
    if (!ok) return JPEGAMD_ERR_ARG;
    *out_bytes = n;
    return JPEGAMD_OK;
}

}  // namespace

extern "C" int32_t jpegamd_encode_files(const char *const *in_paths, const char *const *out_paths, int32_t count, int32_t quality,
                                        int32_t *status, JpegAmdBatchStats *stats) {
    if (count < 0 || (count > 0 && (!in_paths || !out_paths))) return JPEGAMD_ERR_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return JPEGAMD_ERR_NO_DEVICE;
    JpegAmdBatchStats bs;
    std::memset(&bs, 0, sizeof(bs));
    const double t_begin = now_s();
    std::vector<Slot> slots((size_t)(count < kSlots ? (count > 0 ? count : 1) : kSlots));
    int32_t fatal = JPEGAMD_OK;
    for (Slot &s : slots) {
        if (hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&s.size_ready, hipEventDisableTiming) != hipSuccess ||
            hipMalloc((void **)&s.d_size, sizeof(uint64_t)) != hipSuccess ||
            hipHostMalloc((void **)&s.h_size, sizeof(uint64_t), hipHostMallocDefault) != hipSuccess) { fatal = JPEGAMD_ERR_HIP; break; }
    }
    int failed = 0;
    const auto finish_slot = [&](Slot &s) {
        if (s.file < 0) return;
        uint64_t nb = 0;
        const int32_t rc = drain(s, out_paths[s.file], &nb, &bs.seconds_write);
        if (status) status[s.file] = rc;
        if (rc) ++failed; else { bs.bytes_out += nb; bs.bytes_in += s.in_bytes; ++bs.files_ok; }
        s.file = -1;
    };
    if (!fatal) {
        for (int i = 0; i < count; ++i) {
            Slot &s = slots[(size_t)i % slots.size()];
            finish_slot(s);                                        // the slot's previous file (i - slots) must be out first
            const int32_t rc = submit(s, in_paths[i], quality, &bs.seconds_read);
            if (rc) { if (status) status[i] = rc; ++failed; continue; }
            s.file = i;
        }
        for (size_t k = 0; k < slots.size(); ++k) finish_slot(slots[((size_t)count + k) % slots.size()]);   // oldest first
    }
    for (Slot &s : slots) s.release();
    bs.seconds_total = now_s() - t_begin;
    bs.files_failed = failed;
    if (stats) *stats = bs;
    if (fatal) return fatal;
    return failed ? JPEGAMD_ERR_BMP : JPEGAMD_OK;
}
