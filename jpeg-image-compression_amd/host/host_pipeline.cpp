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
// waits, writes the .jpg.  The reads themselves run in the background: kReaders staging threads (reader t fills the slots of
// files t, t + kReaders, ... as soon as the slot's previous upload has left its pinned buffer), each splitting its file over
// kReadLanes pread() calls on disjoint chunks -- one thread copying out of the page cache into pinned memory tops out near
// 5 GB/s, far below PCIe.  The slots (pinned + device buffers, stream, encoder context: hundreds of MB of allocations) are
// kept between calls.
// Host code only; every byte of the stream is produced by the same device path as jpegamd_encode_async.
#include <hip/hip_runtime.h>

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#include "jpeg_compression.h"

namespace {

constexpr int kSlots = 4;
constexpr int kReaders = 2;
constexpr int kReadLanes = 6;                     // pread() calls in flight per file being staged (12 copying threads in all)
constexpr size_t kReadChunk = 8u << 20;

struct Slot {
    JpegAmdEncoder *enc = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t size_ready = nullptr;
    uint8_t *h_in = nullptr, *d_in = nullptr, *d_out = nullptr, *h_out = nullptr;
    uint64_t *d_size = nullptr, *h_size = nullptr;
    size_t in_cap = 0, out_cap = 0;
    size_t d_in_cap = 0;
    JpegAmdImage img_inflight;                 // the file being encoded (for the second attempt with a worst-case buffer)
    int32_t enc_w = 0, enc_h = 0;
    int file = -1;                  // index of the file in flight, -1 = idle
    int32_t status = 0;
    uint64_t in_bytes = 0;
    // hand-over between the reader thread and the submitting thread
    hipEvent_t h2d_done = nullptr;  // the upload has left h_in
    std::mutex mu;
    std::condition_variable cv;
    int staged = -1;                // file whose bytes are complete in h_in (-1: none / consumed)
    int next_file = 0, stride = 1;  // the slot takes files next_file, next_file + stride, ... strictly in that order
    int32_t staged_rc = 0;          // read error of that file
    size_t staged_len = 0;
    bool h2d_pending = false;       // an upload from h_in was enqueued and not yet known to be finished

    void release() {
        if (enc) jpegamd_encoder_destroy(enc);
        if (stream) hipStreamDestroy(stream);
        if (size_ready) hipEventDestroy(size_ready);
        if (h2d_done) hipEventDestroy(h2d_done);
        if (h_in) hipHostFree(h_in);
        if (h_out) hipHostFree(h_out);
        if (h_size) hipHostFree(h_size);
        if (d_in) hipFree(d_in);
        if (d_out) hipFree(d_out);
        if (d_size) hipFree(d_size);
        enc = nullptr; stream = nullptr; size_ready = h2d_done = nullptr;
        h_in = d_in = d_out = h_out = nullptr; d_size = h_size = nullptr;
        in_cap = d_in_cap = out_cap = 0; file = -1;
    }
    // The pinned host buffer is grown by the reader thread (the previous upload from it has completed: h2d_done); the DEVICE
    // buffer only by the submitting thread, after the slot's previous file has been drained -- its kernels read d_in, and a
    // file that did not fit its output buffer is encoded a second time from it.
    bool grow_in(size_t n) {
        if (n <= in_cap) return true;
        if (h_in) hipHostFree(h_in);
        h_in = nullptr; in_cap = 0;
        if (hipHostMalloc((void **)&h_in, n, hipHostMallocDefault) != hipSuccess) return false;
        in_cap = n;
        return true;
    }
    bool grow_dev_in(size_t n) {
        if (n <= d_in_cap) return true;
        if (d_in) hipFree(d_in);
        d_in = nullptr; d_in_cap = 0;
        if (hipMalloc((void **)&d_in, n) != hipSuccess) return false;
        d_in_cap = n;
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

// background: bring `path` into the slot's pinned buffer (grown on demand: only this thread touches h_in / d_in sizes
// while the slot is not staged, and the submitting thread only reads them while it is)
void stage_file(Slot &s, int index, const char *path, std::atomic<long long> *read_ns) {
    {   // the previous upload from this buffer must be through
        std::unique_lock<std::mutex> lk(s.mu);
        s.cv.wait(lk, [&] { return s.staged < 0 && s.next_file == index; });   // readers of one slot may arrive out of order
        if (s.h2d_pending) { hipEventSynchronize(s.h2d_done); s.h2d_pending = false; }
    }
    const double t0 = now_s();
    int32_t rc = JPEGAMD_OK;
    size_t got = 0;
    const int fd = path ? ::open(path, O_RDONLY) : -1;
    if (fd < 0) {
        rc = JPEGAMD_ERR_BMP;
    } else {
        struct stat sb;
        const long long len = ::fstat(fd, &sb) == 0 ? (long long)sb.st_size : -1;
        if (len <= 0) rc = JPEGAMD_ERR_BMP;
        else if (!s.grow_in((size_t)len)) rc = JPEGAMD_ERR_HIP;
        else {
            // chunks of the file, handed out to kReadLanes threads through one counter
            const size_t nchunks = ((size_t)len + kReadChunk - 1) / kReadChunk;
            std::atomic<size_t> next{0}, done{0};
            const auto lane = [&] {
                for (size_t c = next.fetch_add(1); c < nchunks; c = next.fetch_add(1)) {
                    size_t off = c * kReadChunk, left = (size_t)len - off < kReadChunk ? (size_t)len - off : kReadChunk;
                    while (left) {
                        const ssize_t n = ::pread(fd, s.h_in + off, left, (off_t)off);
                        if (n <= 0) break;
                        off += (size_t)n; left -= (size_t)n; done.fetch_add((size_t)n);
                    }
                }
            };
            std::vector<std::thread> lanes;
            const int nl = nchunks < (size_t)kReadLanes ? (int)nchunks : kReadLanes;
            for (int t = 1; t < nl; ++t) lanes.emplace_back(lane);
            lane();
            for (std::thread &t : lanes) t.join();
            got = done.load();
        }
        ::close(fd);
    }
    read_ns->fetch_add((long long)((now_s() - t0) * 1e9));
    std::lock_guard<std::mutex> lk(s.mu);
    s.staged = index; s.staged_rc = rc; s.staged_len = got;
    s.cv.notify_all();
}

// parse and enqueue the staged file of `s`; returns 0 or a JPEGAMD_ERR_* code (the slot stays idle on error)
int32_t submit(Slot &s, int index, int32_t quality) {
    size_t got;
    {
        std::unique_lock<std::mutex> lk(s.mu);
        s.cv.wait(lk, [&] { return s.staged == index; });
        if (s.staged_rc) { const int32_t rc = s.staged_rc; s.staged = -1; s.next_file += s.stride; s.cv.notify_all(); return rc; }
        got = s.staged_len;
    }
    const auto give_back = [&](int32_t rc, bool uploading) {
        std::lock_guard<std::mutex> lk(s.mu);
        s.h2d_pending = uploading;
        s.staged = -1;
        s.next_file += s.stride;
        s.cv.notify_all();
        return rc;
    };
    JpegAmdImage img;
    uint64_t off = 0;
    int32_t rc = jpegamd_parse_bmp(s.h_in, got, &img, &off);
    if (rc) return give_back(rc, false);
    if (!s.enc || img.width > s.enc_w || img.height > s.enc_h) {        // contexts are sized for the largest image seen
        if (s.enc) jpegamd_encoder_destroy(s.enc);
        s.enc = nullptr;
        const int32_t w = img.width > s.enc_w ? img.width : s.enc_w, h = img.height > s.enc_h ? img.height : s.enc_h;
        rc = jpegamd_encoder_create(&s.enc, w, h);
        if (rc) return give_back(rc, false);
        s.enc_w = w; s.enc_h = h;
    }
    const size_t bytes = (size_t)img.row_stride * (size_t)img.height;
    const uint64_t cap = jpegamd_max_jfif_bytes(img.width, img.height);
    // the output buffer is sized for typical content (1 byte per pixel + container); the encoder reports -8 beyond it
    const size_t out_cap = (size_t)img.width * (size_t)img.height + 4096 < cap ? (size_t)img.width * (size_t)img.height + 4096 : (size_t)cap;
    if (!s.grow_out(out_cap) || !s.grow_dev_in(bytes)) return give_back(JPEGAMD_ERR_HIP, false);
    if (hipMemcpyAsync(s.d_in, s.h_in + off, bytes, hipMemcpyHostToDevice, s.stream) != hipSuccess) return give_back(JPEGAMD_ERR_HIP, false);
    if (hipEventRecord(s.h2d_done, s.stream) != hipSuccess) { hipStreamSynchronize(s.stream); return give_back(JPEGAMD_ERR_HIP, false); }
    s.in_bytes = (uint64_t)got;
    img.pixels = s.d_in;
    img.quality = quality;
    s.img_inflight = img;
    rc = jpegamd_encode_async(s.enc, &img, s.d_out, s.out_cap, s.d_size, 1, (void *)s.stream);
    if (rc) return give_back(rc, true);
    if (hipMemcpyAsync(s.h_size, s.d_size, sizeof(uint64_t), hipMemcpyDeviceToHost, s.stream) != hipSuccess) return give_back(JPEGAMD_ERR_HIP, true);
    if (hipEventRecord(s.size_ready, s.stream) != hipSuccess) return give_back(JPEGAMD_ERR_HIP, true);
    return give_back(JPEGAMD_OK, true);
}

// wait for the slot's stream, fetch exactly the produced bytes, write the file
int32_t drain(Slot &s, const char *path, uint64_t *out_bytes, double *t_write) {
    if (hipEventSynchronize(s.size_ready) != hipSuccess) return JPEGAMD_ERR_HIP;
    int32_t rc = jpegamd_encoder_finish(s.enc, nullptr);      // status of the call (-8 when the stream outgrew the buffer)
    uint64_t n = *s.h_size;
    if (rc == JPEGAMD_ERR_HUFF_CAPACITY || (rc == JPEGAMD_OK && n > s.out_cap)) {
        // The buffer is sized for typical content (1 byte per pixel); noise-like content at high quality needs more.
        // Second attempt with the worst-case size (the pixels are still in d_in: nothing touches it before this drain).
        const uint64_t worst = jpegamd_max_jfif_bytes(s.img_inflight.width, s.img_inflight.height);
        if (!s.grow_out((size_t)worst)) return JPEGAMD_ERR_HIP;
        rc = jpegamd_encode_async(s.enc, &s.img_inflight, s.d_out, s.out_cap, s.d_size, 1, (void *)s.stream);
        if (rc == JPEGAMD_OK && hipMemcpyAsync(s.h_size, s.d_size, sizeof(uint64_t), hipMemcpyDeviceToHost, s.stream) != hipSuccess) rc = JPEGAMD_ERR_HIP;
        if (rc == JPEGAMD_OK) rc = jpegamd_encoder_finish(s.enc, nullptr);
        n = *s.h_size;
    }
    if (rc) return rc;
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
    // The slots live as long as the process (one call at a time: the header's threading contract): a second call finds its pinned
    // and device buffers, streams and encoder contexts in place.
    static std::mutex pool_mu;
    static std::vector<Slot> *pool = nullptr;
    std::lock_guard<std::mutex> pool_lk(pool_mu);
    if (!pool) pool = new std::vector<Slot>((size_t)kSlots);
    const size_t nslots = (size_t)(count < kSlots ? (count > 0 ? count : 1) : kSlots);
    std::vector<Slot> &slots = *pool;
    int32_t fatal = JPEGAMD_OK;
    for (size_t k = 0; k < nslots; ++k) {
        Slot &s = slots[k];
        s.next_file = (int)k; s.stride = (int)nslots; s.staged = -1; s.file = -1; s.h2d_pending = false;
        if (s.stream) continue;                                     // set up by an earlier call
        if (hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&s.size_ready, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&s.h2d_done, hipEventDisableTiming) != hipSuccess ||
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
        std::atomic<long long> read_ns{0};
        int dev = 0;
        hipGetDevice(&dev);
        std::vector<std::thread> readers;
        const int nreaders = count < kReaders ? count : kReaders;
        for (int t = 0; t < nreaders; ++t)
            readers.emplace_back([&, t] {
                hipSetDevice(dev);
                for (int i = t; i < count; i += nreaders) stage_file(slots[(size_t)i % nslots], i, in_paths[i], &read_ns);
            });
        for (int i = 0; i < count; ++i) {
            Slot &s = slots[(size_t)i % nslots];
            finish_slot(s);                                        // the slot's previous file (i - slots) must be out first
            const int32_t rc = submit(s, i, quality);
            if (rc) { if (status) status[i] = rc; ++failed; continue; }
            s.file = i;
        }
        for (size_t k = 0; k < nslots; ++k) finish_slot(slots[((size_t)count + k) % nslots]);   // oldest first
        for (std::thread &t : readers) t.join();
        bs.seconds_read = (double)read_ns.load() * 1e-9;
    }
    if (fatal) for (Slot &s : slots) s.release();                  // (a half-built pool is not kept)
    bs.seconds_total = now_s() - t_begin;
    bs.files_failed = failed;
    if (stats) *stats = bs;
    if (fatal) return fatal;
    return failed ? JPEGAMD_ERR_BMP : JPEGAMD_OK;
}
