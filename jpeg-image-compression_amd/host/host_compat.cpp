// host_compat.cpp -- the natural_c library surface on top of the device encoder:
// loadBMPImage / freeBMPImage / saveJPEGGrayscale with the reference's signatures and
// error behaviour (natural_c/include/bmp_handler.h:43-45, jpeg_handler.h:107;
// src/io/bmp_handler.c:5-129, src/io/jpeg_handler.c:119-282), plus the in-memory helpers
// the CLI, tests and bench use.  No pixel arithmetic happens on the host.
#include <hip/hip_runtime_api.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "jpeg_compression.h"
#include "jpegamd_internal.h"

namespace {

uint32_t rd32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
uint16_t rd16(const uint8_t *p) { return (uint16_t)(p[0] | (p[1] << 8)); }

// Header checks in the order the reference performs them (bmp_handler.c:22-52); `why`
// receives the reference's message.
int parse_headers(const uint8_t *f, uint64_t n, int32_t *w, int32_t *h, int *top_down, uint32_t *off,
                  const char **why) {
    if (n < 14) { *why = "Error: Failed to read BMP file header.\n"; return JPEGAMD_ERR_BMP; }
    if (rd16(f) != 0x4D42) { *why = "Error: File is not a valid BMP file.\n"; return JPEGAMD_ERR_BMP; }
    if (n < 54) { *why = "Error: Failed to read BMP info header.\n"; return JPEGAMD_ERR_BMP; }
    if (rd16(f + 28) != 24) { *why = "Error: Only 24-bit BMP images are supported.\n"; return JPEGAMD_ERR_BMP; }
    if (rd32(f + 30) != 0) { *why = "Error: Compressed BMP images are not supported.\n"; return JPEGAMD_ERR_BMP; }
    *w = (int32_t)rd32(f + 18);
    *h = (int32_t)rd32(f + 22);
    *top_down = 0;
    if (*h < 0) { *h = -*h; *top_down = 1; }       // bmp_handler.c:68-72
    *off = rd32(f + 10);                           // bfOffBits, bmp_handler.c:88
    if (*w <= 0 || *h <= 0 || *w > 65535 || *h > 65535) { *why = "Error: unsupported BMP dimensions.\n"; return JPEGAMD_ERR_BMP; }
    return JPEGAMD_OK;
}

struct DeviceBuf {
    void *p = nullptr;
    ~DeviceBuf() { if (p) hipFree(p); }
    bool alloc(size_t n) { return hipMalloc(&p, n ? n : 1) == hipSuccess; }
};

// Encode device-resident pixels into a host vector (JFIF file bytes).
int64_t encode_to_host(const JpegAmdImage &img, std::vector<uint8_t> &out, JpegAmdStats *st) {
    uint64_t *size_dev = nullptr;
    JpegAmdEncoder *enc = jpegamd::shared_context(img.width, img.height, &size_dev);
    if (!enc) return JPEGAMD_ERR_NO_DEVICE;
    // Typical output is << 1 byte/pixel; retry with the hard upper bound if it does not fit.
    uint64_t cap = 4096 + (uint64_t)img.width * img.height;
    for (int attempt = 0; attempt < 2; ++attempt) {
        DeviceBuf d;
        if (!d.alloc(cap)) return JPEGAMD_ERR_HIP;
        int32_t rc = jpegamd_encode_async(enc, &img, d.p, cap, size_dev, 1, nullptr);
        if (rc) return rc;
        JpegAmdStats local;
        rc = jpegamd_encoder_finish(enc, &local);
        if (rc == JPEGAMD_ERR_HUFF_CAPACITY && attempt == 0) { cap = jpegamd_max_jfif_bytes(img.width, img.height); continue; }
        if (rc) return rc;
        out.resize(local.jfif_bytes);
        if (hipMemcpy(out.data(), d.p, local.jfif_bytes, hipMemcpyDeviceToHost) != hipSuccess) return JPEGAMD_ERR_HIP;
        if (st) *st = local;
        return (int64_t)local.jfif_bytes;
    }
    return JPEGAMD_ERR_HUFF_CAPACITY;
}

}  // namespace

extern "C" int32_t jpegamd_parse_bmp(const uint8_t *bmp, uint64_t bmp_len, JpegAmdImage *view, uint64_t *pixel_offset) {
    if (!bmp || !view) return JPEGAMD_ERR_ARG;
    int32_t w, h; int td; uint32_t off; const char *why = "";
    int rc = parse_headers(bmp, bmp_len, &w, &h, &td, &off, &why);
    if (rc) return rc;
    const uint64_t stride = ((uint64_t)w * 3 + 3) & ~(uint64_t)3;      // bmp_handler.c:75
    if ((uint64_t)off + stride * (uint64_t)h > bmp_len) return JPEGAMD_ERR_BMP;   // :104 insufficient data
    view->pixels = (const void *)(uintptr_t)off;
    view->width = w; view->height = h; view->row_stride = (int32_t)stride;
    view->bottom_up = td ? 0 : 1;
    view->channel_order = JPEGAMD_ORDER_BGR;
    view->quality = 0;
    if (pixel_offset) *pixel_offset = off;
    return JPEGAMD_OK;
}

extern "C" int64_t jpegamd_encode_bmp_memory(const uint8_t *bmp, uint64_t bmp_len, int32_t quality, uint8_t *out,
                                             uint64_t out_cap) {
    JpegAmdImage img;
    uint64_t off = 0;
    int32_t rc = jpegamd_parse_bmp(bmp, bmp_len, &img, &off);
    if (rc) return rc;
    const size_t bytes = (size_t)img.row_stride * (size_t)img.height;
    DeviceBuf d;
    if (!jpegamd::shared_context(img.width, img.height, nullptr)) return JPEGAMD_ERR_NO_DEVICE;
    if (!d.alloc(bytes)) return JPEGAMD_ERR_HIP;
    if (hipMemcpy(d.p, bmp + off, bytes, hipMemcpyHostToDevice) != hipSuccess) return JPEGAMD_ERR_HIP;
    img.pixels = d.p;
    img.quality = quality;
    std::vector<uint8_t> jf;
    const int64_t n = encode_to_host(img, jf, nullptr);
    if (n < 0) return n;
    if ((uint64_t)n > out_cap || !out) return JPEGAMD_ERR_HUFF_CAPACITY;
    std::memcpy(out, jf.data(), (size_t)n);
    return n;
}

// ---------------------------------------------------------------------------------------
// natural_c surface
// ---------------------------------------------------------------------------------------
extern "C" void freeBMPImage(BMPImage *image) {
    if (image) { std::free(image->data); std::free(image); }
}

extern "C" BMPImage *loadBMPImage(const char *filename) {
    FILE *fp = filename ? std::fopen(filename, "rb") : nullptr;
    if (!fp) { std::fprintf(stderr, "Error: Unable to open file: %s\n", filename ? filename : "(null)"); return nullptr; }
    std::fseek(fp, 0, SEEK_END);
    const long len = std::ftell(fp);
    std::fseek(fp, 0, SEEK_SET);
    std::vector<uint8_t> file((size_t)(len > 0 ? len : 0));
    const size_t got = file.empty() ? 0 : std::fread(file.data(), 1, file.size(), fp);
    std::fclose(fp);

    int32_t w, h; int td; uint32_t off; const char *why = "";
    if (parse_headers(file.data(), got, &w, &h, &td, &off, &why)) { std::fputs(why, stderr); return nullptr; }
    const size_t stride = ((size_t)w * 3 + 3) & ~(size_t)3;
    BMPImage *img = (BMPImage *)std::malloc(sizeof(BMPImage));
    if (!img) { std::fprintf(stderr, "Error: Memory allocation failed for BMPImage struct.\n"); return nullptr; }
    img->width = w; img->height = h;
    img->data = (uint8_t *)std::malloc((size_t)w * h * 3);
    if (!img->data) { std::fprintf(stderr, "Error: Memory allocation failed for pixel data.\n"); std::free(img); return nullptr; }
    for (int i = 0; i < h; ++i) {                         // bmp_handler.c:103-123
        if ((uint64_t)off + (uint64_t)(i + 1) * stride > got) {
            std::fprintf(stderr, "Error: Insufficient data reading row %d\n", i);
            freeBMPImage(img);
            return nullptr;
        }
        const uint8_t *src = file.data() + off + (size_t)i * stride;
        uint8_t *dst = img->data + (size_t)(td ? i : (h - 1 - i)) * w * 3;
        for (int j = 0; j < w; ++j) { dst[3 * j + 0] = src[3 * j + 2]; dst[3 * j + 1] = src[3 * j + 1]; dst[3 * j + 2] = src[3 * j + 0]; }
    }
    return img;
}

extern "C" bool saveJPEGGrayscale(const char *filename, const BMPImage *img) {
    FILE *fp = std::fopen(filename, "wb");                 // jpeg_handler.c:121: the file is opened FIRST
    if (!fp) { std::perror("Error opening output file"); return false; }
    std::printf("Starting JPEG compression pipeline...\n");
    if (!img || !img->data || img->width <= 0 || img->height <= 0) {
        std::printf("Error: Failed to convert BMP to Grayscale (YImage).\n");   // jpeg_handler.c:134-139
        std::fclose(fp);
        return false;
    }
    if (!jpegamd::shared_context(img->width, img->height, nullptr)) {
        std::printf("Error: no HIP device; this build has no CPU pipeline.\n");
        std::fclose(fp);
        return false;
    }
    // Upload the loaded RGB top-down image with a 4-byte-aligned pitch.
    const size_t tight = (size_t)img->width * 3, pitch = (tight + 3) & ~(size_t)3;
    DeviceBuf d;
    if (!d.alloc(pitch * (size_t)img->height) ||
        hipMemcpy2D(d.p, pitch, img->data, tight, tight, (size_t)img->height, hipMemcpyHostToDevice) != hipSuccess) {
        std::printf("Error: Failed to convert BMP to Grayscale (YImage).\n");
        std::fclose(fp);
        return false;
    }
    JpegAmdImage di;
    di.pixels = d.p; di.width = img->width; di.height = img->height; di.row_stride = (int32_t)pitch;
    di.bottom_up = 0; di.channel_order = JPEGAMD_ORDER_RGB; di.quality = 0;

    // First quantised block, as the reference prints it (jpeg_handler.c:168-175).
    {
        int8_t y[64]; float dct[64]; int16_t zz[64];
        JpegAmdEncoder *enc = jpegamd::shared_context(img->width, img->height, nullptr);
        if (enc && jpegamd::first_block_taps(enc, &di, y, dct, zz) == JPEGAMD_OK) {
            int16_t q[64];
            for (int i = 0; i < 64; ++i) q[jpegamd::kZigzagHost[i]] = zz[i];
            std::printf("Natural C quant (First Block):\n");
            for (int r = 0; r < 8; ++r) { for (int c = 0; c < 8; ++c) std::printf("%d ", q[r * 8 + c]); std::printf("\n"); }
        }
    }

    std::vector<uint8_t> jf;
    const int64_t n = encode_to_host(di, jf, nullptr);
    if (n < 0) {
        std::printf("Error: Failed to perform Huffman encoding.\n");               // jpeg_handler.c:199-210
        std::fclose(fp);
        return false;
    }
    std::printf("Pipeline finished. Writing to file...\n");
    bool ok = std::fwrite(jf.data(), 1, JPEGAMD_JFIF_PREFIX_BYTES, fp) == JPEGAMD_JFIF_PREFIX_BYTES;
    if (!ok) {
        std::printf("Error: Failed to write JPEG headers to file.\n");              // jpeg_handler.c:235-248
        std::fclose(fp);
        return false;
    }
    const size_t seg = (size_t)n - JPEGAMD_JFIF_PREFIX_BYTES - 2;
    const size_t written = std::fwrite(jf.data() + JPEGAMD_JFIF_PREFIX_BYTES, 1, seg, fp);
    if (written != seg) { std::printf("Error: Failed to write bitstream data. Wrote %zu of %zu bytes.\n", written, seg); ok = false; }
    else std::printf("Bitstream written: %zu bytes.\n", written);
    std::fwrite(jf.data() + n - 2, 1, 2, fp);                                      // EOI, jpeg_handler.c:262
    std::fclose(fp);
    if (ok) std::printf("Compression successful. File saved: %s\n", filename);
    return ok;
}
