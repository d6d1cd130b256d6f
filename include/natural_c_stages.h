/* natural_c_stages.h -- the reference's stage-by-stage interface, GPU-backed.
 *
 * natural_c exposes every stage of the pipeline as `X *f(const Y *)` with malloc'd results and a matching
 * `freeX` (natural_c/include/converter.h:13-31, dct.h:14-24, quantization.h:10-17, zigzag.h:7-15, rle.h:8-24,
 * huffman.h:9-38, jpeg_handler.h:109); saveJPEGGrayscale chains them (src/io/jpeg_handler.c:119-282) and it is
 * the reference's own way of comparing implementations stage by stage (dsp_port/jpeg_client/main.c:137-203).
 * libjpegamd.so exports the same names with the same struct layouts.  Every function here uploads its input,
 * runs HIP kernels (host/stage_compat.hip) and downloads the result: NULL on bad input, allocation failure or
 * when no HIP device is present -- there is no CPU implementation behind them.  They are the verification /
 * debugging surface; the production path is the fused pipeline behind jpegamd_encode_async.
 */
#ifndef NATURAL_C_STAGES_H
#define NATURAL_C_STAGES_H

#include <stddef.h>
#include <stdint.h>

#include "jpeg_compression.h" /* BMPImage */

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { int width; int height; uint8_t *data; } YImage;              /* converter.h:21-26, padded to 8 */
typedef struct { int width; int height; int8_t *data; } CenteredYImage;       /* converter.h:13-18 */
typedef struct { int width; int height; float *coefficients; } DCTImage;      /* dct.h:14-19, image layout */
typedef struct { int width; int height; int16_t *data; } QuantizedImage;      /* quantization.h:10-14 */
typedef struct { int numBlocksW; int numBlocksH; int totalBlocks; int16_t *data; } ZigZagData;   /* zigzag.h:7-12 */
typedef struct { uint8_t symbol; uint16_t code; uint8_t codeBits; } RLESymbol;                  /* rle.h:8-14 */
typedef struct { RLESymbol *data; size_t count; size_t capacity; } RLEData;                      /* rle.h:17-21 */
typedef struct { uint8_t *data; size_t size; size_t capacity; } JpegEncoderBuffer;               /* huffman.h:9-13 */

YImage *convertBMPToJPEGGrayscale(const BMPImage *image);                      /* core/converter.c:4-58 */
CenteredYImage *centerYImage(const YImage *source);                            /* core/converter.c:60-90 */
DCTImage *performDCT(const CenteredYImage *image);                             /* core/dct.c:98-151 */
void computeDCTBlock(const int8_t inputBlock[8][8], float outputBlock[8][8]);  /* core/dct.c:63-96 */
QuantizedImage *quantizeImage(const DCTImage *dctImg);                         /* core/quantization.c:3-43 */
ZigZagData *performZigZag(const QuantizedImage *qImg);                         /* core/zigzag.c:21-68 */
RLEData *performRLE(const ZigZagData *zigZagData);                             /* core/rle.c:51-127 */
JpegEncoderBuffer *encodeHuffman(const RLEData *rleData, int totalBlocks);     /* core/huffman.c:121-193 */

void freeYImage(YImage *img);
void freeCenteredYImage(CenteredYImage *img);
void freeDCTImage(DCTImage *img);
void freeQuantizedImage(QuantizedImage *img);
void freeZigZagData(ZigZagData *zData);
void freeRLEData(RLEData *rleData);
void freeJpegEncoderBuffer(JpegEncoderBuffer *buffer);

#ifdef __cplusplus
}
#endif
#endif
