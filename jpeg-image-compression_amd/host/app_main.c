/*
 * app_main.c -- `jpeg_compression_app <input.bmp> <output.jpeg>`.
 *
 * Same argv contract, stdout/stderr lines and exit codes as the reference's CLI
 * (natural_c/src/main.c:4-35, natural_c/Makefile:14): usage error -> 1, load failure -> 1,
 * otherwise 0 (even when the save fails, main.c:24-28,34).  The codec behind
 * saveJPEGGrayscale runs on the MI355X.
 */
#include <stdio.h>

#include "jpeg_compression.h"

int main(int argc, char *argv[]) {
    if (argc != 3) {
        fprintf(stderr, "Usage: %s <input_file_path> <output_file_path>\n", argv[0]);
        return 1;
    }
    printf("Starting processing...\n");
    printf("Input: %s\n", argv[1]);
    BMPImage *img = loadBMPImage(argv[1]);
    if (!img) {
        fprintf(stderr, "Error: Failed to load image from %s\n", argv[1]);
        return 1;
    }
    if (saveJPEGGrayscale(argv[2], img)) printf("Save is sucesfull");
    return 0;   /* like the reference, `img` is not freed and a failed save still exits 0 */
}
