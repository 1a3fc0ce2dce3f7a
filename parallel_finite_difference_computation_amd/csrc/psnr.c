/* psnr -- drop-in for the reference's image comparer cuda_reference_RTM/models/marmousi/psnr (an ELF without source; SURVEY.md section 4:
 * "Usage: ./psnr file1 file2", prints MSE / RMSE / SNR / PSNR).  Behaviour restated from the tool's own output: two raw fp32 files of equal
 * size; MSE = mean (f1-f2)^2, RMSE, SNR = 10 log10(sum f2^2 / sum (f1-f2)^2), PSNR = 20 log10(max |f2| / RMSE), each printed as
 * "%-10s %15e"; the difference f1 - f2 is written to ./dir.output; the messages and the exit status (always 0) are the tool's.  Everything
 * is computed on the GPU (fdw_image_compare), in the tool's own arithmetic, so the printed figures are the tool's digit for digit
 * (FDW_PSNR_EXACT_SUMS=1: a parallel reduction in double instead). */
#include <stdio.h>
#include <stdlib.h>

#include "fdwave.h"

int main(int argc, char **argv)
{
    if (argc <= 2) {
        fputs("\nUsage: ./psnr file1 file2 \nwhere fileN must binary files.\n", stdout);
        return 0;
    }
    FILE *f1 = fopen(argv[1], "rb"), *f2 = fopen(argv[2], "rb");
    if (!f1) {
        fputs("\nfile1 not found!\n", stdout);
        return 0;
    }
    if (!f2) {
        fputs("\nfile2 not found!\n", stdout);
        return 0;
    }
    fseek(f1, 0, SEEK_END);
    fseek(f2, 0, SEEK_END);
    const long b1 = ftell(f1), b2 = ftell(f2);
    rewind(f1);
    rewind(f2);
    if (b1 != b2) {
        fputs("\nFile sizes don't match!\n", stdout);
        return 0;
    }
    const size_t n = (size_t)b1 / sizeof(float);
    float *a = (float *)calloc(n ? n : 1, sizeof(float)), *b = (float *)calloc(n ? n : 1, sizeof(float)), *d = (float *)calloc(n ? n : 1, sizeof(float));
    if (!a || !b || !d || fread(a, sizeof(float), n, f1) != n || fread(b, sizeof(float), n, f2) != n) {
        fprintf(stderr, "psnr: cannot read %zu floats\n", n);
        return EXIT_FAILURE;
    }
    fclose(f1);
    fclose(f2);
    double st[4];
    if (fdw_image_compare(0, a, b, n, d, st, getenv("FDW_PSNR_EXACT_SUMS") != NULL) != FDW_OK) {      /* default: the tool's own serial fp32 sums */
        fprintf(stderr, "psnr: %s\n", fdw_last_error());
        return EXIT_FAILURE;
    }
    printf("%-10s %15e\n", "MSE:", st[0]);
    printf("%-10s %15e\n", "RMSE:", st[1]);
    printf("%-10s %15e\n", "SNR:", st[2]);
    printf("%-10s %15e\n", "PSNR:", st[3]);
    FILE *fo = fopen("dir.output", "wb");
    if (fo) {
        fwrite(d, sizeof(float), n, fo);
        fclose(fo);
    }
    free(a); free(b); free(d);
    return 0;
}
