/* lapfilt -- drop-in for the reference's image filter models/3lay_mod/laplace.f90 (a Fortran program with its sizes and file names
 * built in: nz = nx = 151, dz = dx = 10, dir.image -> dir.imalap).  Without arguments it does exactly that; otherwise
 *     ./lapfilt nz nx dz dx [infile [outfile]]
 * The image is [nx][nz] float32 (the layout rtm_code / rtm_main write). */
#include <stdio.h>
#include <stdlib.h>

#include "fdwave.h"

int main(int argc, char **argv)
{
    int nz = 151, nx = 151;
    float dz = 10.f, dx = 10.f;
    const char *in = "dir.image", *out = "dir.imalap";
    if (argc >= 5) {
        nz = atoi(argv[1]); nx = atoi(argv[2]); dz = (float)atof(argv[3]); dx = (float)atof(argv[4]);
        if (argc >= 6) in = argv[5];
        if (argc >= 7) out = argv[6];
    } else if (argc != 1) {
        fprintf(stderr, "usage: %s [nz nx dz dx [infile [outfile]]]\n", argv[0]);
        return EXIT_FAILURE;
    }
    if (nz < 1 || nx < 1) return EXIT_FAILURE;
    const size_t n = (size_t)nx * nz;
    float *a = (float *)calloc(n, sizeof(float)), *b = (float *)calloc(n, sizeof(float));
    FILE *f = fopen(in, "rb");
    if (!f || !a || !b || fread(a, sizeof(float), n, f) != n) {
        fprintf(stderr, "cannot read %zu floats from '%s'\n", n, in);
        return EXIT_FAILURE;
    }
    fclose(f);
    if (fdw_image_laplacian(0, a, nx, nz, dx, dz, b) != FDW_OK) {
        fprintf(stderr, "fdw_image_laplacian: %s\n", fdw_last_error());
        return EXIT_FAILURE;
    }
    f = fopen(out, "wb");
    if (!f || fwrite(b, sizeof(float), n, f) != n) {
        fprintf(stderr, "cannot write '%s'\n", out);
        return EXIT_FAILURE;
    }
    fclose(f);
    free(a); free(b);
    return 0;
}
