/* stencil_code -- drop-in for cuda_reference_stencil_computation/stencil_code (fd-source-code.cu = S):
 *     ./stencil_code ./input.dat
 * reads the deck (S:34-108: tmpdir = path of the INPUT field, nz nx dz dx nxb nzb order), the fp32
 * field [nxe][nze], applies the Laplacian once on the MI355X (S:325) and writes the result.
 * Output path: `outfile=` key if present; otherwise the reference's hard-coded ../bin/output_cuda.bin
 * (S:337) when ../bin exists, else ./output_teste.bin like the reference's SYCL siblings.
 * Host code is plain C; all device work goes through the C ABI of libfdwave.so (include/fdwave.h). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>

#include "fdw_config.h"
#include "fdwave.h"

int main(int argc, char **argv)
{
    if (argc < 2) {
        fprintf(stderr, "usage: %s <input.dat>\n", argv[0]);
        return EXIT_FAILURE;
    }
    fdw_deck *deck = fdw_deck_read(argv[1]);
    if (!deck) return EXIT_FAILURE; /* S:40-41 */
    const char *file_path = fdw_deck_str(deck, "tmpdir");
    const int nzb = fdw_deck_int(deck, "nzb"), nxb = fdw_deck_int(deck, "nxb");
    const int nz = fdw_deck_int(deck, "nz"), nx = fdw_deck_int(deck, "nx");
    const float dz = fdw_deck_float(deck, "dz"), dx = fdw_deck_float(deck, "dx");
    const int order = fdw_deck_int(deck, "order");

    /* the reference's banner, typo included (S:281-288) */
    printf("Local do arquivo: %s\n", file_path ? file_path : "(null)");
    printf("nzb = %i\n", nzb);
    printf("nzb = %i\n", nxb);
    printf("nz = %i\n", nz);
    printf("nx = %i\n", nx);
    printf("dz = %f\n", dz);
    printf("dx = %f\n", dx);
    printf("order = %i\n", order);
    if (!file_path || nz <= 0 || nx <= 0 || nxb < 0 || nzb < 0) {
        fprintf(stderr, "input deck is missing tmpdir/nz/nx/nxb/nzb\n");
        return EXIT_FAILURE;
    }
    const int nxe = nx + 2 * nxb, nze = nz + 2 * nzb; /* S:290-291 */
    const size_t n = (size_t)nxe * nze;

    fdw_params prm;
    memset(&prm, 0, sizeof prm);
    prm.order = order;
    prm.nxe = nxe; prm.nze = nze;
    prm.nxb = 0; prm.nzb = 0;          /* the stencil program has no taper; borders only size the grid */
    prm.dx = dx; prm.dz = dz; prm.dt = 0.0f; prm.fac = 1.0f;
    prm.compat = 0;
    prm.coef_cxx = 1;                   /* S:184-216 is compiled as C++ */
    {   /* our extension, absent = the reference's arithmetic: numerics=1 (or FDW_NUMERICS=1) selects FAST numerics (include/fdwave.h) */
        int numerics = fdw_deck_int(deck, "numerics");
        if (getenv("FDW_NUMERICS")) numerics = atoi(getenv("FDW_NUMERICS"));
        prm.numerics = numerics == 1 ? FDW_NUMERICS_FAST : FDW_NUMERICS_EXACT;
    }
    fdw_ctx *ctx = NULL;
    if (fdw_create(&prm, 0, &ctx) != FDW_OK) {
        fprintf(stderr, "fdw_create: %s\n", fdw_last_error());
        return EXIT_FAILURE;
    }

    FILE *fin = fopen(file_path, "rb");
    if (!fin) {
        printf("Unable to open file!\n");
        return EXIT_FAILURE;
    }
    printf("Input successfully opened for reading.\n");
    float *in = (float *)calloc(n, sizeof(float)), *out = (float *)calloc(n, sizeof(float));
    if (!in || !out) {
        printf("Input memory allocation error!\n");
        return EXIT_FAILURE;
    }
    printf("Input memory allocation was successful.\n");
    if (fread(in, sizeof(float), n, fin) != n)
        printf("Input reading error!\n");
    else
        printf("Input reading was successful.\n");
    fclose(fin);

    if (fdw_laplacian(ctx, in, out) != FDW_OK) {
        fprintf(stderr, "fdw_laplacian: %s\n", fdw_last_error());
        return EXIT_FAILURE;
    }
    printf("Output memory allocation was successful.\n");

    const char *outfile = fdw_deck_str(deck, "outfile");
    struct stat st;
    if (!outfile) outfile = (stat("../bin", &st) == 0 && S_ISDIR(st.st_mode)) ? "../bin/output_cuda.bin" : "output_teste.bin";
    FILE *fout = fopen(outfile, "wb");
    if (!fout) {
        printf("Unable to open file!\n");
        return EXIT_FAILURE;
    }
    printf("Output successfully opened for writing.\n");
    if (fwrite(out, sizeof(float), n, fout) != n)
        printf("Output writing error!\n");
    else
        printf("Output writing was successful.\n");
    fclose(fout);

    fdw_destroy(ctx);
    free(in);
    free(out);
    fdw_deck_free(deck);
    return 0;
}
