/* mod_main -- drop-in for the forward-modelling producer of the reference's CPU-serial sibling
 * (dpct_gpu_rtm_domain_division/src/mod_main.cpp = M):
 *     ./mod_main par=input.dat          (the SU getpar form its run scripts use; a bare file name works too)
 * Same deck keys and defaults (M:63-86), same input (vpfile [nx][nz]), same output: `datfile` = data[ns][nx][nt] float32, the
 * gather an RTM run reads back as its datfile.  The whole loop M:140-174 of a batch of shots is one device-resident fdw_model_shot_batch() call.
 * Not reproduced: the "* it = ..." progress lines every 100 steps. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/time.h>

#include "fdw_config.h"
#include "fdwave.h"

int main(int argc, char **argv)
{
    struct timeval start, end;
    gettimeofday(&start, NULL);
    const char *par = NULL;
    for (int i = 1; i < argc; i++) {
        if (!strncmp(argv[i], "par=", 4)) par = argv[i] + 4;
        else if (!strchr(argv[i], '=')) par = argv[i];
    }
    if (!par) {
        fprintf(stderr, "usage: %s par=<input.dat>\n", argv[0]);
        return EXIT_FAILURE;
    }
    fdw_deck *deck = fdw_deck_read(par);
    if (!deck) return EXIT_FAILURE;
    const char *tmpdir = fdw_deck_str(deck, "tmpdir"), *vpfile = fdw_deck_str(deck, "vpfile"), *datfile = fdw_deck_str(deck, "datfile");
    const int nz = fdw_deck_int(deck, "nz"), nx = fdw_deck_int(deck, "nx"), nt = fdw_deck_int(deck, "nt");
    const float dz = fdw_deck_float(deck, "dz"), dx = fdw_deck_float(deck, "dx"), dt = fdw_deck_float(deck, "dt");
    const float fpeak = fdw_deck_float(deck, "fpeak");
    int ns = fdw_deck_int(deck, "ns"), sz = fdw_deck_int(deck, "sz"), fsx = fdw_deck_int(deck, "fsx"), ds = fdw_deck_int(deck, "ds");
    int gz = fdw_deck_int(deck, "gz"), order = fdw_deck_int(deck, "order"), nzb = fdw_deck_int(deck, "nzb"), nxb = fdw_deck_int(deck, "nxb");
    float fac = fdw_deck_float(deck, "fac");
    if (ns == -1) ns = 1;       /* M:76-86 */
    if (sz == -1) sz = 0;
    if (fsx == -1) fsx = 0;
    if (ds == -1) ds = 1;
    if (gz == -1) gz = 0;
    if (order == -1) order = 8;
    if (nzb == -1) nzb = 40;
    if (nxb == -1) nxb = 40;
    if (fac == -1.0f) fac = 0.7f;
    if (!tmpdir || !vpfile || !datfile || nz <= 0 || nx <= 0 || nt <= 0 || dz == -1.0f || dx == -1.0f || dt == -1.0f || fpeak == -1.0f) {
        fprintf(stderr, "input deck is missing one of tmpdir/vpfile/datfile/nz/nx/nt/dz/dx/dt/fpeak\n");   /* MUSTGETPAR*, M:64-74 */
        return EXIT_FAILURE;
    }
    printf("## vp = %s \n", vpfile);
    printf("## nz = %d, nx = %d, nt = %d \n", nz, nx, nt);
    printf("## dz = %f, dx = %f, dt = %f \n", dz, dx, dt);
    printf("## ns = %d, sz = %d, fsx = %d, ds = %d, gz = %d \n", ns, sz, fsx, ds, gz);
    printf("## order = %d, nzb = %d, nxb = %d, F = %f \n", order, nzb, nxb, fac);

    float *srce = (float *)malloc((size_t)nt * sizeof(float));
    fdw_mod_ricker_wavelet(nt, dt, fpeak, srce);            /* M:95-96 */
    sz += nzb;
    gz += nzb;
    const int nze = nz + 2 * nzb, nxe = nx + 2 * nxb;
    const size_t ne = (size_t)nxe * nze, ni = (size_t)nx * nz;
    float *vp = (float *)calloc(ni, sizeof(float));
    FILE *fvp = fopen(vpfile, "rb");
    if (!fvp || !vp) {
        fprintf(stderr, "cannot open vpfile '%s'\n", vpfile);
        return EXIT_FAILURE;
    }
    if (fread(vp, sizeof(float), ni, fvp) != ni) fprintf(stderr, "warning: vpfile '%s' is short (rest stays zero)\n", vpfile);
    fclose(fvp);
    float *vel2 = (float *)calloc(ne, sizeof(float));       /* the reference leaves the border uninitialised until extendvel */
    for (int ix = 0; ix < nx; ix++)
        for (int iz = 0; iz < nz; iz++) {
            const float v = vp[(size_t)ix * nz + iz];
            vel2[(size_t)(ix + nxb) * nze + iz + nzb] = v * v; /* M:121-125 */
        }
    fdw_mod_extendvel(nx, nz, nxb, nzb, vel2);              /* M:127 */

    fdw_params prm;
    memset(&prm, 0, sizeof prm);
    prm.order = order; prm.nxe = nxe; prm.nze = nze; prm.nxb = nxb; prm.nzb = nzb; prm.nt = nt;
    prm.dx = dx; prm.dz = dz; prm.dt = dt; prm.fac = fac;
    prm.dialect = FDW_DIALECT_MOD;
    {   /* our extension, absent = the reference's arithmetic: numerics=1 (or FDW_NUMERICS=1) selects FAST numerics (include/fdwave.h) */
        int numerics = fdw_deck_int(deck, "numerics");
        if (getenv("FDW_NUMERICS")) numerics = atoi(getenv("FDW_NUMERICS"));
        prm.numerics = numerics == 1 ? FDW_NUMERICS_FAST : FDW_NUMERICS_EXACT;
    }
    fdw_ctx *ctx = NULL;
    if (fdw_create(&prm, 0, &ctx) != FDW_OK) {              /* fd_init + taper_init, M:130-131 */
        fprintf(stderr, "fdw_create: %s\n", fdw_last_error());
        return EXIT_FAILURE;
    }
    FILE *fdat = fopen(datfile, "w+");                      /* M:137 */
    if (!fdat) {
        fprintf(stderr, "cannot create datfile '%s'\n", datfile);
        return EXIT_FAILURE;
    }
    float *data = (float *)calloc((size_t)ns * nx * nt, sizeof(float));
    /* small decks: a batch of shots per launch (fdw_shot_batch_max says how many fill the chip; 1 = the grid is big enough alone) */
    int bmax = getenv("FDW_NO_SHOT_BATCH") ? 1 : fdw_shot_batch_max(ctx);
    if (bmax < 1) bmax = 1;
    for (int is0 = 0; is0 < ns; is0 += bmax) {
        const int nb = is0 + bmax <= ns ? bmax : ns - is0;
        for (int is = is0; is < is0 + nb; is++) printf("** source %d, at (%d,%d) \n", is + 1, fsx + is * ds, sz - nzb); /* M:99-101 */
        if (fdw_model_shot_batch(ctx, nb, vel2, fsx + is0 * ds + nxb, ds, sz, gz, srce, nt, data + (size_t)is0 * nx * nt) != FDW_OK) {
            fprintf(stderr, "fdw_model_shot_batch: %s\n", fdw_last_error());
            return EXIT_FAILURE;
        }
    }
    fwrite(data, sizeof(float), (size_t)ns * nt * nx, fdat); /* M:171 */
    fclose(fdat);
    gettimeofday(&end, NULL);
    printf("> Exec time = %.2f (s)\n", ((end.tv_sec - start.tv_sec) * 1000000.0 + (end.tv_usec - start.tv_usec)) / 1000000.0);
    fdw_destroy(ctx);
    free(srce); free(vp); free(vel2); free(data);
    fdw_deck_free(deck);
    return 0;
}
