/* rtm_main -- drop-in for the stored-wavefield RTM of the reference's CPU-serial sibling
 * (dpct_gpu_rtm_domain_division/src/rtm_main.cpp = M):
 *     ./rtm_main par=input.dat          (the SU getpar form its run.sh uses; a bare file name works too)
 * Same deck keys and defaults (M:66-89), same inputs (vpfile [nx][nz], datfile = the gather [ns][nx][nt] mod_main wrote), same
 * outputs in the working directory: dir.img (the per-shot images, appended) and dir.image (their stack), both [nx][nz] float32.
 * Per shot the loops M:158-240 are one device-resident fdw_rtm_stored_shot() call.
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
    (void)tmpdir;
    if (!vpfile || !datfile || nz <= 0 || nx <= 0 || nt <= 0 || dz == -1.0f || dx == -1.0f || dt == -1.0f || fpeak == -1.0f) {
        fprintf(stderr, "input deck is missing one of vpfile/datfile/nz/nx/nt/dz/dx/dt/fpeak\n");   /* MUSTGETPAR*, M:66-77 */
        return EXIT_FAILURE;
    }
    printf("## vp = %s \n", vpfile);
    printf("## nz = %d, nx = %d, nt = %d \n", nz, nx, nt);
    printf("## dz = %f, dx = %f, dt = %f \n", dz, dx, dt);
    printf("## ns = %d, sz = %d, fsx = %d, ds = %d, gz = %d \n", ns, sz, fsx, ds, gz);
    printf("## order = %d, nzb = %d, nxb = %d, F = %f \n", order, nzb, nxb, fac);

    float *srce = (float *)malloc((size_t)nt * sizeof(float));
    fdw_mod_ricker_wavelet(nt, dt, fpeak, srce);            /* M:98-99 */
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
            vel2[(size_t)(ix + nxb) * nze + iz + nzb] = v * v; /* M:125-129 */
        }
    fdw_mod_extendvel(nx, nz, nxb, nzb, vel2);              /* M:131 */

    fdw_params prm;
    memset(&prm, 0, sizeof prm);
    prm.order = order; prm.nxe = nxe; prm.nze = nze; prm.nxb = nxb; prm.nzb = nzb; prm.nt = nt;
    prm.dx = dx; prm.dz = dz; prm.dt = dt; prm.fac = fac;
    prm.dialect = FDW_DIALECT_RTM_STORED;
    {   /* our extension, absent = the reference's arithmetic: numerics=1 (or FDW_NUMERICS=1) selects FAST numerics (include/fdwave.h) */
        int numerics = fdw_deck_int(deck, "numerics");
        if (getenv("FDW_NUMERICS")) numerics = atoi(getenv("FDW_NUMERICS"));
        prm.numerics = numerics == 1 ? FDW_NUMERICS_FAST : FDW_NUMERICS_EXACT;
    }
    fdw_ctx *ctx = NULL;
    if (fdw_create(&prm, 0, &ctx) != FDW_OK) {              /* fd_init + taper_init, M:134-135 */
        fprintf(stderr, "fdw_create: %s\n", fdw_last_error());
        return EXIT_FAILURE;
    }
    const size_t nd = (size_t)ns * nx * nt;
    float *dobs = (float *)calloc(nd ? nd : 1, sizeof(float));
    FILE *fdobs = fopen(datfile, "rb");                     /* M:149-153 */
    if (!fdobs || !dobs) {
        fprintf(stderr, "cannot open datfile '%s'\n", datfile);
        return EXIT_FAILURE;
    }
    if (fread(dobs, sizeof(float), nd, fdobs) != nd) fprintf(stderr, "warning: datfile '%s' is short (rest stays zero)\n", datfile);
    fclose(fdobs);
    FILE *flim = fopen("dir.img", "w+"), *fimg = fopen("dir.image", "w+");   /* M:145-146: in the working directory */
    if (!flim || !fimg) {
        fprintf(stderr, "cannot create dir.img / dir.image\n");
        return EXIT_FAILURE;
    }
    float *imloc = (float *)calloc(ni, sizeof(float)), *img = (float *)calloc(ni, sizeof(float));
    for (int is = 0; is < ns; is++) {
        const int sx = fsx + is * ds + nxb;                 /* M:102-104 */
        printf("** source %d, at (%d,%d) \n", is + 1, sx - nxb, sz - nzb);
        printf("** backward propagation %d, at (%d,%d) \n", is + 1, sx - nxb, sz - nzb);
        if (fdw_rtm_stored_shot(ctx, vel2, sx, sz, gz, srce, nt, dobs, nd, is, imloc) != FDW_OK) {
            fprintf(stderr, "fdw_rtm_stored_shot: %s\n", fdw_last_error());
            return EXIT_FAILURE;
        }
        fwrite(imloc, sizeof(float), ni, flim);             /* M:233 */
        for (int iz = 0; iz < nz; iz++)                     /* M:236-240 */
            for (int ix = 0; ix < nx; ix++) img[(size_t)ix * nz + iz] += imloc[(size_t)ix * nz + iz];
    }
    fwrite(img, sizeof(float), ni, fimg);                   /* M:248 */
    fclose(flim);
    fclose(fimg);
    free(dobs); free(imloc); free(img);
    gettimeofday(&end, NULL);
    printf("Execution Time: %.2f seconds\n", ((end.tv_sec - start.tv_sec) * 1000000.0 + (end.tv_usec - start.tv_usec)) / 1000000.0);
    fdw_destroy(ctx);
    free(srce); free(vp); free(vel2);
    fdw_deck_free(deck);
    return 0;
}
