/* fdw_compat.c -- fd_init / fd_forward / fd_back with the reference's signatures (include/fdwave_compat.h)
 * on top of the C ABI.  Built into libfdwave_rtm_compat.so. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "fdwave.h"
#include "fdwave_compat.h"

static fdw_ctx *g_ctx = NULL;
static int g_nxb, g_nzb;

static void die(const char *what)
{
    fprintf(stderr, "%s: %s\n", what, fdw_last_error());
    exit(EXIT_FAILURE);
}

void fd_init(int order, int nxe, int nze, int nxb, int nzb, int nt, int ns, float fac, float dx, float dz, float dt)
{
    (void)ns;
    fdw_params prm;
    memset(&prm, 0, sizeof prm);
    prm.order = order; prm.nxe = nxe; prm.nze = nze; prm.nxb = nxb; prm.nzb = nzb; prm.nt = nt;
    prm.dx = dx; prm.dz = dz; prm.dt = dt; prm.fac = fac;
    prm.compat = 1;
    if (g_ctx) fdw_destroy(g_ctx);
    g_ctx = NULL;
    g_nxb = nxb; g_nzb = nzb;
    if (fdw_create(&prm, 0, &g_ctx) != FDW_OK) die("fd_init");
}

void fd_forward(int order, float **p, float **pp, float **v2, int nze, int nxe, int nt, int is, int sz, int *sx,
                float *srce, int propag)
{
    (void)order; (void)nze; (void)nxe; (void)propag;
    if (!g_ctx) { fprintf(stderr, "fd_forward: fd_init has not been called\n"); exit(EXIT_FAILURE); }
    if (fdw_forward(g_ctx, p[0], pp[0], v2[0], sx[is], sz, srce, nt) != FDW_OK) die("fd_forward");
}

void fd_back(int order, float **p, float **pp, float **pr, float **ppr, float **v2, int nze, int nxe, int nt, int is,
             int sz, int gz, float ***snaps, float **imloc, float **d_obs)
{
    (void)order; (void)p; (void)pp; (void)pr; (void)ppr; (void)nze; (void)nxe; (void)sz;
    if (!g_ctx) { fprintf(stderr, "fd_back: fd_init has not been called\n"); exit(EXIT_FAILURE); }
    /* snaps[0] = P, snaps[1] = PP (fd-code.cu:502-507); d_obs[is] is the [nx][nt] gather (fd-code.cu:426-435) */
    if (fdw_back(g_ctx, v2[0], snaps[0][0], snaps[1][0], d_obs[is], gz, imloc[0], nt) != FDW_OK) die("fd_back");
}

void fd_free(void)
{
    if (g_ctx) fdw_destroy(g_ctx);
    g_ctx = NULL;
}
