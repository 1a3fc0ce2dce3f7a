/*
 * fdw_oracle.c -- TEST INFRASTRUCTURE ONLY.  CPU restatement of the reference's 2-D acoustic
 * finite-difference path, kept deliberately in the reference's own shape (one pass per CUDA
 * kernel, one sequential "thread" per grid point, the same launch extents) so that it can be read
 * side by side with the reference.  Nothing under parallel_finite_difference_computation_amd/
 * may link, import or execute this file: it is the checker for tests/, smoke() and the
 * cpu_baseline leg of bench.py, never the product.
 *
 * Citations: S = cuda_reference_stencil_computation/fd-source-code.cu
 *            R = cuda_reference_RTM/src/fd-code.cu
 *            F = cuda_reference_RTM/lib/src/functions.c
 *
 * Parity pins (see DESIGN.md "Oracle"):
 *   - orc_kernel_lap      bit-exact vs dpct_migrated_stencil_computation/output_teste.bin
 *   - orc_fd_forward      <=1e-5 max-norm-rel vs cuda_reference_stencil_computation/input.bin
 *                         (shot 5 of models/new_mod, P after 1700 steps, real-hardware output)
 *   - host tables         bit-exact vs oracle/_ref (F compiled unmodified by gcc)
 *   - orc_fd_back         PARITY UNPINNED: the reference ships no usable image golden
 *                         (output/dir.image is all zeros, dobs.6 is missing) and the .cu cannot
 *                         be built here (needs nvcc / cuda.h).  fd_back reuses the pinned
 *                         lap/time/taper passes; injection + imaging are restated from R:124-144.
 *
 * Build: gcc -O2 -ffp-contract=off (no FMA contraction: the reference is built with
 * --fmad=false, S Makefile:4) -- see oracle/Makefile.
 *
 * FAST numerics (orc_set_numerics(s, 1)): NOT the reference's arithmetic but the product's documented tolerance mode (include/fdwave.h
 * fdw_params.numerics, csrc/fdw_device.h): the Laplacian as ONE chain of symmetric sums and fused multiply-adds, everything else as
 * above.  Restated here (orc_lap_fast, explicit fmaf calls) so that the FAST kernels are checked bit for bit against a CPU statement of
 * their own formula -- masks, extents, taper and injection included -- and not only to a tolerance against the exact arithmetic.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define ORC_PI (3.141592653589793) /* F header: functions.h:7 */
#define ORC_BLOCK_RTM 8            /* functions.h:6 sizeblock */

/* ------------------------------------------------------------------ host tables */

/* F:160-192 (C, double libm) and S:184-216 (same text compiled as C++ by nvcc's host compiler, so
 * cos/pow on float arguments resolve to the float overloads).  cxx!=0 selects the latter. */
static void orc_makeo2(float *coef, int order, int cxx)
{
    float h_beta, alpha1, alpha2, central_term, coef_filt, arg, coef_wind;
    float alpha = .54, beta = 6.;
    int msign = -1, ix;
    h_beta = 0.5 * beta;
    alpha1 = 2. * alpha - 1.0;
    alpha2 = 2. * (1.0 - alpha);
    central_term = 0.0;
    for (ix = 1; ix <= order / 2; ix++) {
        msign = -msign;
        coef_filt = (2. * msign) / (ix * ix);
        arg = ORC_PI * ix / (2. * (order / 2 + 2));
        if (cxx) {
            float c = cosf(arg);
            coef_wind = powf(alpha1 + alpha2 * c * c, h_beta);
        } else {
            coef_wind = pow((alpha1 + alpha2 * cos(arg) * cos(arg)), h_beta);
        }
        coef[order / 2 + ix] = coef_filt * coef_wind;
        central_term = central_term + coef[order / 2 + ix];
        coef[order / 2 - ix] = coef[order / 2 + ix];
    }
    coef[order / 2] = -2. * central_term;
}

/* F:113-158 == S:137-182: tabulated weights for order 2/4/6/8, windowed series otherwise. */
void orc_calc_coefs(int order, int cxx, float *coef)
{
    int i;
    for (i = 0; i <= order; i++) coef[i] = 0.0f;
    switch (order) {
    case 2:
        coef[0] = 1.; coef[1] = -2.; coef[2] = 1.;
        break;
    case 4:
        coef[0] = -1. / 12.; coef[1] = 4. / 3.; coef[2] = -5. / 2.;
        coef[3] = 4. / 3.; coef[4] = -1. / 12.;
        break;
    case 6:
        coef[0] = 1. / 90.; coef[1] = -3. / 20.; coef[2] = 3. / 2.; coef[3] = -49. / 18.;
        coef[4] = 3. / 2.; coef[5] = -3. / 20.; coef[6] = 1. / 90.;
        break;
    case 8:
        coef[0] = -1. / 560.; coef[1] = 8. / 315.; coef[2] = -1. / 5.; coef[3] = 8. / 5.;
        coef[4] = -205. / 72.;
        coef[5] = 8. / 5.; coef[6] = -1. / 5.; coef[7] = 8. / 315.; coef[8] = -1. / 560.;
        break;
    default:
        orc_makeo2(coef, order, cxx);
    }
}

/* R:203-217 == S:244-257: d?2inv in double then float; coefficient scaling float*float. */
void orc_scaled_coefs(int order, float dx, float dz, int cxx, float *coefs_x, float *coefs_z)
{
    float dx2inv = (1. / dx) * (1. / dx);
    float dz2inv = (1. / dz) * (1. / dz);
    float *c = (float *)malloc((order + 1) * sizeof(float));
    int io;
    orc_calc_coefs(order, cxx, c);
    for (io = 0; io <= order; io++) {
        coefs_z[io] = dz2inv * c[io];
        coefs_x[io] = dx2inv * c[io];
    }
    free(c);
}

/* R:159-166.  The text lives in a .cu (C++): sqrt/log on a float pick the float overloads,
 * pow(float,int) promotes to double, exp(double). */
void orc_taper_tables(int nxb, int nzb, float fac, float *taper_x, float *taper_z)
{
    float dfrac;
    int i;
    dfrac = sqrtf(-logf(fac)) / (1. * nxb);
    for (i = 0; i < nxb; i++) taper_x[i] = exp(-pow((double)(dfrac * (nxb - i)), 2));
    dfrac = sqrtf(-logf(fac)) / (1. * nzb);
    for (i = 0; i < nzb; i++) taper_z[i] = exp(-pow((double)(dfrac * (nzb - i)), 2));
}

/* F:302-326 */
static float orc_ricker(float t, float fpeak)
{
    float x, xx;
    x = ORC_PI * fpeak * t;
    xx = x * x;
    return exp(-xx) * (1.0 - 2.0 * xx);
}

/* F:328-334 */
void orc_ricker_wavelet(int nt, float dt, float peak, float *s)
{
    int it;
    for (it = 0; it < nt; it++) s[it] = orc_ricker(it * dt - 1.0 / peak, peak);
}

/* F:336-394 random-velocity border.  vel is [nxe][nze] contiguous.  Restated with two helpers;
 * what must be preserved is (i) the float expression of each draw and (ii) the ORDER in which
 * glibc rand() is consumed: bottom strip (ix outer, iz inner), then left/right strips interleaved
 * (iz outer, ix inner, left draw before right draw), then the two bottom corners (lower-left
 * completely before lower-right), each walking the triangle ix<=iz with two draws per visit. */
static float orc_ramp(float v, int k, int nb)
{   /* linear descent from v towards the 300 m/s floor: F:348,357,362,379,389 */
    float l_lim = 300.;
    return v - (v - l_lim) * (k) / (nb - 1);
}
static float orc_draw(float v, float v_ave)
{   /* uniform integer in a window that widens as v_ave drops, delta = 200: F:349 */
    float delta = 200.;
    return rand() % (int)(v + delta - (v_ave - delta) + 1) + v_ave - delta;
}
void orc_extendvel_linear(int nx, int nz, int nxb, int nzb, float *vel)
{
    const int nze = nz + 2 * nzb, xl = nxb, xr = nxb + nx - 1, zt = nzb, zb = nzb + nz - 1;
    const int xlast = nx + 2 * nxb - 1, zlast = nz + 2 * nzb - 1;
    int ix, iz, side;
#define V(ix_, iz_) vel[(size_t)(ix_) * nze + (iz_)]
    for (ix = xl; ix <= xr; ix++)           /* top: replicate first interior sample; bottom: draws */
        for (iz = 0; iz < nzb; iz++) {
            float v = V(ix, zb);
            V(ix, iz) = V(ix, zt);
            V(ix, zb + 1 + iz) = orc_draw(v, orc_ramp(v, iz, nzb));
        }
    for (iz = zt; iz <= zb; iz++)           /* left then right, per (iz, ix) */
        for (ix = 0; ix < nxb; ix++) {
            float v = V(xl, iz);
            V(xl - 1 - ix, iz) = orc_draw(v, orc_ramp(v, ix, nxb));
            v = V(xr, iz);
            V(xr + 1 + ix, iz) = orc_draw(v, orc_ramp(v, ix, nxb));
        }
    for (iz = 0; iz < nzb; iz++)            /* top corners: replicate sideways (F:368-373) */
        for (ix = 0; ix < nxb; ix++) {
            V(ix, iz) = V(xl, iz);
            V(xr + 1 + ix, iz) = V(xr, iz);
        }
    for (side = 0; side < 2; side++)        /* bottom corners: left (F:375-383), right (F:385-393) */
        for (iz = 0; iz < nzb; iz++)
            for (ix = 0; ix <= iz; ix++) {
                float v = V(side ? xr : xl, zb);
                float va = orc_ramp(v, nxb - 1 - ix, nzb); /* note: nxb in the index, nzb in the divisor */
                int a = side ? xlast - ix : ix, b = side ? xlast - iz : iz;
                V(a, zlast - iz) = orc_draw(v, va);
                V(b, zlast - ix) = orc_draw(v, va);
            }
#undef V
}

void orc_srand(unsigned seed) { srand(seed); }

/* ------------------------------------------------------------------ launch extents */

/* R:185-195: `int div_x = (float)nxe/(float)sizeblock; gridx = (int)ceil(div_x)` -- the quotient is
 * truncated by the int assignment before ceil sees it, so the RTM grids cover 8*floor(n/8) points.
 * compat=1 reproduces that; compat=0 covers the whole array. */
void orc_extents(int nxe, int nze, int nzb, int compat, int *xlim, int *zlim, int *ztap)
{
    if (compat) {
        *xlim = ORC_BLOCK_RTM * (nxe / ORC_BLOCK_RTM);
        *zlim = ORC_BLOCK_RTM * (nze / ORC_BLOCK_RTM);
        *ztap = ORC_BLOCK_RTM * (nzb / ORC_BLOCK_RTM);
    } else {
        *xlim = nxe; *zlim = nze; *ztap = nzb;
    }
}

/* ------------------------------------------------------------------ kernels (one pass each) */

/* R:53-78 == S:110-135.  gx,gz = number of threads launched per axis (grid*block). */
void orc_kernel_lap(int order, int nx, int nz, int gx, int gz, const float *p, float *lap,
                    const float *coefsx, const float *coefsz)
{
    /* thread (ti, tj) works on i = h + ti, j = h + tj and returns unless i < nx - h, j < nz - h (R:56-64) */
    int half_order = order / 2, ti, tj, io;
    int nti = gx < nx - 2 * half_order ? gx : nx - 2 * half_order;
    int ntj = gz < nz - 2 * half_order ? gz : nz - 2 * half_order;
#ifdef _OPENMP
#pragma omp parallel for private(tj, io) schedule(static)   /* rows are independent; per-point arithmetic unchanged */
#endif
    for (ti = 0; ti < nti; ti++) {
        int i = half_order + ti;
        for (tj = 0; tj < ntj; tj++) {
            int j = half_order + tj;
            size_t mult = (size_t)i * nz;
            float acmx = 0, acmz = 0;
            for (io = 0; io <= order; io++) {
                int aux = io - half_order;
                acmz += p[mult + j + aux] * coefsz[io];
                acmx += p[(size_t)(i + aux) * nz + j] * coefsx[io];
            }
            lap[mult + j] = acmz + acmx;
        }
    }
}

/* FAST numerics (see the header): lap = c0 p + sum_k [cz_k (p(j-k) + p(j+k)) + cx_k (p(i-k) + p(i+k))], c0 = cz_0 + cx_0 in fp32, one chain:
 * acc = c0 * p; per k = 1..h: acc = fma(z sum, cz_k, acc); acc = fma(x sum, cx_k, acc).  p points at the centre, sx = floats between rows. */
static float orc_lap_fast(const float *p, size_t sx, int h, const float *coefsx, const float *coefsz)
{
    float c0 = coefsz[h] + coefsx[h];
    float acc = c0 * p[0];
    int k;
    for (k = 1; k <= h; k++) {
        float sz = p[-k] + p[k];
        float sxs = p[-(long)(k * sx)] + p[k * sx];
        acc = fmaf(sz, coefsz[h - k], acc);
        acc = fmaf(sxs, coefsx[h - k], acc);
    }
    return acc;
}

/* kernel_lap's extents with the FAST Laplacian */
void orc_kernel_lap_fast(int order, int nx, int nz, int gx, int gz, const float *p, float *lap,
                         const float *coefsx, const float *coefsz)
{
    int half_order = order / 2, ti, tj;
    int nti = gx < nx - 2 * half_order ? gx : nx - 2 * half_order;
    int ntj = gz < nz - 2 * half_order ? gz : nz - 2 * half_order;
#ifdef _OPENMP
#pragma omp parallel for private(tj) schedule(static)
#endif
    for (ti = 0; ti < nti; ti++) {
        int i = half_order + ti;
        for (tj = 0; tj < ntj; tj++) {
            int j = half_order + tj;
            lap[(size_t)i * nz + j] = orc_lap_fast(p + (size_t)i * nz + j, (size_t)nz, half_order, coefsx, coefsz);
        }
    }
}

/* R:80-92: the literal 2. makes the sum double; only (v2*dt2)*lap is a float product. */
void orc_kernel_time(int nx, int nz, int gx, int gz, const float *p, float *pp, const float *v2,
                     const float *lap, float dt2)
{
    int i, j, ni = gx < nx ? gx : nx;
#ifdef _OPENMP
#pragma omp parallel for private(j) schedule(static)
#endif
    for (i = 0; i < ni; i++)
        for (j = 0; j < gz && j < nz; j++) {
            size_t k = (size_t)i * nz + j;
            pp[k] = 2. * p[k] - pp[k] + v2[k] * dt2 * lap[k];
        }
}

/* R:94-117.  gx = threads along x, gz = threads along z (already 8*floor(nzb/8) in compat).
 * The reference races on the right-hand columns (thread i scales column nx-1-i while that
 * column's own thread scales it by taperz); the intended result is both multiplies.  We fix the
 * order (p*taperz)*taperx on both sides: pass 1 = every thread's taperz, pass 2 = the taperx pair. */
void orc_kernel_tapper(int nx, int nz, int nxb, int nzb, int gx, int gz, float *p, float *pp,
                       const float *taperx, const float *taperz)
{
    int i, j, itxr = nx - 1;
    for (i = 0; i < gx && i < nx; i++)
        for (j = 0; j < gz && j < nzb; j++) {
            size_t k = (size_t)i * nz + j;
            p[k] *= taperz[j];
            pp[k] *= taperz[j];
        }
    for (i = 0; i < gx && i < nxb; i++)
        for (j = 0; j < gz && j < nzb; j++) {
            size_t k = (size_t)i * nz + j, kr = (size_t)(itxr - i) * nz + j;
            p[k] *= taperx[i];
            pp[k] *= taperx[i];
            p[kr] *= taperx[i];
            pp[kr] *= taperx[i];
        }
}

/* R:119-122: 64 racing threads, one add intended (DD rtm_main.cpp:171 confirms). */
void orc_kernel_src(int nz, float *pp, int sx, int sz, float srce) { pp[(size_t)sx * nz + sz] += srce; }

/* R:124-131: one add per receiver (the 8 threadIdx.y replicas race on the same address). */
void orc_kernel_sism(int nx, int nz, int nxb, int nt, int it, int gz_, int gx, const float *d_obs,
                     float *ppr)
{
    int size = nx - 2 * nxb, i;
    for (i = 0; i < gx && i < size; i++)
        ppr[(size_t)(i + nxb) * nz + gz_] += d_obs[(size_t)i * nt + (nt - 1 - it)];
}

/* R:133-144 */
void orc_kernel_img(int nx, int nz, int nxb, int nzb, int gx, int gz, float *imloc, const float *p,
                    const float *ppr)
{
    int size_x = nx - 2 * nxb, size_z = nz - 2 * nzb, i, j, ni = gx < size_x ? gx : size_x;
#ifdef _OPENMP
#pragma omp parallel for private(j) schedule(static)
#endif
    for (i = 0; i < ni; i++)
        for (j = 0; j < gz && j < size_z; j++) {
            size_t k = (size_t)(i + nxb) * nz + (j + nzb);
            imloc[(size_t)i * size_z + j] += p[k] * ppr[k];
        }
}

/* ------------------------------------------------------------------ propagation */

typedef struct {
    int order, nxe, nze, nxb, nzb, nt;
    int xlim, zlim, ztap; /* launch extents, orc_extents() */
    float dt2;
    float coefs_x[65], coefs_z[65];
    float *taper_x, *taper_z;
    float *d_laplace; /* shared scratch like R:32; calloc'ed (device memory reads as zero) */
    int numerics;     /* 0: the reference's arithmetic; 1: FAST (orc_lap_fast) */
} orc_state;

void orc_set_numerics(orc_state *s, int numerics) { s->numerics = numerics; }
static void orc_lap_pass(const orc_state *s, const float *p, float *lap)
{
    if (s->numerics) orc_kernel_lap_fast(s->order, s->nxe, s->nze, s->xlim, s->zlim, p, lap, s->coefs_x, s->coefs_z);
    else orc_kernel_lap(s->order, s->nxe, s->nze, s->xlim, s->zlim, p, lap, s->coefs_x, s->coefs_z);
}

orc_state *orc_init(int order, int nxe, int nze, int nxb, int nzb, int nt, float fac, float dx,
                    float dz, float dt, int compat)
{
    orc_state *s = (orc_state *)calloc(1, sizeof(orc_state));
    if (order < 2 || order > 64 || (order & 1)) { free(s); return NULL; }
    s->order = order; s->nxe = nxe; s->nze = nze; s->nxb = nxb; s->nzb = nzb; s->nt = nt;
    s->dt2 = dt * dt; /* R:205 */
    orc_scaled_coefs(order, dx, dz, 0, s->coefs_x, s->coefs_z);
    s->taper_x = (float *)calloc(nxb > 0 ? nxb : 1, sizeof(float));
    s->taper_z = (float *)calloc(nzb > 0 ? nzb : 1, sizeof(float));
    orc_taper_tables(nxb, nzb, fac, s->taper_x, s->taper_z);
    orc_extents(nxe, nze, nzb, compat, &s->xlim, &s->zlim, &s->ztap);
    s->d_laplace = (float *)calloc((size_t)nxe * nze, sizeof(float));
    return s;
}

void orc_free(orc_state *s)
{
    if (!s) return;
    free(s->taper_x); free(s->taper_z); free(s->d_laplace); free(s);
}

/* one forward iteration body, R:260-267, on caller-owned d_p/d_pp pointers (swapped in place). */
static void orc_forward_step(orc_state *s, float **d_p, float **d_pp, const float *v2, int sx, int sz,
                             float srce_it)
{
    float *d_swap = *d_pp;
    *d_pp = *d_p;
    *d_p = d_swap;
    orc_kernel_tapper(s->nxe, s->nze, s->nxb, s->nzb, s->xlim, s->ztap, *d_p, *d_pp, s->taper_x, s->taper_z);
    orc_lap_pass(s, *d_p, s->d_laplace);
    orc_kernel_time(s->nxe, s->nze, s->xlim, s->zlim, *d_p, *d_pp, v2, s->d_laplace, s->dt2);
    orc_kernel_src(s->nze, *d_pp, sx, sz, srce_it);
}

/* R:247-288.  p,pp are [nxe][nze] host arrays, overwritten with d_p (u^{nt-1}, tapered) and d_pp
 * (u^{nt}).  nsteps<=nt lets tests stop early; it0 offsets the source sample index. */
void orc_fd_forward(orc_state *s, float *p, float *pp, const float *v2, int sx, int sz,
                    const float *srce, int nsteps)
{
    size_t n = (size_t)s->nxe * s->nze;
    float *a = (float *)malloc(n * sizeof(float)), *b = (float *)malloc(n * sizeof(float));
    float *d_p = a, *d_pp = b;
    int it;
    memcpy(d_p, p, n * sizeof(float));
    memcpy(d_pp, pp, n * sizeof(float));
    for (it = 0; it < nsteps; it++) orc_forward_step(s, &d_p, &d_pp, v2, sx, sz, srce[it]);
    memcpy(p, d_p, n * sizeof(float));
    memcpy(pp, d_pp, n * sizeof(float));
    free(a); free(b);
}

/* R:290-341.  snap0 = P (u^{nt-1}), snap1 = PP (u^{nt}) of the forward pass; d_obs is one shot
 * gather laid out [ix][it] (R:426-435); imloc [nx][nz] is accumulated into (R:243 uploads it).
 * The four fields start at zero exactly as main does (R:511-514). */
void orc_fd_back(orc_state *s, const float *v2, const float *snap0, const float *snap1,
                 const float *d_obs, int gz_, float *imloc, int nsteps)
{
    size_t n = (size_t)s->nxe * s->nze;
    float *d_p = (float *)calloc(n, sizeof(float)), *d_pp = (float *)calloc(n, sizeof(float));
    float *d_pr = (float *)calloc(n, sizeof(float)), *d_ppr = (float *)calloc(n, sizeof(float));
    float *d_swap;
    int it;
    for (it = 0; it < nsteps; it++) {
        if (it == 0 || it == 1) {
            memcpy(d_pp, it == 0 ? snap1 : snap0, n * sizeof(float)); /* R:304-314 */
        } else {
            orc_lap_pass(s, d_p, s->d_laplace);
            orc_kernel_time(s->nxe, s->nze, s->xlim, s->zlim, d_p, d_pp, v2, s->d_laplace, s->dt2);
        }
        d_swap = d_pp; d_pp = d_p; d_p = d_swap; /* R:321-323 */
        orc_kernel_tapper(s->nxe, s->nze, s->nxb, s->nzb, s->xlim, s->ztap, d_pr, d_ppr, s->taper_x, s->taper_z);
        orc_lap_pass(s, d_pr, s->d_laplace);
        orc_kernel_time(s->nxe, s->nze, s->xlim, s->zlim, d_pr, d_ppr, v2, s->d_laplace, s->dt2);
        orc_kernel_sism(s->nxe, s->nze, s->nxb, s->nt, it, gz_, s->xlim, d_obs, d_ppr);
        orc_kernel_img(s->nxe, s->nze, s->nxb, s->nzb, s->xlim, s->zlim, imloc, d_p, d_ppr);
        d_swap = d_ppr; d_ppr = d_pr; d_pr = d_swap; /* R:331-333 */
    }
    free(d_p); free(d_pp); free(d_pr); free(d_ppr);
}

/* S:241-262 + S:325: the stencil program's single launch.  Grid rounds up to 32 (S:231-238) so the
 * whole interior is covered; border cells of the output stay zero (S:153 memset + golden). */
void orc_stencil(int order, int nxe, int nze, float dx, float dz, const float *in, float *out)
{
    float cx[65], cz[65];
    int gx = ((nxe - 1) / 32 + 1) * 32, gz = ((nze - 1) / 32 + 1) * 32;
    orc_scaled_coefs(order, dx, dz, 1, cx, cz);
    memset(out, 0, (size_t)nxe * nze * sizeof(float));
    orc_kernel_lap(order, nxe, nze, gx, gz, in, out, cx, cz);
}

/* One forward iteration (R:260-267 without the pointer swap) on an x-slab of a decomposed grid: the
 * local arrays hold global rows [x_off, x_off+nxl); rows [r0,r1) (local) are updated, rows
 * [r0-h, r1+h) are damped in place first (they are exactly the rows this step reads).  The reference
 * has no decomposition; this is the per-slab restatement used by the CPU (gloo) tests of the halo
 * exchange logic -- with x_off=0, nxl=nxe, r0=t0=0, r1=t1=nxe it is orc_forward_step. */
/* the same launch with the FAST Laplacian (fdw_dev_laplacian on a numerics = 1 context) */
void orc_stencil_fast(int order, int nxe, int nze, float dx, float dz, const float *in, float *out)
{
    float cx[65], cz[65];
    int gx = ((nxe - 1) / 32 + 1) * 32, gz = ((nze - 1) / 32 + 1) * 32;
    orc_scaled_coefs(order, dx, dz, 1, cx, cz);
    memset(out, 0, (size_t)nxe * nze * sizeof(float));
    orc_kernel_lap_fast(order, nxe, nze, gx, gz, in, out, cx, cz);
}

void orc_slab_step(const orc_state *s, int x_off, int nxl, float *p, float *pp, const float *v2, int r0, int r1,
                   int t0, int t1, int sx_global, int sz, float srce_it)
{
    /* [t0,t1): rows damped in place by this call.  A caller that splits one time step into several
     * row ranges damps every row it will read exactly once (first call), t0>=t1 on the others. */
    const int h = s->order / 2, nze = s->nze, nxe = s->nxe;
    int l, j, io;
    float *lap = (float *)calloc((size_t)nxl * nze, sizeof(float));
    /* kernel_tapper with its thread extents, in global coordinates (pass 1 taperz, pass 2 taperx) */
    for (l = t0; l < t1; l++) {
        int g = x_off + l, gm = nxe - 1 - g;
        for (j = 0; j < s->ztap && j < s->nzb; j++) {
            size_t k = (size_t)l * nze + j;
            if (g < s->xlim) { p[k] *= s->taper_z[j]; pp[k] *= s->taper_z[j]; }
        }
        for (j = 0; j < s->ztap && j < s->nzb; j++) {
            size_t k = (size_t)l * nze + j;
            if (g < s->nxb && g < s->xlim) { p[k] *= s->taper_x[g]; pp[k] *= s->taper_x[g]; }
            else if (gm < s->nxb && gm < s->xlim) { p[k] *= s->taper_x[gm]; pp[k] *= s->taper_x[gm]; }
        }
    }
    for (l = r0; l < r1; l++) {
        int g = x_off + l;
        if (g < h || g >= nxe - h || g >= h + s->xlim || l < h || l >= nxl - h) continue;
        for (j = h; j < nze - h && j < h + s->zlim; j++) {
            float acmx = 0, acmz = 0;
            if (s->numerics) {
                lap[(size_t)l * nze + j] = orc_lap_fast(p + (size_t)l * nze + j, (size_t)nze, h, s->coefs_x, s->coefs_z);
                continue;
            }
            for (io = 0; io <= s->order; io++) {
                acmz += p[(size_t)l * nze + j + io - h] * s->coefs_z[io];
                acmx += p[(size_t)(l + io - h) * nze + j] * s->coefs_x[io];
            }
            lap[(size_t)l * nze + j] = acmz + acmx;
        }
    }
    for (l = r0; l < r1; l++) {
        if (x_off + l >= s->xlim) continue;
        for (j = 0; j < s->zlim; j++) {
            size_t k = (size_t)l * nze + j;
            pp[k] = 2. * p[k] - pp[k] + v2[k] * s->dt2 * lap[k];
        }
    }
    l = sx_global - x_off;
    if (sx_global >= 0 && l >= r0 && l < r1) pp[(size_t)l * nze + sz] += srce_it;
    free(lap);
}

/* One iteration of fd_back's loop (R:302-339) on rows [r0,r1) of an x-slab, the per-slab restatement behind the CPU (gloo) tests of
 * decomp.SlabBack -- the reference has no decomposition.  Local arrays hold global rows [x_off, x_off+nxl).
 *   step_source != 0: F_k = leap-frog(f1 = F_{k-1}, f0 = F_{k-2}) written over f0 (kernel_lap + kernel_time, R:317-318)
 *   step_source == 0: the source field is f1 as it stands (iterations 0 and 1: the snapshots, R:304-314)
 *   then kernel_tapper on (pr, ppr) for local rows [t0,t1) (R:325; a caller that splits one iteration into several row ranges damps
 *   every row it will read exactly once), kernel_lap + kernel_time on the receiver pair (R:326-327), kernel_sism with samples[i] =
 *   d_obs[i][nt-1-it] (R:328) and kernel_img (R:329) into img[nxl][nze] on the extended grid.  With x_off = 0, nxl = nxe, full ranges
 *   it is the loop body of orc_fd_back. */
void orc_slab_back_iter(const orc_state *s, int x_off, int nxl, int step_source, float *f1, float *f0, float *pr, float *ppr,
                        const float *v2, int r0, int r1, int t0, int t1, const float *samples, int gz_, float *img)
{
    const int h = s->order / 2, nze = s->nze, nxe = s->nxe, nx = nxe - 2 * s->nxb, nz = nze - 2 * s->nzb;
    int l, j, io, pass;
    float *lap = (float *)calloc((size_t)nxl * nze, sizeof(float));
    const float *F = step_source ? f0 : f1;
    for (l = t0; l < t1; l++) {
        int g = x_off + l, gm = nxe - 1 - g;
        for (j = 0; j < s->ztap && j < s->nzb; j++) {
            size_t k = (size_t)l * nze + j;
            if (g < s->xlim) { pr[k] *= s->taper_z[j]; ppr[k] *= s->taper_z[j]; }
        }
        for (j = 0; j < s->ztap && j < s->nzb; j++) {
            size_t k = (size_t)l * nze + j;
            if (g < s->nxb && g < s->xlim) { pr[k] *= s->taper_x[g]; ppr[k] *= s->taper_x[g]; }
            else if (gm < s->nxb && gm < s->xlim) { pr[k] *= s->taper_x[gm]; ppr[k] *= s->taper_x[gm]; }
        }
    }
    for (pass = step_source ? 0 : 1; pass < 2; pass++) {      /* pass 0: the source pair, pass 1: the receiver pair */
        const float *p = pass == 0 ? f1 : pr;
        float *pp = pass == 0 ? f0 : ppr;
        memset(lap, 0, (size_t)nxl * nze * sizeof(float));
        for (l = r0; l < r1; l++) {
            int g = x_off + l;
            if (g < h || g >= nxe - h || g >= h + s->xlim || l < h || l >= nxl - h) continue;
            for (j = h; j < nze - h && j < h + s->zlim; j++) {
                float acmx = 0, acmz = 0;
                if (s->numerics) {
                    lap[(size_t)l * nze + j] = orc_lap_fast(p + (size_t)l * nze + j, (size_t)nze, h, s->coefs_x, s->coefs_z);
                    continue;
                }
                for (io = 0; io <= s->order; io++) {
                    acmz += p[(size_t)l * nze + j + io - h] * s->coefs_z[io];
                    acmx += p[(size_t)(l + io - h) * nze + j] * s->coefs_x[io];
                }
                lap[(size_t)l * nze + j] = acmz + acmx;
            }
        }
        for (l = r0; l < r1; l++) {
            if (x_off + l >= s->xlim) continue;
            for (j = 0; j < s->zlim; j++) {
                size_t k = (size_t)l * nze + j;
                pp[k] = 2. * p[k] - pp[k] + v2[k] * s->dt2 * lap[k];
            }
        }
    }
    for (l = r0; l < r1; l++) {
        int i = x_off + l - s->nxb;                            /* interior row index of kernel_sism / kernel_img */
        if (i < 0 || i >= nx || i >= s->xlim) continue;
        ppr[(size_t)l * nze + gz_] += samples[i];
        for (j = 0; j < s->zlim && j < nz; j++) {
            size_t k = (size_t)l * nze + (j + s->nzb);
            img[k] += F[k] * ppr[k];
        }
    }
    free(lap);
}

int orc_max_threads(void);
/* Fused single-pass form of the forward iteration body used ONLY as the cpu_baseline "port" timing
 * kernel in bench.py (same arithmetic per point as lap+time above, full extents, no taper/source):
 * reads p, pp, v2 and writes pp -- the 16 B/point shape the GPU kernel is priced on. */
void orc_fused_steps(int order, int nxe, int nze, float *p, float *pp, const float *v2,
                     const float *cx, const float *cz, float dt2, int nsteps)
{
    int h = order / 2, it, i, j, io;
    for (it = 0; it < nsteps; it++) {
        float *t;
        /* rows are independent within a step: the OpenMP build (liborc_native.so, oracle/Makefile) shares them out; the arithmetic
         * per point is unchanged, so the threaded result equals the serial one bit for bit */
#ifdef _OPENMP
#pragma omp parallel for private(j, io) schedule(dynamic, 8)
#endif
        for (i = 0; i < nxe; i++)
            for (j = 0; j < nze; j++) {
                size_t k = (size_t)i * nze + j;
                float lap = 0.0f;
                if (i >= h && i < nxe - h && j >= h && j < nze - h) {
                    float acmx = 0, acmz = 0;
                    for (io = 0; io <= order; io++) {
                        long a = io - h;
                        acmz += p[(long)k + a] * cz[io];
                        acmx += p[(long)k + a * nze] * cx[io];
                    }
                    lap = acmz + acmx;
                }
                pp[k] = 2. * p[k] - pp[k] + v2[k] * dt2 * lap;
            }
        t = p; p = pp; pp = t;
    }
}

#ifdef _OPENMP
#include <omp.h>
int orc_max_threads(void) { return omp_get_max_threads(); }
#else
int orc_max_threads(void) { return 1; }
#endif
