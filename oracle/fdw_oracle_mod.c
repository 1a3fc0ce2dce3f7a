/*
 * fdw_oracle_mod.c -- TEST INFRASTRUCTURE ONLY (same rules as fdw_oracle.c).  CPU restatement of the
 * forward-modelling producer of the reference's CPU-serial sibling: the program
 * that synthesises the `datfile` an RTM run consumes.
 *
 * Citations: M  = dpct_gpu_rtm_domain_division/src/mod_main.cpp
 *            FD = dpct_gpu_rtm_domain_division/src/timestep/fd.c
 *            T  = dpct_gpu_rtm_domain_division/src/boundary/taper.c
 *            PS = dpct_gpu_rtm_domain_division/src/source/ptsrc.c
 * The reference builds ALL of these with g++ (their Makefiles set CC = g++ -fpermissive), so calls such as
 * exp(float) resolve to the C++ float overloads; this file is C and spells those choices out (expf / exp).
 *
 * Parity pins: orc_mod_shot reproduces build/3lay_mod/dobs.bin (151 traces x 1001 samples) BIT-EXACTLY from
 * build/3lay_mod/3layer_151x151.bin + input.dat (tests/test_oracle_golden.py); the individual passes are
 * bit-exact against oracle/_ref/libref_dd.so = FD, T, PS compiled unmodified with the reference's own flags.
 *
 * Build: gcc -O2 -ffp-contract=off (the reference's g++ -O3 for x86-64 has no FMA to contract either).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define MOD_PI (3.141592653589793) /* cwp.h PI */

/* FD:54-92 calc_coefs: table orders 2..8 are the same rationals as the CUDA path's; other orders (makeo2 as C++:
 * cos(float) and pow(float,float) are the float overloads) are not needed by any deck and are left to orc_calc_coefs(cxx=1). */
void orc_calc_coefs(int order, int cxx, float *coef);

/* T:26-44: taper[i] = exp(-pow(F*(nb-i), 2)); F*(nb-i) is a float product, pow(float,int) promotes to double in C++11 */
void orc_mod_taper_tables(int nxb, int nzb, float F, float *taperx, float *taperz)
{
    for (int i = 0; i < nxb; i++) taperx[i] = exp(-pow((double)(F * (nxb - i)), 2));
    for (int i = 0; i < nzb; i++) taperz[i] = exp(-pow((double)(F * (nzb - i)), 2));
}

/* T:7-23: replicate the edge values of the (squared) velocity outwards: first along z for the interior rows (top from the first
 * interior sample, bottom from the last), then whole rows along x (left rows from the first interior row, right rows from the last) */
void orc_mod_extendvel(int nx, int nz, int nxb, int nzb, float *vel)
{
    const size_t nze = (size_t)nz + 2 * (size_t)nzb;
    const int nxe = nx + 2 * nxb;
    for (int row = nxb; row < nxb + nx; row++) {
        float *r = vel + (size_t)row * nze;
        const float top = r[nzb], bottom = r[nzb + nz - 1];
        for (int k = 0; k < nzb; k++) {
            r[k] = top;
            r[nzb + nz + k] = bottom;
        }
    }
    const float *first = vel + (size_t)nxb * nze, *last = vel + (size_t)(nxb + nx - 1) * nze;
    for (int row = 0; row < nxe; row++) {
        if (row >= nxb && row < nxb + nx) continue;
        memcpy(vel + (size_t)row * nze, row < nxb ? first : last, nze * sizeof(float));
    }
}

/* PS:88-99 + PS:60-86: Ricker wavelet delayed by 1/fpeak and cut off after 2/fpeak */
void orc_mod_ricker_wavelet(int nt, float dt, float peak, float *s)
{
    for (int it = 0; it < nt; it++) {
        if (it * dt > 2.0 / peak) {
            s[it] = 0.0;
        } else {
            const float t = it * dt - 1.0 / peak;
            const float x = MOD_PI * peak * t;
            const float xx = x * x;
            s[it] = expf(-xx) * (1.0 - 2.0 * xx); /* exp(float) is the float overload under g++ */
        }
    }
}

/* T:46-66 taper_apply: every sample of the four strips is multiplied by its z factor first (top strip taperz[iz], bottom strip
 * mirrored) and then, in the left / right strips, by its x factor (mirrored on the right); an element-wise operation, so only the
 * order of the two products per sample matters */
static float mod_strip_factor(const float *taper, int nb, int n, int i)
{
    if (i < nb) return taper[i];
    if (i >= n - nb) return taper[n - 1 - i];
    return 1.0f;
}
void orc_mod_taper_apply(float *pp, int nx, int nz, int nxb, int nzb, const float *taperx, const float *taperz)
{
    const int nxe = nx + 2 * nxb, nze = nz + 2 * nzb;
    for (int ix = 0; ix < nxe; ix++) {
        const int xstrip = ix < nxb || ix >= nxe - nxb;
        const float fx = mod_strip_factor(taperx, nxb, nxe, ix);
        float *row = pp + (size_t)ix * nze;
        for (int iz = 0; iz < nze; iz++) {
            if (iz < nzb || iz >= nze - nzb) row[iz] *= mod_strip_factor(taperz, nzb, nze, iz);
            if (xstrip) row[iz] *= fx;
        }
    }
}

/* FAST numerics of the product for these dialects (include/fdwave.h fdw_params.numerics = 1; NOT the sibling's arithmetic, see the header of
 * fdw_oracle.c): the weights carry their spacing (czf_k = c_k * dz2inv, cxf_k = c_k * dx2inv, c0 = czf_0 + cxf_0, all fp32) and the Laplacian is
 * one chain of symmetric sums and fused multiply-adds, exactly the RTM dialect's FAST formula.  orc_mod_set_numerics(1) switches every loop
 * of this file to it. */
static int orc_mod_numerics = 0;
void orc_mod_set_numerics(int numerics) { orc_mod_numerics = numerics; }

/* FD:24-46 fd_step: ONE accumulator, z term then x term per tap, weights scaled per term; Laplacian only inside the
 * order/2 frame (zero elsewhere, FD:19); update on the whole grid */
void orc_mod_fd_step(int order, const float *coefs, float dx2inv, float dz2inv, float dt2, const float *p, float *pp, const float *v2,
                     float *laplace, int nze, int nxe)
{
    const int h = order / 2;
    float acm = 0;
    if (orc_mod_numerics) {
        float czf[65], cxf[65];
        for (int io = 0; io <= order; io++) {
            czf[io] = coefs[io] * dz2inv;
            cxf[io] = coefs[io] * dx2inv;
        }
        const float c0 = czf[h] + cxf[h];
        for (int ix = h; ix < nxe - h; ix++)
            for (int iz = h; iz < nze - h; iz++) {
                const float *q = p + (size_t)ix * nze + iz;
                float acc = c0 * q[0];
                for (int k = 1; k <= h; k++) {
                    acc = fmaf(q[-k] + q[k], czf[h - k], acc);
                    acc = fmaf(q[-(long)k * nze] + q[(long)k * nze], cxf[h - k], acc);
                }
                laplace[ix * nze + iz] = acc;
            }
    } else
    for (int ix = h; ix < nxe - h; ix++)
        for (int iz = h; iz < nze - h; iz++) {
            for (int io = 0; io <= order; io++) {
                acm += p[ix * nze + iz + io - h] * coefs[io] * dz2inv;
                acm += p[(ix + io - h) * nze + iz] * coefs[io] * dx2inv;
            }
            laplace[ix * nze + iz] = acm;
            acm = 0.0;
        }
    for (int ix = 0; ix < nxe; ix++)
        for (int iz = 0; iz < nze; iz++) {
            const int i = ix * nze + iz;
            pp[i] = 2. * p[i] - pp[i] + v2[i] * dt2 * laplace[i];
        }
}

/* PS:12-58 ptsrc: a Gaussian blob of radius 3 cells around (xs, zs), clipped by the array; exp(float) is the float overload under
 * g++, the product and the sum are float */
void orc_mod_ptsrc(int xs, int zs, int nx, int nz, float ts, float *s)
{
    for (int dx = -3; dx <= 3; dx++) {
        const int ix = xs + dx;
        if (ix < 0 || ix > nx - 1) continue;
        for (int dz = -3; dz <= 3; dz++) {
            const int iz = zs + dz;
            if (iz < 0 || iz > nz - 1) continue;
            const float xn = (float)ix - (float)xs, zn = (float)iz - (float)zs;
            s[(size_t)ix * nz + iz] += ts * expf(-xn * xn - zn * zn);
        }
    }
}

/* M:140-174: one shot of the modelling loop.  v2 is the extended squared velocity, data is [nx][nt]. */
void orc_mod_shot(int order, int nx, int nz, int nxb, int nzb, int nt, float dx, float dz, float dt, float fac, const float *v2, int sx, int sz,
                  int gz, const float *srce, float *data)
{
    const int nxe = nx + 2 * nxb, nze = nz + 2 * nzb;
    const size_t ne = (size_t)nxe * nze;
    const float dx2inv = (1. / dx) * (1. / dx), dz2inv = (1. / dz) * (1. / dz), dt2 = dt * dt; /* FD:13-15 */
    float coefs[65];
    orc_calc_coefs(order, 1, coefs);
    float *taperx = (float *)malloc(sizeof(float) * (nxb > 0 ? nxb : 1)), *taperz = (float *)malloc(sizeof(float) * (nzb > 0 ? nzb : 1));
    orc_mod_taper_tables(nxb, nzb, fac, taperx, taperz);
    float *P = (float *)calloc(ne, sizeof(float)), *PP = (float *)calloc(ne, sizeof(float)), *lap = (float *)calloc(ne, sizeof(float));
    for (int it = 0; it < nt; it++) {
        orc_mod_fd_step(order, coefs, dx2inv, dz2inv, dt2, P, PP, v2, lap, nze, nxe);
        orc_mod_ptsrc(sx, sz, nxe, nze, srce[it], PP);
        orc_mod_taper_apply(PP, nx, nz, nxb, nzb, taperx, taperz);
        orc_mod_taper_apply(P, nx, nz, nxb, nzb, taperx, taperz);
        for (int ix = 0; ix < nx; ix++) data[(size_t)ix * nt + it] = P[(size_t)(ix + nxb) * nze + gz];
        float *tmp = PP; PP = P; P = tmp;
    }
    free(P); free(PP); free(lap); free(taperx); free(taperz);
}

/* ======================================================================================================================
 * Stored-wavefield RTM of the same sibling: RM = dpct_gpu_rtm_domain_division/src/rtm_main.cpp.
 * Same fd_step; point source on ONE cell; taper_apply2 (top strip only, x factors only inside it: T:68-83); the source
 * wavefield of every step is kept and the image is formed from the stored fields.
 * Pin: orc_rtm_stored_shot reproduces build/3lay_mod/dir.image bit-exactly from that deck, model and dobs.bin.
 * ====================================================================================================================== */

/* T:68-83 taper_apply2: only the top strip is damped -- z factor for every row, then the x factor in the left / right rows */
void orc_mod_taper_apply2(float *pp, int nx, int nz, int nxb, int nzb, const float *taperx, const float *taperz)
{
    const int nxe = nx + 2 * nxb, nze = nz + 2 * nzb;
    for (int ix = 0; ix < nxe; ix++) {
        const int xstrip = ix < nxb || ix >= nxe - nxb;
        const float fx = mod_strip_factor(taperx, nxb, nxe, ix);
        float *row = pp + (size_t)ix * nze;
        for (int iz = 0; iz < nzb; iz++) {
            row[iz] *= taperz[iz];
            if (xstrip) row[iz] *= fx;
        }
    }
}

/* RM:158-240 for one shot.  dobs_flat is the WHOLE gather file [ns][nx][nt] (n_flat floats) and `is` the shot: the reference reads
 * sample nt-it of trace ix (RM:203), i.e. one sample past the trace at it = 0 -- the first sample of the next trace, or, for the last
 * trace of the last shot, one float past the allocation (taken as 0 here).  It also offsets the receiver rows by nzb, not nxb
 * (RM:203), which is kept.  imloc[nx][nz] is overwritten (RM:189). */
void orc_rtm_stored_shot(int order, int nx, int nz, int nxb, int nzb, int nt, float dx, float dz, float dt, float fac, const float *v2, int sx,
                         int sz, int gz, const float *srce, const float *dobs_flat, size_t n_flat, int is, float *imloc)
{
    const int nxe = nx + 2 * nxb, nze = nz + 2 * nzb;
    const size_t ne = (size_t)nxe * nze, ni = (size_t)nx * nz;
    const float dx2inv = (1. / dx) * (1. / dx), dz2inv = (1. / dz) * (1. / dz), dt2 = dt * dt;
    float coefs[65];
    orc_calc_coefs(order, 1, coefs);
    float *taperx = (float *)malloc(sizeof(float) * (nxb > 0 ? nxb : 1)), *taperz = (float *)malloc(sizeof(float) * (nzb > 0 ? nzb : 1));
    orc_mod_taper_tables(nxb, nzb, fac, taperx, taperz);
    float *P = (float *)calloc(ne, sizeof(float)), *PP = (float *)calloc(ne, sizeof(float)), *lap = (float *)calloc(ne, sizeof(float));
    float *swf = (float *)calloc(ni * (size_t)nt, sizeof(float)), *rwf = (float *)calloc(ni * (size_t)nt, sizeof(float));
    for (int it = 0; it < nt; it++) {
        orc_mod_fd_step(order, coefs, dx2inv, dz2inv, dt2, P, PP, v2, lap, nze, nxe);
        PP[(size_t)sx * nze + sz] += srce[it];
        orc_mod_taper_apply2(PP, nx, nz, nxb, nzb, taperx, taperz);
        orc_mod_taper_apply2(P, nx, nz, nxb, nzb, taperx, taperz);
        for (int ix = 0; ix < nx; ix++)
            for (int iz = 0; iz < nz; iz++) swf[(size_t)it * ni + (size_t)ix * nz + iz] = P[(size_t)(ix + nxb) * nze + iz + nzb];
        float *tmp = PP; PP = P; P = tmp;
    }
    memset(P, 0, ne * sizeof(float));
    memset(PP, 0, ne * sizeof(float));
    memset(imloc, 0, ni * sizeof(float));
    for (int it = 0; it < nt; it++) {
        orc_mod_fd_step(order, coefs, dx2inv, dz2inv, dt2, P, PP, v2, lap, nze, nxe);
        for (int ix = 0; ix < nx; ix++) {
            const size_t k = ((size_t)is * nx + ix) * nt + (size_t)(nt - it);
            PP[(size_t)(ix + nzb) * nze + gz] += k < n_flat ? dobs_flat[k] : 0.0f;
        }
        orc_mod_taper_apply2(PP, nx, nz, nxb, nzb, taperx, taperz);
        orc_mod_taper_apply2(P, nx, nz, nxb, nzb, taperx, taperz);
        for (int ix = 0; ix < nx; ix++)
            for (int iz = 0; iz < nz; iz++) rwf[(size_t)it * ni + (size_t)ix * nz + iz] = P[(size_t)(ix + nxb) * nze + iz + nzb];
        float *tmp = PP; PP = P; P = tmp;
    }
    for (int it = 0; it < nt; it++)
        for (size_t k = 0; k < ni; k++) imloc[k] += swf[(size_t)(nt - it - 1) * ni + k] * rwf[(size_t)it * ni + k];
    free(P); free(PP); free(lap); free(swf); free(rwf); free(taperx); free(taperz);
}

/* ======================================================================================================================
 * Image post-processing: LP = dpct_gpu_rtm_domain_division/build/3lay_mod/laplace.f90 (the same file
 * sits in cuda_reference_RTM/models/3lay_mod), a second-order Laplacian filter of dir.image -> dir.imalap.  LP:25-29, single precision,
 * in the order the Fortran expression spells: ((a - 2*c) + b)/(dz*dz) + ((d - 2*c) + e)/(dx*dx); the frame stays 0.
 * Pin: bit-exact against the program itself built with flang from that source and run on build/3lay_mod/dir.image.
 * ====================================================================================================================== */
void orc_image_laplacian(const float *img, int nx, int nz, float dx, float dz, float *out)
{
    memset(out, 0, (size_t)nx * nz * sizeof(float));
    for (int ix = 1; ix < nx - 1; ix++)
        for (int iz = 1; iz < nz - 1; iz++) {
            const size_t k = (size_t)ix * nz + iz;
            const float c = img[k];
            out[k] = ((img[k + 1] - 2.f * c) + img[k - 1]) / (dz * dz) + ((img[k + nz] - 2.f * c) + img[k - nz]) / (dx * dx);
        }
}

/* ---- the reference's image comparer models/marmousi/psnr (an x86-64 ELF without source; "Usage: ./psnr file1 file2") -------------------
 * Restated from the tool's observable behaviour and pinned to its own output (tests/golden/psnr_reference_output.json, produced by running
 * the binary): the squares (formed in double) are added one after the other into fp32 sums, MSE = sum / n in fp32, RMSE = sqrtf, SNR = 10 log10(sum f2^2 / sum d^2),
 * PSNR = 20 log10(max |f2| / RMSE) narrowed to fp32; d = f1 - f2 is what it writes to ./dir.output.  stats = {MSE, RMSE, SNR, PSNR}. */
void orc_image_compare(const float *f1, const float *f2, size_t n, float *diff, double *stats)
{
    float sd = 0.0f, sb = 0.0f, mx = 0.0f;
    for (size_t i = 0; i < n; i++) {
        const float d = f1[i] - f2[i];
        if (diff) diff[i] = d;
        sd = (float)((double)sd + (double)d * (double)d);      /* the squares are formed in double (pow-like), the running sums kept in fp32 */
        sb = (float)((double)sb + (double)f2[i] * (double)f2[i]);
        if (fabsf(f2[i]) > mx) mx = fabsf(f2[i]);
    }
    const float mse = sd / (float)n, rmse = sqrtf(mse);
    stats[0] = mse;
    stats[1] = rmse;
    stats[2] = 10.0 * log10((double)(sb / sd));
    stats[3] = (float)(20.0 * log10((double)(mx / rmse)));      /* the tool keeps this one in a float (its last printed digit shows it) */
}
