/*
 * fdw_host.c -- host-side formulas of the path (pure C11, no device needed): finite-difference
 * weights, Ricker source, Gaussian taper tables, random-velocity border.  These are the values the
 * reference's libsource.a (cuda_reference_RTM/lib/src/functions.c = F) and fd_init_cuda
 * (cuda_reference_RTM/src/fd-code.cu = R) feed to the kernels, so they must agree to the bit:
 * every intermediate keeps the type (float/double) it has in the reference expression.
 * S = cuda_reference_stencil_computation/fd-source-code.cu.
 */
#include <math.h>
#include <stdlib.h>

#include "fdwave.h"

static const double FDW_PI = 3.141592653589793; /* functions.h:7 */

/* Central second-derivative weights for orders 2..8 as exact rationals; the reference writes them as
 * double quotients narrowed to float (F:120-152).  Half of each (symmetric) stencil, centre last. */
static const struct { int order; double num[5], den[5]; } kTab[] = {
    {2, {1, -2}, {1, 1}},
    {4, {-1, 4, -5}, {12, 3, 2}},
    {6, {1, -3, 3, -49}, {90, 20, 2, 18}},
    {8, {-1, 8, -1, 8, -205}, {560, 315, 5, 5, 72}},
};

/* Windowed series for every other even order (F:160-192; S:184-216 is the same text built as C++,
 * where cos/pow of float operands are the float overloads -- cxx selects that behaviour). */
static void windowed_weights(float *w, int order, int cxx)
{
    const int h = order / 2;
    const float alpha = .54, beta = 6.;
    const float h_beta = 0.5 * beta;
    const float a1 = 2. * alpha - 1.0;
    const float a2 = 2. * (1.0 - alpha);
    float centre = 0.0;
    int sign = -1;
    for (int k = 1; k <= h; k++) {
        sign = -sign;
        const float filt = (2. * sign) / (k * k);
        const float arg = FDW_PI * k / (2. * (h + 2));
        float wind;
        if (cxx) {
            const float c = cosf(arg);
            wind = powf(a1 + a2 * c * c, h_beta);
        } else {
            wind = pow(a1 + a2 * cos(arg) * cos(arg), h_beta);
        }
        w[h + k] = filt * wind;
        centre = centre + w[h + k];
        w[h - k] = w[h + k];
    }
    w[h] = -2. * centre;
}

int fdw_calc_coefs(int order, int cxx, float *coef)
{
    if (order < 2 || order > FDW_MAX_ORDER || (order & 1) || !coef) return FDW_EINVAL;
    for (size_t t = 0; t < sizeof kTab / sizeof kTab[0]; t++)
        if (kTab[t].order == order) {
            const int h = order / 2;
            for (int k = 0; k <= h; k++) coef[k] = coef[order - k] = (float)(kTab[t].num[k] / kTab[t].den[k]);
            return FDW_OK;
        }
    windowed_weights(coef, order, cxx);
    return FDW_OK;
}

/* F:302-334: s[it] = ricker(it*dt - 1/fpeak).  it*dt is a float product, 1.0/fpeak a double. */
void fdw_ricker_wavelet(int nt, float dt, float fpeak, float *s)
{
    for (int it = 0; it < nt; it++) {
        const float t = it * dt - 1.0 / fpeak;
        const float x = FDW_PI * fpeak * t;
        const float xx = x * x;
        s[it] = exp(-xx) * (1.0 - 2.0 * xx);
    }
}

/* R:159-166: exp(-(dfrac*(nb-i))^2), dfrac = sqrt(-log F)/nb.  In the .cu the sqrt/log of a float are
 * the float overloads and pow(float,int) is evaluated in double. */
static void gauss_taper(int nb, float fac, float *t)
{
    const float dfrac = sqrtf(-logf(fac)) / (1. * nb);
    for (int i = 0; i < nb; i++) {
        const float a = dfrac * (nb - i);
        t[i] = exp(-pow((double)a, 2));
    }
}

void fdw_taper_tables(int nxb, int nzb, float fac, float *taper_x, float *taper_z)
{
    if (taper_x) gauss_taper(nxb, fac, taper_x);
    if (taper_z) gauss_taper(nzb, fac, taper_z);
}

/* F:336-394.  Border model: top = first interior sample replicated; bottom / left / right = integer
 * draws around a linear ramp from the edge velocity down to 300 m/s, window +-200 m/s; bottom
 * corners are filled along anti-diagonals.  The glibc rand() stream must be consumed in exactly the
 * reference's order, which is what the loop nesting below preserves. */
static float ramp_to_floor(float v, int k, int nb)
{
    const float floor_v = 300.;
    return v - (v - floor_v) * k / (nb - 1);
}
/* The reference draws from glibc rand() and never seeds it (R:486), i.e. it consumes the seed-1 stream of
 * glibc's default generator from the start of the process.  Inside a HIP process that stream is not ours
 * alone (runtime libraries may draw from it too), so the generator is restated here and kept private:
 * glibc TYPE_3 additive feedback, r[i] = r[i-3] + r[i-31] over 32-bit words, seeded by the minimal-standard
 * LCG (16807 mod 2^31-1), first 310 outputs discarded, result = word >> 1.  fdw_srand(1) == a fresh process. */
static struct { unsigned r[34]; int i; int ready; } g_rng;

void fdw_srand(unsigned seed)
{
    int word = seed ? (int)seed : 1;
    g_rng.r[0] = (unsigned)word;
    for (int k = 1; k < 31; k++) {
        const long hi = word / 127773, lo = word % 127773;
        long t = 16807 * lo - 2836 * hi;
        if (t < 0) t += 2147483647;
        word = (int)t;
        g_rng.r[k] = (unsigned)word;
    }
    /* state as a ring of 31 words: front = r[3], rear = r[0] (glibc: fptr = &state[3], rptr = &state[0]) */
    g_rng.i = 0;
    g_rng.ready = 1;
    for (int k = 0; k < 310; k++) {
        const int f = (g_rng.i + 3) % 31, b = g_rng.i % 31;
        g_rng.r[f] += g_rng.r[b];
        g_rng.i = (g_rng.i + 1) % 31;
    }
}

static int fdw_rand(void)
{
    if (!g_rng.ready) fdw_srand(1);
    const int f = (g_rng.i + 3) % 31, b = g_rng.i % 31;
    g_rng.r[f] += g_rng.r[b];
    g_rng.i = (g_rng.i + 1) % 31;
    return (int)(g_rng.r[f] >> 1);
}

static float draw_near(float v, float centre)
{
    const float half = 200.;
    return fdw_rand() % (int)(v + half - (centre - half) + 1) + centre - half;
}

void fdw_extendvel_linear(int nx, int nz, int nxb, int nzb, float *vel)
{
    const size_t nze = (size_t)nz + 2 * (size_t)nzb;
    const int x_first = nxb, x_last = nxb + nx - 1, z_first = nzb, z_last = nzb + nz - 1;
    const int x_end = nx + 2 * nxb - 1, z_end = nz + 2 * nzb - 1;
#define AT(i, j) vel[(size_t)(i)*nze + (size_t)(j)]
    /* 1. per interior column of x: top replica + bottom draws (x outer, depth inner) */
    for (int i = x_first; i <= x_last; i++)
        for (int d = 0; d < nzb; d++) {
            const float edge = AT(i, z_last);
            AT(i, d) = AT(i, z_first);
            AT(i, z_last + 1 + d) = draw_near(edge, ramp_to_floor(edge, d, nzb));
        }
    /* 2. per interior depth: one left draw then one right draw per distance (depth outer) */
    for (int j = z_first; j <= z_last; j++)
        for (int d = 0; d < nxb; d++) {
            float edge = AT(x_first, j);
            AT(x_first - 1 - d, j) = draw_near(edge, ramp_to_floor(edge, d, nxb));
            edge = AT(x_last, j);
            AT(x_last + 1 + d, j) = draw_near(edge, ramp_to_floor(edge, d, nxb));
        }
    /* 3. top corners: replicate the first / last interior column sideways */
    for (int j = 0; j < nzb; j++)
        for (int d = 0; d < nxb; d++) {
            AT(d, j) = AT(x_first, j);
            AT(x_last + 1 + d, j) = AT(x_last, j);
        }
    /* 4. bottom corners, left completely before right; two draws per (row, col<=row) visit */
    for (int right = 0; right <= 1; right++) {
        const float edge_x = right ? x_last : x_first;
        for (int m = 0; m < nzb; m++)
            for (int n = 0; n <= m; n++) {
                const float edge = AT((int)edge_x, z_last);
                const float centre = ramp_to_floor(edge, nxb - 1 - n, nzb);
                const int xa = right ? x_end - n : n, xb = right ? x_end - m : m;
                AT(xa, z_end - m) = draw_near(edge, centre);
                AT(xb, z_end - n) = draw_near(edge, centre);
            }
    }
#undef AT
}

/* ---- forward-modelling producer of the CPU-serial sibling (DD = dpct_gpu_rtm_domain_division/src) --------------------
 * DD builds its .c files with g++ (every Makefile there sets CC = g++ -fpermissive), so exp(float) is the float overload. */

/* DD boundary/taper.c:26-44: taper[i] = exp(-pow(F*(nb-i), 2)), the product in float, pow and exp in double */
void fdw_mod_taper_tables(int nxb, int nzb, float fac, float *taper_x, float *taper_z)
{
    for (int i = 0; taper_x && i < nxb; i++) taper_x[i] = exp(-pow((double)(fac * (nxb - i)), 2));
    for (int i = 0; taper_z && i < nzb; i++) taper_z[i] = exp(-pow((double)(fac * (nzb - i)), 2));
}

/* DD boundary/taper.c:7-23 */
void fdw_mod_extendvel(int nx, int nz, int nxb, int nzb, float *vel)
{
    const size_t nze = (size_t)nz + 2 * (size_t)nzb;
    for (int ix = nxb; ix < nxb + nx; ix++) {
        float *row = vel + (size_t)ix * nze;
        for (int iz = 0; iz < nzb; iz++) row[iz] = row[nzb];
        for (size_t iz = (size_t)nzb + nz; iz < nze; iz++) row[iz] = row[nz + nzb - 1];
    }
    for (size_t iz = 0; iz < nze; iz++) {
        for (int ix = 0; ix < nxb; ix++) vel[(size_t)ix * nze + iz] = vel[(size_t)nxb * nze + iz];
        for (int ix = nxb + nx; ix < nx + 2 * nxb; ix++) vel[(size_t)ix * nze + iz] = vel[(size_t)(nx + nxb - 1) * nze + iz];
    }
}

/* DD source/ptsrc.c:60-99 */
void fdw_mod_ricker_wavelet(int nt, float dt, float fpeak, float *s)
{
    for (int it = 0; it < nt; it++) {
        if (it * dt > 2.0 / fpeak) {
            s[it] = 0.0f;
        } else {
            const float t = it * dt - 1.0 / fpeak;
            const float x = FDW_PI * fpeak * t;
            const float xx = x * x;
            s[it] = expf(-xx) * (1.0 - 2.0 * xx);
        }
    }
}
