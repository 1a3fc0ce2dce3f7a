/* sanitize_host.c -- TEST INFRASTRUCTURE.  Drives the CPU-side C of this repository under AddressSanitizer + UndefinedBehaviorSanitizer
 * (`make -C oracle asan` links it with csrc/fdw_host.c, csrc/fdw_config.c, oracle/fdw_oracle.c, oracle/fdw_oracle_mod.c, all built with
 * -fsanitize=address,undefined -fno-sanitize-recover=all; tests/test_host_formulas.py::test_cpu_side_c_is_clean_under_the_sanitizers runs it).
 * The reference has no sanitizer configuration at all and checks no return value; this is the "ours" column of
 * SURVEY.md section 5.  GPU AddressSanitizer is not available on the pool, so the HIP side is covered by the parity tests instead.
 *
 *   sanitize_host <dir with decks> <scratch dir>
 *
 * What it walks through: the deck reader on every deck given (and on hostile ones it writes itself: no trailing newline, a 70 000
 * character line, 5 000 keys through the realloc path, empty file, keys without values), the host formulas for every order and several
 * border geometries including the degenerate ones (no border on one axis, border deeper than wide), and the oracle's forward / backward
 * loops, slab steps, the sibling's modelling and stored-wavefield loops and the image tools on small ragged grids in both numerics modes.
 * Any finding aborts with a non-zero status; on success it prints one line and exits 0. */
#include <dirent.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "fdw_config.h"
#include "fdwave.h"

/* oracle/fdw_oracle.c, oracle/fdw_oracle_mod.c */
typedef struct orc_state orc_state;
orc_state *orc_init(int order, int nxe, int nze, int nxb, int nzb, int nt, float fac, float dx, float dz, float dt, int compat);
void orc_set_numerics(orc_state *s, int numerics);
void orc_free(orc_state *s);
void orc_fd_forward(orc_state *s, float *p, float *pp, const float *v2, int sx, int sz, const float *srce, int nsteps);
void orc_fd_back(orc_state *s, const float *v2, const float *snap0, const float *snap1, const float *d_obs, int gz_, float *imloc, int nsteps);
void orc_stencil(int order, int nxe, int nze, float dx, float dz, const float *in, float *out);
void orc_stencil_fast(int order, int nxe, int nze, float dx, float dz, const float *in, float *out);
void orc_slab_step(const orc_state *s, int x_off, int nxl, float *p, float *pp, const float *v2, int r0, int r1, int t0, int t1, int sx_global, int sz, float srce_it);
void orc_slab_back_iter(const orc_state *s, int x_off, int nxl, int step_source, float *f1, float *f0, float *pr, float *ppr, const float *v2, int r0, int r1, int t0,
                        int t1, const float *samples, int gz_, float *img);
void orc_calc_coefs(int order, int cxx, float *coef);
void orc_ricker_wavelet(int nt, float dt, float peak, float *s);
void orc_taper_tables(int nxb, int nzb, float fac, float *taper_x, float *taper_z);
void orc_extendvel_linear(int nx, int nz, int nxb, int nzb, float *vel);
void orc_srand(unsigned seed);
void orc_fused_steps(int order, int nxe, int nze, float *p, float *pp, const float *v2, const float *cx, const float *cz, float dt2, int nsteps);
void orc_scaled_coefs(int order, float dx, float dz, int cxx, float *coefs_x, float *coefs_z);
void orc_mod_shot(int order, int nx, int nz, int nxb, int nzb, int nt, float dx, float dz, float dt, float fac, const float *v2, int sx, int sz, int gz,
                  const float *srce, float *data);
void orc_rtm_stored_shot(int order, int nx, int nz, int nxb, int nzb, int nt, float dx, float dz, float dt, float fac, const float *v2, int sx, int sz, int gz,
                         const float *srce, const float *dobs_flat, size_t n_flat, int is, float *imloc);
void orc_mod_ricker_wavelet(int nt, float dt, float peak, float *s);
void orc_mod_set_numerics(int numerics);
void orc_mod_extendvel(int nx, int nz, int nxb, int nzb, float *vel);
void orc_image_laplacian(const float *img, int nx, int nz, float dx, float dz, float *out);
void orc_image_compare(const float *f1, const float *f2, size_t n, float *diff, double *stats);

static unsigned long long rng_state = 0x5EED0001ull;
static float frand(void)
{   /* splitmix64 -> [0, 1) */
    unsigned long long z = (rng_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (float)((z >> 40) * (1.0 / 16777216.0));
}
static float *randf(size_t n, float lo, float hi)
{
    float *a = (float *)malloc((n ? n : 1) * sizeof(float));
    if (!a) abort();
    for (size_t i = 0; i < n; i++) a[i] = lo + (hi - lo) * frand();
    return a;
}
static double checksum = 0.0;
static void eat(const float *a, size_t n)
{   /* every output is read once: an uninitialised or out-of-bounds result cannot go unnoticed by the sanitizers */
    for (size_t i = 0; i < n; i++) checksum += a[i];
}

static void walk_deck(const char *path)
{
    fdw_deck *d = fdw_deck_read(path);
    if (!d) return;
    static const char *keys[] = {"tmpdir", "vpfile", "datfile", "vel_ext_file", "nz", "nx", "nt", "dz", "dx", "dt", "fpeak", "ns", "iss", "sz", "fsx", "ds", "gz",
                                 "nxb", "nzb", "rnd", "fac", "order", "slabs", "numerics", "no_such_key", ""};
    for (size_t k = 0; k < sizeof keys / sizeof *keys; k++) {
        const char *s = fdw_deck_str(d, keys[k]);
        checksum += fdw_deck_int(d, keys[k]) + fdw_deck_float(d, keys[k]) + (s ? (double)strlen(s) : 0.0) + fdw_deck_has(d, keys[k]);
    }
    fdw_deck_free(d);
}

static void decks(const char *dir, const char *scratch)
{
    char path[4096];
    DIR *dp = opendir(dir);
    if (dp) {
        for (struct dirent *e; (e = readdir(dp));) {
            if (e->d_name[0] == '.') continue;
            snprintf(path, sizeof path, "%s/%s", dir, e->d_name);
            walk_deck(path);
        }
        closedir(dp);
    }
    FILE *f;
    snprintf(path, sizeof path, "%s/hostile1.dat", scratch);
    if ((f = fopen(path, "w"))) {      /* no trailing newline, CRLF, blanks, comments, '=' in a value, keys without values, values without keys */
        fputs("# comment only\r\n  nz = 12 \r\nnzb=3\nnx=7 # trailing\nname=./a b/c=d.bin\nnz=99\n\nbad line\n=5\nempty=\n   \nlast=1", f);
        fclose(f);
        walk_deck(path);
    }
    snprintf(path, sizeof path, "%s/hostile2.dat", scratch);
    if ((f = fopen(path, "w"))) {      /* one 70 000 character line, then 5 000 keys: getline's growth and the table's realloc path */
        fputs("long=", f);
        for (int i = 0; i < 70000; i++) fputc('a' + i % 26, f);
        fputc('\n', f);
        for (int i = 0; i < 5000; i++) fprintf(f, "key%d = %d\n", i, i);
        fputs("nz=5\n", f);
        fclose(f);
        walk_deck(path);
    }
    snprintf(path, sizeof path, "%s/empty.dat", scratch);
    if ((f = fopen(path, "w"))) {
        fclose(f);
        walk_deck(path);
    }
    fdw_deck_free(NULL);
    if (fdw_deck_read("/nonexistent/input.dat")) abort();
}

static void host_formulas(void)
{
    float w[FDW_MAX_ORDER + 1], o[65];
    for (int order = 2; order <= FDW_MAX_ORDER; order += 2)
        for (int cxx = 0; cxx < 2; cxx++) {
            if (fdw_calc_coefs(order, cxx, w) != 0) abort();
            orc_calc_coefs(order, cxx, o);
            eat(w, (size_t)order + 1);
            if (memcmp(w, o, ((size_t)order + 1) * sizeof(float)) != 0) abort();      /* product == oracle, as the CPU tests hold elsewhere */
        }
    if (fdw_calc_coefs(3, 0, w) == 0 || fdw_calc_coefs(0, 0, w) == 0 || fdw_calc_coefs(FDW_MAX_ORDER + 2, 0, w) == 0) abort();
    static const int nts[] = {1, 2, 64, 1700};
    for (size_t i = 0; i < sizeof nts / sizeof *nts; i++) {
        float *s = randf((size_t)nts[i], 0, 0), *t = randf((size_t)nts[i], 0, 0);
        fdw_ricker_wavelet(nts[i], 0.001f, 20.0f, s);
        orc_ricker_wavelet(nts[i], 0.001f, 20.0f, t);
        if (memcmp(s, t, (size_t)nts[i] * sizeof(float)) != 0) abort();
        fdw_mod_ricker_wavelet(nts[i], 0.001f, 30.0f, s);
        orc_mod_ricker_wavelet(nts[i], 0.001f, 30.0f, t);
        if (memcmp(s, t, (size_t)nts[i] * sizeof(float)) != 0) abort();
        eat(s, (size_t)nts[i]);
        free(s);
        free(t);
    }
    static const int nb[][2] = {{50, 50}, {40, 8}, {8, 40}, {1, 1}, {17, 13}};
    for (size_t i = 0; i < sizeof nb / sizeof *nb; i++) {
        float *tx = randf((size_t)nb[i][0], 0, 0), *tz = randf((size_t)nb[i][1], 0, 0);
        fdw_taper_tables(nb[i][0], nb[i][1], 0.75f, tx, tz);
        eat(tx, (size_t)nb[i][0]);
        eat(tz, (size_t)nb[i][1]);
        fdw_taper_tables(nb[i][0], 0, 0.75f, tx, NULL);
        fdw_taper_tables(0, nb[i][1], 0.01f, NULL, tz);
        fdw_mod_taper_tables(nb[i][0], nb[i][1], 0.01f, tx, tz);
        eat(tx, (size_t)nb[i][0]);
        free(tx);
        free(tz);
    }
    /* extendvel_linear: interior nx x nz with borders of every shape (wider than deep, deeper than wide, absent on one axis) */
    static const int geo[][4] = {{24, 20, 6, 5}, {31, 17, 3, 9}, {12, 40, 10, 2}, {20, 20, 0, 7}, {20, 20, 7, 0}, {9, 9, 2, 2}};
    for (size_t g = 0; g < sizeof geo / sizeof *geo; g++) {
        const int nx = geo[g][0], nz = geo[g][1], nxb = geo[g][2], nzb = geo[g][3], nxe = nx + 2 * nxb, nze = nz + 2 * nzb;
        float *a = (float *)calloc((size_t)nxe * nze, sizeof(float)), *b = (float *)calloc((size_t)nxe * nze, sizeof(float));
        if (!a || !b) abort();
        for (int i = 0; i < nx; i++)
            for (int j = 0; j < nz; j++) a[(size_t)(i + nxb) * nze + j + nzb] = b[(size_t)(i + nxb) * nze + j + nzb] = 1500.0f + 2500.0f * frand();
        for (int shot = 0; shot < 3; shot++) {
            if (shot == 0) {
                fdw_srand(1);
                orc_srand(1);
            }
            fdw_extendvel_linear(nx, nz, nxb, nzb, a);
            orc_extendvel_linear(nx, nz, nxb, nzb, b);
            if (memcmp(a, b, (size_t)nxe * nze * sizeof(float)) != 0) abort();      /* the private generator IS glibc's rand() */
        }
        eat(a, (size_t)nxe * nze);
        fdw_mod_extendvel(nx, nz, nxb, nzb, a);
        orc_mod_extendvel(nx, nz, nxb, nzb, b);
        eat(a, (size_t)nxe * nze);
        free(a);
        free(b);
    }
}

static void oracle_loops(void)
{
    /* ragged grids: sizes that are no multiple of 8 (truncated extents), unequal borders, the smallest grid an order admits */
    static const int cases[][6] = {{99, 83, 17, 13, 14, 8}, {41, 52, 8, 9, 9, 4}, {33, 35, 0, 0, 6, 6}, {21, 19, 2, 2, 5, 2}, {40, 37, 5, 7, 7, 12}};
    for (size_t c = 0; c < sizeof cases / sizeof *cases; c++)
        for (int compat = 0; compat < 2; compat++)
            for (int numerics = 0; numerics < 2; numerics++) {
                const int nxe = cases[c][0], nze = cases[c][1], nxb = cases[c][2], nzb = cases[c][3], nt = cases[c][4], order = cases[c][5];
                const int nx = nxe - 2 * nxb, nz = nze - 2 * nzb;
                const size_t ne = (size_t)nxe * nze;
                orc_state *s = orc_init(order, nxe, nze, nxb, nzb, nt, 0.75f, 10.0f, 12.5f, 0.001f, compat);
                if (!s) abort();
                orc_set_numerics(s, numerics);
                float *v2 = randf(ne, 1500.0f * 1500.0f, 3000.0f * 3000.0f), *srce = randf((size_t)nt, -1, 1);
                float *P = (float *)calloc(ne, sizeof(float)), *PP = (float *)calloc(ne, sizeof(float));
                float *dobs = randf((size_t)nx * nt, -1, 1), *img = (float *)calloc((size_t)nx * nz + 1, sizeof(float));
                if (!P || !PP || !img) abort();
                const int sx = nxb + nx / 2, sz = nzb + 1, gz = nzb;
                orc_fd_forward(s, P, PP, v2, sx, sz, srce, nt);
                orc_fd_back(s, v2, P, PP, dobs, gz, img, nt);
                eat(P, ne);
                eat(PP, ne);
                eat(img, (size_t)nx * nz);
                /* the per-slab restatements on two bands with a 2 h k ghost band */
                const int h = order / 2, half = (nxe / 2 / 4) * 4, G = h;
                if (half - G >= h + 1 && nxe - half - G >= h + 1) {
                    for (int r = 0; r < 2; r++) {
                        const int x_off = r == 0 ? 0 : half - G, nxl = r == 0 ? half + G : nxe - half + G;
                        float *p = (float *)calloc((size_t)nxl * nze, sizeof(float)), *pp = (float *)calloc((size_t)nxl * nze, sizeof(float));
                        float *pr = (float *)calloc((size_t)nxl * nze, sizeof(float)), *ppr = (float *)calloc((size_t)nxl * nze, sizeof(float));
                        float *im = (float *)calloc((size_t)nxl * nze, sizeof(float)), *smp = randf((size_t)nx + 1, -1, 1);
                        if (!p || !pp || !pr || !ppr || !im) abort();
                        const int r0 = r == 0 ? 0 : G, r1 = r == 0 ? nxl - G : nxl;
                        orc_slab_step(s, x_off, nxl, p, pp, v2 + (size_t)x_off * nze, r0, r1, 0, nxl, sx, sz, 1.0f);
                        orc_slab_back_iter(s, x_off, nxl, 1, p, pp, pr, ppr, v2 + (size_t)x_off * nze, r0, r1, 0, nxl, smp, gz, im);
                        orc_slab_back_iter(s, x_off, nxl, 0, p, pp, pr, ppr, v2 + (size_t)x_off * nze, r0, r1, 0, nxl, smp, gz, im);
                        eat(pp, (size_t)nxl * nze);
                        eat(im, (size_t)nxl * nze);
                        free(p); free(pp); free(pr); free(ppr); free(im); free(smp);
                    }
                }
                float *lap = (float *)calloc(ne, sizeof(float));
                if (!lap) abort();
                (numerics ? orc_stencil_fast : orc_stencil)(order, nxe, nze, 10.0f, 12.5f, v2, lap);
                eat(lap, ne);
                float cx[65], cz[65];
                orc_scaled_coefs(order, 10.0f, 12.5f, 0, cx, cz);
                orc_fused_steps(order, nxe, nze, P, PP, v2, cx, cz, 1e-6f, 3);
                eat(PP, ne);
                free(lap); free(v2); free(srce); free(P); free(PP); free(dobs); free(img);
                orc_free(s);
            }
    /* the sibling's loops (orders <= 8) and the image tools */
    {
        const int nx = 23, nz = 19, nxb = 6, nzb = 5, nt = 12, nxe = nx + 2 * nxb, nze = nz + 2 * nzb;
        float *v2 = randf((size_t)nxe * nze, 2.0e6f, 9.0e6f), *srce = randf((size_t)nt, -1, 1), *data = (float *)calloc((size_t)2 * nx * nt, sizeof(float));
        float *img = (float *)calloc((size_t)nx * nz, sizeof(float)), *out = (float *)calloc((size_t)nx * nz, sizeof(float));
        if (!data || !img || !out) abort();
        for (int order = 2; order <= 8; order += 2) {
            orc_mod_set_numerics((order / 2) & 1);      /* both numerics modes of these loops */
            orc_mod_shot(order, nx, nz, nxb, nzb, nt, 10.0f, 8.0f, 0.001f, 0.01f, v2, nxb + 3, nzb + 2, nzb + 1, srce, data);
            eat(data, (size_t)nx * nt);
            /* the gather as rtm_main indexes it: one sample past the last trace of the last shot reads as 0 (n_flat bounds it) */
            orc_rtm_stored_shot(order, nx, nz, nxb, nzb, nt, 10.0f, 8.0f, 0.001f, 0.01f, v2, nxb + 3, nzb + 2, nzb + 1, srce, data, (size_t)nx * nt, 0, img);
            eat(img, (size_t)nx * nz);
        }
        orc_mod_set_numerics(0);
        orc_image_laplacian(img, nx, nz, 10.0f, 8.0f, out);
        eat(out, (size_t)nx * nz);
        double st[4];
        orc_image_compare(img, out, (size_t)nx * nz, data, st);
        orc_image_compare(img, img, (size_t)nx * nz, NULL, st);      /* identical images: division by zero inside the tool's formulas (inf / nan, as the ELF prints) */
        free(v2); free(srce); free(data); free(img); free(out);
    }
}

int main(int argc, char **argv)
{
    if (argc < 3) {
        fprintf(stderr, "usage: %s <dir with decks> <scratch dir>\n", argv[0]);
        return 2;
    }
    decks(argv[1], argv[2]);
    host_formulas();
    oracle_loops();
    printf("sanitize_host: clean (checksum %.6e)\n", checksum);
    return 0;
}
