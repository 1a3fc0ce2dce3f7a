/* rtm_code -- drop-in for cuda_reference_RTM/rtm_code (src/fd-code.cu = R):
 *     ./rtm_code ./models/<model>/input.dat
 * Same deck keys and defaults (R:343-378), same input binaries (vpfile [nx][nz], datfile
 * [ns][nx][nt], optional vel_ext_file [ns][nxe][nze]), same outputs: <tmpdir>/dir.image (stacked image
 * [nx][nz]), <tmpdir>/dir.image_lap (zeros, R:477,542), empty dir.snaps / dir.snaps_rec / dir.snapr
 * (R:465-470), ./image.num text dump (R:522-528), and the stdout banners.  The per-shot propagation
 * (fd_forward + fd_back, R:499-518) is one device-resident fdw_shot() call; launch extents are the
 * reference's (compat = 1), so the image equals the reference's.  Shots run side by side on up to FDW_SHOT_WORKERS (default 4) host
 * threads / streams; models are drawn and images stacked in shot order, so every output is what the serial loop writes.
 * Not reproduced: the `file-teste` debug dump at it == 750 (R:268-281) and the in-loop progress lines.
 *
 * Several GPUs (no counterpart in the reference, which drives one): the deck key `slabs=N` (or FDW_SLABS=N in the environment) runs every
 * shot on N GPUs, the grid cut into N bands of rows with halo exchange over RCCL / xGMI inside libfdwave.so (fdw_slabs_shot), one host
 * thread per GPU; FDW_SLABS_LOCAL=1 keeps all N ranks on GPU 0 with device copies instead of RCCL (tests on a one-GPU box).  `gpus=N` (or
 * FDW_GPUS=N) deals whole shots to N GPUs instead.  Every output file is byte for byte the one-GPU program's. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>
#include <sys/time.h>

#include "fdw_config.h"
#include "fdwave.h"

static float *read_floats(const char *path, size_t n, const char *what)
{
    FILE *f = path ? fopen(path, "rb") : NULL;
    if (!f) {
        fprintf(stderr, "cannot open %s '%s'\n", what, path ? path : "(null)");
        return NULL;
    }
    float *a = (float *)calloc(n ? n : 1, sizeof(float)); /* R:414,421,438 memset to 0 then fread */
    const size_t got = a ? fread(a, sizeof(float), n, f) : 0;
    fclose(f);
    if (got != n) fprintf(stderr, "warning: %s '%s' holds %zu of %zu floats (rest stays zero)\n", what, path, got, n);
    return a;
}

static FILE *open_out(const char *dir, const char *name)
{
    char path[4096];
    snprintf(path, sizeof path, "%s/%s", dir, name);
    FILE *f = fopen(path, "w");
    if (!f) fprintf(stderr, "cannot create '%s'\n", path);
    return f;
}

/* one batch of shots shared out to host threads: worker w takes shots w, w + nw, ... of the batch */
typedef struct {
    const fdw_params *prm;
    int ns, nworkers, is0, nb, nx, nt, sz, gz;
    const int *sx;
    const float *srce, *d_obs, *vel2_all;
    const float *vp;      /* dev_border: the interior model every worker keeps resident in HBM */
    long long draws;      /* rand() calls one extendvel_linear consumes */
    int dev_border;
    float *imloc_all;
    size_t ne, ni;
    int gpus;             /* workers are dealt to GPUs 0 .. gpus-1 */
    volatile int failed;
} shot_job;
typedef struct {
    shot_job *job;
    int w, nw;
} shot_worker_arg;

static void *shot_worker(void *p)
{
    shot_worker_arg *a = (shot_worker_arg *)p;
    shot_job *j = a->job;
    fdw_ctx *ctx = NULL;
    if (fdw_create(j->prm, a->w % (j->gpus > 0 ? j->gpus : 1), &ctx) != FDW_OK) { /* fd_init, R:452 (one context = one stream + its own device buffers per worker) */
        fprintf(stderr, "fdw_create: %s\n", fdw_last_error());
        j->failed = 1;
        return NULL;
    }
    if (j->dev_border && fdw_model_resident(ctx, j->vp) != FDW_OK) {
        fprintf(stderr, "fdw_model_resident: %s\n", fdw_last_error());
        j->failed = 1;
    }
    for (int b = a->w; b < j->nb && !j->failed; b += a->nw) {
        const int is = j->is0 + b;
        const float *d_obs = j->d_obs + (size_t)is * j->nx * j->nt;
        float *imloc = j->imloc_all + (size_t)b * j->ni;
        int rc;
        if (j->dev_border) {
            /* R:486-494 in HBM: shot `is` of the serial program consumes draws [is T, (is + 1) T) of the unseeded rand() stream */
            rc = fdw_dev_extendvel_linear(ctx, (unsigned long long)is * (unsigned long long)j->draws, NULL);
            if (rc == FDW_OK) rc = fdw_shot_resident(ctx, j->sx[is], j->sz, j->gz, j->srce, d_obs, imloc, NULL, NULL);
        } else {
            rc = fdw_shot(ctx, j->vel2_all + (size_t)b * j->ne, j->sx[is], j->sz, j->gz, j->srce, d_obs, imloc, NULL, NULL);
        }
        if (rc != FDW_OK) {
            fprintf(stderr, "fdw_shot: %s\n", fdw_last_error());
            j->failed = 1;
        }
    }
    fdw_destroy(ctx);
    return NULL;
}

/* ---- slabs=N: every shot decomposed over N ranks (host threads, one GPU each) ---------------------------------------------- */
typedef struct {
    const fdw_params *prm;
    int world, local, ns, nx, nt, sz, gz;
    const int *sx;
    const float *srce, *d_obs;
    const float *v2;         /* the current shot's squared model [nxe][nze], built by rank 0 between the barriers */
    float *imloc;            /* the current shot's image [nx][nz]: every rank writes its own rows */
    char uid[FDW_COMM_ID_BYTES];
    fdw_comm **local_comms;
    pthread_barrier_t *bar;
    volatile int failed;
} slab_job;
typedef struct {
    slab_job *job;
    int rank;
    fdw_comm *comm;
    fdw_slabs *slabs;
} slab_rank;

/* Opening a rank is collective from ncclCommInitRank on (RCCL has no timeout: a rank that never arrives leaves the others waiting for
 * ever, holding their GPUs), so it goes in two phases with a barrier in between: first everything a rank can check ALONE -- its device
 * exists, is a gfx950 and can be selected --, then, only if no rank failed, the collective calls.  Every rank (the main thread included)
 * calls this exactly once and reaches both barriers. */
static int slab_rank_open(slab_rank *r)
{
    slab_job *j = r->job;
    int rc = j->local ? FDW_OK : fdw_device_usable(r->rank);                         /* rank r drives GPU r */
    if (rc != FDW_OK) {
        fprintf(stderr, "rank %d: %s\n", r->rank, fdw_last_error());
        j->failed = 1;
    }
    pthread_barrier_wait(j->bar);       /* O1: every rank has checked its device */
    if (!j->failed) {
        if (j->local) r->comm = j->local_comms[r->rank];
        else rc = fdw_comm_init_rank(j->uid, r->rank, j->world, r->rank, &r->comm);
        if (rc == FDW_OK) rc = fdw_slabs_create(j->prm, r->comm, 0, 0, &r->slabs);
        if (rc != FDW_OK) {
            fprintf(stderr, "rank %d: %s\n", r->rank, fdw_last_error());
            j->failed = 1;
        }
    }
    pthread_barrier_wait(j->bar);       /* O2: every rank is open (or the job has failed) */
    return j->failed ? FDW_ECOMM : FDW_OK;
}
static void slab_rank_shot(slab_rank *r, int is)
{
    slab_job *j = r->job;
    if (j->failed || !r->slabs) return;
    if (fdw_slabs_shot(r->slabs, j->v2, j->sx[is], j->sz, j->gz, j->srce, j->d_obs + (size_t)is * j->nx * j->nt, j->imloc, NULL, NULL) != FDW_OK) {
        fprintf(stderr, "rank %d, shot %d: %s\n", r->rank, is, fdw_last_error());
        j->failed = 1;
    }
}
static void *slab_rank_thread(void *p)      /* ranks 1 .. N-1; rank 0 is the main thread */
{
    slab_rank *r = (slab_rank *)p;
    slab_job *j = r->job;
    slab_rank_open(r);
    for (int is = 0; is < j->ns; is++) {
        pthread_barrier_wait(j->bar);       /* A: the shot's model is ready */
        slab_rank_shot(r, is);
        pthread_barrier_wait(j->bar);       /* B: every rank's rows of the image are in place */
    }
    if (r->slabs) fdw_slabs_destroy(r->slabs);
    if (r->comm) fdw_comm_destroy(r->comm);
    return NULL;
}

static double now_s(void)
{
    struct timeval t;
    gettimeofday(&t, NULL);
    return t.tv_sec + t.tv_usec * 1e-6;
}

int main(int argc, char **argv)
{
    struct timeval start, end;
    gettimeofday(&start, NULL);
    const int timing = getenv("FDW_TIMING") != NULL;   /* phase times on stderr */
    double t_shots = 0.0, t_stack = 0.0;
    const double t_begin = now_s();
    if (argc < 2) {
        fprintf(stderr, "usage: %s <input.dat>\n", argv[0]);
        return EXIT_FAILURE;
    }
    fdw_deck *deck = fdw_deck_read(argv[1]);
    if (!deck) return EXIT_FAILURE; /* F:51-52 */

    /* init_args, R:343-378 */
    const char *tmpdir = fdw_deck_str(deck, "tmpdir"), *vpfile = fdw_deck_str(deck, "vpfile");
    const char *datfile = fdw_deck_str(deck, "datfile"), *vel_ext_file = fdw_deck_str(deck, "vel_ext_file");
    const int nz = fdw_deck_int(deck, "nz"), nx = fdw_deck_int(deck, "nx"), nt = fdw_deck_int(deck, "nt");
    int ns = fdw_deck_int(deck, "ns"), sz = fdw_deck_int(deck, "sz"), fsx = fdw_deck_int(deck, "fsx");
    int ds = fdw_deck_int(deck, "ds"), gz = fdw_deck_int(deck, "gz"), order = fdw_deck_int(deck, "order");
    int nzb = fdw_deck_int(deck, "nzb"), nxb = fdw_deck_int(deck, "nxb"), iss = fdw_deck_int(deck, "iss");
    const int rnd = fdw_deck_int(deck, "rnd");
    const float dz = fdw_deck_float(deck, "dz"), dx = fdw_deck_float(deck, "dx"), dt = fdw_deck_float(deck, "dt");
    const float fpeak = fdw_deck_float(deck, "fpeak");
    float fac = fdw_deck_float(deck, "fac");
    const int vel_ext_flag = vel_ext_file != NULL;
    if (iss == -1) iss = 0;
    if (ns == -1) ns = 1;
    if (sz == -1) sz = 0;
    if (fsx == -1) fsx = 0;
    if (ds == -1) ds = 1;
    if (gz == -1) gz = 0;
    if (order == -1) order = 8;
    if (nzb == -1) nzb = 40;
    if (nxb == -1) nxb = 40;
    if (fac == -1.0f) fac = 0.7f;
    (void)iss;

    printf("## vp = %s, d_obs = %s, vel_ext_file = %s, vel_ext_flag = %d \n", vpfile, datfile, vel_ext_file, vel_ext_flag);
    printf("## nz = %d, nx = %d, nt = %d \n", nz, nx, nt);
    printf("## dz = %f, dx = %f, dt = %f \n", dz, dx, dt);
    printf("## ns = %d, sz = %d, fsx = %d, ds = %d, gz = %d \n", ns, sz, fsx, ds, gz);
    printf("## order = %d, nzb = %d, nxb = %d, F = %f, rnd = %d \n", order, nzb, nxb, fac, rnd);
    if (nz <= 0 || nx <= 0 || nt <= 0 || !tmpdir || !vpfile || !datfile) {
        fprintf(stderr, "input deck is missing one of tmpdir/vpfile/datfile/nz/nx/nt\n");
        return EXIT_FAILURE;
    }

    /* R:402-411 */
    float *srce = (float *)malloc((size_t)nt * sizeof(float));
    fdw_ricker_wavelet(nt, dt, fpeak, srce);
    int *sx = (int *)malloc((size_t)ns * sizeof(int));
    for (int is = 0; is < ns; is++) sx[is] = fsx + is * ds + nxb;
    sz += nzb;
    gz += nzb;
    const int nze = nz + 2 * nzb, nxe = nx + 2 * nxb;
    const size_t ne = (size_t)nxe * nze, ni = (size_t)nx * nz;

    float *vel_ext_rnd = NULL;
    if (vel_ext_flag && !(vel_ext_rnd = read_floats(vel_ext_file, ne * ns, "vel_ext_file"))) return EXIT_FAILURE; /* R:412-418 */
    float *d_obs = read_floats(datfile, (size_t)ns * nx * nt, "datfile");                                        /* R:420-424 */
    float *vp = read_floats(vpfile, ni, "vpfile");                                                               /* R:437-441 */
    if (!d_obs || !vp) return EXIT_FAILURE;
    float *vpe = (float *)calloc(ne, sizeof(float)); /* the reference leaves the border uninitialised (malloc) until extendvel */
    for (int ix = 0; ix < nx; ix++)
        for (int iz = 0; iz < nz; iz++) vpe[(size_t)(ix + nxb) * nze + iz + nzb] = vp[(size_t)ix * nz + iz]; /* R:445-449 */

    fdw_params prm;
    memset(&prm, 0, sizeof prm);
    prm.order = order; prm.nxe = nxe; prm.nze = nze; prm.nxb = nxb; prm.nzb = nzb; prm.nt = nt;
    prm.dx = dx; prm.dz = dz; prm.dt = dt; prm.fac = fac;
    prm.compat = 1;   /* the reference's launch extents, R:185-195 */
    prm.coef_cxx = 0; /* libsource.a is C, F:160-192 */
    {                 /* our extension, absent = the reference's arithmetic: numerics=1 (or FDW_NUMERICS=1) selects FAST numerics (fdwave.h) */
        int numerics = fdw_deck_int(deck, "numerics");
        if (getenv("FDW_NUMERICS")) numerics = atoi(getenv("FDW_NUMERICS"));
        prm.numerics = numerics == 1 ? FDW_NUMERICS_FAST : FDW_NUMERICS_EXACT;
        if (prm.numerics) printf("## numerics = FAST (symmetric sums + fused multiply-adds in the Laplacian; within 1e-5 of the reference's arithmetic)\n");
    }

    float *img = (float *)calloc(ni, sizeof(float)), *img_lap = (float *)calloc(ni, sizeof(float));
    FILE *fsns = open_out(tmpdir, "dir.snaps"), *fsns2 = open_out(tmpdir, "dir.snaps_rec"), *fsnr = open_out(tmpdir, "dir.snapr");
    FILE *fimg = open_out(tmpdir, "dir.image"), *fimg_lap = open_out(tmpdir, "dir.image_lap"); /* R:464-474 */
    FILE *fnum = fopen("image.num", "w");                                                       /* R:478-479 */
    if (!fimg || !fimg_lap || !fnum) return EXIT_FAILURE;

    /* Shots are independent (R:480-529 only couples them through the running image sum), and a shot of a deck this size fills a
     * few percent of an MI355X: up to FDW_SHOT_WORKERS (default 4) host threads, each with its own context and stream, propagate
     * shots side by side.  What must stay serial does: the border model draws from ONE rand() stream in shot order (R:486), so all
     * squared-velocity models are built first, and the images are stacked (and image.num written) in shot order afterwards. */
    /* The border model itself is generated on the device from the resident interior model (fdw_dev_extendvel_linear: the rand() stream is
     * addressed by position, so no worker waits for another's draws); FDW_HOST_BORDER=1 or a geometry the device path refuses (a one-cell
     * border) keeps the host loop below. */
    int slabs = fdw_deck_int(deck, "slabs"), gpus = fdw_deck_int(deck, "gpus");      /* our extensions; absent = -1 */
    if (getenv("FDW_SLABS")) slabs = atoi(getenv("FDW_SLABS"));
    if (getenv("FDW_GPUS")) gpus = atoi(getenv("FDW_GPUS"));
    if (slabs > 64 || gpus > 64) {
        fprintf(stderr, "slabs / gpus: at most 64\n");
        return EXIT_FAILURE;
    }
    if (slabs > 1) {
        /* ---- every shot on `slabs` GPUs: bands of rows, halo exchange inside the library ---- */
        slab_job sj;
        memset(&sj, 0, sizeof sj);
        sj.local = getenv("FDW_SLABS_LOCAL") != NULL;
        const int ndev = fdw_device_count();
        if (ndev < 1 || (!sj.local && slabs > ndev)) {      /* before any thread or communicator exists: RCCL would wait for the missing ranks for ever */
            fprintf(stderr, "slabs=%d needs %d GPUs, one rank each; %d visible%s\n", slabs, slabs, ndev < 0 ? 0 : ndev,
                    ndev >= 1 ? " (FDW_SLABS_LOCAL=1 runs the ranks as threads sharing GPU 0: rehearsals only)" : "");
            return EXIT_FAILURE;
        }
        sj.prm = &prm; sj.world = slabs; sj.ns = ns; sj.nx = nx; sj.nt = nt; sj.sz = sz; sj.gz = gz;
        sj.sx = sx; sj.srce = srce; sj.d_obs = d_obs;
        float *v2 = (float *)malloc(ne * sizeof(float)), *imloc = (float *)calloc(ni, sizeof(float));
        fdw_comm *lc[64];
        pthread_barrier_t bar;
        pthread_t th[64];
        slab_rank rk[64];
        if (!v2 || !imloc || pthread_barrier_init(&bar, NULL, (unsigned)slabs) != 0) return EXIT_FAILURE;
        sj.imloc = imloc; sj.v2 = v2; sj.bar = &bar; sj.local_comms = lc;
        if ((sj.local ? fdw_comm_init_local(slabs, NULL, lc) : fdw_comm_get_unique_id(sj.uid)) != FDW_OK) {
            fprintf(stderr, "communicator: %s\n", fdw_last_error());
            return EXIT_FAILURE;
        }
        for (int r = 0; r < slabs; r++) {
            memset(&rk[r], 0, sizeof rk[r]);
            rk[r].job = &sj; rk[r].rank = r;
            if (r > 0 && pthread_create(&th[r], NULL, slab_rank_thread, &rk[r]) != 0) return EXIT_FAILURE;
        }
        slab_rank_open(&rk[0]);
        for (int is = 0; is < ns; is++) {
            const float *v = vpe;
            if (vel_ext_flag) v = vel_ext_rnd + (size_t)is * ne;      /* R:484 */
            else fdw_extendvel_linear(nx, nz, nxb, nzb, vpe);         /* R:486 */
            for (size_t k = 0; k < ne; k++) v2[k] = v[k] * v[k];      /* R:490-494 */
            memset(imloc, 0, ni * sizeof(float));                     /* R:515 */
            pthread_barrier_wait(&bar);
            slab_rank_shot(&rk[0], is);
            pthread_barrier_wait(&bar);
            fprintf(stdout, "** source %d, at (%d,%d) \n\n** backward propagation %d, at (%d,%d) \n\n", is + 1, sx[is] - nxb, sz - nzb, is + 1, sx[is] - nxb, sz - nzb);
            fprintf(fnum, "======== %i ========\n", is);
            for (int iz = 0; iz < nz; iz++)
                for (int ix = 0; ix < nx; ix++) {
                    img[(size_t)ix * nz + iz] += imloc[(size_t)ix * nz + iz];
                    fprintf(fnum, " %f \n", img[(size_t)ix * nz + iz]);
                }
        }
        for (int r = 1; r < slabs; r++) pthread_join(th[r], NULL);
        if (rk[0].slabs) fdw_slabs_destroy(rk[0].slabs);
        if (rk[0].comm) fdw_comm_destroy(rk[0].comm);
        pthread_barrier_destroy(&bar);
        free(v2); free(imloc);
        if (sj.failed) return EXIT_FAILURE;
        goto outputs;
    }
    const int dev_border = !vel_ext_flag && !getenv("FDW_HOST_BORDER") && nxb != 1 && nzb != 1 && nzb <= nxe;
    int nworkers = 4;
    if (getenv("FDW_SHOT_WORKERS")) nworkers = atoi(getenv("FDW_SHOT_WORKERS"));
    if (nworkers < 1) nworkers = 1;
    if (gpus > 1 && nworkers < gpus) nworkers = gpus;      /* at least one worker per GPU */
    if (nworkers > ns) nworkers = ns;
    if (nworkers > 64) nworkers = 64;
    while (nworkers > 1 && (size_t)ns * (ne + ni) * sizeof(float) > ((size_t)8 << 30)) nworkers = 1;   /* big decks: one shot fills the GPU anyway */
    /* Small decks: a whole batch of shots advances through ONE launch per time step (fdw_shot_batch; the library says how many shots fill
     * the chip for this geometry, 1 = the grid is big enough by itself).  FDW_NO_SHOT_BATCH=1 keeps one shot per launch sequence. */
    fdw_ctx *bctx = NULL;
    int bmax = 1;
    if (ns > 1 && !getenv("FDW_NO_SHOT_BATCH") && gpus <= 1) {      /* (shots dealt to several GPUs go one context per worker instead) */
        if (fdw_create(&prm, 0, &bctx) != FDW_OK) {
            fprintf(stderr, "fdw_create: %s\n", fdw_last_error());
            return EXIT_FAILURE;
        }
        bmax = fdw_shot_batch_max(bctx);
        if (bmax > ns) bmax = ns;
        if (bmax > 1 && dev_border && fdw_model_resident(bctx, vp) != FDW_OK) {
            fprintf(stderr, "fdw_model_resident: %s\n", fdw_last_error());
            return EXIT_FAILURE;
        }
        if (bmax <= 1) {
            fdw_destroy(bctx);
            bctx = NULL;
        }
    }
    const int batch = bctx ? bmax : (nworkers > 1 ? ns : 1);      /* shots whose model and image are held at once */
    float *vel2_all = (float *)malloc((size_t)batch * ne * sizeof(float)), *imloc_all = (float *)calloc((size_t)batch * ni, sizeof(float));
    if (!vel2_all || !imloc_all) {
        fprintf(stderr, "out of host memory\n");
        return EXIT_FAILURE;
    }
    shot_job job;
    job.prm = &prm; job.ns = ns; job.nworkers = nworkers; job.sx = sx; job.sz = sz; job.gz = gz; job.srce = srce; job.d_obs = d_obs;
    job.nx = nx; job.nt = nt; job.ne = ne; job.ni = ni; job.vel2_all = vel2_all; job.imloc_all = imloc_all; job.failed = 0;
    job.vp = vp; job.draws = fdw_border_draws(nx, nz, nxb, nzb); job.dev_border = dev_border; job.gpus = gpus;

    for (int is0 = 0; is0 < ns; is0 += batch) {
        const int nb = is0 + batch <= ns ? batch : ns - is0;
        for (int b = 0; b < nb && !dev_border; b++) { /* models in shot order: the rand() stream is sequential */
            const int is = is0 + b;
            const float *v = vpe;
            if (vel_ext_flag)
                v = vel_ext_rnd + (size_t)is * ne; /* R:484 */
            else
                fdw_extendvel_linear(nx, nz, nxb, nzb, vpe); /* R:486: glibc rand(), never seeded */
            float *v2 = vel2_all + (size_t)b * ne;
            for (size_t k = 0; k < ne; k++) v2[k] = v[k] * v[k]; /* R:490-494 */
        }
        const double t0 = now_s();
        job.is0 = is0; job.nb = nb;
        memset(imloc_all, 0, (size_t)nb * ni * sizeof(float));                                 /* R:515 */
        if (bctx) {
            /* shots is0 .. is0 + nb - 1: source rows sx[is0] + b ds (R:405-407), border models from draws [(is0 + b) T, ...) of the stream */
            /* models: drawn on the device, or the host-built ones of this batch (vel_ext_file decks, FDW_HOST_BORDER=1) */
            if (fdw_shot_batch(bctx, nb, dev_border ? NULL : vel2_all, (unsigned long long)is0 * (unsigned long long)job.draws, sx[is0], ds, sz, gz, srce,
                               d_obs + (size_t)is0 * nx * nt, imloc_all) != FDW_OK) {
                fprintf(stderr, "fdw_shot_batch: %s\n", fdw_last_error());
                return EXIT_FAILURE;
            }
        } else {
            const int nw = nb < nworkers ? nb : nworkers;
            pthread_t th[64];
            shot_worker_arg wa[64];
            for (int w = 0; w < nw; w++) {
                wa[w].job = &job; wa[w].w = w; wa[w].nw = nw;
                if (w > 0 && pthread_create(&th[w], NULL, shot_worker, &wa[w]) != 0) {
                    fprintf(stderr, "pthread_create failed\n");
                    return EXIT_FAILURE;
                }
            }
            shot_worker(&wa[0]);
            for (int w = 1; w < nw; w++) pthread_join(th[w], NULL);
            if (job.failed) return EXIT_FAILURE;
        }
        const double t1 = now_s();
        t_shots += t1 - t0;
        for (int b = 0; b < nb; b++) {               /* R:480-529 in shot order */
            const int is = is0 + b;
            const float *imloc = imloc_all + (size_t)b * ni;
            fprintf(stdout, "** source %d, at (%d,%d) \n", is + 1, sx[is] - nxb, sz - nzb);
            fprintf(stdout, "\n");
            fprintf(stdout, "** backward propagation %d, at (%d,%d) \n", is + 1, sx[is] - nxb, sz - nzb);
            fprintf(stdout, "\n");
            fprintf(fnum, "======== %i ========\n", is); /* R:522-528: iz outer, ix inner, running sum */
            for (int iz = 0; iz < nz; iz++)
                for (int ix = 0; ix < nx; ix++) {
                    img[(size_t)ix * nz + iz] += imloc[(size_t)ix * nz + iz];
                    fprintf(fnum, " %f \n", img[(size_t)ix * nz + iz]);
                }
        }
        t_stack += now_s() - t1;
    }
    if (bctx) fdw_destroy(bctx);
    free(vel2_all);
    free(imloc_all);
outputs:
    if (timing)
        fprintf(stderr, "[timing] total %.3f s: shots (contexts, border models, propagation) %.3f s, stacking + image.num %.3f s, rest (deck, inputs) %.3f s\n",
                now_s() - t_begin, t_shots, t_stack, now_s() - t_begin - t_shots - t_stack);
    /* opt-in extension (deck key image_lap=1): fill dir.image_lap with the reference's own offline filter (models/3lay_mod/laplace.f90)
     * of the stacked image instead of the zeros the reference writes (R:477, R:542) */
    if (fdw_deck_int(deck, "image_lap") == 1 && fdw_image_laplacian(0, img, nx, nz, dx, dz, img_lap) != FDW_OK) {
        fprintf(stderr, "fdw_image_laplacian: %s\n", fdw_last_error());
        return EXIT_FAILURE;
    }
    gettimeofday(&end, NULL);
    /* the reference divides integers (whole seconds, R:536); we print the real value */
    const double exec = ((end.tv_sec - start.tv_sec) * 1000000.0 + (end.tv_usec - start.tv_usec)) / 1000000.0;
    printf("> Exec time = %.2f (s)\n", exec);

    fwrite(img, sizeof(float), ni, fimg);         /* R:540 */
    fwrite(img_lap, sizeof(float), ni, fimg_lap); /* R:542 */
    if (fsns) fclose(fsns);
    if (fsns2) fclose(fsns2);
    if (fsnr) fclose(fsnr);
    fclose(fimg);
    fclose(fimg_lap);
    fclose(fnum);
    free(srce); free(sx); free(vel_ext_rnd); free(d_obs); free(vp); free(vpe);
    free(img); free(img_lap);
    fdw_deck_free(deck);
    return 0;
}
