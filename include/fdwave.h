/*
 * fdwave.h -- C ABI of libfdwave.so: the MI355X (gfx950) implementation of the 2-D acoustic
 * finite-difference hot path of FernandoSchett/parallel_finite_difference_computation.
 *
 * Plain C: pointers, ints, floats.  No HIP or torch types cross this boundary (a stream is a void*
 * that is really a hipStream_t; NULL = the context's own stream).  Every function returns 0 on
 * success or a negative FDW_E* code, with a human-readable message in fdw_last_error().  There is
 * NO CPU fallback: without a usable HIP device fdw_create() fails with FDW_ENODEVICE.
 *
 * Each entry point cites the reference interface it replaces.  File tags:
 *   S = cuda_reference_stencil_computation/fd-source-code.cu
 *   R = cuda_reference_RTM/src/fd-code.cu
 *   H = cuda_reference_RTM/lib/include/functions.h
 *   F = cuda_reference_RTM/lib/src/functions.c
 *
 * Array conventions are the reference's: a field is float[nxe][nze], x slow, z contiguous
 * (idx = ix*nze + iz, R:58), fp32 little endian, nxe = nx + 2*nxb, nze = nz + 2*nzb (R:410-411).
 * "Host" arrays are dense with that layout.  "Device" arrays (fdw_dev_*) are pitched:
 * float[nxl][pitch] with pitch = fdw_pitch(ctx) >= nze, padding columns must be zero.
 */
#ifndef FDWAVE_H
#define FDWAVE_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FDW_VERSION 2
#define FDW_MAX_ORDER 32

/* error codes */
#define FDW_OK 0
#define FDW_EINVAL (-1)    /* bad argument (message says which) */
#define FDW_ENODEVICE (-2) /* no HIP device / wrong architecture / HIP runtime failure at create */
#define FDW_EHIP (-3)      /* a HIP call failed later */
#define FDW_ENOMEM (-4)
#define FDW_ESTATE (-5)    /* call sequence error (e.g. back() before forward() in a device-resident shot) */
#define FDW_ECOMM (-6)     /* librccl missing, an RCCL call failed, or the ranks of an exchange disagree */

typedef struct fdw_ctx fdw_ctx;

/* The arguments of the reference's fd_init (R:200, H:15; stencil variant S:241) plus the switches
 * that the reference fixes at build time. */
typedef struct fdw_params {
    int order;      /* even, 2..FDW_MAX_ORDER (R:374 default 8).  2/4/6/8 use the register-window kernel */
    int nxe, nze;   /* extended grid (R:410-411) */
    int nxb, nzb;   /* absorbing border widths (R:375-376); 0 allowed (stencil program does not use them) */
    int nt;         /* time steps per propagation (R:351) */
    float dx, dz, dt;
    float fac;      /* taper strength F (R:377), in (0,1] */
    int compat;     /* 1: reproduce the reference's truncated RTM launch extents, gridx = floor(nxe/8)
                       etc. (R:185-195) -- required for parity with rtm_code output;
                       0: update the whole array */
    int coef_cxx;   /* 1: generic-order weights with float cosf/powf as in the stencil program (S:184-216
                       is compiled as C++); 0: double libm as in libsource.a (F:160-192).  Irrelevant
                       for order 2/4/6/8 */
    int dialect;    /* 0 (FDW_DIALECT_RTM): the CUDA programs' arithmetic and one-sided taper (everything above);
                       1 (FDW_DIALECT_MOD): the CPU-serial sibling's forward modelling -- dpct_gpu_rtm_domain_division/src:
                       fd_step's single accumulator (timestep/fd.c:24-46), four-sided taper_apply with taper = exp(-(F*(nb-i))^2)
                       (boundary/taper.c:26-66), whole-grid update; order <= 8; only fdw_model_shot and the host helpers use it */
    int numerics;   /* 0 (FDW_NUMERICS_EXACT, the default of a zeroed struct): the reference's arithmetic operation for operation (nvcc
                       --fmad=false: every product and every sum of the Laplacian rounded, R:66-72) -- results identical to the no-FMA
                       CUDA build bit for bit;
                       1 (FDW_NUMERICS_FAST): the same stencil with the symmetric taps summed first and fused multiply-adds,
                       lap = c0 p + sum_k [cz_k (p(j-k) + p(j+k)) + cx_k (p(i-k) + p(i+k))] (csrc/fdw_device.h), about half the vector
                       instructions; the fp64 leap-frog (R:89), the taper, the launch extents and every other quirk are unchanged.
                       Results agree with EXACT to rounding: <= 1e-5 max-norm-relative over the 1 700 steps of the reference's new_mod
                       deck (tests), the size of the reference's own FMA / no-FMA difference.  Dialects 1 and 2 take the same formula
                       with the spacings folded into the weights (c_k * d?2inv once instead of inside every term): the sibling's committed
                       3lay_mod gather is met to 2.2e-6 in the max norm (1.3e-5 in the L2 norm) over its 1 001 steps. */
} fdw_params;
enum { FDW_NUMERICS_EXACT = 0, FDW_NUMERICS_FAST = 1 };
enum { FDW_DIALECT_RTM = 0, FDW_DIALECT_MOD = 1, FDW_DIALECT_RTM_STORED = 2 };
/* 2 (FDW_DIALECT_RTM_STORED): the same sibling's stored-wavefield RTM (src/rtm_main.cpp): fd_step arithmetic as dialect 1, one-cell
 * source, taper_apply2 (top strip only, taper.c:68-83) with the taper table of dialect 1; only fdw_rtm_stored_shot uses it. */

/* A slab of the global grid owned by one device (domain decomposition along x, the slow axis).
 * The reference has no multi-GPU path; slab 0..nxe is the single-GPU case. */
typedef struct fdw_slab {
    int x_off; /* global row index of local row 0 (may include ghost rows) */
    int nxl;   /* local rows held on this device, ghost rows included */
} fdw_slab;

const char *fdw_last_error(void);
int fdw_version(void);

/* ---- life cycle ------------------------------------------------------------------------------
 * fdw_create       replaces fd_init + fd_init_cuda (R:146-224, S:218-262): derived constants
 *                  (d?2inv, dt2, scaled coefficients, taper tables), device buffers, launch
 *                  geometry.  device = HIP ordinal.
 * fdw_create_slab  same for one x-slab of a decomposed grid; slab rows must hold order/2 ghost
 *                  rows (or more) towards every neighbouring slab.
 * fdw_destroy      replaces the cudaFree block R:569-582 / S:264-275. */
int fdw_create(const fdw_params *prm, int device, fdw_ctx **out);
/* fdw_device_count   HIP devices visible to this process (0 when there is none or HIP cannot start).
 * fdw_device_usable  0 if `device` exists, is a gfx950 and can be selected -- what a rank of a multi-GPU job checks ALONE before it enters
 *                    a collective call (rtm_code slabs=N). */
int fdw_device_count(void);
int fdw_device_usable(int device);
int fdw_create_slab(const fdw_params *prm, const fdw_slab *slab, int device, fdw_ctx **out);
void fdw_destroy(fdw_ctx *ctx);

/* ---- host-array entry points (the reference's L2 seam; synchronous like the reference) ---------
 *
 * fdw_laplacian   S:320-333: H2D(p), kernel_lap, D2H(lap).  lap border cells are written as 0
 *                 (the committed golden has zeros there).  Uses coef_cxx weights.
 *
 * fdw_forward     fd_forward R:247-288.  p,pp in/out (the reference uploads them R:230-231 and
 *                 downloads d_p,d_pp R:285-286); v2 = squared velocity; sx,sz = source position on
 *                 the extended grid (R:406,408); srce[nsteps]; nsteps normally = nt.
 *
 * fdw_back        fd_back R:290-341.  snap0 = P and snap1 = PP of the forward pass (R:502-507),
 *                 d_obs = one shot gather [nx][nt] (R:426-435), gz on the extended grid (R:409),
 *                 imloc[nx][nz] in/out (uploaded R:243, downloaded R:340).
 *
 * fdw_shot        forward + back for one shot with everything device-resident (no snapshot round
 *                 trip): what main does per shot R:496-518 minus the host memsets.  Results are
 *                 identical to fdw_forward followed by fdw_back.  P/PP may be NULL.
 */
int fdw_laplacian(fdw_ctx *ctx, const float *p, float *lap);
int fdw_forward(fdw_ctx *ctx, float *p, float *pp, const float *v2, int sx, int sz, const float *srce, int nsteps);
int fdw_back(fdw_ctx *ctx, const float *v2, const float *snap0, const float *snap1, const float *d_obs, int gz,
             float *imloc, int nsteps);
int fdw_shot(fdw_ctx *ctx, const float *v2, int sx, int sz, int gz, const float *srce, const float *d_obs,
             float *imloc, float *P, float *PP);

/* ---- device-array entry points (benchmarks, multi-GPU drivers; asynchronous on `stream`) -------
 * Buffers are caller-owned device memory laid out [nxl][fdw_pitch()] (e.g. a torch tensor).
 *
 * PRECONDITION of every entry point that takes wavefields (host or device) on a context with compat = 1 and nxe not a multiple of 8:
 * inside the damped strip (columns < 8*floor(nzb/8)) the rows the reference never time-steps (>= 8*floor(nxe/8)) must be ZERO.  The
 * reference damps those rows with taperx every step although nothing ever rewrites them (R:94-117 with the grids of R:185-195); the kernels
 * here damp on load instead of in place (csrc/fdw_device.h, "lazy taper") and cannot reproduce that for rows they never store.  Every call
 * site of the reference satisfies it (fields start at zero, R:496-497, R:511-514, and snapshots come from such runs).  The host-array entry
 * points check it and return FDW_EINVAL; the fdw_dev_* ones do not look at the data (it is on the device, and they are asynchronous) -- a
 * caller that violates it gets values in those few rows of the strip that differ from the reference's.  fdw_dev_check_field(ctx, d_field,
 * stream) runs the same check ON the device for such a caller (synchronises `stream`; FDW_EINVAL with the count of offending cells).
 * A source row in those rows is refused (FDW_EINVAL) by every path.
 *
 * fdw_dev_step    one fused time step on rows [r0,r1) of the slab:
 *                   mode 0 FWD   taper + Laplacian + leap-frog + point source  (R:264-267)
 *                   mode 1 PLAIN Laplacian + leap-frog                          (R:317-318)
 *                   mode 2 RECV  taper + Laplacian + leap-frog + receivers + imaging (R:325-329)
 *                 d_p is read, d_pp is read and overwritten with the new field (the caller swaps
 *                 roles afterwards, R:260-262).  pp_twice: 0 on the first step after fresh data,
 *                 1 afterwards (see csrc/fdw_device.h "lazy taper").  d_inj: FWD: device pointer to
 *                 the source sample of this step, inj_x/inj_z its GLOBAL position (inj_x < 0: no
 *                 source); RECV: device pointer to nx receiver samples of this step (row-contiguous,
 *                 i.e. d_obs transposed to [it][ix]), inj_z = gz.  d_psrc/d_img: RECV only.
 * fdw_dev_taper_finalize  applies the one taper pass the lazy scheme still owes to a field that was
 *                 last used as d_p (needed before it is exported or used untapered).
 * fdw_dev_laplacian  mode 3: d_lap = Laplacian(d_p), zero outside the interior.
 * fdw_dev_step2   TWO forward iterations in one pass (temporal blocking, order 8, full slab): reads d_p = newest
 *                 field u^n (the reference's d_p AFTER its swap), d_pp = u^{n-1}, writes u^{n+1} to d_out1 and
 *                 u^{n+2} to d_out2 (no aliasing allowed), d_srce_it -> {srce[it], srce[it+1]}.  10 B/point/step
 *                 instead of 16; results are bit-identical to two fdw_dev_step calls.
 * fdw_dev_steps2  nsteps iterations over FOUR rotating buffers, in pairs through fdw_dev_step2 (odd remainder:
 *                 one-step kernel).  ip and ipp index the reference's (d_p, d_pp) before the first swap on entry
 *                 and after the loop on return.
 * fdw_dev_step4   FOUR forward iterations in one pass of the wave-pipeline kernel (order 8, fields < 2 GiB): d_p = u^n,
 *                 d_pp = u^{n-1} -> d_out1 = u^{n+3}, d_out2 = u^{n+4} on local rows [r0, r1) and, optionally, [r0b, r1b)
 *                 (r1 < 0: every row the reference time-steps); rows within 16 of a range end are read from d_p / d_pp, so
 *                 a slab driver shrinks the range by 16 rows per pass between two halo exchanges (decomp.py).
 *                 d_srce_it -> srce[it .. it+3]; xchunk 0 = automatic.  Bit-identical to four fdw_dev_step calls.
 * fdw_dev_steps_shrink  like fdw_dev_steps for one slab of a decomposed grid between two halo exchanges:
 *                 step j = j0.. of the cycle updates rows [h*j, nxl - h*j) on the sides that have a
 *                 neighbour (shrink_lo / shrink_hi), see decomp.py.
 * fdw_dev_back_iter  ONE iteration of fd_back's loop (R:302-339) on local rows [r0, r1) of a slab (or of the whole grid): with
 *                 step_source = 1 the source field is reconstructed one step back in time first, F_k = leap-frog(d_f1 = F_{k-1},
 *                 d_f0 = F_{k-2}) written over d_f0 (R:317-318: no taper, no source); with step_source = 0 (iterations 0 and 1, whose
 *                 source fields are the two snapshots, R:304-314) d_f1 is used as it stands and d_f0 is ignored.  Then the receiver
 *                 step (taper + Laplacian + leap-frog, d_pr read, d_ppr overwritten, R:325-327), the injection of d_samples[0..nx)
 *                 = d_obs[.][nt-1-it] on column gz of the interior rows (R:328) and img += F_k * new receiver field (R:329), where
 *                 d_img is [nxl][pitch] on the extended grid.  The caller swaps (d_f1, d_f0) when step_source and (d_pr, d_ppr) always.
 *                 Rows of different calls of one iteration must be disjoint; pp_twice as for fdw_dev_step.  One launch where the fused
 *                 backward kernel exists (order <= 8), two otherwise.
 * fdw_dev_back4   FOUR iterations of that loop (no snapshot iterations among them) as two passes of the wave-pipeline kernel (order 8, fields
 *                 < 2 GiB, no receiver rows beyond the time-stepped rows; fdw_back_pipe_active says whether fdw_back itself takes this path on this
 *                 grid): pass 1 reconstructs F_it .. F_{it+3} from d_f1 = F_{it-1}, d_f0 = F_{it-2} into d_lvl0, d_lvl1, d_fo1, d_fo2; pass 2
 *                 advances the receiver field four times (d_pr = r^it, d_ppr = r^{it-1} -> d_ro1 = r^{it+3}, d_ro2 = r^{it+4}), injecting
 *                 d_samples + j * sample_stride at iteration it + j, and adds the four imaging products to d_img in iteration order.  Local
 *                 rows [r0, r1) and [r0b, r1b) (r1 < 0: all); rows within 16 of a range end are read from the inputs, so a slab driver
 *                 shrinks the range by 16 rows per call between two halo exchanges.  No buffer may alias another.  Bit-identical to four
 *                 fdw_dev_back_iter calls.
 * fdw_dev_steps   nsteps FWD steps with internal role swapping; *d_srce is srce[] on the device
 *                 (may be NULL = no source).  After an odd number of steps the newest field is in
 *                 the buffer passed as d_pp, after an even number in d_p (as in the reference loop).
 */
int fdw_pitch(const fdw_ctx *ctx);           /* floats per row of a device array */
size_t fdw_field_bytes(const fdw_ctx *ctx);  /* nxl * pitch * 4 */
int fdw_dev_step(fdw_ctx *ctx, int mode, const float *d_p, float *d_pp, const float *d_v2, int r0, int r1,
                 int pp_twice, const float *d_inj, int inj_x, int inj_z, const float *d_psrc, float *d_img,
                 void *stream);
int fdw_dev_back_iter(fdw_ctx *ctx, int step_source, const float *d_f1, float *d_f0, const float *d_pr, float *d_ppr, const float *d_v2,
                      int r0, int r1, int pp_twice, const float *d_samples, int gz, float *d_img, void *stream);
int fdw_dev_back4(fdw_ctx *ctx, const float *d_f1, const float *d_f0, float *d_fo1, float *d_fo2, float *d_lvl0, float *d_lvl1, const float *d_pr,
                  const float *d_ppr, float *d_ro1, float *d_ro2, const float *d_v2, const float *d_samples, int sample_stride, int gz, float *d_img,
                  int pp_twice, int r0, int r1, int r0b, int r1b, int xchunk, void *stream);
int fdw_back_pipe_active(const fdw_ctx *ctx);
int fdw_dev_steps(fdw_ctx *ctx, float *d_p, float *d_pp, const float *d_v2, const float *d_srce, int sx, int sz,
                  int it0, int nsteps, int first_pp_twice, void *stream);
int fdw_dev_steps_shrink(fdw_ctx *ctx, float *d_p, float *d_pp, const float *d_v2, const float *d_srce, int sx, int sz,
                         int it0, int nsteps, int first_pp_twice, int j0, int shrink_lo, int shrink_hi, void *stream);
int fdw_dev_step2(fdw_ctx *ctx, const float *d_p, const float *d_pp, const float *d_v2, float *d_out1, float *d_out2, int pp_twice,
                  const float *d_srce_it, int sx, int sz, void *stream);
int fdw_dev_step4(fdw_ctx *ctx, const float *d_p, const float *d_pp, const float *d_v2, float *d_out1, float *d_out2, int pp_twice,
                  const float *d_srce_it, int sx, int sz, int r0, int r1, int r0b, int r1b, int xchunk, void *stream);
int fdw_dev_steps2(fdw_ctx *ctx, float *const *d_buf, const float *d_v2, const float *d_srce, int sx, int sz, int it0, int nsteps,
                   int first_pp_twice, int *ip, int *ipp, void *stream);
int fdw_dev_taper_finalize(fdw_ctx *ctx, float *d_f, void *stream);
int fdw_dev_check_field(fdw_ctx *ctx, const float *d_f, void *stream);
int fdw_dev_laplacian(fdw_ctx *ctx, const float *d_p, float *d_lap, void *stream);

/* ---- forward-modelling producer -------------------------------------------------------
 * fdw_model_shot  one shot of mod_main's loop (dpct_gpu_rtm_domain_division/src/mod_main.cpp:140-174) on a context created with
 *                 dialect = FDW_DIALECT_MOD: P = PP = 0; nt x { fd_step; ptsrc (7x7 Gaussian around (sx, sz), source/ptsrc.c:12-58);
 *                 taper_apply(PP); taper_apply(P); data[ix][it] = P[ix+nxb][gz] }.  vel2[nxe][nze] is the extended squared
 *                 velocity (after fdw_mod_extendvel), srce[nt], data[nx][nt].  Device resident, one launch per step.
 * fdw_mod_extendvel       taper.c:7-23: replicate the edge values outwards (in place, [nxe][nze])
 * fdw_mod_ricker_wavelet  ptsrc.c:88-99: Ricker delayed by 1/fpeak, zero after 2/fpeak
 * fdw_mod_taper_tables    taper.c:26-44 */
int fdw_model_shot(fdw_ctx *ctx, const float *vel2, int sx, int sz, int gz, const float *srce, int nt, float *data);
/* the same loop on caller-owned device arrays (d_p / d_pp = mod_main's P / PP, swapped every step; d_rec[it][nx] or NULL) */
int fdw_dev_model_steps(fdw_ctx *ctx, float *d_p, float *d_pp, const float *d_v2, const float *d_srce, int sx, int sz, int gz,
                        float *d_rec, int it0, int nsteps, void *stream);
void fdw_mod_extendvel(int nx, int nz, int nxb, int nzb, float *vel);
void fdw_mod_ricker_wavelet(int nt, float dt, float fpeak, float *srce);
void fdw_mod_taper_tables(int nxb, int nzb, float fac, float *taper_x, float *taper_z);

/* ---- stored-wavefield RTM of the CPU-serial sibling ----------------------------------------
 * fdw_rtm_stored_shot  one shot of rtm_main's loop (dpct_gpu_rtm_domain_division/src/rtm_main.cpp:158-240) on a context created with
 *                 dialect = FDW_DIALECT_RTM_STORED: the source pass keeps the field of every step on the device (nt fields: fails with
 *                 FDW_ENOMEM when they do not fit), the receiver pass injects sample nt-it of every trace of shot `is` -- read from the
 *                 WHOLE gather dobs[ns][nx][nt] of n_floats floats exactly as the reference indexes it, i.e. one sample past the trace at
 *                 it = 0 (a sample past the end of the file counts as 0) and with its nzb row offset (rtm_main.cpp:203) -- and the image
 *                 accumulates swf[nt-it-1] * rwf[it] in iteration order (same sums as rtm_main.cpp:224-230).  imloc[nx][nz] is overwritten.
 *                 When the nt fields do not fit -- the budget of fdw_set_store_budget (or FDW_STORE_BUDGET_MB), else what hipMalloc grants --
 *                 the source pass CHECKPOINTS: it keeps the pair of fields every m-th step starts from, and the receiver pass recomputes
 *                 one segment of m fields at a time from its pair (2 ceil(nt/m) + m + 1 fields, least near m = sqrt(2 nt); one more forward
 *                 pass of launches).  The recomputation repeats the same launches on the same inputs: the image is bit-identical to the
 *                 unconstrained run.  FDW_ENOMEM only when not even that fits.
 * fdw_set_store_budget  bytes the stored source fields may occupy (0 = no limit of our own); fdw_store_segments: how many segments the last
 *                 fdw_rtm_stored_shot was cut into (1 = every field was kept). */
int fdw_rtm_stored_shot(fdw_ctx *ctx, const float *vel2, int sx, int sz, int gz, const float *srce, int nt, const float *dobs,
                        size_t n_floats, int is, float *imloc);
int fdw_set_store_budget(fdw_ctx *ctx, size_t bytes);
int fdw_store_segments(const fdw_ctx *ctx);

/* ---- image post-processing -----------------------------------------------------------------
 * fdw_image_laplacian  the reference's offline filter models/3lay_mod/laplace.f90:25-29 (dir.image -> dir.imalap): second-order
 *                 Laplacian of img[nx][nz] with the frame left at zero, on `device`. */
int fdw_image_laplacian(int device, const float *img, int nx, int nz, float dx, float dz, float *out);
/* fdw_image_compare  the reference's image comparer models/marmousi/psnr ("./psnr file1 file2"; it ships as an ELF without source, so its
 *                 behaviour is restated from its output): stats = {MSE = mean (a-b)^2, RMSE, SNR = 10 log10(sum b^2 / sum (a-b)^2) dB,
 *                 PSNR = 20 log10(max |b| / RMSE) dB}; diff (may be NULL) = a - b, what the tool writes to ./dir.output.  On `device`.
 *                 exact_sums = 0: the tool's own arithmetic -- the squares added one after the other into fp32 sums (a serial recurrence: one
 *                 lane walks the arrays, ~10 ns per element), fp32 quotient and root -- so the values are the tool's, digit for digit;
 *                 exact_sums = 1: a parallel reduction carrying the same terms in double (fast; differs from the tool's figures by the
 *                 rounding the tool accumulates, up to a few 1e-5 relative on the reference's own images). */
int fdw_image_compare(int device, const float *a, const float *b, size_t n, float *diff, double stats[4], int exact_sums);

/* host <-> pitched device copies (dense [rows][nze] on the host side), synchronous */
int fdw_upload_field(fdw_ctx *ctx, float *d_dst, const float *h_src);
int fdw_download_field(fdw_ctx *ctx, float *h_dst, const float *d_src);

/* ---- random-border model generated on the device ---------------------
 * The reference rebuilds its extended velocity model on the host for every shot (extendvel_linear, F:336-394, called at
 * R:486; vel2 = vpe * vpe at R:488-494) and uploads it (R:205).  Here the interior model is uploaded once and each
 * shot's border is generated in HBM from the same unseeded glibc rand() stream, addressed by position: shot s of a
 * fresh process consumes draws [s T, (s+1) T), T = fdw_border_draws(...).  Cells the reference's loops never write
 * (bottom corners when nxb > nzb) are zero, as in the reference's calloc'ed array.
 * fdw_border_draws          rand() calls one extendvel_linear consumes: nx nzb + 2 nz nxb + 2 nzb (nzb + 1)
 * fdw_model_resident        uploads the interior velocity vp[nx][nz] (not squared); RTM dialect, full-grid context
 * fdw_dev_extendvel_linear  fills the context's resident vel2 (and vel_out[nxe][nze] on the host if not NULL) from
 *                           draws [draw_offset, draw_offset + T) of the seed-1 stream
 * fdw_shot_resident         fdw_shot (R:496-520) on the resident vel2; FDW_ESTATE if a host model was uploaded since
 * fdw_rand_stream           out[i] = draw number draw_offset + i of that stream, produced by the device generator (tests)
 * fdw_shot_batch            `nshots` consecutive shots of the reference's loop (R:480-520) through ONE launch per time step: shot b has
 *                           source row sx0 + b dsx (R:405-407), gather d_obs + b nx nt, image imloc + b nx nz (accumulated into, as
 *                           fdw_shot does), model v2_all + b nxe nze or, v2_all == NULL, the border model of draws
 *                           [draw_offset + b T, ...) drawn on the device.  Results are those of the shots run one by one, bit for bit;
 *                           contexts whose regime the batched launches do not cover (large grids, orders above 8, receiver rows
 *                           outside the truncated extents) run them one by one.
 * fdw_shot_batch_max        batch size that fills the chip for this geometry (1: batching gains nothing)
 */
long long fdw_border_draws(int nx, int nz, int nxb, int nzb);
int fdw_shot_batch(fdw_ctx *ctx, int nshots, const float *v2_all, unsigned long long draw_offset, int sx0, int dsx, int sz, int gz,
                   const float *srce, const float *d_obs, float *imloc);
int fdw_shot_batch_max(const fdw_ctx *ctx);
/* mod_main's shot loop (mod_main.cpp:140-174) for `nshots` consecutive shots (source rows sx0 + b dsx, M:99-101) on its one velocity
 * model, one launch per time step for all of them; data[nshots][nx][nt].  Same results as fdw_model_shot per shot, bit for bit. */
int fdw_model_shot_batch(fdw_ctx *ctx, int nshots, const float *vel2, int sx0, int dsx, int sz, int gz, const float *srce, int nt, float *data);
int fdw_model_resident(fdw_ctx *ctx, const float *vp);
int fdw_dev_extendvel_linear(fdw_ctx *ctx, unsigned long long draw_offset, float *vel_out);
int fdw_shot_resident(fdw_ctx *ctx, int sx, int sz, int gz, const float *srce, const float *d_obs, float *imloc, float *P, float *PP);
int fdw_rand_stream(fdw_ctx *ctx, unsigned long long draw_offset, long long n, int *out);

/* ---- multi-GPU: communicators and the slab-decomposed loops (csrc/fdw_comm.cpp, csrc/fdw_slabs.cpp) ------------------------
 * The reference has no multi-GPU path.  The grid is decomposed along x, the slow axis, into one band of rows per
 * rank; a rank is one GPU, driven by one process (RCCL backend) or by one host thread of a process (RCCL or local backend).
 *
 * fdw_comm_get_unique_id  ncclGetUniqueId: rank 0 calls it and hands the 128 bytes to the other ranks (a file, an environment
 *                         variable, torch.distributed ...).  librccl.so.1 is opened on first use (dlopen); FDW_ECOMM if it is missing.
 * fdw_comm_init_rank      ncclCommInitRank on `device`: the RCCL backend, one rank per GPU.  Halo blocks travel as ncclSend / ncclRecv
 *                         pairs inside one ncclGroupStart / ncclGroupEnd on the slab driver's communication stream (neighbours only).
 * fdw_comm_init_local     `world` ranks inside ONE process, out[r] for the host thread that drives rank r on devices[r] (NULL: all on
 *                         device 0).  A halo transfer is a device copy on the receiver's stream ordered by events after the sender's
 *                         stream.  Ranks may share a device (tests and rehearsals on a one-GPU box, where RCCL refuses duplicate
 *                         devices) or sit on different GPUs (peer copies over xGMI).  Every rank must be destroyed.
 * fdw_comm_init_stub      rank `rank` of a world whose other ranks do not exist: exchanges move nothing.  TIMING EXPERIMENTS ONLY (what
 *                         one rank of an N-way decomposition costs without its links: scripts/probe_slabs_c.py); results are wrong.
 * fdw_comm_init_shm       one rank per PROCESS without RCCL: halo blocks are staged through the POSIX shared-memory segment `name` ("/..."; rank
 *                         0 creates it, it is unlinked as soon as all ranks hold it), box_bytes = the largest message of one exchange
 *                         (fields x ghost rows x pitch x 4).  A TEST TRANSPORT (ranks may share one GPU, which RCCL refuses): the ranks'
 *                         streams are tied together by nothing but the arrival of a block, as under RCCL.  Collective.
 * fdw_comm_kind           FDW_COMM_RCCL / FDW_COMM_LOCAL / FDW_COMM_SHM, 0 for a stub or NULL
 * fdw_comm_allreduce      one double per rank, summed (op_max = 0) or the maximum (op_max = 1); blocks the host.
 * fdw_comm_selftest       one block sent to the OWN rank through the backend's send / receive path on a stream, and compared.
 *
 * fdw_slabs_create        this rank's share of the decomposition: its band of rows plus order/2 * ksteps ghost rows towards each
 *                         neighbour (ksteps = time steps per halo exchange; 0 = chosen from the band size, the same on every rank), a
 *                         slab context (fdw_create_slab), three streams.  Collective: every rank of `comm` calls it.  comm == NULL: one
 *                         rank holding the whole grid on `device`.
 * fdw_slabs_geometry      x_off = global row of local row 0, nxl = local rows (ghosts included), [own0, own1) = owned global rows,
 *                         nbuf = field buffers fdw_slabs_dev_forward rotates over (4 where it runs four time steps per pass, else 2).
 * fdw_slabs_dev_forward   fd_forward's loop (R:259-267) on caller-owned device arrays [nxl][fdw_pitch(fdw_slabs_ctx())]: buf[*ip], buf[*ipp]
 *                         are the reference's (d_p, d_pp) before its first swap on entry and after the loop on return.  Halo exchanges
 *                         included, overlapped with the interior rows; asynchronous (fdw_slabs_synchronize; fdw_slabs_stream is the
 *                         compute stream, for events).
 * fdw_slabs_dev_back      fd_back's loop (R:302-339) on caller-owned device arrays: f[role[0]], f[role[1]] = (F_{k-1}, F_{k-2}) -- before iteration 2
 *                         the (P, PP) of the forward pass, P damped (fdw_dev_taper_finalize) --, r[role[2]], r[role[3]] = (r^k, r^{k-1}), zero
 *                         before iteration 0; role[] is updated on return.  f holds nfb buffers and r nrb (fdw_slabs_back_buffers: 6 and 4
 *                         where the slab runs four iterations per pair of pipeline passes, else 2 and 2).  d_samples[nt][nx] with row it =
 *                         d_obs[.][nt-1-it], d_img [nxl][pitch] (owned rows meaningful).
 * fdw_slabs_shot          one shot of rtm_code's loop (R:496-520) on host arrays: every rank passes the GLOBAL v2[nxe][nze], srce[nt],
 *                         d_obs[nx][nt]; imloc[nx][nz] (global; accumulated into) and the optional P, PP [nxe][nze] receive this rank's
 *                         OWNED rows only.  Bit-identical to fdw_shot on the whole grid.
 * fdw_slabs_set_stub      on = 1: this rank's halo exchanges move nothing from now on (the cycles keep their launches and stream hand-overs).
 *                         MEASUREMENT ONLY: bench.py times the same window with and without the transfers to report the exposed
 *                         communication fraction; results computed while it is on are wrong.  Every rank must switch together. */
typedef struct fdw_comm fdw_comm;
typedef struct fdw_slabs fdw_slabs;
#define FDW_COMM_ID_BYTES 128
int fdw_comm_get_unique_id(char id[FDW_COMM_ID_BYTES]);
int fdw_comm_init_rank(const char id[FDW_COMM_ID_BYTES], int rank, int world, int device, fdw_comm **out);
int fdw_comm_init_local(int world, const int *devices, fdw_comm **out /* [world] */);
int fdw_comm_init_stub(int rank, int world, int device, fdw_comm **out);
int fdw_comm_init_shm(const char *name, int rank, int world, int device, size_t box_bytes, fdw_comm **out);
enum { FDW_COMM_RCCL = 1, FDW_COMM_LOCAL = 2, FDW_COMM_SHM = 3 };
int fdw_comm_kind(const fdw_comm *comm);
void fdw_comm_destroy(fdw_comm *comm);
int fdw_comm_rank(const fdw_comm *comm);
int fdw_comm_world(const fdw_comm *comm);
int fdw_comm_device(const fdw_comm *comm);
int fdw_comm_is_local(const fdw_comm *comm);
int fdw_comm_allreduce(fdw_comm *comm, double *value, int op_max);
int fdw_comm_barrier(fdw_comm *comm);
int fdw_comm_selftest(fdw_comm *comm);
int fdw_slabs_create(const fdw_params *prm, fdw_comm *comm, int device, int ksteps, fdw_slabs **out);
void fdw_slabs_destroy(fdw_slabs *s);
fdw_ctx *fdw_slabs_ctx(fdw_slabs *s);
int fdw_slabs_geometry(const fdw_slabs *s, int *x_off, int *nxl, int *own0, int *own1, int *ksteps, int *nbuf);
void *fdw_slabs_stream(fdw_slabs *s);
int fdw_slabs_synchronize(fdw_slabs *s);
int fdw_slabs_dev_forward(fdw_slabs *s, float *const *buf, const float *d_v2, const float *d_srce, int sx, int sz, int it0, int nsteps,
                          int first_pp_twice, int *ip, int *ipp);
int fdw_slabs_back_buffers(const fdw_slabs *s, int *nfb, int *nrb);
int fdw_slabs_dev_back(fdw_slabs *s, float *const *f, float *const *r, const float *d_v2, const float *d_samples, int gz, float *d_img, int it0,
                       int nsteps, int role[4]);
int fdw_slabs_shot(fdw_slabs *s, const float *v2, int sx, int sz, int gz, const float *srce, const float *d_obs, float *imloc, float *P, float *PP);
int fdw_slabs_set_stub(fdw_slabs *s, int on);

/* ---- tuning / introspection --------------------------------------------------------------------
 * fdw_set_tuning  xchunk = rows marched per wave (0 = auto), wz = waves of a block laid along z
 *                 (1,2,4; 0 = auto), use_generic = force the generic-order kernel (tests),
 *                 prefetch = software prefetch distance in rows (0 = default; 1..3, order 8 only),
 *                 two_step = temporal blocking in the forward loops: 0 auto (by grid size), 1 two steps per pass always,
 *                 4 the four-steps-per-pass wave pipeline always, -1 never.
 * fdw_get_tables  copies of the derived host tables (any pointer may be NULL):
 *                 coefs_x/z[order+1] (R:214-217), taper_x[nxb], taper_z[nzb] (R:159-166).
 * fdw_get_extents xlim/zlim = rows/columns the time update covers, ztap = damped columns (R:185-195).
 * fdw_selftest    checks on the device the two hardware behaviours the kernels rely on (one-lane
 *                 __shfl_up/down, range-predicated buffer stores); 0 if they are as assumed.
 */
int fdw_set_tuning(fdw_ctx *ctx, int xchunk, int wz, int use_generic, int prefetch, int two_step);
int fdw_get_tables(const fdw_ctx *ctx, float *coefs_x, float *coefs_z, float *taper_x, float *taper_z);
int fdw_get_extents(const fdw_ctx *ctx, int *xlim, int *zlim, int *ztap);
int fdw_two_step_active(const fdw_ctx *ctx); /* 1 if the forward loops of this context use the two-step kernel */
int fdw_steps_per_pass(const fdw_ctx *ctx);  /* time steps one launch of the forward loops advances: 4 (wave pipeline), 2 or 1 */
int fdw_selftest(fdw_ctx *ctx);
/* 1 if the host loops emit roctx ranges (forward loop, backward loop, halo exchange, shot): they do when a marker library
 * (librocprofiler-sdk-roctx / libroctx64) is already in the process -- rocprofv3 --marker-trace -- or FDW_ROCTX=1 asks for it. */
int fdw_trace_active(void);

/* ---- host formulas of libsource.a restated (pure C, usable without a device) --------------------
 * fdw_calc_coefs        calc_coefs + makeo2, F:113-192 / S:137-216 (cxx selects the float variant)
 * fdw_ricker_wavelet    ricker_wavelet, F:302-334
 * fdw_taper_tables      taper tables of fd_init_cuda, R:159-166
 * fdw_extendvel_linear  extendvel_linear, F:336-394; vel is [nxe][nze] contiguous; draws the stream
 *                       glibc rand() yields (the reference never seeds it, R:486) from a PRIVATE restatement
 *                       of that generator, so nothing else in the process can perturb the border model
 * fdw_srand             reseeds that private generator like srand(); a fresh process behaves as fdw_srand(1)
 */
int fdw_calc_coefs(int order, int cxx, float *coef /* [order+1] */);
void fdw_ricker_wavelet(int nt, float dt, float fpeak, float *s);
void fdw_taper_tables(int nxb, int nzb, float fac, float *taper_x, float *taper_z);
void fdw_extendvel_linear(int nx, int nz, int nxb, int nzb, float *vel);
void fdw_srand(unsigned seed);

#ifdef __cplusplus
}
#endif
#endif /* FDWAVE_H */
