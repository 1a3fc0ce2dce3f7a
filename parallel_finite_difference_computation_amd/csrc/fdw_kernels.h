// fdw_kernels.h -- internal interface between the C-ABI layer (fdw_api.cpp) and the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>

namespace fdw {

enum {
    FDW_MODE_FWD = 0,    // taper + Laplacian + leap-frog + point source   (fd_forward body, R:264-267)
    FDW_MODE_PLAIN = 1,  // Laplacian + leap-frog only                     (fd_back source field, R:317-318)
    FDW_MODE_RECV = 2,   // taper + Laplacian + leap-frog + receivers + imaging (R:325-329)
    FDW_MODE_LAP = 3,    // Laplacian only, written to `pp`                (stencil_code, S:325)
    FDW_MODE_MOD = 4,    // forward-modelling step of the CPU-serial sibling (mod_main: fd_step + ptsrc + taper_apply + trace sample)
    FDW_MODE_DD_FWD = 5, // its stored-wavefield RTM, source pass (rtm_main.cpp:165-184: fd_step + one-cell source + taper_apply2)
    FDW_MODE_BACK = 7,   // one whole backward iteration of fd_back in a single pass: source-field step + receiver step + imaging (R:317-329)
    FDW_MODE_BACK4 = 9,  // wave-pipeline kernel only: four whole iterations of fd_back in one pass of an eight-wave workgroup (source + receiver fields)
    FDW_MODE_PLAIN_ALL = 8, // wave-pipeline kernel only: PLAIN with all four time levels stored (source field of the backward loop)
    FDW_MODE_DD_RECV = 6 // its receiver pass (rtm_main.cpp:197-220) + img += stored source field * CURRENT receiver field (rtm_main.cpp:224-230)
};

constexpr int kMaxFastHalfOrder = 4;   // register-window kernel is instantiated for order 2,4,6,8
constexpr int kMaxPipeSteps = 4;       // time levels of one pass of the wave-pipeline kernel

// Everything one launch needs, passed by value (lives in SGPRs / the scalar cache).
// All row indices are LOCAL rows of this device's slab; the host translates global extents.
struct StepArgs {
    const float* p;        // [nxl][pitch] newest field (read only in this launch)
    float* pp;             // [nxl][pitch] older field, overwritten with the new one (or Laplacian out)
    float* out;            // NULL, or where the new field goes INSTEAD of over pp (pp is then only read: stored-wavefield loops keep every field)
    const float* v2;       // [nxl][pitch] squared velocity
    const float* psrc;     // IMG: source wavefield to correlate with; BACK: F_{k-1}, the newer source field (read only)
    float* fpp;            // BACK: F_{k-2}, overwritten with the reconstructed F_k
    int img_z1;            // IMG: columns >= img_z1 lie outside kernel_img's launch extent (R:133-144) and keep their image value
    float* img;            // IMG: image accumulator on the extended grid
    const float* taperz;   // [ztap] z damping factors
    const float* txfac;    // [nxl] per-row x damping factor (1.0f where none applies)
    const float* inj;      // INJ=1: one source sample; INJ=2: inj_n receiver samples of this step
    const float* gcx;      // generic kernel: device copies of the scaled coefficients
    const float* gcz;
    int pitch, nxl;
    int r0, r1;            // rows updated by this launch (already clipped to rows < upd_x1)
    int lap_x0, lap_x1;    // rows / columns where the Laplacian is evaluated (0 elsewhere)
    int lap_z0, lap_z1;
    int upd_z1;            // columns >= upd_z1 keep their old value (reference launch extent)
    int ztap, tz_x1;       // damped strip: z < ztap; rows < tz_x1 get the z factor
    int xt_lo, xt_hi;      // rows in [xt_lo, xt_hi) have txfac == 1 and get the z factor (the common case)
    int pp_twice;          // pp owes the previous step's T() (all steps but the first after an upload)
    int inj_x, inj_z, inj_n;
    int xchunk, wz, nzblk, nblk, nper;  // launch geometry (fast kernel)
    float dt2;
    float cx[2 * kMaxFastHalfOrder + 1], cz[2 * kMaxFastHalfOrder + 1];
    float c0;              // FAST numerics: cz[H] + cx[H], the weight of the centre point (formed on the host in fp32)
    int numerics;          // 0 EXACT (the reference's operations), 1 FAST (symmetric sums + fma; fdw_device.h): picks the instantiation
    // FDW_MODE_MOD only: cz holds the UNSCALED weights, the spacings come separately (fd.c:24-36 scales per term)
    float dx2inv, dz2inv;
    float gw[4][4];        // expf(-(i*i + j*j)): the 7x7 Gaussian point source of ptsrc.c:49-55
    float* rec;            // this step's trace samples [rec_n]: rec[r - rec_x0] = p(r, rec_z)   (mod_main.cpp:155-157)
    int rec_z, rec_x0, rec_n;
    // batch of independent shots on one geometry (gridDim.y = shots; 0 / 1 = a single shot): shot b works on field pointers + b * bstride,
    // samples inj + b * inj_bstride, source row inj_x + b * inj_dx
    // (v2 + b * v2_bstride: one model per shot in the RTM loop, one for all in the modelling loop; rec + b * rec_bstride)
    long long bstride, inj_bstride, v2_bstride, rec_bstride;
    int inj_dx, nbatch;
};

// Two-steps-per-pass kernel (temporal blocking, order 8, forward mode): see fdw_step2_kernel.
struct Step2Args {
    const float* p;        // u^n
    const float* pp;       // u^{n-1}
    const float* v2;
    float* out1;           // u^{n+1}
    float* out2;           // u^{n+2}
    const float* taperz;
    const float* txfac;
    const float* inj;      // FWD: two source samples srce[it], srce[it+1]; RECV: receiver samples of iteration it (step 1)
    const float* inj2;     // RECV: receiver samples of iteration it+1 (step 2)
    const float* psrc_a;   // RECV: source wavefield of iteration it   (imaged against u^{n+1})
    const float* psrc_b;   // RECV: source wavefield of iteration it+1 (imaged against u^{n+2})
    float* img;            // RECV: image accumulator on the extended grid (in place, owned cells only)
    int img_z1;            // RECV: columns >= img_z1 lie outside kernel_img's launch extent and keep their image value
    int pitch, nxl;
    int r0, r1;            // rows whose u^{n+1}, u^{n+2} this launch produces
    int r0b, r1b, chunks_a; // pipeline kernel only: optional second row range (chunks >= chunks_a walk [r0b, r1b))
    int lap_x0, lap_x1, lap_z0, lap_z1;
    int upd_x1, upd_z1;
    int ztap, tz_x1, xt_lo, xt_hi;
    int zt_lo, zt_hi;      // pipeline kernel: columns [zt_lo, zt_hi) carry no damping factor (zt_hi < 0: no x factor outside the damped strip either)
    int pp_twice;
    int inj_x, inj_z, inj_n;
    int xchunk, nstrip, nzblk, nblk, nper;
    float dt2;
    float cx[2 * kMaxFastHalfOrder + 1], cz[2 * kMaxFastHalfOrder + 1];
    float c0;              // FAST numerics: cz[H] + cx[H]
    int numerics;          // 0 EXACT, 1 FAST (see StepArgs)
    // pipeline kernel in FDW_MODE_MOD only (see StepArgs): unscaled weights in cz, spacings, Gaussian source weights, trace samples
    float dx2inv, dz2inv;
    float gw[4][4];
    float* rec;            // [steps of the pass][rec_n]: wave k writes row k
    int rec_z, rec_x0, rec_n;
    // pipeline kernel, backward loop (fd_back, R:302-339), two passes per kPipeSteps iterations:
    //   FDW_MODE_PLAIN_ALL  the source field: like PLAIN, but EVERY wave stores its time level (wave 0 -> lvl0, wave 1 -> lvl1, then out1, out2),
    //                       because the imaging condition of iteration it+k needs F_{it+k} at every point
    //   FDW_MODE_RECV       the receiver field: wave k adds the trace samples of iteration it+k (inj + k * inj_stride) on column inj_z of rows
    //                       [inj_x, inj_x+inj_n) and multiplies its new row with the same row of plev[k] = F_{it+k}; the four products are
    //                       added to the image row in iteration order as it travels from wave to wave through an LDS FIFO
    float *lvl0, *lvl1;
    const float* plev[kMaxPipeSteps];
    int inj_stride;
    //   FDW_MODE_BACK4      both in one pass (fdw_back4_kernel): p / pp / out1 / out2 are the source field's, these the receiver field's
    const float *rp, *rpp;
    float *rout1, *rout2;
};
hipError_t launch_step2(const Step2Args& a, int half_order, int mode, hipStream_t s);

// kPipeSteps time steps per pass (wave pipeline through LDS, order 8, FWD / PLAIN): see fdw_stepn_kernel.  Uses Step2Args with
// out1 = u^{n+kPipeSteps-1}, out2 = u^{n+kPipeSteps}, inj -> kPipeSteps source samples, nstrip = strips of 64-2*kPipeSteps cells,
// nblk = nstrip * chunks workgroups of kPipeSteps waves.
#ifndef FDW_PIPE_STEPS
#define FDW_PIPE_STEPS 4      // experiments: 2 = a pipeline of two waves (build_variants.sh)
#endif
constexpr int kPipeSteps = FDW_PIPE_STEPS;
hipError_t launch_stepn(const Step2Args& a, int half_order, int mode, hipStream_t s);

hipError_t launch_step_fast(const StepArgs& a, int half_order, int mode, int prefetch, hipStream_t s);
hipError_t launch_step_generic(const StepArgs& a, int half_order, int mode, hipStream_t s);
hipError_t launch_taper_finalize(float* f, const float* taperz, const float* txfac, int pitch, int nxl, int ztap,
                                 int tz_x1, hipStream_t s);
hipError_t launch_selftest(const float* src, float* out, hipStream_t s);
// receiver rows the reference injects and images but never time-steps (truncated launch extents with a narrow x border): see fdw_static_rows_kernel
hipError_t launch_static_rows(float* pp, const float* psrc, float* img, const float* samples, int pitch, int row0, int nrows, int gz,
                              int img_z0, int img_z1, hipStream_t s);
hipError_t launch_image_laplacian(const float* d_img, float* d_out, int nx, int nz, float dx, float dz, hipStream_t s);
// d_part: 3 * nblocks doubles of scratch; d_out: {sum (a-b)^2, sum b^2, max |b|} carried in double and, with `serial`, [3], [4] = the two
// sums as the reference tool forms them (one term after the other into fp32 sums)
hipError_t launch_image_compare(const float* a, const float* b, size_t n, float* diff, double* d_part, int nblocks, double* d_out, int serial, hipStream_t s);

// ---- random-border velocity model on the device (fdw_border.hip) ----
constexpr int kRandLag = 31;          // glibc TYPE_3: y[t] = y[t-31] + y[t-3]
struct RandWindow {                   // (y[K-31] .. y[K-1]) for a stream position K
    unsigned w[kRandLag];
};
struct BorderArgs {
    const float* vp;                  // interior model [nx][nz]
    const int* draws;                 // the rand() values one extendvel_linear call consumes, in its order
    float* vel;                       // extended model [nxe][pitch] (may be null)
    float* vel2;                      // its square [nxe][pitch]
    int nx, nz, nxb, nzb, pitch;
};
struct RandBase {                     // y[K-31] .. y[K+29]: the window at the position of draw 0 and the next 30 words
    unsigned y[2 * kRandLag - 1];
};
// d_tab: x^(31 l) mod P for l < 64 as [31][64] (coefficient-major), then x^(31 64 h) mod P for block h as [blocks][31];
// P = x^31 - x^28 - 1 is the characteristic polynomial of the generator.  d_out[i] = draw number i counted from the base.
hipError_t launch_rand_stream(const RandBase& base, const unsigned* d_tab, long long n, int* d_out, hipStream_t s);
hipError_t launch_extendvel(const BorderArgs& a, hipStream_t s);
// d_in [nshots][nx][nt] -> d_out [nshots][nt][nx]
hipError_t launch_gather_transpose(const float* d_in, float* d_out, int nx, int nt, int nshots, hipStream_t s);

}  // namespace fdw
