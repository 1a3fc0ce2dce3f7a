// fdw_api.cpp -- the C ABI of libfdwave.so (include/fdwave.h) on top of the gfx950 kernels.
//
// Replaces the reference's L2 seam: fd_init / fd_init_cuda / write_buffers / fd_forward / fd_back of
// cuda_reference_RTM/src/fd-code.cu (R) and fd_init / the single launch of
// cuda_reference_stencil_computation/fd-source-code.cu (S).  There is no CPU compute path in this
// library: if HIP cannot be initialised on a gfx950 device fdw_create fails.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include "fdw_internal.h"
#include "fdw_kernels.h"
#include "fdwave.h"

using namespace fdw;

// ------------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

int fdw_fail(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}
#define fail fdw_fail

#define HIP_TRY(call)                                                                                     \
    do {                                                                                                  \
        hipError_t e_ = (call);                                                                           \
        if (e_ != hipSuccess) return fail(FDW_EHIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                                          __FILE__, __LINE__);                                            \
    } while (0)

extern "C" const char* fdw_last_error(void) { return g_err; }
extern "C" int fdw_version(void) { return FDW_VERSION; }

// ------------------------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------------------------
struct fdw_ctx {
    fdw_params prm{};
    fdw_slab slab{};
    int device = 0;
    int h = 0, pitch = 0, nxl = 0;
    int nx = 0, nz = 0;            // interior size of the GLOBAL grid
    int xlim = 0, zlim = 0, ztap = 0;  // global launch extents (R:185-195)
    // local-row translations
    int lap_x0 = 0, lap_x1 = 0, lap_z0 = 0, lap_z1 = 0;     // RTM modes
    int slap_x0 = 0, slap_x1 = 0, slap_z0 = 0, slap_z1 = 0; // stencil program (full interior)
    int upd_x1 = 0, upd_z1 = 0, tz_x1 = 0, xt_lo = 0, xt_hi = 0;
    float dt2 = 0.f, dx2inv = 0.f, dz2inv = 0.f;
    float c0 = 0.f;                // FAST numerics: cz[h] + cx[h]
    float fcx[FDW_MAX_ORDER + 1]{}, fcz[FDW_MAX_ORDER + 1]{};  // FAST numerics of dialects 1, 2: the weights with their spacing folded in (fp32 products)
    float cx[FDW_MAX_ORDER + 1]{}, cz[FDW_MAX_ORDER + 1]{};    // RTM weights (C libm variant unless coef_cxx); dialect MOD: unscaled
    float* d_rec = nullptr;     // dialect MOD: trace samples [nt][nx] of one shot
    size_t rec_cap = 0;
    std::vector<float> taper_x, taper_z, txfac;
    // device tables
    float *d_taperz = nullptr, *d_txfac = nullptr, *d_gcx = nullptr, *d_gcz = nullptr;
    hipStream_t stream = nullptr;
    // lazily allocated work buffers of the host-array API
    float* fld[10] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};      // 8, 9: source-field levels of a backward pipeline pass
    float *d_v2 = nullptr, *d_img = nullptr, *d_srce = nullptr, *d_dobs = nullptr;
    size_t srce_cap = 0, dobs_cap = 0;
    // tuning
    int xchunk = 0, wz = 0, use_generic = 0, prefetch = 0, force_edge = 0, xchunk2 = 0;
    int tb = 0;   // two-steps-per-pass kernel: 0 auto (large grids), 1 always, -1 never
    // random-border model generated on the device (row f4): interior model, one call's draws, jump tables, the extended model
    float *d_vp = nullptr, *d_vpe = nullptr;
    int* d_draws = nullptr;
    long long draws_cap = 0;
    unsigned* d_jump = nullptr;   // the generator's jump table (fdw_border.hip), sized for njump blocks of 64 threads
    int njump = 0;
    bool model_resident = false, v2_resident = false;
    // a batch of shots through one launch per time step (fdw_shot_batch): per-shot copies of the eight fields, v2, image, gather
    int nbatch = 1, batch_dsx = 0, batch_cap = 0, batch_nt = 0;
    bool batch_all = false;      // the batch buffers hold the RTM loop's full set (else: the modelling loop's two fields + gathers)
    float* bfld[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    float *b_v2 = nullptr, *b_img = nullptr, *b_dobs = nullptr;
    float* d_raw = nullptr;      // gathers as the caller holds them ([shot][nx][nt]) before the transposition on the device
    size_t raw_cap = 0;
    int no_fused_back = 0;   // experiments / tests: backward iterations as two launches (source step, receiver step) -- FDW_NO_FUSED_BACK=1
    int no_back_pipe = 0;    // experiments / tests: no wave-pipeline passes in the backward loop -- FDW_NO_BACK_PIPE=1
    int no_back_fused = 0;   // experiments / tests: the backward pipeline as two passes (source field, receiver field) instead of the fused kernel -- FDW_NO_BACK_FUSED=1
    size_t store_budget = 0; // fdw_rtm_stored_shot: bytes the stored source fields may take (0: whatever the device grants); fdw_set_store_budget
    int store_segments = 0;  // ... segments the last such shot was cut into (1: every field kept)
};

static size_t field_elems(const fdw_ctx* c) { return (size_t)c->nxl * (size_t)c->pitch; }

extern "C" int fdw_pitch(const fdw_ctx* c) { return c ? c->pitch : 0; }
extern "C" size_t fdw_field_bytes(const fdw_ctx* c) { return c ? field_elems(c) * sizeof(float) : 0; }

static int is_full_grid(const fdw_ctx* c) { return c->slab.x_off == 0 && c->nxl == c->prm.nxe; }

static hipStream_t pick_stream(fdw_ctx* c, void* s) { return s ? (hipStream_t)s : c->stream; }

static int alloc_zero(float** p, size_t elems)
{
    if (*p) return FDW_OK;
    hipError_t e = hipMalloc((void**)p, elems * sizeof(float));
    if (e != hipSuccess) return fail(FDW_ENOMEM, "hipMalloc(%zu bytes) failed: %s", elems * sizeof(float), hipGetErrorString(e));
    HIP_TRY(hipMemset(*p, 0, elems * sizeof(float)));
    HIP_TRY(hipDeviceSynchronize());   // the context stream is non-blocking: do not let it overtake the memset
    return FDW_OK;
}

static int ensure_work_buffers(fdw_ctx* c, int nfields, bool need_img)
{
    for (int i = 0; i < nfields; i++) {
        int rc = alloc_zero(&c->fld[i], field_elems(c));
        if (rc) return rc;
    }
    int rc = alloc_zero(&c->d_v2, field_elems(c));
    if (rc) return rc;
    if (need_img) {
        rc = alloc_zero(&c->d_img, field_elems(c));
        if (rc) return rc;
    }
    return FDW_OK;
}

static int ensure_cap(float** p, size_t* cap, size_t elems)
{
    if (*cap >= elems && *p) return FDW_OK;
    if (*p) (void)hipFree(*p);
    *p = nullptr;
    *cap = 0;
    hipError_t e = hipMalloc((void**)p, std::max<size_t>(elems, 1) * sizeof(float));
    if (e != hipSuccess) return fail(FDW_ENOMEM, "hipMalloc failed: %s", hipGetErrorString(e));
    *cap = elems;
    return FDW_OK;
}

// ------------------------------------------------------------------------------------------------
// create / destroy
// ------------------------------------------------------------------------------------------------
static int validate(const fdw_params* p, const fdw_slab* s)
{
    if (!p) return fail(FDW_EINVAL, "params is NULL");
    if (p->order < 2 || p->order > FDW_MAX_ORDER || (p->order & 1))
        return fail(FDW_EINVAL, "order=%d must be even and in [2,%d]", p->order, FDW_MAX_ORDER);
    if (p->nxe <= p->order || p->nze <= p->order)
        return fail(FDW_EINVAL, "grid %dx%d too small for order %d", p->nxe, p->nze, p->order);
    if (p->nxb < 0 || p->nzb < 0 || 2 * p->nxb >= p->nxe || 2 * p->nzb >= p->nze)
        return fail(FDW_EINVAL, "borders nxb=%d nzb=%d do not leave an interior in %dx%d", p->nxb, p->nzb, p->nxe, p->nze);
    if ((p->nxb > 0 || p->nzb > 0) && !(p->fac > 0.0f && p->fac <= 1.0f))
        return fail(FDW_EINVAL, "fac=%g must be in (0,1]", (double)p->fac);
    if (!(p->dx > 0.0f) || !(p->dz > 0.0f)) return fail(FDW_EINVAL, "dx, dz must be positive");
    if (p->nt < 0) return fail(FDW_EINVAL, "nt=%d is negative", p->nt);
    if (p->dialect < FDW_DIALECT_RTM || p->dialect > FDW_DIALECT_RTM_STORED) return fail(FDW_EINVAL, "dialect=%d is unknown", p->dialect);
    if (p->dialect != FDW_DIALECT_RTM && p->order > 2 * kMaxFastHalfOrder)
        return fail(FDW_EINVAL, "dialects 1 and 2 are built for orders 2..%d", 2 * kMaxFastHalfOrder);
    if (p->numerics != FDW_NUMERICS_EXACT && p->numerics != FDW_NUMERICS_FAST) return fail(FDW_EINVAL, "numerics=%d is unknown", p->numerics);
    if (s->nxl <= p->order || s->x_off < 0 || s->x_off + s->nxl > p->nxe)
        return fail(FDW_EINVAL, "slab [%d,%d) does not fit the grid (nxe=%d) or is thinner than the stencil",
                    s->x_off, s->x_off + s->nxl, p->nxe);
    return FDW_OK;
}

extern "C" int fdw_create_slab(const fdw_params* prm, const fdw_slab* slab, int device, fdw_ctx** out)
{
    if (!out) return fail(FDW_EINVAL, "out is NULL");
    *out = nullptr;
    if (!slab) return fail(FDW_EINVAL, "slab is NULL");
    int rc = validate(prm, slab);
    if (rc) return rc;

    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(FDW_ENODEVICE, "no HIP device available (%s); libfdwave has no CPU path",
                    e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
    if (device < 0 || device >= ndev) return fail(FDW_ENODEVICE, "device %d out of range (have %d)", device, ndev);
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, device)) != hipSuccess)
        return fail(FDW_ENODEVICE, "hipGetDeviceProperties: %s", hipGetErrorString(e));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(FDW_ENODEVICE, "device %d is %s; libfdwave is built for gfx950 (MI355X) only", device, prop.gcnArchName);
    if ((e = hipSetDevice(device)) != hipSuccess) return fail(FDW_ENODEVICE, "hipSetDevice: %s", hipGetErrorString(e));

    fdw_ctx* c = new (std::nothrow) fdw_ctx();
    if (!c) return fail(FDW_ENOMEM, "out of host memory");
    c->prm = *prm;
    c->slab = *slab;
    c->device = device;
    c->h = prm->order / 2;
    c->nxl = slab->nxl;
    c->nx = prm->nxe - 2 * prm->nxb;
    c->nz = prm->nze - 2 * prm->nzb;
    c->pitch = ((prm->nze + 63) / 64) * 64;  // 256-B aligned rows: every lane's float4 is aligned
    if (const char* nf = getenv("FDW_NO_FUSED_BACK")) c->no_fused_back = atoi(nf);
    if (const char* nf = getenv("FDW_NO_BACK_PIPE")) c->no_back_pipe = atoi(nf);
    if (const char* nf = getenv("FDW_NO_BACK_FUSED")) c->no_back_fused = atoi(nf);
    if (const char* pad = getenv("FDW_PITCH_PAD")) {   // experiment knob: extra floats per row (multiple of 4)
        const int extra = atoi(pad);
        if (extra > 0 && extra % 4 == 0) c->pitch += extra;
    }

    // launch extents, R:185-195 (the int assignment truncates before ceil)
    const bool mod = prm->dialect != FDW_DIALECT_RTM;              // the CPU-serial sibling's arithmetic (fd.c)
    const bool four_sided = prm->dialect == FDW_DIALECT_MOD;       // taper_apply (mod_main) vs taper_apply2 (rtm_main)
    if (mod) {                    // fd.c:24-46 and taper.c walk the whole array; taper_apply's z table spans the whole row
        c->xlim = prm->nxe;
        c->zlim = prm->nze;
        c->ztap = four_sided ? prm->nze : prm->nzb;
    } else if (prm->compat) {
        c->xlim = 8 * (prm->nxe / 8);
        c->zlim = 8 * (prm->nze / 8);
        c->ztap = 8 * (prm->nzb / 8);
    } else {
        c->xlim = prm->nxe;
        c->zlim = prm->nze;
        c->ztap = prm->nzb;
    }
    const int h = c->h, xo = slab->x_off;
    // kernel_lap covers i = h + t, t < xlim threads, and i < nxe - h (R:56-64)
    const int glx0 = h, glx1 = std::min(prm->nxe - h, h + c->xlim);
    c->lap_x0 = std::max(h, glx0 - xo);
    c->lap_x1 = std::min(c->nxl - h, glx1 - xo);
    c->lap_z0 = h;
    c->lap_z1 = std::min(prm->nze - h, h + c->zlim);
    // the stencil program rounds its grid up to 32 (S:231-238): whole interior
    c->slap_x0 = std::max(h, h - xo);
    c->slap_x1 = std::min(c->nxl - h, prm->nxe - h - xo);
    c->slap_z0 = h;
    c->slap_z1 = prm->nze - h;
    c->upd_x1 = std::max(0, std::min(c->nxl, c->xlim - xo));
    c->upd_z1 = c->zlim;
    c->tz_x1 = c->upd_x1;
    // rows outside [xt_lo, xt_hi) carry an x damping factor or miss the z factor (corner tiles)
    c->xt_lo = std::max(0, std::min(c->nxl, prm->nxb - xo));
    c->xt_hi = std::max(c->xt_lo, std::min(c->nxl, std::min(prm->nxe - prm->nxb, c->xlim) - xo));

    // derived constants, R:203-217 (double quotient narrowed to float; float*float scaling)
    const float dx2inv = (1. / prm->dx) * (1. / prm->dx);
    const float dz2inv = (1. / prm->dz) * (1. / prm->dz);
    c->dt2 = prm->dt * prm->dt;
    c->dx2inv = dx2inv;
    c->dz2inv = dz2inv;
    float w[FDW_MAX_ORDER + 1];
    fdw_calc_coefs(prm->order, mod ? 1 : prm->coef_cxx, w);
    for (int io = 0; io <= prm->order; io++) {
        c->cz[io] = mod ? w[io] : dz2inv * w[io];      // fd.c:33-34 scales inside every term
        c->cx[io] = mod ? w[io] : dx2inv * w[io];
    }
    for (int io = 0; io <= prm->order; io++) {      // FAST numerics of the sibling's dialects: c_k * d?2inv once, instead of inside every term
        c->fcz[io] = mod ? w[io] * dz2inv : c->cz[io];
        c->fcx[io] = mod ? w[io] * dx2inv : c->cx[io];
    }
    c->c0 = c->fcz[c->h] + c->fcx[c->h];    // FAST numerics: the centre point's weight (one fp32 add, part of that mode's definition)
    c->taper_x.assign(std::max(prm->nxb, 1), 1.0f);
    c->taper_z.assign(std::max(prm->nzb, 1), 1.0f);
    if (mod) {
        fdw_mod_taper_tables(prm->nxb, prm->nzb, prm->fac, c->taper_x.data(), c->taper_z.data());
        if (four_sided) {         // taper.c:50-56 as ONE factor per column: top strip, 1.0f, mirrored bottom strip
            std::vector<float> full(prm->nze, 1.0f);
            for (int i = 0; i < prm->nzb; i++) {
                full[i] = c->taper_z[i];
                full[prm->nze - 1 - i] = c->taper_z[i];
            }
            c->taper_z.swap(full);
        }                         // taper_apply2 (taper.c:68-83) has the CUDA path's shape: z factors on the top strip, x factors inside it only
    } else {
        if (prm->nxb > 0) fdw_taper_tables(prm->nxb, 0, prm->fac, c->taper_x.data(), nullptr);
        if (prm->nzb > 0) fdw_taper_tables(0, prm->nzb, prm->fac, nullptr, c->taper_z.data());
    }
    // per-row x factor: thread i < nxb (and < xlim) scales columns i and nxe-1-i by taperx[i] (R:108-115)
    c->txfac.assign(c->nxl, 1.0f);
    for (int l = 0; l < c->nxl; l++) {
        const int g = xo + l, gm = prm->nxe - 1 - g;
        if (g < prm->nxb && g < c->xlim) c->txfac[l] = c->taper_x[g];
        else if (gm < prm->nxb && gm < c->xlim) c->txfac[l] = c->taper_x[gm];
    }

#define CREATE_TRY(call)                                                                                  \
    do {                                                                                                  \
        hipError_t e2_ = (call);                                                                          \
        if (e2_ != hipSuccess) {                                                                          \
            fail(FDW_EHIP, "%s failed: %s", #call, hipGetErrorString(e2_));                               \
            fdw_destroy(c);                                                                               \
            return FDW_EHIP;                                                                              \
        }                                                                                                 \
    } while (0)
    CREATE_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    const size_t ntz = std::max(c->ztap, 1);
    CREATE_TRY(hipMalloc((void**)&c->d_taperz, ntz * sizeof(float)));
    CREATE_TRY(hipMalloc((void**)&c->d_txfac, c->nxl * sizeof(float)));
    CREATE_TRY(hipMalloc((void**)&c->d_gcx, (FDW_MAX_ORDER + 1) * sizeof(float)));
    CREATE_TRY(hipMalloc((void**)&c->d_gcz, (FDW_MAX_ORDER + 1) * sizeof(float)));
    CREATE_TRY(hipMemcpy(c->d_taperz, c->taper_z.data(), std::min<size_t>(ntz, c->taper_z.size()) * sizeof(float), hipMemcpyHostToDevice));
    CREATE_TRY(hipMemcpy(c->d_txfac, c->txfac.data(), c->nxl * sizeof(float), hipMemcpyHostToDevice));
    CREATE_TRY(hipMemcpy(c->d_gcx, c->cx, (FDW_MAX_ORDER + 1) * sizeof(float), hipMemcpyHostToDevice));
    CREATE_TRY(hipMemcpy(c->d_gcz, c->cz, (FDW_MAX_ORDER + 1) * sizeof(float), hipMemcpyHostToDevice));
#undef CREATE_TRY
    *out = c;
    return FDW_OK;
}

extern "C" int fdw_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

extern "C" int fdw_device_usable(int device)
{
    const int n = fdw_device_count();
    if (device < 0 || device >= n) return fail(FDW_ENODEVICE, "device %d does not exist (%d visible)", device, n);
    hipDeviceProp_t prop;
    hipError_t e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) return fail(FDW_ENODEVICE, "hipGetDeviceProperties(%d): %s", device, hipGetErrorString(e));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return fail(FDW_ENODEVICE, "device %d is %s; libfdwave is built for gfx950 (MI355X) only", device, prop.gcnArchName);
    if ((e = hipSetDevice(device)) != hipSuccess) return fail(FDW_ENODEVICE, "hipSetDevice(%d): %s", device, hipGetErrorString(e));
    return FDW_OK;
}

extern "C" int fdw_create(const fdw_params* prm, int device, fdw_ctx** out)
{
    if (!prm) return fail(FDW_EINVAL, "params is NULL");
    fdw_slab s{0, prm->nxe};
    return fdw_create_slab(prm, &s, device, out);
}

extern "C" void fdw_destroy(fdw_ctx* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    float* bufs[] = {c->d_taperz, c->d_txfac, c->d_gcx, c->d_gcz, c->fld[0], c->fld[1], c->fld[2], c->fld[3],
                     c->fld[4], c->fld[5], c->fld[6], c->fld[7], c->fld[8], c->fld[9], c->d_v2, c->d_img, c->d_srce, c->d_dobs, c->d_rec,
                     c->d_vp, c->d_vpe, (float*)c->d_draws, (float*)c->d_jump, c->bfld[0], c->bfld[1], c->bfld[2], c->bfld[3],
                     c->bfld[4], c->bfld[5], c->bfld[6], c->bfld[7], c->b_v2, c->b_img, c->b_dobs, c->d_raw};
    for (float* b : bufs)
        if (b) (void)hipFree(b);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

// ------------------------------------------------------------------------------------------------
// launch geometry + one step
// ------------------------------------------------------------------------------------------------
// ring size of the register-ring kernel for (half order, prefetch) -- must match RingGeom in fdw_device.h
static int ring_rows(int h, int pf) { return ((2 * h + pf + pf - 1) / pf) * pf; }
static int effective_prefetch(const fdw_ctx* c) { return (c->h == 4 && c->prefetch >= 1 && c->prefetch <= 3) ? c->prefetch : 2; }

static void fill_geometry(const fdw_ctx* c, StepArgs& a, int rows, int batch = 1)
{
    const int nstrips = (c->pitch + 255) / 256;
    int wz = c->wz;
    if (wz != 1 && wz != 2 && wz != 4) wz = nstrips >= 4 ? 4 : (nstrips >= 2 ? 2 : 1);
    int xchunk = c->xchunk;
    if (xchunk <= 0) {
        // Measured on MI355X (scripts/probe_step.py, order 8): once every tile runs the same
        // branch-free body the step time is flat in the chunk length to within run-to-run noise,
        // with a shallow optimum at short chunks (many waves per CU slot = good balance; the 2H
        // halo rows a neighbouring chunk re-reads come from L2 / Infinity Cache, not HBM).
        //   HBM-resident grids (>= 8192^2): one ring turn (10 rows at prefetch 2)
        //   Infinity-Cache-resident grids (4096^2): 24 rows
        //   small decks (new_mod 415x295 ... 2048^2): as many waves as the grid gives, down to ONE row per
        //   wave -- a launch is a latency chain (prologue + rows), 5.1 us/step at 1 row vs 12.7 at 8 on new_mod
        const long strip_rows = (long)rows * nstrips * batch;   // a batch of shots counts as one grid of that many rows
        if (strip_rows >= 200000) xchunk = ring_rows(c->h, effective_prefetch(c));
        else if (strip_rows >= 60000) xchunk = 24;   // 4096^2 class (Infinity-Cache resident)
        else xchunk = (int)std::min<long>(std::max<long>((strip_rows + 4095) / 4096, 1), 10);   // >= ~4096 waves, down to 1 row each;
                                                                                                  // a 1056 x 8192 slab (N = 8) lands on 9: 293 Gpt/s vs 268 at 24
    }
    a.xchunk = xchunk;
    a.wz = wz;
    a.nzblk = (nstrips + wz - 1) / wz;
    const int chunks = (rows + xchunk - 1) / xchunk;
    const int wpx = 4 / wz;
    const int nxblk = (chunks + wpx - 1) / wpx;
    a.nblk = a.nzblk * nxblk;
    a.nper = (a.nblk + 7) / 8;
}

static int step_impl(fdw_ctx* c, int mode, const float* d_p, float* d_pp, const float* d_v2, int r0, int r1,
                     int pp_twice, const float* d_inj, int inj_x_global, int inj_z, const float* d_psrc, float* d_img,
                     hipStream_t s, float* d_rec_row = nullptr, int rec_z = 0, float* d_fpp = nullptr, float* d_out = nullptr)
{
    const bool lap = (mode == FDW_MODE_LAP);
    if (!d_p || !d_pp) return fail(FDW_EINVAL, "step: field pointer is NULL");
    if (!lap && !d_v2) return fail(FDW_EINVAL, "step: v2 is NULL");
    if ((mode == FDW_MODE_RECV || mode == FDW_MODE_DD_RECV || mode == FDW_MODE_BACK) && (!d_psrc || !d_img || !d_inj)) return fail(FDW_EINVAL, "step: RECV needs d_inj, d_psrc and d_img");
    if (mode == FDW_MODE_BACK && (!d_fpp || d_fpp == d_pp || d_fpp == d_p || d_psrc == d_pp || c->h > kMaxFastHalfOrder || c->use_generic))
        return fail(FDW_EINVAL, "step: BACK needs the older source field as a fourth, distinct buffer and an order <= 8");
    if (mode < FDW_MODE_FWD || mode > FDW_MODE_BACK) return fail(FDW_EINVAL, "step: unknown mode %d", mode);
    const int mode_dialect = mode == FDW_MODE_MOD ? FDW_DIALECT_MOD : ((mode == FDW_MODE_DD_FWD || mode == FDW_MODE_DD_RECV) ? FDW_DIALECT_RTM_STORED : FDW_DIALECT_RTM);
    if (mode_dialect != c->prm.dialect)
        return fail(FDW_ESTATE, "step: mode %d does not belong to this context's dialect %d", mode, c->prm.dialect);
    if (r0 < 0 || r1 > c->nxl || r0 > r1) return fail(FDW_EINVAL, "step: rows [%d,%d) outside the slab (%d rows)", r0, r1, c->nxl);
    // d_out (the new field stored somewhere else than over pp) is honoured by the register-ring kernel's DD_FWD instantiation only -- the one
    // caller, fdw_rtm_stored_shot, whose store slots are zero-filled so that cells the kernel never stores equal what an in-place update
    // would have left there.  Anything else would silently update in place: refuse.
    if (d_out && (mode != FDW_MODE_DD_FWD || c->h > kMaxFastHalfOrder || c->use_generic || d_out == d_p || d_out == d_pp))
        return fail(FDW_EINVAL, "step: a separate output array is supported by the DD_FWD register-ring kernel only, and must not alias the inputs");

    StepArgs a{};
    a.p = d_p; a.pp = d_pp; a.v2 = d_v2; a.psrc = d_psrc; a.img = d_img; a.fpp = d_fpp;
    a.out = d_out;            // NULL: in place over pp
    a.taperz = c->d_taperz; a.txfac = c->d_txfac; a.inj = d_inj; a.gcx = c->d_gcx; a.gcz = c->d_gcz;
    a.pitch = c->pitch; a.nxl = c->nxl;
    a.r0 = r0;
    a.r1 = lap ? r1 : std::min(r1, c->upd_x1);   // rows >= xlim are never time-stepped (R:83-87)
    a.lap_x0 = lap ? c->slap_x0 : c->lap_x0; a.lap_x1 = lap ? c->slap_x1 : c->lap_x1;
    a.lap_z0 = lap ? c->slap_z0 : c->lap_z0; a.lap_z1 = lap ? c->slap_z1 : c->lap_z1;
    a.upd_z1 = c->upd_z1;
    a.ztap = c->ztap; a.tz_x1 = c->tz_x1; a.xt_lo = c->xt_lo; a.xt_hi = c->xt_hi;
    a.pp_twice = pp_twice ? 1 : 0;
    a.img_z1 = c->prm.nzb + std::min(c->nz, c->zlim);      // kernel_img: interior columns j < zlim && j < nz (R:133-144)
    a.inj_x = -1; a.inj_z = inj_z; a.inj_n = 0;
    if (mode == FDW_MODE_DD_FWD || mode == FDW_MODE_DD_RECV) { a.dx2inv = c->dx2inv; a.dz2inv = c->dz2inv; }
    if ((mode == FDW_MODE_FWD || mode == FDW_MODE_DD_FWD) && d_inj && inj_x_global >= 0) {
        if (inj_z < 0 || inj_z >= c->prm.nze || inj_x_global >= c->prm.nxe)
            return fail(FDW_EINVAL, "step: source (%d,%d) outside the grid", inj_x_global, inj_z);
        a.inj_x = inj_x_global - c->slab.x_off;   // may fall outside this slab: then no row matches
        if (a.inj_x >= c->upd_x1 && a.inj_x < c->nxl)
            return fail(FDW_EINVAL, "step: source row %d lies in rows the reference never time-steps (>= %d)", inj_x_global, c->xlim);
    } else if (mode == FDW_MODE_MOD) {
        if (d_inj) {
            if (inj_z < 0 || inj_z >= c->prm.nze || inj_x_global < 0 || inj_x_global >= c->prm.nxe)
                return fail(FDW_EINVAL, "step: source (%d,%d) outside the grid", inj_x_global, inj_z);
            a.inj_x = inj_x_global - c->slab.x_off;
        } else {
            a.inj_x = -1000000;
        }
        for (int i = 0; i < 4; i++)
            for (int j = 0; j < 4; j++) a.gw[i][j] = expf(-(float)(i * i) - (float)(j * j));   // ptsrc.c:53 with exp(float) as g++ resolves it
        a.dx2inv = c->dx2inv; a.dz2inv = c->dz2inv;
        a.rec = d_rec_row; a.rec_z = rec_z;
        a.rec_x0 = c->prm.nxb - c->slab.x_off; a.rec_n = c->nx;
        if (d_rec_row && (rec_z < 0 || rec_z >= c->prm.nze)) return fail(FDW_EINVAL, "step: receiver depth %d outside the grid", rec_z);
    } else if (mode == FDW_MODE_RECV || mode == FDW_MODE_DD_RECV || mode == FDW_MODE_BACK) {
        if (inj_z < 0 || inj_z >= c->prm.nze) return fail(FDW_EINVAL, "step: receiver depth %d outside the grid", inj_z);
        // receivers sit on interior columns nxb..nxb+nx-1 (R:126-129); clip to this slab.  The sibling's rtm_main offsets them by
        // nzb instead (PP[ix+nzb][gz], rtm_main.cpp:203) -- the same thing only when both borders are equally wide; kept as is.
        const int off = mode == FDW_MODE_DD_RECV ? c->prm.nzb : c->prm.nxb;
        if (off + c->nx > c->prm.nxe) return fail(FDW_EINVAL, "step: receiver rows [%d,%d) leave the grid", off, off + c->nx);
        const int g0 = off, g1 = off + std::min(c->nx, c->xlim);
        const int l0 = std::max(g0 - c->slab.x_off, 0), l1 = std::min(g1 - c->slab.x_off, c->nxl);
        a.inj_x = l0;
        a.inj_n = std::max(0, l1 - l0);
        a.inj = d_inj + (l0 + c->slab.x_off - g0);
    }
    a.dt2 = c->dt2;
    a.c0 = c->c0;
    a.numerics = c->prm.numerics;
    const bool fastw = c->prm.numerics == FDW_NUMERICS_FAST;      // (for the RTM dialect fcx / fcz are cx / cz)
    for (int io = 0; io <= 2 * kMaxFastHalfOrder; io++) {
        a.cx[io] = io <= c->prm.order ? (fastw ? c->fcx[io] : c->cx[io]) : 0.0f;
        a.cz[io] = io <= c->prm.order ? (fastw ? c->fcz[io] : c->cz[io]) : 0.0f;
    }
    if (a.r1 <= a.r0) return FDW_OK;
    if (c->nbatch > 1) {      // fdw_shot_batch: shot b = these pointers + b fields, its own gather, its own source row
        a.nbatch = c->nbatch;
        a.bstride = (long long)field_elems(c);
        const bool one_source = mode == FDW_MODE_FWD || mode == FDW_MODE_MOD;     // the wavelet is shared, its row moves; receivers: a gather each
        a.inj_bstride = one_source ? 0 : (long long)c->nx * c->prm.nt;
        a.inj_dx = one_source ? c->batch_dsx : 0;
        a.v2_bstride = mode == FDW_MODE_MOD ? 0 : a.bstride;                       // mod_main models every shot on one velocity model (M:140-174)
        a.rec_bstride = (long long)c->nx * c->batch_nt;
    }
    hipError_t e;
    if (c->h <= kMaxFastHalfOrder && !c->use_generic) {
        fill_geometry(c, a, a.r1 - a.r0, std::max(c->nbatch, 1));
        e = launch_step_fast(a, c->h, mode, effective_prefetch(c), s);
    } else {
        if (mode >= FDW_MODE_MOD) return fail(FDW_EINVAL, "step: mode %d has no generic-order kernel", mode);
        e = launch_step_generic(a, c->h, mode, s);
    }
    if (e != hipSuccess) return fail(FDW_EHIP, "kernel launch failed: %s", hipGetErrorString(e));
    if ((mode == FDW_MODE_RECV || mode == FDW_MODE_BACK) && r1 > c->upd_x1) {
        // receiver rows beyond the time-stepped rows (narrow x border + truncated extents): the reference still injects and images them
        const int s0 = std::max(a.inj_x, c->upd_x1), s1 = std::min(a.inj_x + a.inj_n, std::min(r1, c->nxl));
        if (s1 > s0) {
            e = launch_static_rows(d_pp, mode == FDW_MODE_BACK ? d_fpp : d_psrc, d_img, a.inj + (s0 - a.inj_x), c->pitch, s0, s1 - s0, inj_z,
                                   c->prm.nzb, a.img_z1, s);
            if (e != hipSuccess) return fail(FDW_EHIP, "static-row launch failed: %s", hipGetErrorString(e));
        }
    }
    return FDW_OK;
}

extern "C" int fdw_dev_step(fdw_ctx* c, int mode, const float* d_p, float* d_pp, const float* d_v2, int r0, int r1,
                            int pp_twice, const float* d_inj, int inj_x, int inj_z, const float* d_psrc, float* d_img,
                            void* stream)
{
    if (!c) return fail(FDW_EINVAL, "ctx is NULL");
    if (mode == FDW_MODE_LAP) return fail(FDW_EINVAL, "use fdw_dev_laplacian for mode 3");
    return step_impl(c, mode, d_p, d_pp, d_v2, r0, r1, pp_twice, d_inj, inj_x, inj_z, d_psrc, d_img, pick_stream(c, stream));
}

// One iteration of fd_back's loop (R:302-339) on rows [r0, r1) of the slab, on caller-owned device arrays.
extern "C" int fdw_dev_back_iter(fdw_ctx* c, int step_source, const float* d_f1, float* d_f0, const float* d_pr, float* d_ppr, const float* d_v2,
                                 int r0, int r1, int pp_twice, const float* d_samples, int gz, float* d_img, void* stream)
{
    if (!c) return fail(FDW_EINVAL, "ctx is NULL");
    if (c->prm.dialect != FDW_DIALECT_RTM) return fail(FDW_ESTATE, "fdw_dev_back_iter belongs to the RTM dialect");
    if (!d_f1 || !d_pr || !d_ppr || !d_v2 || !d_samples || !d_img || (step_source && !d_f0)) return fail(FDW_EINVAL, "back_iter: NULL buffer");
    hipStream_t s = pick_stream(c, stream);
    if (!step_source)      // iterations 0 and 1: the source field is a snapshot as it stands (R:304-314)
        return step_impl(c, FDW_MODE_RECV, d_pr, d_ppr, d_v2, r0, r1, pp_twice, d_samples, 0, gz, d_f1, d_img, s);
    if (c->h <= kMaxFastHalfOrder && !c->use_generic && !c->no_fused_back)
        return step_impl(c, FDW_MODE_BACK, d_pr, d_ppr, d_v2, r0, r1, pp_twice, d_samples, 0, gz, d_f1, d_img, s, nullptr, 0, d_f0);
    int rc = step_impl(c, FDW_MODE_PLAIN, d_f1, d_f0, d_v2, r0, r1, 0, nullptr, -1, 0, nullptr, nullptr, s);      // F_k overwrites F_{k-2} (R:317-318)
    if (rc) return rc;
    return step_impl(c, FDW_MODE_RECV, d_pr, d_ppr, d_v2, r0, r1, pp_twice, d_samples, 0, gz, d_f0, d_img, s);
}

extern "C" int fdw_dev_laplacian(fdw_ctx* c, const float* d_p, float* d_lap, void* stream)
{
    if (!c) return fail(FDW_EINVAL, "ctx is NULL");
    return step_impl(c, FDW_MODE_LAP, d_p, d_lap, nullptr, 0, c->nxl, 0, nullptr, -1, 0, nullptr, nullptr, pick_stream(c, stream));
}

extern "C" int fdw_dev_steps(fdw_ctx* c, float* d_p, float* d_pp, const float* d_v2, const float* d_srce, int sx, int sz,
                             int it0, int nsteps, int first_pp_twice, void* stream)
{
    if (!c) return fail(FDW_EINVAL, "ctx is NULL");
    hipStream_t s = pick_stream(c, stream);
    for (int k = 0; k < nsteps; k++) {
        std::swap(d_p, d_pp);  // R:260-262
        int rc = step_impl(c, FDW_MODE_FWD, d_p, d_pp, d_v2, 0, c->nxl, (k > 0) || first_pp_twice,
                           d_srce ? d_srce + it0 + k : nullptr, d_srce ? sx : -1, sz, nullptr, nullptr, s);
        if (rc) return rc;
    }
    return FDW_OK;
}

// ---- two time steps per pass (temporal blocking) --------------------------------------------------
// The two-step kernel wins where a launch is bandwidth bound (>= 8192^2: +18 %, 16384^2: +47 %); on small decks a
// launch is a latency chain and the longer march loses (new_mod: 12 vs 5 us/step), so the one-step kernel stays.
// Auto-selection threshold of the wave-pipeline kernel, in strip rows (rows x strips of 56 cells).  Measured (scripts/probe_sizes.py,
// probe_slabsize.py): 2048^2 (20k) one-step 280 vs pipeline 253 Gpt/s; 1056x8192 (39k) 312 vs 371; 4096^2 (78k) 337 (two-step 360) vs 452.
constexpr long kPipeAutoStripRows = 30000;
static bool two_step_pays(const fdw_ctx* c)
{
    if (c->h != kMaxFastHalfOrder || c->use_generic || c->tb < 0 || c->prm.dialect != FDW_DIALECT_RTM) return false;
    if (c->tb > 0) return true;
    return (long)c->upd_x1 * ((c->pitch / 4 + 59) / 60) >= 200000;
}

struct Step2Extra {          // what the receiver / imaging variant needs on top of the forward one
    const float* inj2 = nullptr;
    const float* psrc_a = nullptr;
    const float* psrc_b = nullptr;
    float* img = nullptr;
};

// mode FWD: d_inj -> {srce[it], srce[it+1]}, inj_x_global / inj_z = source position (inj_x_global < 0: none)
// mode PLAIN: no taper, no injection
// mode RECV: d_inj / ex.inj2 -> receiver samples of iterations it / it+1 (nx each), inj_z = gz, imaging with ex.psrc_a/b
static int step2_impl(fdw_ctx* c, int mode, const float* d_p, const float* d_pp, const float* d_v2, float* d_out1, float* d_out2, int pp_twice,
                      const float* d_inj, int inj_x_global, int inj_z, const Step2Extra& ex, hipStream_t s)
{
    if (c->h != kMaxFastHalfOrder) return fail(FDW_EINVAL, "step2: the two-step kernel is built for order 8 only");
    if (!d_p || !d_pp || !d_v2 || !d_out1 || !d_out2) return fail(FDW_EINVAL, "step2: NULL buffer");
    if (d_out1 == d_p || d_out1 == d_pp || d_out2 == d_p || d_out2 == d_pp || d_out1 == d_out2)
        return fail(FDW_EINVAL, "step2: outputs must not alias the inputs (tiles re-read each other's input rows)");
    Step2Args a{};
    a.p = d_p; a.pp = d_pp; a.v2 = d_v2; a.out1 = d_out1; a.out2 = d_out2;
    a.taperz = c->d_taperz; a.txfac = c->d_txfac; a.inj = d_inj; a.inj2 = ex.inj2;
    a.psrc_a = ex.psrc_a; a.psrc_b = ex.psrc_b; a.img = ex.img;
    a.pitch = c->pitch; a.nxl = c->nxl;
    a.r0 = 0; a.r1 = c->upd_x1;
    a.lap_x0 = c->lap_x0; a.lap_x1 = c->lap_x1; a.lap_z0 = c->lap_z0; a.lap_z1 = c->lap_z1;
    a.upd_x1 = c->upd_x1; a.upd_z1 = c->upd_z1;
    a.ztap = c->ztap; a.tz_x1 = c->tz_x1; a.xt_lo = c->xt_lo; a.xt_hi = c->xt_hi;
    a.pp_twice = pp_twice ? 1 : 0;
    a.img_z1 = c->prm.nzb + std::min(c->nz, c->zlim);
    a.inj_x = -1000000; a.inj_z = inj_z; a.inj_n = 0;
    if (mode == FDW_MODE_FWD && d_inj && inj_x_global >= 0) {
        if (inj_z < 0 || inj_z >= c->prm.nze || inj_x_global >= c->prm.nxe) return fail(FDW_EINVAL, "step2: source (%d,%d) outside the grid", inj_x_global, inj_z);
        a.inj_x = inj_x_global - c->slab.x_off;
        if (a.inj_x >= c->upd_x1 && a.inj_x < c->nxl)
            return fail(FDW_EINVAL, "step2: source row %d lies in rows the reference never time-steps (>= %d)", inj_x_global, c->xlim);
    } else if (mode == FDW_MODE_RECV) {
        if (!d_inj || !ex.inj2 || !ex.psrc_a || !ex.psrc_b || !ex.img) return fail(FDW_EINVAL, "step2: RECV needs both sample rows, both source fields and the image");
        if (inj_z < 0 || inj_z >= c->prm.nze) return fail(FDW_EINVAL, "step2: receiver depth %d outside the grid", inj_z);
        const int g0 = c->prm.nxb, g1 = c->prm.nxb + std::min(c->nx, c->xlim);       // receivers on interior columns (R:126-129)
        const int l0 = std::max(g0 - c->slab.x_off, 0), l1 = std::min(g1 - c->slab.x_off, c->nxl);
        a.inj_x = l0;
        a.inj_n = std::max(0, l1 - l0);
        a.inj = d_inj + (l0 + c->slab.x_off - g0);
        a.inj2 = ex.inj2 + (l0 + c->slab.x_off - g0);
    }
    a.dt2 = c->dt2;
    a.c0 = c->c0;
    a.numerics = c->prm.numerics;
    for (int io = 0; io <= 2 * kMaxFastHalfOrder; io++) {
        const bool fastw = c->prm.numerics == FDW_NUMERICS_FAST;
        a.cx[io] = fastw ? c->fcx[io] : c->cx[io];
        a.cz[io] = fastw ? c->fcz[io] : c->cz[io];
    }
    const int ncells = c->pitch / 4;
    a.nstrip = (ncells + 59) / 60;
    a.nzblk = (a.nstrip + 3) / 4;
    const int rows = a.r1 - a.r0;
    if (rows <= 0) return FDW_OK;
    // whole ring turns only (the march is branch-free, a partial turn still costs a full one): xchunk + 2H = 10k.
    // Measured (scripts/probe_tb.py / probe_slabsize.py, noise-filled fields, same-box A/B; +-5 % between boxes): 16384^2 449 Gpt/s
    // at 42 (443 at 22, 444 at 72); 8192^2 442 at 32 (435 at 22, 408 at 42); 4128x8192 416-423 at 32; 2080x8192 375 at 32;
    // 1056x8192 301 at 22 (267 at 32: too few waves).
    const long strip_rows = (long)rows * a.nstrip;
    int xchunk = c->xchunk2 > 0 ? c->xchunk2 : (strip_rows >= 1000000 ? 42 : (strip_rows >= 70000 ? 32 : (strip_rows >= 16384 ? 22 : 2)));
    a.xchunk = xchunk;
    const int chunks = (rows + xchunk - 1) / xchunk;
    a.nblk = a.nzblk * chunks;
    a.nper = (a.nblk + 7) / 8;
    hipError_t e = launch_step2(a, c->h, mode, s);
    if (e != hipSuccess) return fail(FDW_EHIP, "step2 launch failed: %s", hipGetErrorString(e));
    // rows the reference never time-steps (compat, nxe not a multiple of 8): both fields are static there and
    // swap roles every step, so after two steps out1 carries pp's rows and out2 p's rows
    if (c->upd_x1 < c->nxl) {
        const size_t off = (size_t)c->upd_x1 * c->pitch, n = (size_t)(c->nxl - c->upd_x1) * c->pitch * sizeof(float);
        HIP_TRY(hipMemcpyAsync(d_out1 + off, d_pp + off, n, hipMemcpyDeviceToDevice, s));
        HIP_TRY(hipMemcpyAsync(d_out2 + off, d_p + off, n, hipMemcpyDeviceToDevice, s));
        if (mode == FDW_MODE_RECV) {
            // receiver rows among them: iteration it injects into (and images) what is now out1, iteration it+1 what is now out2
            const int s0 = std::max(a.inj_x, c->upd_x1), s1 = std::min(a.inj_x + a.inj_n, c->nxl);
            if (s1 > s0) {
                hipError_t e2 = launch_static_rows(d_out1, ex.psrc_a, ex.img, a.inj + (s0 - a.inj_x), c->pitch, s0, s1 - s0, inj_z, c->prm.nzb, a.img_z1, s);
                if (e2 == hipSuccess)
                    e2 = launch_static_rows(d_out2, ex.psrc_b, ex.img, a.inj2 + (s0 - a.inj_x), c->pitch, s0, s1 - s0, inj_z, c->prm.nzb, a.img_z1, s);
                if (e2 != hipSuccess) return fail(FDW_EHIP, "static-row launch failed: %s", hipGetErrorString(e2));
            }
        }
    }
    return FDW_OK;
}

// ---- kPipeSteps time steps per pass (wave pipeline through LDS) -------------------------------------
static bool pipe_pays(const fdw_ctx* c)
{
    if (c->h != kMaxFastHalfOrder || c->use_generic || c->tb < 0 || c->prm.dialect == FDW_DIALECT_RTM_STORED) return false;
    if ((size_t)c->nxl * c->pitch * sizeof(float) >= (1ull << 31)) return false;   // the kernel addresses a field through one 2 GiB buffer descriptor
    if (c->tb == kPipeSteps) return true;
    if (c->tb > 0) return false;                                   // two-step forced
    return (long)c->upd_x1 * ((c->pitch / 4 + 55) / 56) >= kPipeAutoStripRows;
}

// FWD: d_inj -> kPipeSteps source samples srce[it .. it+kPipeSteps-1]; PLAIN: no taper, no injection.
// d_out1 = u^{n+kPipeSteps-1}, d_out2 = u^{n+kPipeSteps}.
struct StepnBack {      // the passes of the backward loop (Step2Args: lvl0, lvl1, plev, img, inj_stride, rp ...)
    float *lvl0 = nullptr, *lvl1 = nullptr;       // PLAIN_ALL: where waves 0 and 1 store their levels
    const float* plev[kPipeSteps] = {};           // RECV: the source field of iterations it .. it+3
    float* img = nullptr;
    int inj_stride = 0;                           // RECV, BACK4: floats between the sample rows of consecutive iterations
    const float *rp = nullptr, *rpp = nullptr;    // BACK4: the receiver pair (the positional arguments are the source field's)
    float *rout1 = nullptr, *rout2 = nullptr;
};
struct RowRanges {      // rows the pass produces: [r0, r1) and optionally [r0b, r1b); r1 < 0 = all rows the reference time-steps
    int r0 = 0, r1 = -1, r0b = 0, r1b = 0, xchunk = 0;
};
static int stepn_impl(fdw_ctx* c, int mode, const float* d_p, const float* d_pp, const float* d_v2, float* d_out1, float* d_out2, int pp_twice,
                      const float* d_inj, int inj_x_global, int inj_z, hipStream_t s, const RowRanges& rr = RowRanges{}, float* d_rec = nullptr,
                      int rec_z = 0, const StepnBack* bk = nullptr)
{
    if (c->h != kMaxFastHalfOrder) return fail(FDW_EINVAL, "stepn: the pipelined kernel is built for order 8 only");
    if (mode != FDW_MODE_FWD && mode != FDW_MODE_PLAIN && mode != FDW_MODE_MOD && mode != FDW_MODE_PLAIN_ALL && mode != FDW_MODE_RECV && mode != FDW_MODE_BACK4)
        return fail(FDW_EINVAL, "stepn: FWD, PLAIN, PLAIN_ALL, RECV, BACK4 or MOD only");
    if (mode == FDW_MODE_BACK4) {
        if (!bk || !bk->img || !d_inj || !bk->rp || !bk->rpp || !bk->rout1 || !bk->rout2) return fail(FDW_EINVAL, "stepn: BACK4 needs the receiver pair, its outputs, samples and image");
        const float* all[8] = {d_p, d_pp, d_out1, d_out2, bk->rp, bk->rpp, bk->rout1, bk->rout2};
        for (int i = 0; i < 8; i++)
            for (int j = i + 1; j < 8; j++)
                if (all[i] == all[j]) return fail(FDW_EINVAL, "stepn: BACK4 buffers must not alias");
    }
    if ((mode == FDW_MODE_PLAIN_ALL && (!bk || !bk->lvl0 || !bk->lvl1)) ||
        (mode == FDW_MODE_RECV && (!bk || !bk->img || !d_inj || !bk->plev[0] || !bk->plev[1] || !bk->plev[2] || !bk->plev[3])))
        return fail(FDW_EINVAL, "stepn: the backward passes need their level buffers / source fields, samples and image");
    if ((mode == FDW_MODE_MOD) != (c->prm.dialect == FDW_DIALECT_MOD)) return fail(FDW_ESTATE, "stepn: mode %d does not belong to dialect %d", mode, c->prm.dialect);
    if (!d_p || !d_pp || !d_v2 || !d_out1 || !d_out2) return fail(FDW_EINVAL, "stepn: NULL buffer");
    if (d_out1 == d_p || d_out1 == d_pp || d_out2 == d_p || d_out2 == d_pp || d_out1 == d_out2)
        return fail(FDW_EINVAL, "stepn: outputs must not alias the inputs (tiles re-read each other's input rows)");
    Step2Args a{};
    a.p = d_p; a.pp = d_pp; a.v2 = d_v2; a.out1 = d_out1; a.out2 = d_out2;
    a.taperz = c->d_taperz; a.txfac = c->d_txfac; a.inj = d_inj;
    a.pitch = c->pitch; a.nxl = c->nxl;
    const bool whole = rr.r1 < 0;
    if (!whole && (rr.r0 < 0 || rr.r1 > c->nxl || rr.r0b < 0 || rr.r1b > c->nxl || (rr.r1b > rr.r0b && rr.r0b < rr.r1)))
        return fail(FDW_EINVAL, "stepn: row ranges [%d,%d) [%d,%d) outside the slab or overlapping", rr.r0, rr.r1, rr.r0b, rr.r1b);
    a.r0 = whole ? 0 : rr.r0; a.r1 = std::min(whole ? c->upd_x1 : rr.r1, c->upd_x1);
    a.r0b = whole ? 0 : rr.r0b; a.r1b = whole ? 0 : std::min(rr.r1b, c->upd_x1);
    a.lap_x0 = c->lap_x0; a.lap_x1 = c->lap_x1; a.lap_z0 = c->lap_z0; a.lap_z1 = c->lap_z1;
    a.upd_x1 = c->upd_x1; a.upd_z1 = c->upd_z1;
    a.ztap = c->ztap; a.tz_x1 = c->tz_x1; a.xt_lo = c->xt_lo; a.xt_hi = c->xt_hi;
    {   // columns without a damping factor (taper_apply's table is 1.0f between the two strips; the RTM dialects damp the top strip only)
        const bool four_sided = c->prm.dialect == FDW_DIALECT_MOD;
        a.zt_lo = four_sided ? c->prm.nzb : c->ztap;
        a.zt_hi = four_sided ? c->prm.nze - c->prm.nzb : -1;
    }
    a.pp_twice = pp_twice ? 1 : 0;
    a.inj_x = -1000000; a.inj_z = inj_z; a.inj_n = 0;
    if (mode == FDW_MODE_FWD && d_inj && inj_x_global >= 0) {
        if (inj_z < 0 || inj_z >= c->prm.nze || inj_x_global >= c->prm.nxe) return fail(FDW_EINVAL, "stepn: source (%d,%d) outside the grid", inj_x_global, inj_z);
        a.inj_x = inj_x_global - c->slab.x_off;
        if (a.inj_x >= c->upd_x1 && a.inj_x < c->nxl)
            return fail(FDW_EINVAL, "stepn: source row %d lies in rows the reference never time-steps (>= %d)", inj_x_global, c->xlim);
    }
    if (mode == FDW_MODE_PLAIN_ALL) {
        a.lvl0 = bk->lvl0; a.lvl1 = bk->lvl1;
        for (float* o : {d_out1, d_out2}) if (bk->lvl0 == o || bk->lvl1 == o || bk->lvl0 == bk->lvl1 || bk->lvl0 == d_p || bk->lvl0 == d_pp || bk->lvl1 == d_p || bk->lvl1 == d_pp)
            return fail(FDW_EINVAL, "stepn: level buffers must not alias the inputs or outputs");
    }
    if (mode == FDW_MODE_BACK4) {
        a.rp = bk->rp; a.rpp = bk->rpp; a.rout1 = bk->rout1; a.rout2 = bk->rout2;
    }
    if (mode == FDW_MODE_RECV || mode == FDW_MODE_BACK4) {
        if (inj_z < 0 || inj_z >= c->prm.nze) return fail(FDW_EINVAL, "stepn: receiver depth %d outside the grid", inj_z);
        const int g0 = c->prm.nxb, g1 = c->prm.nxb + std::min(c->nx, c->xlim);       // receivers on interior rows (R:126-129)
        const int l0 = std::max(g0 - c->slab.x_off, 0), l1 = std::min(g1 - c->slab.x_off, c->nxl);
        a.inj_x = l0;
        a.inj_n = std::max(0, l1 - l0);
        a.inj = d_inj + (l0 + c->slab.x_off - g0);
        a.inj_stride = bk->inj_stride;
        a.img = bk->img;
        a.img_z1 = c->prm.nzb + std::min(c->nz, c->zlim);
        for (int i = 0; i < kPipeSteps; i++) a.plev[i] = bk->plev[i];
    }
    if (mode == FDW_MODE_MOD) {
        if (d_inj) {
            if (inj_z < 0 || inj_z >= c->prm.nze || inj_x_global < 0 || inj_x_global >= c->prm.nxe)
                return fail(FDW_EINVAL, "stepn: source (%d,%d) outside the grid", inj_x_global, inj_z);
            a.inj_x = inj_x_global - c->slab.x_off;
        }
        for (int i = 0; i < 4; i++)
            for (int j = 0; j < 4; j++) a.gw[i][j] = expf(-(float)(i * i) - (float)(j * j));
        a.dx2inv = c->dx2inv; a.dz2inv = c->dz2inv;
        a.rec = d_rec; a.rec_z = rec_z; a.rec_x0 = c->prm.nxb - c->slab.x_off; a.rec_n = c->nx;
    }
    a.dt2 = c->dt2;
    a.c0 = c->c0;
    a.numerics = c->prm.numerics;
    for (int io = 0; io <= 2 * kMaxFastHalfOrder; io++) {
        const bool fastw = c->prm.numerics == FDW_NUMERICS_FAST;
        a.cx[io] = fastw ? c->fcx[io] : c->cx[io];
        a.cz[io] = fastw ? c->fcz[io] : c->cz[io];
    }
    const int ncells = c->pitch / 4, own = 64 - 2 * kPipeSteps;
    a.nstrip = (ncells + own - 1) / own;
    a.nzblk = a.nstrip;
    const int rows_a = std::max(a.r1 - a.r0, 0), rows_b = std::max(a.r1b - a.r0b, 0), rows = rows_a + rows_b;
    if (rows <= 0) return FDW_OK;
    // whole ring turns: xchunk + (NS-1)(2H+1) = 10k  ->  xchunk = 10k - 27 (13, 23, ... 83, 93, ...)
    const long strip_rows = (long)rows * a.nstrip;
    // measured: 16384^2 579 Gpt/s at 253 (560 at 173); 8192^2 566 at 173 (509 at 83, 553 at 253); 4096^2 452 at 83 (382 at 43, 388 at 173);
    // 1056x8192 371 at 43 (355 at 63, 340 at 33)
    // end of round 2 (scripts/probe_pipe.py, PIPE_SHAPE): 4128x8192 (half of the bench grid) 584 at 83, 592 at 123, 609 at 143, 601 at 173;
    // 2144x8192 538 at 83 (499 at 53, 521 at 103)
    // end of round 3 (same probe, FAST numerics): 4128x8192 677 at 143, 710 at 173, 700 at 83 -- the memory-bound FAST kernel prefers the longer chunk there
    const bool fastnum = c->prm.numerics == FDW_NUMERICS_FAST;
    int xchunk = c->xchunk2 > 0 ? c->xchunk2 : (strip_rows >= 1000000 ? 253 : (strip_rows >= 250000 ? 173 : (strip_rows >= 120000 ? (fastnum ? 173 : 143) : (strip_rows >= 60000 ? 83 : 43))));
    if (mode == FDW_MODE_BACK4 && c->xchunk2 <= 0) {
        // the eight-wave kernel holds two workgroups per CU (512 at a time): longer chunks than the forward kernel's at the same grid size
        // (scripts/probe_slabs_c.py, us per iteration by chunk length 43 / 83 / 123 / 173: 8192 rows 363 / 309 / 289 / 277; 4160 rows 192 / 172 / 157 / 162;
        //  2176 rows 109 / 94 / 102 / 93; 1152 rows 61 / 64 / 69 / 80)
        // end of round 2 (lean bodies, scripts/probe_back.py, 8192 rows, us per iteration): 93: 240, 103: 233, 123: 228, 133: 230, 143: 234, 153: 226,
        //  163: 234, 173: 234, 193: 233, 213: 225, 223: 231 -- 54 chunks x 37 strips fill the 512 workgroup slots 3.9 times over
        xchunk = strip_rows >= 250000 ? 153 : (strip_rows >= 120000 ? 123 : (strip_rows >= 60000 ? 83 : 43));
        if (const char* e = getenv("FDW_BACK4_XCHUNK")) xchunk = atoi(e) > 0 ? atoi(e) : xchunk;      // experiments
    }
    if (rr.xchunk > 0) xchunk = rr.xchunk;
    a.xchunk = xchunk;
    a.chunks_a = (rows_a + xchunk - 1) / xchunk;
    const int chunks = a.chunks_a + (rows_b + xchunk - 1) / xchunk;
    a.nblk = a.nstrip * chunks;
    a.nper = (a.nblk + 7) / 8;
    hipError_t e = launch_stepn(a, c->h, mode, s);
    if (e != hipSuccess) return fail(FDW_EHIP, "stepn launch failed: %s", hipGetErrorString(e));
    // rows the reference never time-steps swap roles every step: after an even number of steps out1 carries pp's rows, out2 p's
    static_assert(kPipeSteps % 2 == 0, "static-row bookkeeping assumes an even number of steps per pass");
    if (c->upd_x1 < c->nxl && (whole || rr.r1 >= c->upd_x1 || rr.r1b >= c->upd_x1)) {
        const size_t off = (size_t)c->upd_x1 * c->pitch, n = (size_t)(c->nxl - c->upd_x1) * c->pitch * sizeof(float);
        HIP_TRY(hipMemcpyAsync(d_out1 + off, d_pp + off, n, hipMemcpyDeviceToDevice, s));
        HIP_TRY(hipMemcpyAsync(d_out2 + off, d_p + off, n, hipMemcpyDeviceToDevice, s));
        if (mode == FDW_MODE_PLAIN_ALL) {      // the levels in between alternate the same way: level 0 carries pp's rows, level 1 p's
            HIP_TRY(hipMemcpyAsync(bk->lvl0 + off, d_pp + off, n, hipMemcpyDeviceToDevice, s));
            HIP_TRY(hipMemcpyAsync(bk->lvl1 + off, d_p + off, n, hipMemcpyDeviceToDevice, s));
        }
        if (mode == FDW_MODE_BACK4) {          // and so do the receiver field's
            HIP_TRY(hipMemcpyAsync(bk->rout1 + off, bk->rpp + off, n, hipMemcpyDeviceToDevice, s));
            HIP_TRY(hipMemcpyAsync(bk->rout2 + off, bk->rp + off, n, hipMemcpyDeviceToDevice, s));
        }
    }
    return FDW_OK;
}

// FOUR iterations of fd_back's loop (R:317-329) as two passes of the wave-pipeline kernel on row ranges of the slab (see back_loop)
extern "C" int fdw_dev_back4(fdw_ctx* c, const float* d_f1, const float* d_f0, float* d_fo1, float* d_fo2, float* d_lvl0, float* d_lvl1, const float* d_pr,
                             const float* d_ppr, float* d_ro1, float* d_ro2, const float* d_v2, const float* d_samples, int sample_stride, int gz,
                             float* d_img, int pp_twice, int r0, int r1, int r0b, int r1b, int xchunk, void* stream)
{
    if (!c) return fail(FDW_EINVAL, "ctx is NULL");
    if (c->prm.dialect != FDW_DIALECT_RTM || c->h != kMaxFastHalfOrder || (size_t)c->nxl * c->pitch * sizeof(float) >= (1ull << 31))
        return fail(FDW_EINVAL, "back4: needs the RTM dialect, order 8 and fields below 2 GiB");
    if (c->prm.nxb + c->nx > c->xlim) return fail(FDW_EINVAL, "back4: receiver rows beyond the time-stepped rows are not covered by the pipelined passes");
    hipStream_t s = pick_stream(c, stream);
    RowRanges rr;
    rr.r0 = r0; rr.r1 = r1; rr.r0b = r0b; rr.r1b = r1b; rr.xchunk = xchunk;
    if (!c->no_back_fused) {      // both fields in one pass of the eight-wave kernel: the levels in between never leave the chip
        StepnBack b4;
        b4.rp = d_pr; b4.rpp = d_ppr; b4.rout1 = d_ro1; b4.rout2 = d_ro2; b4.img = d_img; b4.inj_stride = sample_stride;
        return stepn_impl(c, FDW_MODE_BACK4, d_f1, d_f0, d_v2, d_fo1, d_fo2, pp_twice, d_samples, -1, gz, s, rr, nullptr, 0, &b4);
    }
    StepnBack fb;
    fb.lvl0 = d_lvl0; fb.lvl1 = d_lvl1;
    int rc = stepn_impl(c, FDW_MODE_PLAIN_ALL, d_f1, d_f0, d_v2, d_fo1, d_fo2, 0, nullptr, -1, 0, s, rr, nullptr, 0, &fb);
    if (rc) return rc;
    StepnBack rb;
    rb.plev[0] = d_lvl0; rb.plev[1] = d_lvl1; rb.plev[2] = d_fo1; rb.plev[3] = d_fo2;
    rb.img = d_img;
    rb.inj_stride = sample_stride;
    return stepn_impl(c, FDW_MODE_RECV, d_pr, d_ppr, d_v2, d_ro1, d_ro2, pp_twice, d_samples, -1, gz, s, rr, nullptr, 0, &rb);
}

extern "C" int fdw_back_pipe_active(const fdw_ctx* c)
{
    return c && pipe_pays(c) && c->prm.dialect == FDW_DIALECT_RTM && c->prm.nxb + c->nx <= c->xlim && !c->no_back_pipe ? 1 : 0;
}

extern "C" int fdw_two_step_active(const fdw_ctx* c) { return c && two_step_pays(c) ? 1 : 0; }
extern "C" int fdw_steps_per_pass(const fdw_ctx* c) { return !c ? 0 : (pipe_pays(c) ? kPipeSteps : (two_step_pays(c) ? 2 : 1)); }

extern "C" int fdw_dev_step2(fdw_ctx* c, const float* d_p, const float* d_pp, const float* d_v2, float* d_out1, float* d_out2, int pp_twice,
                             const float* d_srce_it, int sx, int sz, void* stream)
{
    if (!c) return fail(FDW_EINVAL, "ctx is NULL");
    return step2_impl(c, FDW_MODE_FWD, d_p, d_pp, d_v2, d_out1, d_out2, pp_twice, d_srce_it, d_srce_it ? sx : -1, sz, Step2Extra{}, pick_stream(c, stream));
}

// kPipeSteps (4) forward iterations in one pass of the wave-pipeline kernel on the given row ranges of a slab
extern "C" int fdw_dev_step4(fdw_ctx* c, const float* d_p, const float* d_pp, const float* d_v2, float* d_out1, float* d_out2, int pp_twice,
                             const float* d_srce_it, int sx, int sz, int r0, int r1, int r0b, int r1b, int xchunk, void* stream)
{
    if (!c) return fail(FDW_EINVAL, "ctx is NULL");
    if (c->h != kMaxFastHalfOrder || (size_t)c->nxl * c->pitch * sizeof(float) >= (1ull << 31))
        return fail(FDW_EINVAL, "step4: needs order 8 and fields below 2 GiB");
    RowRanges rr;
    rr.r0 = r0; rr.r1 = r1; rr.r0b = r0b; rr.r1b = r1b; rr.xchunk = xchunk;
    return stepn_impl(c, FDW_MODE_FWD, d_p, d_pp, d_v2, d_out1, d_out2, pp_twice, d_srce_it, d_srce_it ? sx : -1, sz, pick_stream(c, stream), rr);
}

// nsteps reference iterations (R:259-267) over four rotating buffers: pairs of steps through the two-step
// kernel, an odd last step through the one-step kernel.  On entry buf[*ip], buf[*ipp] are the reference's
// (d_p, d_pp) BEFORE the first swap; on return they index (d_p, d_pp) after the loop.
extern "C" int fdw_dev_steps2(fdw_ctx* c, float* const* buf, const float* d_v2, const float* d_srce, int sx, int sz, int it0, int nsteps,
                              int first_pp_twice, int* ip, int* ipp, void* stream)
{
    if (!c || !buf || !ip || !ipp) return fail(FDW_EINVAL, "NULL argument");
    if (*ip < 0 || *ip > 3 || *ipp < 0 || *ipp > 3 || *ip == *ipp) return fail(FDW_EINVAL, "steps2: bad buffer indices");
    hipStream_t s = pick_stream(c, stream);
    int k = 0;
    while (k < nsteps) {
        const int twice = (k > 0) || first_pp_twice;
        if (nsteps - k >= kPipeSteps && pipe_pays(c)) {
            int o1 = 0, o2 = 0;
            for (int i = 0, n = 0; i < 4; i++)
                if (i != *ip && i != *ipp) { (n++ == 0 ? o1 : o2) = i; }
            int rc = stepn_impl(c, FDW_MODE_FWD, buf[*ipp], buf[*ip], d_v2, buf[o1], buf[o2], twice, d_srce ? d_srce + it0 + k : nullptr,
                                d_srce ? sx : -1, sz, s);
            if (rc) return rc;
            *ip = o1; *ipp = o2;   // d_p = u^{n+kPipeSteps-1}, d_pp = u^{n+kPipeSteps}
            k += kPipeSteps;
        } else if (nsteps - k >= 2 && two_step_pays(c)) {
            int o1 = 0, o2 = 0;   // the two buffers not holding the current pair
            for (int i = 0, n = 0; i < 4; i++)
                if (i != *ip && i != *ipp) { (n++ == 0 ? o1 : o2) = i; }
            // after the swap the kernel's p is the old d_pp (newest field), its pp the old d_p
            int rc = step2_impl(c, FDW_MODE_FWD, buf[*ipp], buf[*ip], d_v2, buf[o1], buf[o2], twice, d_srce ? d_srce + it0 + k : nullptr,
                                d_srce ? sx : -1, sz, Step2Extra{}, s);
            if (rc) return rc;
            *ip = o1; *ipp = o2;   // d_p = u^{n+1}, d_pp = u^{n+2}
            k += 2;
        } else {
            std::swap(*ip, *ipp);
            int rc = step_impl(c, FDW_MODE_FWD, buf[*ip], buf[*ipp], d_v2, 0, c->nxl, twice, d_srce ? d_srce + it0 + k : nullptr,
                               d_srce ? sx : -1, sz, nullptr, nullptr, s);
            if (rc) return rc;
            k += 1;
        }
    }
    return FDW_OK;
}

// ksteps-cycle of the slab decomposition in one call: step j (1-based, j = j0 .. j0+nsteps-1) updates the
// rows still valid on the interior sides, [h*j, nxl - h*j) (decomp.py "deep halos").
extern "C" int fdw_dev_steps_shrink(fdw_ctx* c, float* d_p, float* d_pp, const float* d_v2, const float* d_srce, int sx, int sz,
                                    int it0, int nsteps, int first_pp_twice, int j0, int shrink_lo, int shrink_hi, void* stream)
{
    if (!c) return fail(FDW_EINVAL, "ctx is NULL");
    hipStream_t s = pick_stream(c, stream);
    for (int k = 0; k < nsteps; k++) {
        std::swap(d_p, d_pp);  // R:260-262
        const int j = j0 + k;
        const int r0 = shrink_lo ? c->h * j : 0;
        const int r1 = c->nxl - (shrink_hi ? c->h * j : 0);
        if (r1 <= r0) return fail(FDW_EINVAL, "steps_shrink: slab exhausted at cycle step %d", j);
        int rc = step_impl(c, FDW_MODE_FWD, d_p, d_pp, d_v2, r0, r1, (k > 0) || first_pp_twice, d_srce ? d_srce + it0 + k : nullptr,
                           d_srce ? sx : -1, sz, nullptr, nullptr, s);
        if (rc) return rc;
    }
    return FDW_OK;
}

extern "C" int fdw_dev_taper_finalize(fdw_ctx* c, float* d_f, void* stream)
{
    if (!c || !d_f) return fail(FDW_EINVAL, "ctx or field is NULL");
    hipError_t e = launch_taper_finalize(d_f, c->d_taperz, c->d_txfac, c->pitch, c->nxl, c->ztap, c->tz_x1, pick_stream(c, stream));
    if (e != hipSuccess) return fail(FDW_EHIP, "taper finalize launch failed: %s", hipGetErrorString(e));
    return FDW_OK;
}

// ------------------------------------------------------------------------------------------------
// host <-> device transfers
// ------------------------------------------------------------------------------------------------
static int upload_rows(fdw_ctx* c, float* d_dst, const float* h_src, hipStream_t s)
{
    if (d_dst == c->d_v2) c->v2_resident = false;
    HIP_TRY(hipMemcpy2DAsync(d_dst, (size_t)c->pitch * sizeof(float), h_src, (size_t)c->prm.nze * sizeof(float),
                             (size_t)c->prm.nze * sizeof(float), c->nxl, hipMemcpyHostToDevice, s));
    return FDW_OK;
}
static int download_rows(fdw_ctx* c, float* h_dst, const float* d_src, hipStream_t s)
{
    HIP_TRY(hipMemcpy2DAsync(h_dst, (size_t)c->prm.nze * sizeof(float), d_src, (size_t)c->pitch * sizeof(float),
                             (size_t)c->prm.nze * sizeof(float), c->nxl, hipMemcpyDeviceToHost, s));
    return FDW_OK;
}

extern "C" int fdw_upload_field(fdw_ctx* c, float* d_dst, const float* h_src)
{
    if (!c || !d_dst || !h_src) return fail(FDW_EINVAL, "NULL argument");
    HIP_TRY(hipSetDevice(c->device));
    int rc = upload_rows(c, d_dst, h_src, c->stream);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));
    return FDW_OK;
}
extern "C" int fdw_download_field(fdw_ctx* c, float* h_dst, const float* d_src)
{
    if (!c || !h_dst || !d_src) return fail(FDW_EINVAL, "NULL argument");
    HIP_TRY(hipSetDevice(c->device));
    int rc = download_rows(c, h_dst, d_src, c->stream);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));
    return FDW_OK;
}

// The reference damps rows >= xlim of the top strip with taperx only, every step, in place; the lazy
// scheme cannot reproduce that for rows that are never rewritten.  Those cells are zero at every call
// site of the reference (R:496-497, R:511-514 and snapshots of such runs), so we require it.
static int check_static_taper_rows(const fdw_ctx* c, const float* f, const char* name)
{
    if (c->xlim >= c->prm.nxe || c->ztap <= 0) return FDW_OK;
    for (int i = c->xlim; i < c->prm.nxe; i++)
        for (int j = 0; j < c->ztap; j++)
            if (f[(size_t)i * c->prm.nze + j] != 0.0f)
                return fail(FDW_EINVAL, "%s[%d][%d] != 0: in compat mode rows >= %d of the damped strip must be zero "
                                        "(they are never time-stepped by the reference, R:185-195)", name, i, j, c->xlim);
    return FDW_OK;
}

// the same precondition for a DEVICE array (fdw_border.hip; declared here rather than in fdw_kernels.h, whose text is part of the kernel-source hash of bench.py)
namespace fdw {
hipError_t launch_static_strip_check(const float* f, int pitch, int row0, int nrows, int ztap, unsigned* flag, hipStream_t s);
}
extern "C" int fdw_dev_check_field(fdw_ctx* c, const float* d_f, void* stream)
{
    if (!c || !d_f) return fail(FDW_EINVAL, "ctx or field is NULL");
    const int row0 = std::max(c->xlim - c->slab.x_off, 0);              // local rows the reference never time-steps
    if (row0 >= c->nxl || c->ztap <= 0) return FDW_OK;                  // nothing to check on this grid / slab
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t s = pick_stream(c, stream);
    unsigned* d_flag = nullptr;
    HIP_TRY(hipMalloc((void**)&d_flag, sizeof(unsigned)));
    unsigned bad = 0;
    hipError_t e = hipMemsetAsync(d_flag, 0, sizeof(unsigned), s);
    if (e == hipSuccess) e = launch_static_strip_check(d_f, c->pitch, row0, c->nxl - row0, c->ztap, d_flag, s);
    if (e == hipSuccess) e = hipMemcpyAsync(&bad, d_flag, sizeof(unsigned), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(d_flag);
    if (e != hipSuccess) return fail(FDW_EHIP, "fdw_dev_check_field: %s", hipGetErrorString(e));
    if (bad)
        return fail(FDW_EINVAL, "%u cells of the damped strip (columns < %d) are non-zero on rows the reference never time-steps (global rows >= %d): "
                                "in compat mode they must be zero (R:185-195; include/fdwave.h, PRECONDITION)", bad, c->ztap, c->xlim);
    return FDW_OK;
}

// ------------------------------------------------------------------------------------------------
// host-array entry points
// ------------------------------------------------------------------------------------------------
extern "C" int fdw_laplacian(fdw_ctx* c, const float* p, float* lap)
{
    if (!c || !p || !lap) return fail(FDW_EINVAL, "NULL argument");
    if (!is_full_grid(c)) return fail(FDW_EINVAL, "fdw_laplacian needs a full-grid context");
    HIP_TRY(hipSetDevice(c->device));
    int rc = ensure_work_buffers(c, 2, false);
    if (rc) return rc;
    if ((rc = upload_rows(c, c->fld[0], p, c->stream))) return rc;
    if ((rc = step_impl(c, FDW_MODE_LAP, c->fld[0], c->fld[1], nullptr, 0, c->nxl, 0, nullptr, -1, 0, nullptr, nullptr, c->stream))) return rc;
    if ((rc = download_rows(c, lap, c->fld[1], c->stream))) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));
    return FDW_OK;
}

static int upload_source(fdw_ctx* c, const float* srce, int n)
{
    int rc = ensure_cap(&c->d_srce, &c->srce_cap, (size_t)std::max(n, 1));
    if (rc) return rc;
    if (n > 0) HIP_TRY(hipMemcpyAsync(c->d_srce, srce, (size_t)n * sizeof(float), hipMemcpyHostToDevice, c->stream));
    return FDW_OK;
}

// fd_forward's loop body R:259-267 for nsteps iterations over the context's four field buffers (pairs of steps go
// through the two-step kernel where it pays); *ip / *ipp index (d_p, d_pp) before the loop and after it.
static int forward_loop(fdw_ctx* c, int* ip, int* ipp, int sx, int sz, int nsteps)
{
    FDW_RANGE("fdw: forward loop (fd_forward)");
    int rc = fdw_dev_steps2(c, c->fld, c->d_v2, c->d_srce, sx, sz, 0, nsteps, 0, ip, ipp, c->stream);
    if (rc) return rc;
    if (nsteps > 0) return fdw_dev_taper_finalize(c, c->fld[*ip], c->stream);   // the T() d_p still owes (R:285 downloads the damped d_p)
    return FDW_OK;
}

extern "C" int fdw_forward(fdw_ctx* c, float* p, float* pp, const float* v2, int sx, int sz, const float* srce, int nsteps)
{
    if (!c || !p || !pp || !v2 || (!srce && nsteps > 0)) return fail(FDW_EINVAL, "NULL argument");
    if (!is_full_grid(c)) return fail(FDW_EINVAL, "fdw_forward needs a full-grid context");
    if (nsteps < 0) return fail(FDW_EINVAL, "nsteps=%d", nsteps);
    int rc;
    if ((rc = check_static_taper_rows(c, p, "p")) || (rc = check_static_taper_rows(c, pp, "pp"))) return rc;
    HIP_TRY(hipSetDevice(c->device));
    if ((rc = ensure_work_buffers(c, 4, false))) return rc;
    int ip = 0, ipp = 1;
    if ((rc = upload_rows(c, c->fld[0], p, c->stream)) || (rc = upload_rows(c, c->fld[1], pp, c->stream)) ||
        (rc = upload_rows(c, c->d_v2, v2, c->stream)) || (rc = upload_source(c, srce, nsteps)))
        return rc;
    if ((rc = forward_loop(c, &ip, &ipp, sx, sz, nsteps))) return rc;
    if ((rc = download_rows(c, p, c->fld[ip], c->stream)) || (rc = download_rows(c, pp, c->fld[ipp], c->stream))) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));
    return FDW_OK;
}

// d_obs [nx][nt] (R:426-435) -> device [nt][nx] so that one step's samples are contiguous
static int gathers_to_device(fdw_ctx* c, const float* d_obs, float* d_dst, int nshots)
{
    const size_t n = (size_t)c->nx * c->prm.nt * nshots;
    int rc = ensure_cap(&c->d_raw, &c->raw_cap, n);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(c->d_raw, d_obs, n * sizeof(float), hipMemcpyHostToDevice, c->stream));
    hipError_t e = launch_gather_transpose(c->d_raw, d_dst, c->nx, c->prm.nt, nshots, c->stream);
    if (e != hipSuccess) return fail(FDW_EHIP, "gather transposition launch failed: %s", hipGetErrorString(e));
    return FDW_OK;
}
static int upload_gather(fdw_ctx* c, const float* d_obs)
{
    int rc = ensure_cap(&c->d_dobs, &c->dobs_cap, (size_t)c->nx * c->prm.nt);
    if (rc) return rc;
    return gathers_to_device(c, d_obs, c->d_dobs, 1);
}

static int image_to_device(fdw_ctx* c, const float* imloc)
{
    HIP_TRY(hipMemsetAsync(c->d_img, 0, field_elems(c) * sizeof(float), c->stream));
    float* dst = c->d_img + (size_t)c->prm.nxb * c->pitch + c->prm.nzb;
    HIP_TRY(hipMemcpy2DAsync(dst, (size_t)c->pitch * sizeof(float), imloc, (size_t)c->nz * sizeof(float),
                             (size_t)c->nz * sizeof(float), c->nx, hipMemcpyHostToDevice, c->stream));
    return FDW_OK;
}
static int image_to_host(fdw_ctx* c, float* imloc)
{
    const float* src = c->d_img + (size_t)c->prm.nxb * c->pitch + c->prm.nzb;
    HIP_TRY(hipMemcpy2DAsync(imloc, (size_t)c->nz * sizeof(float), src, (size_t)c->pitch * sizeof(float),
                             (size_t)c->nz * sizeof(float), c->nx, hipMemcpyDeviceToHost, c->stream));
    return FDW_OK;
}

// fd_back's loop R:302-339 on device buffers.
//   F_k  = source wavefield of iteration k:  F_0 = snap1 (u^nt), F_1 = snap0 (u^{nt-1}, R:304-314),
//          F_k = leap-frog(F_{k-1}, F_{k-2}) for k >= 2 (no taper, no source, R:317-318)
//   r^k  = receiver field: r^{k+1} = damped leap-frog(r^k, r^{k-1}) + d_obs[.][nt-1-k] on row gz (R:325-328)
//   img += F_k * r^{k+1}  (R:329)
// Iterations are taken in PAIRS through the two-step kernel where it pays (one pass reconstructs F_k, F_{k+1}; one
// pass advances the receiver field twice and applies both imaging conditions), singly through the one-step kernels
// otherwise.  src[0..3] / rcv[0..3]: rotating buffers; on entry src[0] = snap0, src[1] = snap1, the receivers are zero.
static int back_loop(fdw_ctx* c, float* const src[4], float* const rcv[4], int gz, int nsteps)
{
    FDW_RANGE("fdw: backward loop + imaging (fd_back)");
    const int nt = c->prm.nt;
    const size_t nxs = (size_t)c->nx;
    auto samples = [&](int it) { return c->d_dobs + (size_t)(nt - 1 - it) * nxs; };
    int f1 = 0, f0 = 1;          // indices into src of F_{k-1} (newer) and F_{k-2}: before iteration 2 these are snap0, snap1
    int rn = 0, ro = 1;          // indices into rcv of r^k (d_pr) and r^{k-1} (d_ppr)
    int rc;
    int it = 0;
    const bool pairs = two_step_pays(c);
    // Four iterations per pair of passes through the wave pipeline where it pays: pass 1 reconstructs F_it .. F_{it+3} (all four levels are
    // kept: the imaging condition needs each of them), pass 2 advances the receiver field four times, injects each iteration's samples and
    // adds the four products F_{it+j} r^{it+j+1} to the image in iteration order.  Not where receiver rows lie beyond the time-stepped rows.
    const bool pipe = pipe_pays(c) && c->prm.dialect == FDW_DIALECT_RTM && c->nbatch <= 1 && c->prm.nxb + c->nx <= c->upd_x1 && !c->no_back_pipe;
    if (pipe && c->no_back_fused && nsteps >= 2 + kPipeSteps) {      // the two-pass form keeps two levels in memory
        if ((rc = alloc_zero(&c->fld[8], field_elems(c))) || (rc = alloc_zero(&c->fld[9], field_elems(c)))) return rc;
    }
    while (it < nsteps) {
        if (pipe && it >= 2 && nsteps - it >= kPipeSteps) {
            int o1 = -1, o2 = -1, q1 = -1, q2 = -1;
            for (int i = 0; i < 4; i++) {
                if (i != f1 && i != f0) { (o1 < 0 ? o1 : o2) = i; }
                if (i != rn && i != ro) { (q1 < 0 ? q1 : q2) = i; }
            }
            if (!c->no_back_fused) {      // one pass of the eight-wave kernel for both fields
                StepnBack b4;
                b4.rp = rcv[rn]; b4.rpp = rcv[ro]; b4.rout1 = rcv[q1]; b4.rout2 = rcv[q2]; b4.img = c->d_img; b4.inj_stride = -(int)nxs;
                if ((rc = stepn_impl(c, FDW_MODE_BACK4, src[f1], src[f0], c->d_v2, src[o1], src[o2], it > 0, samples(it), -1, gz, c->stream, RowRanges{}, nullptr, 0, &b4))) return rc;
                f0 = o1; f1 = o2;
                ro = q1; rn = q2;
                it += kPipeSteps;
                continue;
            }
            StepnBack fb;
            fb.lvl0 = c->fld[8]; fb.lvl1 = c->fld[9];
            if ((rc = stepn_impl(c, FDW_MODE_PLAIN_ALL, src[f1], src[f0], c->d_v2, src[o1], src[o2], 0, nullptr, -1, 0, c->stream, RowRanges{}, nullptr, 0, &fb))) return rc;
            StepnBack rb;
            rb.plev[0] = c->fld[8]; rb.plev[1] = c->fld[9]; rb.plev[2] = src[o1]; rb.plev[3] = src[o2];
            rb.img = c->d_img;
            rb.inj_stride = -(int)nxs;                       // iteration it+1 reads the sample row before (time-reversed traces, R:328)
            if ((rc = stepn_impl(c, FDW_MODE_RECV, rcv[rn], rcv[ro], c->d_v2, rcv[q1], rcv[q2], it > 0, samples(it), -1, gz, c->stream, RowRanges{}, nullptr, 0, &rb))) return rc;
            f0 = o1; f1 = o2;
            ro = q1; rn = q2;
            it += kPipeSteps;
            continue;
        }
        if (pairs && nsteps - it >= 2) {
            const float *Fa, *Fb;   // source fields of iterations it, it+1
            if (it == 0) {
                Fa = src[1]; Fb = src[0];                      // u^nt, u^{nt-1}
            } else {
                int o1 = -1, o2 = -1;
                for (int i = 0; i < 4; i++)
                    if (i != f1 && i != f0) { (o1 < 0 ? o1 : o2) = i; }
                if ((rc = step2_impl(c, FDW_MODE_PLAIN, src[f1], src[f0], c->d_v2, src[o1], src[o2], 0, nullptr, -1, 0, Step2Extra{}, c->stream))) return rc;
                Fa = src[o1]; Fb = src[o2];
                f0 = o1; f1 = o2;
            }
            int q1 = -1, q2 = -1;
            for (int i = 0; i < 4; i++)
                if (i != rn && i != ro) { (q1 < 0 ? q1 : q2) = i; }
            Step2Extra ex;
            ex.inj2 = samples(it + 1); ex.psrc_a = Fa; ex.psrc_b = Fb; ex.img = c->d_img;
            if ((rc = step2_impl(c, FDW_MODE_RECV, rcv[rn], rcv[ro], c->d_v2, rcv[q1], rcv[q2], it > 0, samples(it), -1, gz, ex, c->stream))) return rc;
            ro = q1; rn = q2;
            it += 2;
        } else {
            const float* F;
            if (it == 0) F = src[1];
            else if (it == 1) F = src[0];
            else if (c->h <= kMaxFastHalfOrder && !c->use_generic && !c->no_fused_back) {
                // the whole iteration in ONE pass: the source field is stepped in place (F_k overwrites F_{k-2}) in the same kernel
                // that steps the receiver field, and meets the new receiver row in registers for the imaging condition
                if ((rc = step_impl(c, FDW_MODE_BACK, rcv[rn], rcv[ro], c->d_v2, 0, c->nxl, it > 0, samples(it), 0, gz, src[f1], c->d_img, c->stream,
                                    nullptr, 0, src[f0])))
                    return rc;
                std::swap(f1, f0);
                std::swap(rn, ro);
                it += 1;
                continue;
            } else {
                // one-step reconstruction in place: the new field overwrites F_{k-2}
                if ((rc = step_impl(c, FDW_MODE_PLAIN, src[f1], src[f0], c->d_v2, 0, c->nxl, 0, nullptr, -1, 0, nullptr, nullptr, c->stream))) return rc;
                std::swap(f1, f0);
                F = src[f1];
            }
            if ((rc = step_impl(c, FDW_MODE_RECV, rcv[rn], rcv[ro], c->d_v2, 0, c->nxl, it > 0, samples(it), 0, gz, F, c->d_img, c->stream))) return rc;
            std::swap(rn, ro);
            it += 1;
        }
    }
    return FDW_OK;
}

static int zero_fields(fdw_ctx* c, float* const f[], int n)
{
    for (int i = 0; i < n; i++) HIP_TRY(hipMemsetAsync(f[i], 0, field_elems(c) * sizeof(float), c->stream));
    return FDW_OK;
}

extern "C" int fdw_back(fdw_ctx* c, const float* v2, const float* snap0, const float* snap1, const float* d_obs, int gz,
                        float* imloc, int nsteps)
{
    if (!c || !v2 || !snap0 || !snap1 || !d_obs || !imloc) return fail(FDW_EINVAL, "NULL argument");
    if (!is_full_grid(c)) return fail(FDW_EINVAL, "fdw_back needs a full-grid context");
    if (nsteps < 0 || nsteps > c->prm.nt) return fail(FDW_EINVAL, "nsteps=%d outside [0,nt=%d]", nsteps, c->prm.nt);
    if (c->nx <= 0 || c->nz <= 0) return fail(FDW_EINVAL, "no interior to image");
    HIP_TRY(hipSetDevice(c->device));
    int rc;
    if ((rc = ensure_work_buffers(c, 8, true))) return rc;
    if ((rc = upload_rows(c, c->d_v2, v2, c->stream)) || (rc = upload_gather(c, d_obs)) || (rc = image_to_device(c, imloc))) return rc;
    // the reference uploads the two snapshots into d_pp at it = 0 and 1 (R:304-314); having both on the device up front is the same
    if ((rc = upload_rows(c, c->fld[0], snap0, c->stream)) || (rc = upload_rows(c, c->fld[1], snap1, c->stream))) return rc;
    float* const src[4] = {c->fld[0], c->fld[1], c->fld[2], c->fld[3]};
    float* const rcv[4] = {c->fld[4], c->fld[5], c->fld[6], c->fld[7]};
    if ((rc = zero_fields(c, rcv, 2))) return rc;                   // R:513-514
    if ((rc = back_loop(c, src, rcv, gz, nsteps))) return rc;
    if ((rc = image_to_host(c, imloc))) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));
    return FDW_OK;
}

// v2 == nullptr: the squared model already resident in c->d_v2 (fdw_dev_extendvel_linear)
static int shot_impl(fdw_ctx* c, const float* v2, int sx, int sz, int gz, const float* srce, const float* d_obs, float* imloc, float* P, float* PP)
{
    if (!c || !srce || !d_obs || !imloc) return fail(FDW_EINVAL, "NULL argument");
    FDW_RANGE("fdw: shot (uploads, forward, backward, image download)");
    if (!is_full_grid(c)) return fail(FDW_EINVAL, "fdw_shot needs a full-grid context");
    if (c->nx <= 0 || c->nz <= 0) return fail(FDW_EINVAL, "no interior to image");
    HIP_TRY(hipSetDevice(c->device));
    int rc;
    const int nt = c->prm.nt;
    if ((rc = ensure_work_buffers(c, 8, true))) return rc;
    int ip = 0, ipp = 1;
    HIP_TRY(hipMemsetAsync(c->fld[0], 0, field_elems(c) * sizeof(float), c->stream));    // R:496-497
    HIP_TRY(hipMemsetAsync(c->fld[1], 0, field_elems(c) * sizeof(float), c->stream));
    if ((v2 && (rc = upload_rows(c, c->d_v2, v2, c->stream))) || (rc = upload_source(c, srce, nt)) || (rc = upload_gather(c, d_obs)) ||
        (rc = image_to_device(c, imloc)))
        return rc;
    if ((rc = forward_loop(c, &ip, &ipp, sx, sz, nt))) return rc;
    float *d_p = c->fld[ip], *d_pp = c->fld[ipp];
    if (P && (rc = download_rows(c, P, d_p, c->stream))) return rc;
    if (PP && (rc = download_rows(c, PP, d_pp, c->stream))) return rc;
    // the snapshots stay where the forward pass left them; the other two of the first four buffers are the spares
    float* src[4] = {d_p, d_pp, nullptr, nullptr};
    for (int i = 0, n = 2; i < 4; i++)
        if (c->fld[i] != d_p && c->fld[i] != d_pp) src[n++] = c->fld[i];
    float* const rcv[4] = {c->fld[4], c->fld[5], c->fld[6], c->fld[7]};
    if ((rc = zero_fields(c, rcv, 2))) return rc;
    if ((rc = back_loop(c, src, rcv, gz, nt))) return rc;
    if ((rc = image_to_host(c, imloc))) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));
    return FDW_OK;
}

extern "C" int fdw_shot(fdw_ctx* c, const float* v2, int sx, int sz, int gz, const float* srce, const float* d_obs,
                        float* imloc, float* P, float* PP)
{
    if (!v2) return fail(FDW_EINVAL, "NULL argument");
    if (c) c->v2_resident = false;
    return shot_impl(c, v2, sx, sz, gz, srce, d_obs, imloc, P, PP);
}

// ------------------------------------------------------------------------------------------------
// tuning / introspection
// ------------------------------------------------------------------------------------------------
extern "C" int fdw_set_store_budget(fdw_ctx* c, size_t bytes)
{
    if (!c) return fail(FDW_EINVAL, "ctx is NULL");
    c->store_budget = bytes;
    return FDW_OK;
}
extern "C" int fdw_store_segments(const fdw_ctx* c) { return c ? c->store_segments : 0; }

extern "C" int fdw_set_tuning(fdw_ctx* c, int xchunk, int wz, int use_generic, int prefetch, int two_step)
{
    if (!c) return fail(FDW_EINVAL, "ctx is NULL");
    if (xchunk < 0 || (wz != 0 && wz != 1 && wz != 2 && wz != 4) || prefetch < 0 || prefetch > 3)
        return fail(FDW_EINVAL, "bad tuning xchunk=%d wz=%d prefetch=%d", xchunk, wz, prefetch);
    c->xchunk = xchunk;
    c->wz = wz;
    c->use_generic = use_generic ? 1 : 0;
    c->prefetch = prefetch;
    c->tb = two_step < 0 ? -1 : (two_step == kPipeSteps ? kPipeSteps : (two_step > 0 ? 1 : 0));   // kPipeSteps: force the wave-pipeline kernel
    c->xchunk2 = xchunk;   // the two-step kernel shares the knob (0 = its own default)
    return FDW_OK;
}

// ------------------------------------------------------------------------------------------------
// forward-modelling producer (dialect MOD): mod_main.cpp:140-174
// ------------------------------------------------------------------------------------------------
// nsteps iterations of M:147-164 on caller-owned device arrays: d_p / d_pp are M's P / PP (roles swap every step, as there),
// d_rec [>= (it0+nsteps)][nx] receives the trace samples of iteration it at row it (NULL: not recorded).
extern "C" int fdw_dev_model_steps(fdw_ctx* c, float* d_p, float* d_pp, const float* d_v2, const float* d_srce, int sx, int sz, int gz,
                                   float* d_rec, int it0, int nsteps, void* stream)
{
    if (!c) return fail(FDW_EINVAL, "ctx is NULL");
    if (c->prm.dialect != FDW_DIALECT_MOD) return fail(FDW_ESTATE, "fdw_dev_model_steps needs a context created with dialect = FDW_DIALECT_MOD");
    hipStream_t s = pick_stream(c, stream);
    int k = 0;
    if (pipe_pays(c) && nsteps >= kPipeSteps) {
        // four steps per pass, out of place: the caller's two buffers plus two of the context's rotate.  In M's convention P is
        // the current field; a pass reads (p = P, pp = PP) and leaves PP' = u^{n+3} in out1, P' = u^{n+4} in out2.
        int rc = ensure_work_buffers(c, 4, false);
        if (rc) return rc;
        float* B[4] = {d_p, d_pp, c->fld[2], c->fld[3]};
        int iP = 0, iPP = 1;
        for (; nsteps - k >= kPipeSteps; k += kPipeSteps) {
            int o1 = -1, o2 = -1;
            for (int i = 0; i < 4; i++)
                if (i != iP && i != iPP) { (o1 < 0 ? o1 : o2) = i; }
            const int it = it0 + k;
            rc = stepn_impl(c, FDW_MODE_MOD, B[iP], B[iPP], d_v2, B[o1], B[o2], 1, d_srce ? d_srce + it : nullptr, sx, sz, s, RowRanges{},
                            d_rec ? d_rec + (size_t)it * c->nx : nullptr, gz);
            if (rc) return rc;
            iPP = o1; iP = o2;
        }
        // hand the state back in the caller's buffers: an even number of M's swaps leaves P in d_p and PP in d_pp
        const size_t bytes = field_elems(c) * sizeof(float);
        auto move = [&](int dst, int src) { return hipMemcpyAsync(B[dst], B[src], bytes, hipMemcpyDeviceToDevice, s); };
        if (iP == 1 && iPP == 0) {            // crossed: go through a context buffer (both are free here)
            HIP_TRY(move(2, 0)); HIP_TRY(move(0, 1)); HIP_TRY(move(1, 2));
        } else if (iPP == 0) {                // PP sits where P must go: B[1] is free, clear B[0] first
            HIP_TRY(move(1, 0));
            HIP_TRY(move(0, iP));
        } else {                              // B[0] is free or already holds P
            if (iP != 0) HIP_TRY(move(0, iP));
            if (iPP != 1) HIP_TRY(move(1, iPP));
        }
    }
    for (; k < nsteps; k++) {
        const int it = it0 + k;
        int rc = step_impl(c, FDW_MODE_MOD, d_p, d_pp, d_v2, 0, c->nxl, 1, d_srce ? d_srce + it : nullptr, sx, sz, nullptr, nullptr, s,
                           d_rec ? d_rec + (size_t)it * c->nx : nullptr, gz);
        if (rc) return rc;
        std::swap(d_p, d_pp);
    }
    return FDW_OK;
}

extern "C" int fdw_model_shot(fdw_ctx* c, const float* vel2, int sx, int sz, int gz, const float* srce, int nt, float* data)
{
    if (!c || !vel2 || !data || (!srce && nt > 0)) return fail(FDW_EINVAL, "NULL argument");
    if (c->prm.dialect != FDW_DIALECT_MOD) return fail(FDW_ESTATE, "fdw_model_shot needs a context created with dialect = FDW_DIALECT_MOD");
    if (!is_full_grid(c)) return fail(FDW_EINVAL, "fdw_model_shot needs a full-grid context");
    if (nt < 0) return fail(FDW_EINVAL, "nt=%d", nt);
    if (gz < c->prm.nzb || gz >= c->prm.nzb + c->nz)
        return fail(FDW_EINVAL, "receiver depth %d is not an interior column [%d,%d)", gz, c->prm.nzb, c->prm.nzb + c->nz);
    HIP_TRY(hipSetDevice(c->device));
    int rc;
    if ((rc = ensure_work_buffers(c, 2, false))) return rc;
    const size_t nx = c->nx, nrec = nx * (size_t)nt;
    if ((rc = ensure_cap(&c->d_rec, &c->rec_cap, nrec))) return rc;
    if ((rc = upload_rows(c, c->d_v2, vel2, c->stream)) || (rc = upload_source(c, srce, nt))) return rc;
    HIP_TRY(hipMemsetAsync(c->fld[0], 0, field_elems(c) * sizeof(float), c->stream));   // M:144-145
    HIP_TRY(hipMemsetAsync(c->fld[1], 0, field_elems(c) * sizeof(float), c->stream));
    // lazy damping: memory holds raw fields; as "p" a field owes one taper_apply, as "pp" two (it was damped once as the new
    // field and once more as P, M:151-152).  At it = 0 both are zero, so the count does not matter.
    if ((rc = fdw_dev_model_steps(c, c->fld[0], c->fld[1], c->d_v2, c->d_srce, sx, sz, gz, c->d_rec, 0, nt, c->stream))) return rc;
    std::vector<float> t(nrec);
    HIP_TRY(hipMemcpyAsync(t.data(), c->d_rec, nrec * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (size_t ix = 0; ix < nx; ix++)            // device [it][ix] -> data[ix][it] (M:156)
        for (size_t it = 0; it < (size_t)nt; it++) data[ix * nt + it] = t[it * nx + ix];
    return FDW_OK;
}

// ------------------------------------------------------------------------------------------------
// stored-wavefield RTM (dialect RTM_STORED): rtm_main.cpp:158-240
// ------------------------------------------------------------------------------------------------
extern "C" int fdw_rtm_stored_shot(fdw_ctx* c, const float* vel2, int sx, int sz, int gz, const float* srce, int nt, const float* dobs,
                                   size_t n_floats, int is, float* imloc)
{
    if (!c || !vel2 || !imloc || !dobs || (!srce && nt > 0)) return fail(FDW_EINVAL, "NULL argument");
    if (c->prm.dialect != FDW_DIALECT_RTM_STORED) return fail(FDW_ESTATE, "fdw_rtm_stored_shot needs a context created with dialect = FDW_DIALECT_RTM_STORED");
    if (!is_full_grid(c)) return fail(FDW_EINVAL, "fdw_rtm_stored_shot needs a full-grid context");
    if (nt < 0 || is < 0) return fail(FDW_EINVAL, "nt=%d is=%d", nt, is);
    HIP_TRY(hipSetDevice(c->device));
    int rc;
    if ((rc = ensure_work_buffers(c, 2, true))) return rc;
    const size_t nx = c->nx, fe = field_elems(c);
    // receiver samples as the reference indexes them: step it reads dobs[is][ix][nt - it]
    std::vector<float> rows(std::max<size_t>(nx * (size_t)nt, 1));
    for (int it = 0; it < nt; it++)
        for (size_t ix = 0; ix < nx; ix++) {
            const size_t k = ((size_t)is * nx + ix) * (size_t)nt + (size_t)(nt - it);
            rows[(size_t)it * nx + ix] = k < n_floats ? dobs[k] : 0.0f;
        }
    if ((rc = ensure_cap(&c->d_dobs, &c->dobs_cap, rows.size()))) return rc;
    HIP_TRY(hipMemcpy(c->d_dobs, rows.data(), rows.size() * sizeof(float), hipMemcpyHostToDevice));
    if ((rc = upload_rows(c, c->d_v2, vel2, c->stream)) || (rc = upload_source(c, srce, nt))) return rc;
    float *d_p = c->fld[0], *d_pp = c->fld[1];
    // The source wavefield of every step (rtm_main.cpp:177-181 keeps the interior; whole pitched fields here so that the imaging epilogue of
    // the step kernel can read them like any other field): swf[it] = P of step it, it = 0 .. nt-1, swf[0] = 0.  The receiver pass consumes
    // them LAST FIRST (iteration it images against swf[nt-it-1], rtm_main.cpp:224-230).
    //   * They fit the store budget (fdw_set_store_budget; default: whatever hipMalloc grants): all nt fields are kept, one segment.
    //   * They do not: CHECKPOINTING.  The steps are cut into S segments of m; the forward pass keeps, per segment, only the pair of fields it
    //     starts from (2 S fields) and runs through ONE segment buffer of m + 1 fields; the receiver pass takes the segments last first and
    //     recomputes each from its pair into that buffer before consuming it.  The recomputation is the same launches on the same inputs, so
    //     every stored field -- and with it the image -- is bit-identical to the unconstrained run; the price is one more forward pass
    //     (2 nt + nt launches instead of nt + nt).  Memory 2 S + m + 1 fields, least at m ~ sqrt(2 nt): 1 001 steps fit in 92 fields.
    const size_t fbytes = fe * sizeof(float);
    size_t budget_fields = c->store_budget ? c->store_budget / fbytes : (size_t)-1;
    if (const char* e = getenv("FDW_STORE_BUDGET_MB")) budget_fields = (size_t)atoll(e) * ((size_t)1 << 20) / fbytes;
    int m = std::max(nt, 1), S = 1;                       // segment length, segments
    float *d_seg = nullptr, *d_ck = nullptr;
    struct Free { float*& p; ~Free() { if (p) (void)hipFree(p); } } g1{d_seg}, g2{d_ck};
    bool whole = (size_t)nt + 1 <= budget_fields;
    const int forced_m = getenv("FDW_STORE_SEGMENT") ? atoi(getenv("FDW_STORE_SEGMENT")) : 0;      // tests: checkpoint with this segment length whatever fits
    if (forced_m > 0 && forced_m < nt) whole = false;
    if (whole && nt > 0 && hipMalloc((void**)&d_seg, fbytes * ((size_t)nt + 1)) != hipSuccess) {      // no budget given and the device says no: checkpoint into what is free
        (void)hipGetLastError();
        d_seg = nullptr;
        whole = false;
        size_t fr = 0, tot = 0;
        HIP_TRY(hipMemGetInfo(&fr, &tot));
        budget_fields = (size_t)(0.9 * (double)fr) / fbytes;
    }
    if (!whole && nt > 0) {
        // the longest segment that fits: fewest restarts, same recomputation cost (one forward pass) whatever m is
        int best = 0;
        for (int mm = nt; mm >= 1; mm--)
            if (2 * (size_t)((nt + mm - 1) / mm) + (size_t)mm + 1 <= budget_fields) { best = mm; break; }
        if (forced_m > 0 && forced_m < nt) best = forced_m;
        if (best == 0)
            return fail(FDW_ENOMEM, "the stored source fields of %d steps need at least %d fields of %zu bytes with checkpointing (%d + 1 without); the store budget holds %zu",
                        nt, (int)(2 * std::ceil(std::sqrt(nt / 2.0)) + std::ceil(std::sqrt(2.0 * nt)) + 1), fbytes, nt, budget_fields);
        m = best;
        S = (nt + m - 1) / m;
        hipError_t e1 = hipMalloc((void**)&d_seg, fbytes * ((size_t)m + 1));
        hipError_t e2 = e1 == hipSuccess ? hipMalloc((void**)&d_ck, fbytes * 2 * (size_t)S) : e1;
        if (e1 != hipSuccess || e2 != hipSuccess) return fail(FDW_ENOMEM, "checkpointed store (%d segments of %d steps, %zu bytes per field): %s", S, m, fbytes, hipGetErrorString(e2));
    }
    c->store_segments = S;
    auto seg = [&](int j) { return d_seg + (size_t)j * fe; };
    auto ck = [&](int s, int which) { return d_ck + ((size_t)2 * s + which) * fe; };      // which 0: swf[s m - 1], 1: swf[s m]
    // one segment of the forward loop: fields swf[s0 .. s0+len] into seg[0 .. len]; prev = swf[s0 - 1], seg[0] must hold swf[s0].
    // Step it reads p = swf[it], pp = swf[it-1] and writes the new field straight into the next slot -- no copy per step.  (Cells a launch
    // never stores -- the padding columns -- are zero in every slot from the memset below and stay so.)
    auto run_segment = [&](int s0, int len, const float* prev) -> int {
        for (int j = 0; j < len; j++) {
            const int it = s0 + j;
            float* pp_in = j > 0 ? seg(j - 1) : const_cast<float*>(prev);                  // only read (out is given)
            int r = step_impl(c, FDW_MODE_DD_FWD, seg(j), pp_in, c->d_v2, 0, c->nxl, 1, c->d_srce + it, sx, sz, nullptr, nullptr, c->stream, nullptr, 0, nullptr, seg(j + 1));
            if (r) return r;
        }
        return FDW_OK;
    };
    if (nt > 0) {
        HIP_TRY(hipMemsetAsync(d_seg, 0, fbytes * ((size_t)m + 1), c->stream));          // swf[0] = 0 (rtm_main.cpp:161-162) and every padding column
        HIP_TRY(hipMemsetAsync(d_pp, 0, fbytes, c->stream));                              // the field before step 0
        if (d_ck) HIP_TRY(hipMemsetAsync(d_ck, 0, fbytes * 2 * (size_t)S, c->stream));
        for (int s = 0; s < S; s++) {
            const int s0 = s * m, len = std::min(m, nt - s0);
            if (s > 0) HIP_TRY(hipMemcpyAsync(seg(0), ck(s, 1), fbytes, hipMemcpyDeviceToDevice, c->stream));
            if ((rc = run_segment(s0, len, s > 0 ? ck(s, 0) : d_pp))) return rc;
            if (s + 1 < S) {                              // what the next segment starts from: swf[s0 + m - 1], swf[s0 + m]
                HIP_TRY(hipMemcpyAsync(ck(s + 1, 0), seg(m - 1), fbytes, hipMemcpyDeviceToDevice, c->stream));
                HIP_TRY(hipMemcpyAsync(ck(s + 1, 1), seg(m), fbytes, hipMemcpyDeviceToDevice, c->stream));
            }
        }
    }
    HIP_TRY(hipMemsetAsync(d_p, 0, fbytes, c->stream));     // rtm_main.cpp:187-189
    HIP_TRY(hipMemsetAsync(d_pp, 0, fbytes, c->stream));
    HIP_TRY(hipMemsetAsync(c->d_img, 0, fbytes, c->stream));
    float* d_zero = nullptr;                                // checkpointing: the zero field before step 0 (d_pp is a receiver field from here on)
    struct Free g3{d_zero};
    if (S > 1) {
        if (hipMalloc((void**)&d_zero, fbytes) != hipSuccess) return fail(FDW_ENOMEM, "hipMalloc(%zu) failed", fbytes);
        HIP_TRY(hipMemsetAsync(d_zero, 0, fbytes, c->stream));
    }
    int it = 0;
    for (int s = S - 1; s >= 0 && nt > 0; s--) {
        const int s0 = s * m, len = std::min(m, nt - s0);
        if (s < S - 1) {                                    // (the last segment is still in the buffer from the forward pass)
            HIP_TRY(hipMemcpyAsync(seg(0), s > 0 ? ck(s, 1) : d_zero, fbytes, hipMemcpyDeviceToDevice, c->stream));
            if ((rc = run_segment(s0, len, s > 0 ? ck(s, 0) : d_zero))) return rc;
        }
        for (int j = len - 1; j >= 0; j--, it++) {          // receiver iteration `it` images against swf[nt - it - 1] = swf[s0 + j]
            rc = step_impl(c, FDW_MODE_DD_RECV, d_p, d_pp, c->d_v2, 0, c->nxl, 1, c->d_dobs + (size_t)it * nx, -1, gz, seg(j), c->d_img, c->stream);
            if (rc) return rc;
            std::swap(d_p, d_pp);
        }
    }
    if ((rc = image_to_host(c, imloc))) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));
    return FDW_OK;
}

// ------------------------------------------------------------------------------------------------
// image post-processing (row f3): laplace.f90
// ------------------------------------------------------------------------------------------------
extern "C" int fdw_image_laplacian(int device, const float* img, int nx, int nz, float dx, float dz, float* out)
{
    if (!img || !out || nx < 1 || nz < 1 || !(dx > 0.0f) || !(dz > 0.0f)) return fail(FDW_EINVAL, "image_laplacian: bad argument");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || device < 0 || device >= ndev)
        return fail(FDW_ENODEVICE, "image_laplacian: no HIP device %d (%s); libfdwave has no CPU path", device, e != hipSuccess ? hipGetErrorString(e) : "out of range");
    HIP_TRY(hipSetDevice(device));
    const size_t bytes = (size_t)nx * nz * sizeof(float);
    float *d_in = nullptr, *d_out = nullptr;
    if (hipMalloc((void**)&d_in, bytes) != hipSuccess || hipMalloc((void**)&d_out, bytes) != hipSuccess) {
        if (d_in) (void)hipFree(d_in);
        return fail(FDW_ENOMEM, "image_laplacian: hipMalloc(%zu) failed", bytes);
    }
    int rc = FDW_OK;
    if ((e = hipMemcpy(d_in, img, bytes, hipMemcpyHostToDevice)) != hipSuccess || (e = launch_image_laplacian(d_in, d_out, nx, nz, dx, dz, nullptr)) != hipSuccess ||
        (e = hipMemcpy(out, d_out, bytes, hipMemcpyDeviceToHost)) != hipSuccess)
        rc = fail(FDW_EHIP, "image_laplacian: %s", hipGetErrorString(e));
    (void)hipFree(d_in);
    (void)hipFree(d_out);
    return rc;
}

// The reference's image comparer models/marmousi/psnr (an ELF without source; behaviour read off its output: MSE = mean (a-b)^2, RMSE, SNR =
// 10 log10(sum b^2 / sum (a-b)^2), PSNR = 20 log10(max |b| / RMSE), the difference a - b written out) on the device.
extern "C" int fdw_image_compare(int device, const float* a, const float* b, size_t n, float* diff, double stats[4], int exact_sums)
{
    if (!a || !b || !stats || n == 0) return fail(FDW_EINVAL, "image_compare: bad argument");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || device < 0 || device >= ndev)
        return fail(FDW_ENODEVICE, "image_compare: no HIP device %d (%s); libfdwave has no CPU path", device, e != hipSuccess ? hipGetErrorString(e) : "out of range");
    HIP_TRY(hipSetDevice(device));
    const int nblocks = (int)std::min<size_t>(1024, (n + 255) / 256);
    float *d_a = nullptr, *d_b = nullptr, *d_d = nullptr;
    double* d_w = nullptr;
    const size_t bytes = n * sizeof(float);
    int rc = FDW_OK;
    if (hipMalloc((void**)&d_a, bytes) != hipSuccess || hipMalloc((void**)&d_b, bytes) != hipSuccess || (diff && hipMalloc((void**)&d_d, bytes) != hipSuccess) ||
        hipMalloc((void**)&d_w, (3 * (size_t)nblocks + 8) * sizeof(double)) != hipSuccess)
        rc = fail(FDW_ENOMEM, "image_compare: hipMalloc failed");
    double h[5] = {0, 0, 0, 0, 0};
    if (rc == FDW_OK) {
        if ((e = hipMemcpy(d_a, a, bytes, hipMemcpyHostToDevice)) != hipSuccess || (e = hipMemcpy(d_b, b, bytes, hipMemcpyHostToDevice)) != hipSuccess ||
            (e = launch_image_compare(d_a, d_b, n, d_d, d_w + 8, nblocks, d_w, exact_sums ? 0 : 1, nullptr)) != hipSuccess ||
            (e = hipMemcpy(h, d_w, sizeof h, hipMemcpyDeviceToHost)) != hipSuccess || (diff && (e = hipMemcpy(diff, d_d, bytes, hipMemcpyDeviceToHost)) != hipSuccess))
            rc = fail(FDW_EHIP, "image_compare: %s", hipGetErrorString(e));
    }
    for (void* p : {(void*)d_a, (void*)d_b, (void*)d_d, (void*)d_w})
        if (p) (void)hipFree(p);
    if (rc != FDW_OK) return rc;
    if (exact_sums) {
        const double mse = h[0] / (double)n, rmse = std::sqrt(mse);
        stats[0] = mse;
        stats[1] = rmse;
        stats[2] = 10.0 * std::log10(h[1] / h[0]);
        stats[3] = 20.0 * std::log10(h[2] / rmse);
    } else {      // the tool's own arithmetic (pinned to its output by the oracle's restatement): fp32 sums, fp32 quotient and root, PSNR kept in fp32
        const float sd = (float)h[3], sb = (float)h[4], mx = (float)h[2];
        const float mse = sd / (float)n, rmse = sqrtf(mse);
        stats[0] = mse;
        stats[1] = rmse;
        stats[2] = 10.0 * std::log10((double)(sb / sd));
        stats[3] = (float)(20.0 * std::log10((double)(mx / rmse)));
    }
    return FDW_OK;
}

extern "C" int fdw_get_tables(const fdw_ctx* c, float* coefs_x, float* coefs_z, float* taper_x, float* taper_z)
{
    if (!c) return fail(FDW_EINVAL, "ctx is NULL");
    if (coefs_x) memcpy(coefs_x, c->cx, (c->prm.order + 1) * sizeof(float));
    if (coefs_z) memcpy(coefs_z, c->cz, (c->prm.order + 1) * sizeof(float));
    if (taper_x && c->prm.nxb > 0) memcpy(taper_x, c->taper_x.data(), c->prm.nxb * sizeof(float));
    if (taper_z && c->prm.nzb > 0) memcpy(taper_z, c->taper_z.data(), c->prm.nzb * sizeof(float));
    return FDW_OK;
}

extern "C" int fdw_get_extents(const fdw_ctx* c, int* xlim, int* zlim, int* ztap)
{
    if (!c) return fail(FDW_EINVAL, "ctx is NULL");
    if (xlim) *xlim = c->xlim;
    if (zlim) *zlim = c->zlim;
    if (ztap) *ztap = c->ztap;
    return FDW_OK;
}

extern "C" int fdw_selftest(fdw_ctx* c)
{
    if (!c) return fail(FDW_EINVAL, "ctx is NULL");
    HIP_TRY(hipSetDevice(c->device));
    float h_src[64], h_out[384];
    for (int i = 0; i < 64; i++) h_src[i] = 100.0f + i;
    for (int i = 0; i < 384; i++) h_out[i] = -7.0f;
    float* d = nullptr;
    HIP_TRY(hipMalloc((void**)&d, (64 + 384) * sizeof(float)));
    hipError_t e = hipMemcpy(d, h_src, sizeof h_src, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d + 64, h_out, sizeof h_out, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = launch_selftest(d, d + 64, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = hipMemcpy(h_out, d + 64, sizeof h_out, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(FDW_EHIP, "selftest: %s", hipGetErrorString(e));
    for (int i = 0; i < 64; i++) {
        // lane 0 (63) has no neighbour to take from: the kernels never use what it receives (a halo lane / replaced by the strip halo)
        const float up = i == 0 ? h_out[0] : h_src[i - 1], dn = i == 63 ? h_out[127] : h_src[i + 1];
        if (h_out[i] != up || h_out[64 + i] != dn)
            return fail(FDW_EHIP, "selftest: lane exchange mismatch at lane %d (up %g want %g, down %g want %g)", i,
                        (double)h_out[i], (double)up, (double)h_out[64 + i], (double)dn);
        for (int k = 0; k < 4; k++) {
            const float want = (i >= 2 && i <= 61) ? h_src[i] : -7.0f;   // out-of-range lanes must leave memory alone
            if (h_out[128 + 4 * i + k] != want)
                return fail(FDW_EHIP, "selftest: range-predicated buffer store wrote %g at cell %d (want %g)",
                            (double)h_out[128 + 4 * i + k], i, (double)want);
        }
    }
    return FDW_OK;
}

// ------------------------------------------------------------------------------------------------
// random-border velocity model generated on the device
// ------------------------------------------------------------------------------------------------
namespace {
constexpr int L = fdw::kRandLag;
using Mat = std::vector<unsigned>;     // 31 x 31 over Z/2^32, row-major

Mat mat_mul(const Mat& a, const Mat& b)
{
    Mat r((size_t)L * L, 0u);
    for (int i = 0; i < L; i++)
        for (int k = 0; k < L; k++) {
            const unsigned aik = a[(size_t)i * L + k];
            if (!aik) continue;
            for (int j = 0; j < L; j++) r[(size_t)i * L + j] += aik * b[(size_t)k * L + j];
        }
    return r;
}

struct RandTables {
    Mat pow2[64];        // M^(2^j): one stream position per application
    unsigned glo[65][L]; // x^(31 l) mod P, l = 0..64, P = x^31 - x^28 - 1 (the generator's characteristic polynomial)
    unsigned w0[L];      // window before the first step of a seed-1 generator (before glibc's 310 discarded outputs)
};

// The window W = (y[K-31] .. y[K-1]) advances by  W' = (w1 .. w30, w0 + w28)  (y[K] = y[K-31] + y[K-3]).
const RandTables& rand_tables()
{
    static RandTables t;
    static std::once_flag once;
    std::call_once(once, [] {
        Mat m((size_t)L * L, 0u);
        for (int i = 0; i + 1 < L; i++) m[(size_t)i * L + i + 1] = 1;
        m[(size_t)(L - 1) * L + 0] = 1;
        m[(size_t)(L - 1) * L + 28] = 1;
        t.pow2[0] = m;
        for (int j = 1; j < 64; j++) t.pow2[j] = mat_mul(t.pow2[j - 1], t.pow2[j - 1]);
        std::memset(t.glo, 0, sizeof t.glo);
        t.glo[0][0] = 1u;
        for (int l = 0; l < 64; l++) {
            unsigned g[L];
            std::memcpy(g, t.glo[l], sizeof g);
            for (int k = 0; k < L; k++) {                      // times x: the x^31 term folds into x^28 + 1
                const unsigned top = g[L - 1];
                for (int i = L - 1; i > 0; i--) g[i] = g[i - 1];
                g[0] = top;
                g[28] += top;
            }
            std::memcpy(t.glo[l + 1], g, sizeof g);
        }
        // glibc srandom_r(1): r[0] = seed, r[i] = 16807 r[i-1] mod (2^31 - 1) by Schrage's method; front pointer at r[3], rear at r[0]
        unsigned r[L];
        int word = 1;
        r[0] = 1u;
        for (int k = 1; k < L; k++) {
            const long hi = word / 127773, lo = word % 127773;
            long v = 16807 * lo - 2836 * hi;
            if (v < 0) v += 2147483647;
            word = (int)v;
            r[k] = (unsigned)word;
        }
        // step t adds slot t mod 31 into slot (t + 3) mod 31, i.e. y[t] = y[t-31] + y[t-3] with y[s] = r[(s + 34) mod 31] for s in [-31, 0)
        for (int s = 0; s < L; s++) t.w0[s] = r[(s + 3) % L];
    });
    return t;
}

fdw::RandWindow window_at(unsigned long long k)
{
    const RandTables& t = rand_tables();
    unsigned w[L], v[L];
    std::memcpy(w, t.w0, sizeof w);
    for (int j = 0; j < 64; j++)
        if ((k >> j) & 1) {
            const Mat& m = t.pow2[j];
            for (int i = 0; i < L; i++) {
                unsigned acc = 0;
                for (int c = 0; c < L; c++) acc += m[(size_t)i * L + c] * w[c];
                v[i] = acc;
            }
            std::memcpy(w, v, sizeof w);
        }
    fdw::RandWindow out;
    std::memcpy(out.w, w, sizeof w);
    return out;
}

constexpr unsigned long long kGlibcDiscard = 310;     // outputs srandom_r() throws away

// a * b mod P
void poly_mul(const unsigned* a, const unsigned* b, unsigned* out)
{
    unsigned c[2 * L - 1] = {0};
    for (int i = 0; i < L; i++)
        for (int j = 0; j < L; j++) c[i + j] += a[i] * b[j];
    for (int k = 2 * L - 2; k >= L; k--) {                    // x^k = x^(k-3) + x^(k-31)
        c[k - 3] += c[k];
        c[k - L] += c[k];
    }
    std::memcpy(out, c, L * sizeof(unsigned));
}

fdw::RandBase base_at(unsigned long long k)
{
    const fdw::RandWindow w = window_at(k);
    fdw::RandBase b;
    std::memcpy(b.y, w.w, sizeof w.w);
    for (int s = 0; s < L - 1; s++) b.y[L + s] = b.y[s] + b.y[s + 28];     // y[K+s] = y[K+s-31] + y[K+s-3]
    return b;
}

// device table for up to ndraws draws per launch: [31][64] lane factors, then one block factor per 64 threads
int ensure_rand_tables(fdw_ctx* c, long long ndraws)
{
    const long long threads = (ndraws + L - 1) / L;
    const int need = (int)std::max<long long>((threads + 63) / 64, 1);
    if (c->d_jump && c->njump >= need) return FDW_OK;
    if (c->d_jump) (void)hipFree(c->d_jump);
    c->d_jump = nullptr;
    const RandTables& t = rand_tables();
    std::vector<unsigned> flat((size_t)L * 64 + (size_t)need * L);
    for (int i = 0; i < L; i++)
        for (int l = 0; l < 64; l++) flat[(size_t)i * 64 + l] = t.glo[l][i];
    unsigned* hi = &flat[(size_t)L * 64];
    std::memset(hi, 0, L * sizeof(unsigned));
    hi[0] = 1u;
    for (int h = 1; h < need; h++) poly_mul(hi + (size_t)(h - 1) * L, t.glo[64], hi + (size_t)h * L);
    hipError_t e = hipMalloc((void**)&c->d_jump, flat.size() * sizeof(unsigned));
    if (e != hipSuccess) return fail(FDW_ENOMEM, "hipMalloc failed: %s", hipGetErrorString(e));
    HIP_TRY(hipMemcpy(c->d_jump, flat.data(), flat.size() * sizeof(unsigned), hipMemcpyHostToDevice));
    c->njump = need;
    return FDW_OK;
}
}  // namespace

static int ensure_draws(fdw_ctx* c, long long n)
{
    n = std::max<long long>(n, 1);
    int rc = ensure_rand_tables(c, n);
    if (rc || c->draws_cap >= n) return rc;
    if (c->d_draws) (void)hipFree(c->d_draws);
    c->d_draws = nullptr;
    c->draws_cap = 0;
    hipError_t e = hipMalloc((void**)&c->d_draws, (size_t)n * sizeof(int));
    if (e != hipSuccess) return fail(FDW_ENOMEM, "hipMalloc failed: %s", hipGetErrorString(e));
    c->draws_cap = n;
    return FDW_OK;
}

extern "C" long long fdw_border_draws(int nx, int nz, int nxb, int nzb)
{
    if (nx < 0 || nz < 0 || nxb < 0 || nzb < 0) return -1;
    return (long long)nx * nzb + 2ll * nz * nxb + 2ll * nzb * (nzb + 1);
}

extern "C" int fdw_rand_stream(fdw_ctx* c, unsigned long long draw_offset, long long n, int* out)
{
    if (!c || (!out && n > 0) || n < 0) return fail(FDW_EINVAL, "bad argument");
    if (n == 0) return FDW_OK;
    HIP_TRY(hipSetDevice(c->device));
    int rc = ensure_rand_tables(c, n);
    if (rc) return rc;
    int* d = nullptr;
    hipError_t e = hipMalloc((void**)&d, (size_t)n * sizeof(int));
    if (e != hipSuccess) return fail(FDW_ENOMEM, "hipMalloc failed: %s", hipGetErrorString(e));
    e = launch_rand_stream(base_at(kGlibcDiscard + draw_offset), c->d_jump, n, d, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(out, d, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(FDW_EHIP, "rand_stream: %s", hipGetErrorString(e));
    return FDW_OK;
}

extern "C" int fdw_model_resident(fdw_ctx* c, const float* vp)
{
    if (!c || !vp) return fail(FDW_EINVAL, "NULL argument");
    if (c->prm.dialect != FDW_DIALECT_RTM) return fail(FDW_ESTATE, "the random-border model belongs to the RTM dialect");
    if (!is_full_grid(c)) return fail(FDW_EINVAL, "fdw_model_resident needs a full-grid context");
    if (c->nx <= 0 || c->nz <= 0) return fail(FDW_EINVAL, "no interior");
    const int nxb = c->prm.nxb, nzb = c->prm.nzb;
    if (nxb == 1 || nzb == 1) return fail(FDW_EINVAL, "a border of one cell divides by zero in the reference's ramp (nb - 1)");
    if (nzb > c->prm.nxe) return fail(FDW_EINVAL, "nzb=%d > nxe=%d: the reference's corner loops leave the array", nzb, c->prm.nxe);
    HIP_TRY(hipSetDevice(c->device));
    int rc = ensure_work_buffers(c, 0, false);
    if (rc) return rc;
    const size_t n = (size_t)c->nx * c->nz;
    if (!c->d_vp) {
        hipError_t e = hipMalloc((void**)&c->d_vp, n * sizeof(float));
        if (e != hipSuccess) return fail(FDW_ENOMEM, "hipMalloc failed: %s", hipGetErrorString(e));
    }
    if ((rc = ensure_draws(c, fdw_border_draws(c->nx, c->nz, nxb, nzb)))) return rc;
    if ((rc = alloc_zero(&c->d_vpe, field_elems(c)))) return rc;
    HIP_TRY(hipMemcpyAsync(c->d_vp, vp, n * sizeof(float), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->model_resident = true;
    c->v2_resident = false;
    return FDW_OK;
}

extern "C" int fdw_dev_extendvel_linear(fdw_ctx* c, unsigned long long draw_offset, float* vel_out)
{
    if (!c) return fail(FDW_EINVAL, "ctx is NULL");
    if (!c->model_resident) return fail(FDW_ESTATE, "no resident model: call fdw_model_resident first");
    HIP_TRY(hipSetDevice(c->device));
    const long long n = fdw_border_draws(c->nx, c->nz, c->prm.nxb, c->prm.nzb);
    hipError_t e = launch_rand_stream(base_at(kGlibcDiscard + draw_offset), c->d_jump, n, c->d_draws, c->stream);
    if (e != hipSuccess) return fail(FDW_EHIP, "rand_stream launch failed: %s", hipGetErrorString(e));
    BorderArgs a{c->d_vp, c->d_draws, c->d_vpe, c->d_v2, c->nx, c->nz, c->prm.nxb, c->prm.nzb, c->pitch};
    e = launch_extendvel(a, c->stream);
    if (e != hipSuccess) return fail(FDW_EHIP, "extendvel launch failed: %s", hipGetErrorString(e));
    c->v2_resident = true;
    if (vel_out) {
        int rc = download_rows(c, vel_out, c->d_vpe, c->stream);
        if (rc) return rc;
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    return FDW_OK;
}

extern "C" int fdw_shot_resident(fdw_ctx* c, int sx, int sz, int gz, const float* srce, const float* d_obs, float* imloc, float* P, float* PP)
{
    if (!c) return fail(FDW_EINVAL, "ctx is NULL");
    if (!c->v2_resident) return fail(FDW_ESTATE, "no resident squared model: call fdw_dev_extendvel_linear first");
    return shot_impl(c, nullptr, sx, sz, gz, srce, d_obs, imloc, P, PP);
}

// ------------------------------------------------------------------------------------------------
// a batch of shots through ONE launch per time step
// ------------------------------------------------------------------------------------------------
// A shot of a deck of the reference's size (new_mod: 495 x 375 extended) is 3400 dependent launches of kernels that fill a few percent
// of the chip and last 3.5-7 us each: latency, not bandwidth.  Shots are independent (R:480-529 couples them only through the image sum),
// so `nshots` of them on one geometry go through the same launches side by side: gridDim.y = shot, every field pointer offset by
// shot * field, the gather by shot * nx * nt, the source row by shot * dsx (the reference's sx = fsx + is * ds, R:405-407).
static bool batch_ok(const fdw_ctx* c)
{
    if (c->prm.dialect == FDW_DIALECT_MOD) return c->h <= kMaxFastHalfOrder && !c->use_generic && !pipe_pays(c);
    return c->prm.dialect == FDW_DIALECT_RTM && c->h <= kMaxFastHalfOrder && !c->use_generic && !c->no_fused_back && !two_step_pays(c) &&
           !pipe_pays(c) && c->prm.nxb + c->nx <= c->upd_x1 /* no receiver rows outside the time-stepped extent */;
}

extern "C" int fdw_shot_batch_max(const fdw_ctx* c)
{
    if (!c || !is_full_grid(c) || !batch_ok(c)) return 1;
    const long waves = (long)c->nxl * ((c->pitch + 255) / 256);        // one-row-per-wave regime: waves one shot launches
    const long by_fill = 32768 / std::max<long>(waves, 1);            // measured on new_mod-sized shots: 18.6 ms per shot alone, 5.6 at 8, 4.4 at 16, 3.7 at 32
    const long by_mem = (long)(((size_t)6 << 30) / (11 * field_elems(c) * sizeof(float)));
    return (int)std::max<long>(1, std::min<long>({by_fill, by_mem, 64}));
}

// Per-shot copies of what a batch needs: the RTM loop (need_all) its eight fields, v2, image and gathers; the modelling loop only two
// fields and the gathers.  On an allocation failure whatever was allocated is released again (the callers then run the shots one by one).
static int ensure_batch_buffers(fdw_ctx* c, int n, bool need_all)
{
    const size_t nx = c->nx, nt = std::max(c->prm.nt, 1);
    if (c->batch_cap >= n && (c->batch_all || !need_all)) return FDW_OK;
    float** all[] = {&c->bfld[0], &c->bfld[1], &c->bfld[2], &c->bfld[3], &c->bfld[4], &c->bfld[5], &c->bfld[6], &c->bfld[7], &c->b_v2, &c->b_img, &c->b_dobs};
    auto release = [&] {
        for (float** q : all) {
            if (*q) (void)hipFree(*q);
            *q = nullptr;
        }
        c->batch_cap = 0;
        c->batch_all = false;
    };
    release();
    for (float** q : all) {
        const bool wanted = need_all || q == &c->bfld[0] || q == &c->bfld[1] || q == &c->b_dobs;
        if (!wanted) continue;
        const size_t elems = (q == &c->b_dobs ? nx * nt : field_elems(c)) * (size_t)n;
        hipError_t e = hipMalloc((void**)q, elems * sizeof(float));
        if (e == hipSuccess) e = hipMemset(*q, 0, elems * sizeof(float));
        if (e != hipSuccess) {
            (void)hipGetLastError();
            release();
            return fail(FDW_ENOMEM, "batch of %d shots: hipMalloc / hipMemset(%zu bytes) failed: %s", n, elems * sizeof(float), hipGetErrorString(e));
        }
    }
    HIP_TRY(hipDeviceSynchronize());
    c->batch_cap = n;
    c->batch_all = need_all;
    return FDW_OK;
}

namespace {
struct BatchScope {      // the context's single-shot buffers step aside for the batch ones while a batch runs
    fdw_ctx* c;
    explicit BatchScope(fdw_ctx* ctx, int n, int dsx) : c(ctx) { swap(); c->nbatch = n; c->batch_dsx = dsx; c->batch_nt = c->prm.nt; }
    ~BatchScope() { swap(); c->nbatch = 1; c->batch_dsx = 0; }
    void swap()
    {
        for (int i = 0; i < 8; i++) std::swap(c->fld[i], c->bfld[i]);
        std::swap(c->d_v2, c->b_v2);
        std::swap(c->d_img, c->b_img);
        std::swap(c->d_dobs, c->b_dobs);
    }
};
}  // namespace

extern "C" int fdw_shot_batch(fdw_ctx* c, int nshots, const float* v2_all, unsigned long long draw_offset, int sx0, int dsx, int sz, int gz,
                              const float* srce, const float* d_obs, float* imloc)
{
    if (!c || !srce || !d_obs || !imloc) return fail(FDW_EINVAL, "NULL argument");
    if (nshots < 1) return fail(FDW_EINVAL, "nshots=%d", nshots);
    if (!is_full_grid(c)) return fail(FDW_EINVAL, "fdw_shot_batch needs a full-grid context");
    if (c->nx <= 0 || c->nz <= 0) return fail(FDW_EINVAL, "no interior to image");
    if (!v2_all && !c->model_resident) return fail(FDW_ESTATE, "no model: pass v2_all or call fdw_model_resident first");
    const int nt = c->prm.nt;
    const size_t ni = (size_t)c->nx * c->nz, ne = (size_t)c->prm.nxe * c->prm.nze, ng = (size_t)c->nx * nt, fe = field_elems(c);
    const int sx_last = sx0 + (nshots - 1) * dsx;
    if (std::min(sx0, sx_last) < 0 || std::max(sx0, sx_last) >= c->prm.nxe || std::max(sx0, sx_last) >= c->upd_x1)
        return fail(FDW_EINVAL, "source rows %d..%d leave the rows the reference time-steps (< %d)", sx0, sx_last, c->upd_x1);
    HIP_TRY(hipSetDevice(c->device));
    int rc;
    if (nshots == 1 || !batch_ok(c)) {      // nothing to gain or a regime the batched launches do not cover: the shots one after the other
        for (int b = 0; b < nshots; b++) {
            if (!v2_all && (rc = fdw_dev_extendvel_linear(c, draw_offset + (unsigned long long)b * fdw_border_draws(c->nx, c->nz, c->prm.nxb, c->prm.nzb), nullptr)))
                return rc;
            if ((rc = shot_impl(c, v2_all ? v2_all + b * ne : nullptr, sx0 + b * dsx, sz, gz, srce, d_obs + b * ng, imloc + b * ni, nullptr, nullptr))) return rc;
        }
        return FDW_OK;
    }
    if ((rc = ensure_work_buffers(c, 8, true))) return rc;
    if ((rc = ensure_batch_buffers(c, nshots, true)) == FDW_ENOMEM) {          // no room for the batch: the shots one after the other
        for (int b = 0; b < nshots; b++) {
            if (!v2_all && (rc = fdw_dev_extendvel_linear(c, draw_offset + (unsigned long long)b * fdw_border_draws(c->nx, c->nz, c->prm.nxb, c->prm.nzb), nullptr)))
                return rc;
            if ((rc = shot_impl(c, v2_all ? v2_all + b * ne : nullptr, sx0 + b * dsx, sz, gz, srce, d_obs + b * ng, imloc + b * ni, nullptr, nullptr))) return rc;
        }
        return FDW_OK;
    }
    if (rc || (rc = upload_source(c, srce, nt))) return rc;
    hipStream_t s = c->stream;
    if ((rc = gathers_to_device(c, d_obs, c->b_dobs, nshots))) return rc;      // [shot][nx][nt] -> [shot][nt][nx]
    const long long draws = fdw_border_draws(c->nx, c->nz, c->prm.nxb, c->prm.nzb);
    if (!v2_all) {      // the shots' draws are consecutive in the stream: one launch generates them all
        HIP_TRY(hipStreamSynchronize(s));                     // a larger draw buffer replaces one the stream may still be reading
        if ((rc = ensure_draws(c, draws * nshots))) return rc;
        hipError_t e = launch_rand_stream(base_at(kGlibcDiscard + draw_offset), c->d_jump, draws * nshots, c->d_draws, s);
        if (e != hipSuccess) return fail(FDW_EHIP, "rand_stream launch failed: %s", hipGetErrorString(e));
    }
    for (int b = 0; b < nshots; b++) {
        if (v2_all) {
            HIP_TRY(hipMemcpy2DAsync(c->b_v2 + b * fe, (size_t)c->pitch * sizeof(float), v2_all + b * ne, (size_t)c->prm.nze * sizeof(float),
                                     (size_t)c->prm.nze * sizeof(float), c->nxl, hipMemcpyHostToDevice, s));
        } else {
            BorderArgs ba{c->d_vp, c->d_draws + (size_t)b * draws, nullptr, c->b_v2 + b * fe, c->nx, c->nz, c->prm.nxb, c->prm.nzb, c->pitch};
            hipError_t e = launch_extendvel(ba, s);
            if (e != hipSuccess) return fail(FDW_EHIP, "border model launch failed: %s", hipGetErrorString(e));
        }
    }
    for (int i = 0; i < 8; i++) HIP_TRY(hipMemsetAsync(c->bfld[i], 0, fe * nshots * sizeof(float), s));    // R:496-497, R:511-514
    HIP_TRY(hipMemsetAsync(c->b_img, 0, fe * nshots * sizeof(float), s));
    for (int b = 0; b < nshots; b++)
        HIP_TRY(hipMemcpy2DAsync(c->b_img + b * fe + (size_t)c->prm.nxb * c->pitch + c->prm.nzb, (size_t)c->pitch * sizeof(float), imloc + b * ni,
                                 (size_t)c->nz * sizeof(float), (size_t)c->nz * sizeof(float), c->nx, hipMemcpyHostToDevice, s));
    {
        BatchScope scope(c, nshots, dsx);
        int ip = 0, ipp = 1;
        if ((rc = fdw_dev_steps2(c, c->fld, c->d_v2, c->d_srce, sx0, sz, 0, nt, 0, &ip, &ipp, s))) return rc;
        for (int b = 0; b < nshots && nt > 0; b++)
            if ((rc = fdw_dev_taper_finalize(c, c->fld[ip] + b * fe, s))) return rc;
        float *d_p = c->fld[ip], *d_pp = c->fld[ipp];
        float* src[4] = {d_p, d_pp, nullptr, nullptr};
        for (int i = 0, n = 2; i < 4; i++)
            if (c->fld[i] != d_p && c->fld[i] != d_pp) src[n++] = c->fld[i];
        float* const rcv[4] = {c->fld[4], c->fld[5], c->fld[6], c->fld[7]};
        if ((rc = back_loop(c, src, rcv, gz, nt))) return rc;
    }
    for (int b = 0; b < nshots; b++)
        HIP_TRY(hipMemcpy2DAsync(imloc + b * ni, (size_t)c->nz * sizeof(float), c->b_img + b * fe + (size_t)c->prm.nxb * c->pitch + c->prm.nzb,
                                 (size_t)c->pitch * sizeof(float), (size_t)c->nz * sizeof(float), c->nx, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return FDW_OK;
}

// mod_main's shot loop (M:140-174) for `nshots` consecutive shots on its one velocity model, one launch per time step for all of them
extern "C" int fdw_model_shot_batch(fdw_ctx* c, int nshots, const float* vel2, int sx0, int dsx, int sz, int gz, const float* srce, int nt, float* data)
{
    if (!c || !vel2 || !data || (!srce && nt > 0)) return fail(FDW_EINVAL, "NULL argument");
    if (c->prm.dialect != FDW_DIALECT_MOD) return fail(FDW_ESTATE, "fdw_model_shot_batch needs a context created with dialect = FDW_DIALECT_MOD");
    if (!is_full_grid(c)) return fail(FDW_EINVAL, "fdw_model_shot_batch needs a full-grid context");
    if (nshots < 1 || nt < 0) return fail(FDW_EINVAL, "nshots=%d nt=%d", nshots, nt);
    if (gz < c->prm.nzb || gz >= c->prm.nzb + c->nz)
        return fail(FDW_EINVAL, "receiver depth %d is not an interior column [%d,%d)", gz, c->prm.nzb, c->prm.nzb + c->nz);
    const int sx_last = sx0 + (nshots - 1) * dsx;
    if (std::min(sx0, sx_last) < 0 || std::max(sx0, sx_last) >= c->prm.nxe) return fail(FDW_EINVAL, "source rows %d..%d leave the grid", sx0, sx_last);
    const size_t nx = c->nx, ng = nx * (size_t)nt, fe = field_elems(c);
    int rc;
    if (nshots == 1 || nt == 0 || nt > c->prm.nt || !batch_ok(c)) {
        for (int b = 0; b < nshots; b++)
            if ((rc = fdw_model_shot(c, vel2, sx0 + b * dsx, sz, gz, srce, nt, data + b * ng))) return rc;
        return FDW_OK;
    }
    HIP_TRY(hipSetDevice(c->device));
    if ((rc = ensure_work_buffers(c, 2, false))) return rc;
    if ((rc = ensure_batch_buffers(c, nshots, false)) == FDW_ENOMEM) {         // no room for the batch: the shots one after the other
        for (int b = 0; b < nshots; b++)
            if ((rc = fdw_model_shot(c, vel2, sx0 + b * dsx, sz, gz, srce, nt, data + b * ng))) return rc;
        return FDW_OK;
    }
    if (rc || (rc = upload_source(c, srce, nt))) return rc;
    if ((rc = ensure_cap(&c->d_raw, &c->raw_cap, ng * nshots))) return rc;
    hipStream_t s = c->stream;
    if ((rc = upload_rows(c, c->d_v2, vel2, s))) return rc;
    HIP_TRY(hipMemsetAsync(c->bfld[0], 0, fe * nshots * sizeof(float), s));   // M:144-145
    HIP_TRY(hipMemsetAsync(c->bfld[1], 0, fe * nshots * sizeof(float), s));
    c->nbatch = nshots; c->batch_dsx = dsx; c->batch_nt = nt;
    rc = fdw_dev_model_steps(c, c->bfld[0], c->bfld[1], c->d_v2, c->d_srce, sx0, sz, gz, c->b_dobs, 0, nt, s);
    c->nbatch = 1; c->batch_dsx = 0;
    if (rc) return rc;
    // device [shot][it][ix] -> data[shot][ix][it] (M:156)
    hipError_t e = launch_gather_transpose(c->b_dobs, c->d_raw, nt, (int)nx, nshots, s);
    if (e != hipSuccess) return fail(FDW_EHIP, "gather transposition launch failed: %s", hipGetErrorString(e));
    HIP_TRY(hipMemcpyAsync(data, c->d_raw, ng * nshots * sizeof(float), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return FDW_OK;
}
