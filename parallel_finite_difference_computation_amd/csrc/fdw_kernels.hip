// fdw_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the 2-D acoustic FD path.
//
// What the reference does in 4-5 launches per time step (kernel_tapper R:94-117, kernel_lap
// R:53-78, kernel_time R:80-92, kernel_src R:119-122 / kernel_sism R:124-131, kernel_img R:133-144
// of cuda_reference_RTM/src/fd-code.cu, with threadIdx.x on the strided axis) is ONE streaming pass
// here:
//
//   * lanes run along z (the contiguous axis); a lane owns one float4 (4 consecutive z), a wave owns
//     a 256-wide z strip -> every global access is a 1 KiB coalesced global_load/store_dwordx4;
//   * a wave marches along x over `xchunk` rows keeping the 2H+1 rows of p it needs in REGISTERS
//     (x taps never touch memory twice inside a chunk);
//   * z taps come from the two neighbouring lanes by DPP wave shifts (v_mov_b32_dpp wave_shr:1 /
//     wave_shl:1) of the centre row; only lanes 0 and 63 fetch a 16-B strip halo, in one
//     exec-masked load whose result is the DPP `old` operand (so the merge is free);
//   * taper, Laplacian, leap-frog update, point-source / receiver injection and the imaging
//     condition are fused: p, pp, v2 are read once and pp written once = 16 B/point/step
//     (+12 B/point for the imaging epilogue).  No LDS, no MFMA: the stencil is HBM-bound
//     (about 2.5 flop/B), the register window is the cheapest tile there is.
//
// Arithmetic is the reference's, operation for operation, so results are IEEE-identical to the
// no-FMA CUDA build (nvcc --fmad=false --ftz=false, Makefile:4): two fp32 accumulators summed
// in io order, `acmz + acmx`, then the update in double with a single rounding (R:89).
// This file MUST be compiled with -ffp-contract=off; the pragma below is a second guard.
//
// Lazy taper: the reference damps d_p and d_pp IN PLACE before each step.  Doing that inside a
// fused kernel would race with the neighbouring waves that read those rows as stencil taps, so the
// damping is applied on load instead and never written back: a value that sits in memory as "p"
// gets T() once, and when the same memory is read one step later as "pp" it gets T(T()) -- the same
// sequence of fp32 multiplies the reference performs.  The host owes one T() when it finally
// downloads d_p (fdw_taper_finalize_kernel).
#include <hip/hip_runtime.h>
#include "fdw_kernels.h"

#pragma clang fp contract(off)

namespace fdw {

// ------------------------------------------------------------------------------------------------
// cross-lane helpers (wave64)
// ------------------------------------------------------------------------------------------------
// lane n <- lane n-1; lane 0 keeps `old` (its strip halo).  DPP_WF_SR1 = 0x138.
__device__ __forceinline__ float wave_shr1(float src, float old)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old),
                                                                   __builtin_bit_cast(int, src),
                                                                   0x138, 0xf, 0xf, false));
}
// lane n <- lane n+1; lane 63 keeps `old`.  DPP_WF_SL1 = 0x130.
__device__ __forceinline__ float wave_shl1(float src, float old)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old),
                                                                   __builtin_bit_cast(int, src),
                                                                   0x130, 0xf, 0xf, false));
}

struct f4 {
    float v[4];
};

__device__ __forceinline__ f4 f4_zero()
{
    f4 r;
    r.v[0] = r.v[1] = r.v[2] = r.v[3] = 0.0f;
    return r;
}
__device__ __forceinline__ f4 f4_load(const float* p)
{
    const float4 t = *reinterpret_cast<const float4*>(p);
    f4 r;
    r.v[0] = t.x; r.v[1] = t.y; r.v[2] = t.z; r.v[3] = t.w;
    return r;
}
__device__ __forceinline__ void f4_store(float* p, const f4& a)
{
    *reinterpret_cast<float4*>(p) = make_float4(a.v[0], a.v[1], a.v[2], a.v[3]);
}

// one application of the reference's taper to a value: (v * taperz[j]) * taperx[i]  (R:103-114).
// tz is already 1.0f outside the damped strip; zone says whether z < ztap; rowtz / txr are per row.
__device__ __forceinline__ float taper1(float v, float tz, bool zone, bool rowtz, float txr)
{
    const float fz = rowtz ? tz : 1.0f;
    const float fx = zone ? txr : 1.0f;
    return (v * fz) * fx;
}

// ------------------------------------------------------------------------------------------------
// fused step kernel
//   H       half order (1..4)
//   TAPER   apply the lazy top-strip damping to p / pp on load
//   INJ     0 none, 1 point source (kernel_src), 2 receiver row (kernel_sism)
//   IMG     img += psrc * pp_new epilogue (kernel_img)
//   LAPONLY store the Laplacian itself into a.pp (stencil_code path, S:110-135); no update
// block = 256 threads = 4 independent waves (no LDS, no barrier).
// ------------------------------------------------------------------------------------------------
template <int H, bool TAPER, int INJ, bool IMG, bool LAPONLY>
__global__ __launch_bounds__(256) void fdw_step_kernel(const StepArgs a)
{
    constexpr int NW = 2 * H + 1;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

    // XCD-aware block remap: blocks b and b+8 share an XCD (round-robin dispatch), so give each XCD
    // a contiguous run of logical blocks = a contiguous band of x rows whose chunk halos it re-reads
    // from its own L2.  Placement only changes speed, never results.
    const int bid = blockIdx.x;
    const int L = (bid & 7) * a.nper + (bid >> 3);
    if (L >= a.nblk) return;
    const int zb = L % a.nzblk;
    const int xb = L / a.nzblk;
    const int wz = a.wz;              // waves of a block laid along z: 1, 2 or 4
    const int strip = zb * wz + (w & (wz - 1));
    const int chunk = xb * (4 / wz) + (w / wz);
    const int zs = strip * 256;       // first z of this wave's strip
    if (zs >= a.pitch) return;
    const int xa = a.r0 + chunk * a.xchunk;
    const int xe = min(xa + a.xchunk, a.r1);
    if (xa >= xe) return;

    const int z0 = zs + lane * 4;
    const bool act = z0 < a.pitch;    // pitch is a multiple of 4: a float4 never straddles a row end
    // strip halo: lane 0 fetches the 4 columns left of the strip, lane 63 the 4 columns right of it
    const int hz = (lane == 0) ? z0 - 4 : z0 + 4;
    const bool hact = (lane == 0) ? (zs >= 4) : (lane == 63 && hz < a.pitch);

    // per-lane damping factors along z (constant over the march)
    const bool wave_tap = TAPER && (zs - 4 < a.ztap);
    float tzc[4], tzh[4];
    bool znc[4], znh[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        tzc[e] = tzh[e] = 1.0f;
        znc[e] = znh[e] = false;
    }
    if (wave_tap) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int zc = z0 + e, zh = hz + e;
            znc[e] = zc < a.ztap;
            znh[e] = hact && zh < a.ztap;
            if (znc[e]) tzc[e] = a.taperz[zc];
            if (znh[e]) tzh[e] = a.taperz[zh];
        }
    }

    // row loaders -------------------------------------------------------------------------------
    auto rowptr = [&](const float* base, int row) -> const float* {
        return base + (size_t)row * (size_t)a.pitch;
    };
    // a row of p for the register window, damped once when TAPER
    auto load_p = [&](int row) -> f4 {
        f4 v = f4_zero();
        if (row >= 0 && row < a.nxl) {
            if (act) v = f4_load(rowptr(a.p, row) + z0);
            if (wave_tap) {
                const float txr = a.txfac[row];
                const bool rowtz = row < a.tz_x1;
#pragma unroll
                for (int e = 0; e < 4; ++e) v.v[e] = taper1(v.v[e], tzc[e], znc[e], rowtz, txr);
            }
        }
        return v;
    };
    // strip halo of the centre row (only lanes 0 / 63 hold data), damped once when TAPER
    auto load_halo = [&](int row) -> f4 {
        f4 v = f4_zero();
        if (hact) v = f4_load(rowptr(a.p, row) + hz);
        if (wave_tap) {
            const float txr = a.txfac[row];
            const bool rowtz = row < a.tz_x1;
#pragma unroll
            for (int e = 0; e < 4; ++e) v.v[e] = taper1(v.v[e], tzh[e], znh[e], rowtz, txr);
        }
        return v;
    };
    auto load_plain = [&](const float* base, int row) -> f4 {
        f4 v = f4_zero();
        if (act) v = f4_load(rowptr(base, row) + z0);
        return v;
    };

    // prologue: fill the window with rows xa-H .. xa+H ------------------------------------------
    f4 win[NW];
#pragma unroll
    for (int k = 0; k < NW; ++k) win[k] = load_p(xa - H + k);
    f4 hal = load_halo(xa);
    f4 cpp = f4_zero(), cv2 = f4_zero(), cps = f4_zero(), cim = f4_zero();
    if (!LAPONLY) {
        cpp = load_plain(a.pp, xa);
        cv2 = load_plain(a.v2, xa);
    }
    if (IMG) {
        cps = load_plain(a.psrc, xa);
        cim = load_plain(a.img, xa);
    }

    for (int rb = xa; rb < xe; rb += NW) {
#pragma unroll
        for (int u = 0; u < NW; ++u) {
            const int r = rb + u;
            if (r < xe) {
                // ---- prefetch everything the NEXT row needs before touching this one ----------
                const bool more = (r + 1 < xe);
                f4 nxt = load_p(r + H + 1);
                f4 nhal = f4_zero(), npp = f4_zero(), nv2 = f4_zero(), nps = f4_zero(), nim = f4_zero();
                if (more) {
                    nhal = load_halo(r + 1);
                    if (!LAPONLY) {
                        npp = load_plain(a.pp, r + 1);
                        nv2 = load_plain(a.v2, r + 1);
                    }
                    if (IMG) {
                        nps = load_plain(a.psrc, r + 1);
                        nim = load_plain(a.img, r + 1);
                    }
                }

                // ---- this row ------------------------------------------------------------------
                const f4 c = win[(u + H) % NW];
                float W[12];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    W[e] = wave_shr1(c.v[e], hal.v[e]);
                    W[4 + e] = c.v[e];
                    W[8 + e] = wave_shl1(c.v[e], hal.v[e]);
                }
                f4 ppt = cpp;
                if (!LAPONLY && wave_tap) {
                    const float txr = a.txfac[r];
                    const bool rowtz = r < a.tz_x1;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float t = taper1(ppt.v[e], tzc[e], znc[e], rowtz, txr);
                        if (a.pp_twice) t = taper1(t, tzc[e], znc[e], rowtz, txr);
                        ppt.v[e] = t;
                    }
                }
                const bool lapx = (r >= a.lap_x0) && (r < a.lap_x1);
                float injv = 0.0f;
                bool injrow = false;
                if (INJ == 1) {
                    injrow = (r == a.inj_x);
                    if (injrow) injv = a.inj[0];
                } else if (INJ == 2) {
                    injrow = (r >= a.inj_x) && (r < a.inj_x + a.inj_n);
                    if (injrow) injv = a.inj[r - a.inj_x];
                }
                f4 res, imr;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int z = z0 + e;
                    float acmz = 0.0f, acmx = 0.0f;
#pragma unroll
                    for (int io = 0; io < NW; ++io) {
                        acmz = acmz + W[4 + e - H + io] * a.cz[io];
                        acmx = acmx + win[(u + io) % NW].v[e] * a.cx[io];
                    }
                    float lap = acmz + acmx;
                    const bool inl = lapx && (z >= a.lap_z0) && (z < a.lap_z1);
                    lap = inl ? lap : 0.0f;
                    float out;
                    if (LAPONLY) {
                        out = lap;
                    } else {
                        const float prod = (cv2.v[e] * a.dt2) * lap;
                        const double d = 2.0 * (double)c.v[e] - (double)ppt.v[e] + (double)prod;
                        out = (z < a.upd_z1) ? (float)d : ppt.v[e];
                        if (INJ != 0) {
                            if (injrow && z == a.inj_z) out = out + injv;
                        }
                    }
                    res.v[e] = out;
                    if (IMG) imr.v[e] = cim.v[e] + cps.v[e] * out;
                }
                if (act) {
                    f4_store(const_cast<float*>(rowptr(a.pp, r)) + z0, res);
                    if (IMG) f4_store(const_cast<float*>(rowptr(a.img, r)) + z0, imr);
                }

                // ---- rotate --------------------------------------------------------------------
                win[u] = nxt;
                hal = nhal;
                cpp = npp;
                cv2 = nv2;
                cps = nps;
                cim = nim;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// generic-order kernel: any even order up to FDW_MAX_ORDER, one thread per point, every tap from
// global memory (L1/L2 absorb the reuse).  Same arithmetic, same lazy-taper rules; used for orders
// the register-window kernel is not instantiated for, and as an independent cross-check of it.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float generic_p(const StepArgs& a, int row, int z, bool taper)
{
    float v = a.p[(size_t)row * a.pitch + z];
    if (taper && z < a.ztap) v = taper1(v, a.taperz[z], true, row < a.tz_x1, a.txfac[row]);
    return v;
}

__global__ __launch_bounds__(256) void fdw_generic_kernel(const StepArgs a, int h, int taper, int injmode,
                                                          int img, int laponly)
{
    const int z = blockIdx.x * 256 + threadIdx.x;
    const int r = a.r0 + blockIdx.y;
    if (z >= a.pitch || r >= a.r1) return;
    const size_t k = (size_t)r * a.pitch + z;
    float lap = 0.0f;
    if (r >= a.lap_x0 && r < a.lap_x1 && z >= a.lap_z0 && z < a.lap_z1) {
        float acmz = 0.0f, acmx = 0.0f;
        for (int io = 0; io <= 2 * h; ++io) {
            acmz = acmz + generic_p(a, r, z + io - h, taper) * a.gcz[io];
            acmx = acmx + generic_p(a, r + io - h, z, taper) * a.gcx[io];
        }
        lap = acmz + acmx;
    }
    float out;
    if (laponly) {
        out = lap;
    } else {
        const float pc = generic_p(a, r, z, taper);
        float ppv = a.pp[k];
        if (taper && z < a.ztap) {
            const bool rowtz = r < a.tz_x1;
            const float txr = a.txfac[r], tz = a.taperz[z];
            ppv = taper1(ppv, tz, true, rowtz, txr);
            if (a.pp_twice) ppv = taper1(ppv, tz, true, rowtz, txr);
        }
        const float prod = (a.v2[k] * a.dt2) * lap;
        const double d = 2.0 * (double)pc - (double)ppv + (double)prod;
        out = (z < a.upd_z1) ? (float)d : ppv;
        if (injmode == 1) {
            if (r == a.inj_x && z == a.inj_z) out = out + a.inj[0];
        } else if (injmode == 2) {
            if (r >= a.inj_x && r < a.inj_x + a.inj_n && z == a.inj_z) out = out + a.inj[r - a.inj_x];
        }
    }
    a.pp[k] = out;
    if (img) a.img[k] = a.img[k] + a.psrc[k] * out;
}

// ------------------------------------------------------------------------------------------------
// the one T() the lazy scheme owes d_p before it leaves the device (fd_forward's D2H of d_p, R:285)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void fdw_taper_finalize_kernel(float* f, const float* taperz, const float* txfac,
                                                                 int pitch, int nxl, int ztap, int tz_x1)
{
    const int z = blockIdx.x * 256 + threadIdx.x;
    const int r = blockIdx.y;
    if (z >= ztap || z >= pitch || r >= nxl) return;
    const size_t k = (size_t)r * pitch + z;
    f[k] = taper1(f[k], taperz[z], true, r < tz_x1, txfac[r]);
}

// ------------------------------------------------------------------------------------------------
// wave-shift self test: out[lane] = {wave_shr1(src, old), wave_shl1(src, old)} -- lets the host
// verify on the actual device that DPP wave shifts behave as the step kernel assumes.
// ------------------------------------------------------------------------------------------------
__global__ void fdw_dpp_selftest_kernel(const float* src, const float* old, float* out)
{
    const int t = threadIdx.x;
    out[t] = wave_shr1(src[t], old[t]);
    out[64 + t] = wave_shl1(src[t], old[t]);
}

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
template <int H>
static hipError_t launch_fast_h(const StepArgs& a, int mode, hipStream_t s)
{
    const dim3 grid(8 * a.nper), block(256);
    switch (mode) {
    case FDW_MODE_FWD:   hipLaunchKernelGGL((fdw_step_kernel<H, true, 1, false, false>), grid, block, 0, s, a); break;
    case FDW_MODE_PLAIN: hipLaunchKernelGGL((fdw_step_kernel<H, false, 0, false, false>), grid, block, 0, s, a); break;
    case FDW_MODE_RECV:  hipLaunchKernelGGL((fdw_step_kernel<H, true, 2, true, false>), grid, block, 0, s, a); break;
    case FDW_MODE_LAP:   hipLaunchKernelGGL((fdw_step_kernel<H, false, 0, false, true>), grid, block, 0, s, a); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_step_fast(const StepArgs& a, int h, int mode, hipStream_t s)
{
    if (a.nper <= 0) return hipSuccess;
    switch (h) {
    case 1: return launch_fast_h<1>(a, mode, s);
    case 2: return launch_fast_h<2>(a, mode, s);
    case 3: return launch_fast_h<3>(a, mode, s);
    case 4: return launch_fast_h<4>(a, mode, s);
    default: return hipErrorInvalidValue;
    }
}

hipError_t launch_step_generic(const StepArgs& a, int h, int mode, hipStream_t s)
{
    if (a.r1 <= a.r0) return hipSuccess;
    const dim3 grid((a.pitch + 255) / 256, a.r1 - a.r0), block(256);
    const int taper = (mode == FDW_MODE_FWD || mode == FDW_MODE_RECV);
    const int inj = (mode == FDW_MODE_FWD) ? 1 : (mode == FDW_MODE_RECV ? 2 : 0);
    hipLaunchKernelGGL(fdw_generic_kernel, grid, block, 0, s, a, h, taper, inj, mode == FDW_MODE_RECV ? 1 : 0,
                       mode == FDW_MODE_LAP ? 1 : 0);
    return hipGetLastError();
}

hipError_t launch_taper_finalize(float* f, const float* taperz, const float* txfac, int pitch, int nxl, int ztap,
                                 int tz_x1, hipStream_t s)
{
    if (ztap <= 0 || nxl <= 0) return hipSuccess;
    const dim3 grid((ztap + 255) / 256, nxl), block(256);
    hipLaunchKernelGGL(fdw_taper_finalize_kernel, grid, block, 0, s, f, taperz, txfac, pitch, nxl, ztap, tz_x1);
    return hipGetLastError();
}

hipError_t launch_dpp_selftest(const float* src, const float* old, float* out, hipStream_t s)
{
    hipLaunchKernelGGL(fdw_dpp_selftest_kernel, dim3(1), dim3(64), 0, s, src, old, out);
    return hipGetLastError();
}

}  // namespace fdw
