// fdw_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the 2-D acoustic FD path.
//
// What the reference does in 4-5 launches per time step (kernel_tapper R:94-117, kernel_lap
// R:53-78, kernel_time R:80-92, kernel_src R:119-122 / kernel_sism R:124-131, kernel_img R:133-144
// of cuda_reference_RTM/src/fd-code.cu, with threadIdx.x on the strided axis) is ONE streaming pass
// here:
//
//   * lanes run along z (the contiguous axis); a lane owns one float4 (4 consecutive z), a wave owns
//     a 256-wide z strip -> every global access is a 1 KiB coalesced global_load/store_dwordx4;
//   * a wave marches along x over `xchunk` rows keeping the rows of p it needs in a REGISTER ring
//     (2H+1 stencil rows + look-ahead rows that are still in flight), so x taps never touch memory
//     twice inside a chunk and HBM latency is covered by explicit software prefetch, not occupancy;
//   * z taps come from the two neighbouring lanes (ds_bpermute via __shfl_up/down by one lane; DPP
//     wave shifts were measured 8 % slower on gfx950); the 4+4 halo columns of a strip come from ONE
//     extra load issued by all lanes (lane 0 the left piece, every other lane the right piece);
//   * taper, Laplacian, leap-frog update, point-source / receiver injection and the imaging
//     condition are fused: p, pp, v2 are read once and pp written once = 16 B/point/step
//     (+12 B/point for the imaging epilogue).  No MFMA: the stencil is HBM-bound (about 2.5 flop/B),
//     the register ring is the cheapest tile there is;
//   * every global load of the march is unconditional (clamped addresses instead of predication) so
//     that hipcc's s_waitcnt bookkeeping stays exact (vmcnt(4..7), never 0); edge handling is
//     wave-uniform branches around VALU / scalar-cache loads only;
//   * fdw_step2_kernel does TWO time steps per pass (temporal blocking, 10 B/point/step): overlapped
//     60-cell tiles, a second register ring for u^{n+1}, v2 rows parked in LDS, range-predicated stores.
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

#include <type_traits>

#include "fdw_kernels.h"

#pragma clang fp contract(off)

namespace fdw {

// ------------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------------
// Timing experiments only (scripts/build_ablations.sh builds throw-away libraries with -DFDW_ABL_BITS=n; any bit
// breaks the results): 1 fp32 update, 2 no strip-halo load, 4 no store, 8 no lane exchange, 16 trivial Laplacian,
// 32 every row aliases row 0 (loads become cache hits: pure issue/VALU time), 64 pipeline kernel without its barriers;
// pipeline kernel only: 128 no LDS hand-over, 256 trivial Laplacian, 512 fp32 update, 1024 no global loads/stores in the march, 2048 no lane exchange.
#ifndef FDW_ABL_BITS
#define FDW_ABL_BITS 0
#endif

struct f4 {
    float v[4];
};

__device__ __forceinline__ f4 f4_zero()
{
    f4 r;
    r.v[0] = r.v[1] = r.v[2] = r.v[3] = 0.0f;
    return r;
}
// Cache policy of the streams (FDW_NT bitmask; measured in scripts/ubench/rwmix.hip and on the kernel):
//   1 pointwise inputs (pp, v2, psrc, img) are read once per step -> nontemporal loads
//   2 the result is not read again in this launch               -> nontemporal store
//   4 p rows (re-read by the neighbouring chunk as halo)        -> default policy unless set
#ifndef FDW_NT
#define FDW_NT 3
#endif
typedef float v4f __attribute__((ext_vector_type(4)));

// scalar (wave-uniform) row base + 32-bit per-lane byte offset: global_load saddr + voffset form
template <bool NT>
__device__ __forceinline__ f4 f4_load_t(const float* row, unsigned voff_bytes)
{
    const v4f* ptr = reinterpret_cast<const v4f*>(reinterpret_cast<const char*>(row) + voff_bytes);
    const v4f t = NT ? __builtin_nontemporal_load(ptr) : *ptr;
    f4 r;
    r.v[0] = t.x; r.v[1] = t.y; r.v[2] = t.z; r.v[3] = t.w;
    return r;
}
__device__ __forceinline__ f4 f4_load(const float* row, unsigned voff_bytes) { return f4_load_t<(FDW_NT & 4) != 0>(row, voff_bytes); }
__device__ __forceinline__ f4 f4_load_stream(const float* row, unsigned voff_bytes) { return f4_load_t<(FDW_NT & 1) != 0>(row, voff_bytes); }
__device__ __forceinline__ void f4_store(float* row, unsigned voff_bytes, const f4& a)
{
    v4f* ptr = reinterpret_cast<v4f*>(reinterpret_cast<char*>(row) + voff_bytes);
    const v4f t = {a.v[0], a.v[1], a.v[2], a.v[3]};
    if (FDW_NT & 2) __builtin_nontemporal_store(t, ptr); else *ptr = t;
}

template <int N, class F>
__device__ __forceinline__ void static_for(F&& f)
{
    if constexpr (N > 0) {
        static_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

// one application of the reference's taper to a value: (v * taperz[j]) * taperx[i]  (R:103-114).
// tz is already 1.0f outside the damped strip; zone says whether z < ztap; rowtz / txr are per row.
__device__ __forceinline__ float taper1(float v, float tz, bool zone, bool rowtz, float txr)
{
    const float fz = rowtz ? tz : 1.0f;
    const float fx = zone ? txr : 1.0f;
    return (v * fz) * fx;
}

// The arithmetic of one grid point, exactly as kernel_lap + kernel_time spell it (R:66-72, R:89).
//   W    12 consecutive z values of the centre row: W[4+e] is the point itself
//   col  the 2H+1 x taps of this point, oldest row first
template <int H>
__device__ __forceinline__ float laplacian_pt(const float* W, int e, const float (&col)[2 * H + 1], const float* cx,
                                              const float* cz)
{
    float acmz = 0.0f, acmx = 0.0f;
#if FDW_ABL_BITS & 16
    return W[e] + W[8 + e] + col[0] + col[2 * H];   // keep every input alive, almost no arithmetic
#endif
#pragma unroll
    for (int io = 0; io <= 2 * H; ++io) {
        acmz = acmz + W[4 + e - H + io] * cz[io];
        acmx = acmx + col[io] * cx[io];
    }
    return acmz + acmx;
}
// The CPU-serial sibling's Laplacian (dpct_gpu_rtm_domain_division/src/timestep/fd.c:28-36): ONE accumulator, per tap the z term
// then the x term, each weight scaled by its inverse spacing squared inside the term.  c = unscaled weights.
template <int H>
__device__ __forceinline__ float laplacian_dd_pt(const float* W, int e, const float (&col)[2 * H + 1], const float* c, float dx2inv, float dz2inv)
{
    float acm = 0.0f;
#pragma unroll
    for (int io = 0; io <= 2 * H; ++io) {
        acm = acm + (W[4 + e - H + io] * c[io]) * dz2inv;
        acm = acm + (col[io] * c[io]) * dx2inv;
    }
    return acm;
}
// the update once prod = (v2*dt2)*lap is formed (fp32, as the reference's float expression does; R:89)
__device__ __forceinline__ float leapfrog_prod(float p, float pp, float prod)
{
    // 2.*p - pp: the product is exact in double, so the fused form rounds once exactly like the reference's two operations
    const double d = __builtin_fma(2.0, (double)p, -(double)pp) + (double)prod;
    float r = (float)d;
    // Keep the update unconditional: without this hipcc turns the caller's "mask ? update : old" select into a branch
    // around the fp64 chain (one serial basic block per cell, nothing to interleave with).
    asm volatile("" : "+v"(r));
    return r;
}
__device__ __forceinline__ float leapfrog_pt(float p, float pp, float v2, float dt2, float lap)
{
    const float prod = (v2 * dt2) * lap;
#if FDW_ABL_BITS & 1
    return (2.0f * p - pp) + prod;
#endif
    const double d = 2.0 * (double)p - (double)pp + (double)prod;
    return (float)d;
}

// ---- packed fp32 (v_pk_mul_f32 / v_pk_add_f32: two IEEE fp32 operations per lane and instruction) --------------
// The two-step kernel is VALU-issue bound (SQ counters: ~48 VALU instructions per point and step, 85 % VALU busy at
// 8192^2), so the Laplacian of a lane's four cells is formed as two PAIRS.  Each product and each sum is still an
// individually rounded fp32 operation in the reference's order -- bit-identical to laplacian_pt.
typedef float v2f __attribute__((ext_vector_type(2)));

// Window of 12 consecutive z values as aligned pairs: E[k] = (W[2k], W[2k+1]), O[k] = (W[2k+1], W[2k+2]).
struct ZPairs {
    v2f E[6], O[5];
};
__device__ __forceinline__ ZPairs zpairs(const f4& left, const f4& c, const f4& right)
{
    // (the odd pairs cost two v_mov_b32 each; forming them with v_pk_mov_b32 through asm made hipcc keep half of the ring in scratch)
    ZPairs z;
    z.E[0] = v2f{left.v[0], left.v[1]};   z.E[1] = v2f{left.v[2], left.v[3]};
    z.E[2] = v2f{c.v[0], c.v[1]};         z.E[3] = v2f{c.v[2], c.v[3]};
    z.E[4] = v2f{right.v[0], right.v[1]}; z.E[5] = v2f{right.v[2], right.v[3]};
    z.O[0] = v2f{left.v[1], left.v[2]};   z.O[1] = v2f{left.v[3], c.v[0]};
    z.O[2] = v2f{c.v[1], c.v[2]};         z.O[3] = v2f{c.v[3], right.v[0]};
    z.O[4] = v2f{right.v[1], right.v[2]};
    return z;
}
// pair * weight, the weight taken from the low (SEL 0) or high (SEL 1) half of an SGPR pair for BOTH lanes (VOP3P op_sel):
// written as asm because hipcc materialises a splat (c, c) SGPR pair per weight otherwise and then spills SGPRs.
template <int SEL>
__device__ __forceinline__ v2f pk_mul_sel(v2f w, v2f cpair)
{
    v2f r;
    if constexpr (SEL == 0)
        asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(r) : "v"(w), "s"(cpair));
    else
        asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(r) : "v"(w), "s"(cpair));
    return r;
}
// The weights are symmetric (fdw_host.c), so H+1 distinct values per direction travel as (H+2)/2 SGPR pairs.
template <int H>
struct CoefPairs {
    v2f z[(H + 2) / 2], x[(H + 2) / 2];
};
template <int H>
__device__ __forceinline__ CoefPairs<H> coef_pairs(const float* cx, const float* cz)
{
    CoefPairs<H> c;
    static_for<(H + 2) / 2>([&](auto K) {
        constexpr int k = decltype(K)::value;
        constexpr int k1 = (2 * k + 1 <= H) ? 2 * k + 1 : H;
        c.z[k] = v2f{cz[2 * k], cz[k1]};
        c.x[k] = v2f{cx[2 * k], cx[k1]};
    });
    return c;
}
// Laplacian of cells (2P, 2P+1) of the lane: same accumulation order as laplacian_pt (R:66-72).
template <int H, int P, class Col>
__device__ __forceinline__ v2f laplacian_pair(const ZPairs& z, Col&& col, const CoefPairs<H>& c)
{
    v2f acmz = {0.0f, 0.0f}, acmx = {0.0f, 0.0f};
    static_for<2 * H + 1>([&](auto IO) {
        constexpr int io = decltype(IO)::value;
        constexpr int k = 4 + 2 * P - H + io;                       // W index of the pair's first cell for this tap
        constexpr int ic = io <= H ? io : 2 * H - io;
        const v2f w = (k & 1) ? z.O[k >> 1] : z.E[k >> 1];
        acmz = acmz + pk_mul_sel<ic & 1>(w, c.z[ic >> 1]);
        acmx = acmx + pk_mul_sel<ic & 1>(col(IO), c.x[ic >> 1]);
    });
    return acmz + acmx;
}
__device__ __forceinline__ v2f f4_pair(const f4& a, int P) { return v2f{a.v[2 * P], a.v[2 * P + 1]}; }
// Both pairs of a lane at once: the four accumulator chains (z and x of each pair) advance tap by tap side by side, which gives a
// lone wave four independent dependency chains to issue from instead of two (same operations, same order within each chain).
template <int H, class Row>
__device__ __forceinline__ void laplacian_quad(const ZPairs& z, Row&& row, const CoefPairs<H>& c, v2f& lap01, v2f& lap23)
{
    v2f az0 = {0.0f, 0.0f}, ax0 = {0.0f, 0.0f}, az1 = {0.0f, 0.0f}, ax1 = {0.0f, 0.0f};
    static_for<2 * H + 1>([&](auto IO) {
        constexpr int io = decltype(IO)::value;
        constexpr int k0 = 4 - H + io, k1 = 6 - H + io;
        constexpr int ic = io <= H ? io : 2 * H - io;
        const f4 r = row(IO);
        az0 = az0 + pk_mul_sel<ic & 1>((k0 & 1) ? z.O[k0 >> 1] : z.E[k0 >> 1], c.z[ic >> 1]);
        ax0 = ax0 + pk_mul_sel<ic & 1>(v2f{r.v[0], r.v[1]}, c.x[ic >> 1]);
        az1 = az1 + pk_mul_sel<ic & 1>((k1 & 1) ? z.O[k1 >> 1] : z.E[k1 >> 1], c.z[ic >> 1]);
        ax1 = ax1 + pk_mul_sel<ic & 1>(v2f{r.v[2], r.v[3]}, c.x[ic >> 1]);
    });
    lap01 = az0 + ax0;
    lap23 = az1 + ax1;
}

// ring geometry: PF rows of pointwise look-ahead (pp, v2, halo, ...); the p ring holds R rows,
// R a multiple of PF (so queue slots are compile-time constants) and >= 2H+PF.
template <int H, int PF>
struct RingGeom {
    static constexpr int NW = 2 * H + 1;
    static constexpr int R = ((2 * H + PF + PF - 1) / PF) * PF;
    static constexpr int LOOK = R - 2 * H;   // rows of p look-ahead
};

// wave-uniform table read through the scalar cache (s_load_dword, counted by lgkmcnt, so it never
// disturbs the vmcnt bookkeeping of the streaming loads).  Only for tables no kernel writes.
__device__ __forceinline__ float sload(const float* p, int i)
{
    typedef const float __attribute__((address_space(4))) * cptr;
    return ((cptr)p)[i];
}

// ------------------------------------------------------------------------------------------------
// fused step kernel
//   H       half order (1..4)
//   TAPER   apply the lazy top-strip damping to p / pp
//   INJ     0 none, 1 point source (kernel_src), 2 receiver row (kernel_sism), 3 7x7 Gaussian point source (ptsrc.c of the CPU-serial sibling)
//   IMG     img += psrc * pp_new epilogue (kernel_img)
//   LAPONLY store the Laplacian itself into a.pp (stencil_code path, S:110-135); no update
//   DD      arithmetic of the CPU-serial sibling's fd_step (single accumulator, per-term scaling) + one trace sample per row
//   PF      software prefetch distance in rows
// block = 256 threads = 4 independent waves (no LDS, no barrier).
//
// ONE code path for every tile.  Every global load of the march is unconditional (addresses are
// clamped into the slab instead of being predicated), so the compiler's s_waitcnt bookkeeping stays
// exact and the look-ahead loads really stay in flight.  Whatever a clamped load brings in only ever
// reaches outputs that the column / row masks zero: rows outside the slab are taps of rows whose
// Laplacian is masked (lap_x0 >= H, lap_x1 <= nxl-H), columns outside the grid are taps of columns
// >= nze-H.  Edge handling (masks, damping, injection) is wave-uniform branches around VALU / scalar
// loads only.
// ------------------------------------------------------------------------------------------------
template <int H, bool TAPER, int INJ, bool IMG, bool LAPONLY, int PF, bool DD = false>
__device__ __forceinline__ void march(const StepArgs& a, const int lane, const int zs, const int xa, const int xe)
{
    using G = RingGeom<H, PF>;
    constexpr int R = G::R, LOOK = G::LOOK;
#if FDW_ABL_BITS & 32
    const size_t pitch = 0;   // every row aliases row 0: loads become L1 hits -> pure issue/VALU time
#else
    const size_t pitch = (size_t)a.pitch;
#endif
    const int z0 = zs + lane * 4;
    const bool partial = (zs + 256 > a.pitch);              // wave-uniform: last strip of a ragged row
    const bool act = z0 < a.pitch;                          // pitch % 4 == 0: a float4 never straddles a row end
    const unsigned voff = (unsigned)min(z0, a.pitch - 4) * 4u;
    // strip halo in ONE load by all lanes: lane 0 fetches the 4 columns left of the strip, every other
    // lane the 4 columns right of it (a single 16-B piece; only lane 63 consumes it).
    const unsigned hoff = (unsigned)((lane == 0) ? max(zs - 4, 0) : min(zs + 256, a.pitch - 4)) * 4u;
    const bool lane_first = (lane == 0), lane_last = (lane == 63);
    const int rowmax = min(a.nxl, xe + H) - 1;              // last row of p this wave can need

    // wave-uniform classification of the tile
    const bool zedge = (zs < a.lap_z0) || (zs + 256 > a.lap_z1) || (!LAPONLY && zs + 256 > a.upd_z1);
    const bool xedge = (xa < a.lap_x0) || (xe > a.lap_x1);
    const bool wave_tap = TAPER && (zs - 4 < a.ztap);
    const bool xtap = wave_tap && ((xa - H < a.xt_lo) || (xe + H > a.xt_hi));   // rows with an x factor / no z factor
    bool inj_here = false;
    if (INJ == 1) inj_here = (a.inj_x >= xa) && (a.inj_x < xe) && (a.inj_z >= zs) && (a.inj_z < zs + 256);
    if (INJ == 2) inj_here = (a.inj_z >= zs) && (a.inj_z < zs + 256) && (a.inj_x < xe) && (a.inj_x + a.inj_n > xa);
    if (INJ == 3) inj_here = (a.inj_z + 3 >= zs) && (a.inj_z - 3 < zs + 256) && (a.inj_x + 3 >= xa) && (a.inj_x - 3 < xe);   // 7x7 blob
    const float inj_src = ((INJ == 1 || INJ == 3) && inj_here) ? sload(a.inj, 0) : 0.0f;
    const bool rec_here = DD && (a.rec != nullptr) && (a.rec_z >= zs) && (a.rec_z < zs + 256);
    const CoefPairs<H> cpk = coef_pairs<H>(a.cx, a.cz);

    // per-lane column masks and damping factors
    bool mlap[4], mupd[4], znc[4], znh[4], ihit[4];
    float tzc[4], tzh[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int z = z0 + e;
        mlap[e] = (z >= a.lap_z0) && (z < a.lap_z1);
        mupd[e] = z < a.upd_z1;
        ihit[e] = (z == a.inj_z);
        znc[e] = znh[e] = false;
        tzc[e] = tzh[e] = 1.0f;
    }
    if (wave_tap) {
        const int hz = (lane == 0) ? zs - 4 : zs + 256;     // true (unclamped) column of this lane's halo piece
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int zc = z0 + e, zh = hz + e;
            znc[e] = zc < a.ztap;
            znh[e] = (zh >= 0) && (zh < a.ztap);
            if (znc[e]) tzc[e] = a.taperz[zc];
            if (znh[e]) tzh[e] = a.taperz[zh];
        }
    }
    // one application of the reference's damping to a row held in registers (R:103-114)
    auto taper_row = [&](f4& v, const float* tz, const bool* zone, int row) {
        if (!xtap) {   // common case: no x factor on these rows, every row gets the z factor; *1.0f is exact
#pragma unroll
            for (int e = 0; e < 4; ++e) v.v[e] = v.v[e] * tz[e];
        } else {
            const int rc = min(max(row, 0), a.nxl - 1);
            const float txr = sload(a.txfac, rc);
            const bool rowtz = row < a.tz_x1;
#pragma unroll
            for (int e = 0; e < 4; ++e) v.v[e] = taper1(v.v[e], tz[e], zone[e], rowtz, txr);
        }
    };

    // ---- loaders: unconditional, clamped --------------------------------------------------------
    auto load_p = [&](int row) -> f4 { return f4_load(a.p + (size_t)min(max(row, 0), rowmax) * pitch, voff); };
    auto load_halo = [&](int row) -> f4 {
#if FDW_ABL_BITS & 2
        return f4_zero();
#endif
        return f4_load(a.p + (size_t)row * pitch, hoff);
    };
    auto load_plain = [&](const float* base, int row) -> f4 { return f4_load_stream(base + (size_t)row * pitch, voff); };

    // ---- prologue: ring rows xa-H .. xa-H+R-1; pointwise rows xa .. xa+PF-1 --------------------
    // Issue order matters: the loop-header s_waitcnt is the stricter of (prologue state, end-of-turn
    // state).  Issuing the look-ahead loads in the order the steady state would have issued them
    // ("virtual steps" -LOOK..-1) makes the two states agree, so no turn starts with a pipeline drain.
    f4 ring[R];
    f4 qhal[PF], qpp[PF], qv2[PF], qps[PF], qim[PF];
    constexpr int NV = LOOK > PF ? LOOK : PF;
    static_for<2 * H>([&](auto K) {
        constexpr int k = decltype(K)::value;
        ring[k] = load_p(xa - H + k);
        __builtin_amdgcn_sched_barrier(0);
    });
    static_for<NV>([&](auto JJ) {
        constexpr int j = decltype(JJ)::value - NV;   // virtual step -NV .. -1
        if constexpr (j >= -LOOK) ring[j + 2 * H + LOOK] = load_p(xa + j + H + LOOK);
        if constexpr (j >= -PF) {
            constexpr int m = j + PF;
            const int row = min(xa + m, xe - 1);
            qhal[m] = load_halo(row);
            if constexpr (!LAPONLY) {
                qpp[m] = load_plain(a.pp, row);
                qv2[m] = load_plain(a.v2, row);
            }
            if constexpr (IMG) {
                qps[m] = load_plain(a.psrc, row);
                qim[m] = load_plain(a.img, row);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    });
    if (wave_tap) {   // the first 2H window rows never "enter" the window during the march: damp them here
        static_for<2 * H>([&](auto K) {
            constexpr int k = decltype(K)::value;
            taper_row(ring[k], tzc, znc, xa - H + k);
        });
    }

    // One row of the march.  GUARD=false: the row is known to exist; GUARD=true: wave-uniform test.
    auto row_step = [&](const int rb, auto UU, auto GG) {
        constexpr int U = decltype(UU)::value;
        constexpr bool GUARD = decltype(GG)::value;
        constexpr int Q = U % PF;          // pointwise queue slot of this row (R % PF == 0)
        const int r = rb + U;
        if (!GUARD || r < xe) {
            // ---- damping of what enters the computation this step -----------------------------
            f4 hal = qhal[Q];
            f4 ppt = qpp[Q];
            if (wave_tap) {
                taper_row(ring[(U + 2 * H) % R], tzc, znc, r + H);   // row r+H enters the window
                taper_row(hal, tzh, znh, r);
                if constexpr (!LAPONLY) {
                    taper_row(ppt, tzc, znc, r);
                    if (a.pp_twice) taper_row(ppt, tzc, znc, r);
                }
            }
            // ---- z neighbours from the adjacent lanes (ds_bpermute), strip halo at the ends ----
            const f4 c = ring[(U + H) % R];
            f4 lft, rgt;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
#if FDW_ABL_BITS & 8
                lft.v[e] = c.v[e] + hal.v[e];
                rgt.v[e] = c.v[e] - hal.v[e];
#else
                const float up = __shfl_up(c.v[e], 1, 64), dn = __shfl_down(c.v[e], 1, 64);
                lft.v[e] = lane_first ? hal.v[e] : up;
                rgt.v[e] = lane_last ? hal.v[e] : dn;
#endif
            }
            const bool rowok = (r >= a.lap_x0) && (r < a.lap_x1);
            f4 res, imr;
            if constexpr (DD) {
                if (rec_here && r >= a.rec_x0 && r < a.rec_x0 + a.rec_n) {      // the trace sample of this step: the current field at depth rec_z
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (z0 + e == a.rec_z) a.rec[r - a.rec_x0] = c.v[e];      // interior point: its damping factors are 1.0f
                }
                float W[12];
#pragma unroll
                for (int e = 0; e < 4; ++e) { W[e] = lft.v[e]; W[4 + e] = c.v[e]; W[8 + e] = rgt.v[e]; }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float col[2 * H + 1];
#pragma unroll
                    for (int io = 0; io <= 2 * H; ++io) col[io] = ring[(U + io) % R].v[e];
                    float lap = laplacian_dd_pt<H>(W, e, col, a.cz, a.dx2inv, a.dz2inv);
                    if (zedge || xedge) lap = (rowok && mlap[e]) ? lap : 0.0f;
                    const float upd = leapfrog_prod(c.v[e], ppt.v[e], (qv2[Q].v[e] * a.dt2) * lap);
                    res.v[e] = zedge ? (mupd[e] ? upd : ppt.v[e]) : upd;
                }
            } else {
                // packed pairs: same products and sums in the same order as laplacian_pt (see laplacian_pair)
                const ZPairs zp = zpairs(lft, c, rgt);
                static_for<2>([&](auto PP) {
                    constexpr int P = decltype(PP)::value;
                    v2f lap2 = laplacian_pair<H, P>(zp, [&](auto IO) { return f4_pair(ring[(U + decltype(IO)::value) % R], P); }, cpk);
                    if (zedge || xedge) lap2 = v2f{(rowok && mlap[2 * P]) ? lap2.x : 0.0f, (rowok && mlap[2 * P + 1]) ? lap2.y : 0.0f};
                    if constexpr (LAPONLY) {
                        res.v[2 * P] = lap2.x;
                        res.v[2 * P + 1] = lap2.y;
                    } else {
                        const v2f prod2 = (f4_pair(qv2[Q], P) * a.dt2) * lap2;
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            const int e = 2 * P + q;
                            const float upd = leapfrog_prod(c.v[e], ppt.v[e], q ? prod2.y : prod2.x);
                            res.v[e] = zedge ? (mupd[e] ? upd : ppt.v[e]) : upd;
                        }
                    }
                });
            }
            if constexpr (INJ != 0) {
                if (inj_here) {   // wave-uniform, rare
                    const bool injrow = (INJ == 1) ? (r == a.inj_x) : ((r >= a.inj_x) && (r < a.inj_x + a.inj_n));
                    if (injrow) {
                        const float injv = (INJ == 1) ? inj_src : sload(a.inj, r - a.inj_x);
#pragma unroll
                        for (int e = 0; e < 4; ++e) res.v[e] = ihit[e] ? res.v[e] + injv : res.v[e];
                    }
                }
            }
            if constexpr (INJ == 3) {
                if (inj_here && r >= a.inj_x - 3 && r <= a.inj_x + 3) {   // ptsrc.c:49-55: s += ts * exp(-xn*xn - zn*zn), all float
                    const int dxa = r > a.inj_x ? r - a.inj_x : a.inj_x - r;
                    const float g0 = a.gw[dxa][0], g1 = a.gw[dxa][1], g2 = a.gw[dxa][2], g3 = a.gw[dxa][3];   // wave-uniform kernarg reads
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int dz = z0 + e - a.inj_z, dza = dz < 0 ? -dz : dz;
                        const float g = dza == 0 ? g0 : (dza == 1 ? g1 : (dza == 2 ? g2 : g3));
                        if (dza <= 3) res.v[e] = res.v[e] + inj_src * g;
                    }
                }
            }
            if constexpr (IMG) {
                // kernel_img (R:133-144) correlates with the NEW receiver field; the sibling's rtm_main stores the CURRENT one
                // (rwf[it] = P, rtm_main.cpp:211-215), an interior point of which carries damping factors of exactly 1.0f
#pragma unroll
                for (int e = 0; e < 4; ++e) imr.v[e] = qim[Q].v[e] + qps[Q].v[e] * (DD ? c.v[e] : res.v[e]);
            }
#if FDW_ABL_BITS & (4 | 32)
            if (res.v[0] == 123.456f)
#endif
            if (!partial) {
                f4_store(a.pp + (size_t)r * pitch, voff, res);
                if constexpr (IMG) f4_store(a.img + (size_t)r * pitch, voff, imr);
            } else if (act) {
                f4_store(a.pp + (size_t)r * pitch, voff, res);
                if constexpr (IMG) f4_store(a.img + (size_t)r * pitch, voff, imr);
            }

            // ---- refill the slots this row just freed (look-ahead loads) ----------------------
            ring[U] = load_p(r - H + R);
            {
                const int nr = min(r + PF, xe - 1);
                qhal[Q] = load_halo(nr);
                if constexpr (!LAPONLY) {
                    qpp[Q] = load_plain(a.pp, nr);
                    qv2[Q] = load_plain(a.v2, nr);
                }
                if constexpr (IMG) {
                    qps[Q] = load_plain(a.psrc, nr);
                    qim[Q] = load_plain(a.img, nr);
                }
            }
        }
        // keep the look-ahead loads where they were issued: without this the machine scheduler
        // sinks each load to one row before its first use to save registers, which turns the
        // software prefetch into a load-use stall every row
        __builtin_amdgcn_sched_barrier(0);
    };

    int rb = xa;
    // bulk: whole turns of the ring, no tests, exact s_waitcnt bookkeeping
    for (; rb + R <= xe; rb += R)
        static_for<R>([&](auto UU) { row_step(rb, UU, std::false_type{}); });
    // remaining rows (< R)
    if (rb < xe)
        static_for<R>([&](auto UU) { row_step(rb, UU, std::true_type{}); });
}

template <int H, bool TAPER, int INJ, bool IMG, bool LAPONLY, int PF, bool DD = false>
__global__ __launch_bounds__(256) void fdw_step_kernel(const StepArgs a)
{
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
    march<H, TAPER, INJ, IMG, LAPONLY, PF, DD>(a, lane, zs, xa, xe);
}

// ------------------------------------------------------------------------------------------------
// TWO time steps per pass (temporal blocking): u^{n+1} and u^{n+2} from one read of u^n, u^{n-1}, v2.
// Algorithmic traffic drops from 16 to (12 + 8) / 2 = 10 B/point/step.
//
// A wave owns 60 cells (float4 = 4 columns) x xchunk rows of output but loads 64 cells: step 1 is valid
// on lanes 1..62 (a lane's z neighbours come from the adjacent lanes, no halo load at all), step 2 on
// lanes 2..61, so adjacent strips overlap by 4 cells and start 240 columns = 960 B apart (64-B aligned).
// Along x the wave marches step-1 rows s = xa-H .. xe+H-1 and, H rows behind, step-2 rows r = s-H:
// ring1 holds u^n (2H+1 rows + look-ahead, as in the one-step kernel), ring2 the last 2H+1 rows of
// u^{n+1} it has just computed.  Neighbouring tiles recompute the overlap (2H rows, 4 cells): 4/64 of
// the lanes and 2H/xchunk of the step-1 rows are redundant, the price of never synchronising waves.
// Because tiles read each other's input rows/cells, NOTHING is updated in place: u^{n+1} -> out1,
// u^{n+2} -> out2 (four field buffers rotate).  Non-owned lanes are predicated off by the buffer
// descriptor's range check (per-row SRSRC, offset 0xFFFFFFF0), so the stores are unconditional too.
// Arithmetic per point and per step is exactly the one-step kernel's (same helpers), hence bit-identical.
// ------------------------------------------------------------------------------------------------
typedef unsigned v4u __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void f4_store_rsrc(float* row, unsigned row_bytes, unsigned voff_bytes, const f4& a)
{
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(row, 0, row_bytes, 0x00020000);
    const v4f t = {a.v[0], a.v[1], a.v[2], a.v[3]};
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, t), rs, voff_bytes, 0, (FDW_NT & 2) ? 2 : 0);
}

template <int H, bool TAPER, int INJ, bool IMG, int PF>
__device__ __forceinline__ void march2(const Step2Args& a, const int lane, const int cs, const int xa, const int xe, f4* stash)
{
    constexpr int R = ((2 * H + PF + PF - 1) / PF) * PF;   // ring turns == unroll factor (10 for H=4, PF=2)
    constexpr int LOOK = R - 2 * H;
    const size_t pitch = (size_t)a.pitch;
    const int cell = cs + lane;
    const int z0 = cell * 4;                                  // true first column of this lane (may be outside the row)
    const unsigned voff = (unsigned)min(max(z0, 0), a.pitch - 4) * 4u;
    const bool own = (lane >= 2) && (lane <= 61) && (z0 >= 0) && (z0 < a.pitch);
    const unsigned soff = own ? voff : 0xFFFFFFF0u;           // out of range -> the store of this lane is dropped
    const unsigned row_bytes = (unsigned)a.pitch * 4u;
    const int rowmax = a.nxl - 1;

    const bool wave_tap = TAPER && (cs * 4 < a.ztap);
    const bool xtap = wave_tap && ((xa - 2 * H < a.xt_lo) || (xe + 2 * H > a.xt_hi));
    const CoefPairs<H> cpk = coef_pairs<H>(a.cx, a.cz);
    const bool inj_cols = (a.inj_z >= cs * 4) && (a.inj_z < cs * 4 + 256);
    // INJ == 1: point source at (inj_x, inj_z), samples inj[0] -> u^{n+1}, inj[1] -> u^{n+2}          (kernel_src, R:119-122)
    // INJ == 2: receiver row z = inj_z, rows [inj_x, inj_x+inj_n): inj[row-inj_x] -> u^{n+1}, inj2[..] -> u^{n+2} (kernel_sism)
    bool inj_here = false;
    if (INJ == 1) inj_here = inj_cols && (a.inj_x >= xa - H) && (a.inj_x < xe + H);
    if (INJ == 2) inj_here = inj_cols && (a.inj_x < xe + H) && (a.inj_x + a.inj_n > xa - H);
    const float inj0 = (INJ == 1 && inj_here) ? sload(a.inj, 0) : 0.0f;
    const float inj1 = (INJ == 1 && inj_here) ? sload(a.inj, 1) : 0.0f;

    bool mlap[4], mupd[4], znc[4], ihit[4];
    float tzc[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int z = z0 + e;
        mlap[e] = (z >= a.lap_z0) && (z < a.lap_z1);
        mupd[e] = (z >= 0) && (z < a.upd_z1);
        ihit[e] = (z == a.inj_z);
        znc[e] = (z >= 0) && (z < a.ztap);
        tzc[e] = 1.0f;
    }
    if (wave_tap) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (znc[e]) tzc[e] = a.taperz[z0 + e];
    }
    auto taper_row = [&](f4& v, int row) {
        if (!xtap) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v.v[e] = v.v[e] * tzc[e];
        } else {
            const int rc = min(max(row, 0), a.nxl - 1);
            const float txr = sload(a.txfac, rc);
            const bool rowtz = row < a.tz_x1;
#pragma unroll
            for (int e = 0; e < 4; ++e) v.v[e] = taper1(v.v[e], tzc[e], znc[e], rowtz, txr);
        }
    };
    auto rowoff = [&](int row) -> size_t { return (size_t)min(max(row, 0), rowmax) * pitch; };
    auto load_p = [&](int row) -> f4 { return f4_load(a.p + rowoff(row), voff); };
    auto load_pw = [&](const float* base, int row) -> f4 { return f4_load(base + rowoff(row), voff); };

    // march counter m = 0 .. M-1; step-1 row s = s0 + m, step-2 row r = s - H; ring1 row (b0 + k) lives in slot k % R
    const int s0 = xa - H, b0 = xa - 2 * H;
    const int M = (xe - xa) + 2 * H;
    f4 ring1[R], ring2[R];
    f4 qpp[PF], qv2[PF], qsa[PF], qsb[PF], qim[PF];
    static_for<R>([&](auto K) { constexpr int k = decltype(K)::value; ring2[k] = f4_zero(); });
    constexpr int NV = LOOK > PF ? LOOK : PF;
    static_for<2 * H>([&](auto K) {
        constexpr int k = decltype(K)::value;
        ring1[k] = load_p(b0 + k);
        __builtin_amdgcn_sched_barrier(0);
    });
    static_for<NV>([&](auto JJ) {
        constexpr int j = decltype(JJ)::value - NV;
        if constexpr (j >= -LOOK) ring1[j + 2 * H + LOOK] = load_p(b0 + j + 2 * H + LOOK);
        if constexpr (j >= -PF) {
            constexpr int m = j + PF;
            qpp[m] = load_pw(a.pp, s0 + m);
            qv2[m] = load_pw(a.v2, s0 + m);
            if constexpr (IMG) {
                qsa[m] = load_pw(a.psrc_a, s0 + m - H);
                qsb[m] = load_pw(a.psrc_b, s0 + m - H);
                qim[m] = load_pw(a.img, s0 + m - H);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    });
    if (wave_tap) {
        static_for<2 * H>([&](auto K) {
            constexpr int k = decltype(K)::value;
            taper_row(ring1[k], b0 + k);
        });
    }

    // One march step.  There is a single, branch-free variant: rows past the end of the tile, step-1 rows outside
    // the tile's own rows and step-2 rows formed before ring2 is full are simply computed on clamped loads and
    // their stores dropped by an out-of-range buffer offset.
    auto row_step = [&](const int mb, auto UU) {
        constexpr int U = decltype(UU)::value;
        constexpr int Q = U % PF;
        const int m = mb + U;
        {
            const int s = s0 + m, r = s - H;
            const bool live = m < M;
            const unsigned soff1 = (live && (s >= xa) && (s < xe)) ? soff : 0xFFFFFFF0u;   // u^{n+1} row belongs to this tile
            const unsigned soff2 = (live && (r >= xa)) ? soff : 0xFFFFFFF0u;               // ring2 full: u^{n+2} row r is valid
            // ================= step 1: u^{n+1}(s) =================================================
            f4 ppt = qpp[Q];
            if (wave_tap) {
                taper_row(ring1[(U + 2 * H) % R], s + H);               // row s+H enters the u^n window
                taper_row(ppt, s);
                if (a.pp_twice) taper_row(ppt, s);
            }
            const f4 c1 = ring1[(U + H) % R];
            f4 lft, rgt;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                lft.v[e] = __shfl_up(c1.v[e], 1, 64);
                rgt.v[e] = __shfl_down(c1.v[e], 1, 64);
            }
            const bool rowok1 = (s >= a.lap_x0) && (s < a.lap_x1);
            const bool rowupd1 = (s >= 0) && (s < a.upd_x1);
            f4 u1;
            {
                const ZPairs zp = zpairs(lft, c1, rgt);
                static_for<2>([&](auto PP) {
                    constexpr int P = decltype(PP)::value;
                    const v2f lap2 = laplacian_pair<H, P>(zp, [&](auto IO) { return f4_pair(ring1[(U + decltype(IO)::value) % R], P); }, cpk);
                    const v2f prod2 = (f4_pair(qv2[Q], P) * a.dt2) * v2f{(rowok1 && mlap[2 * P]) ? lap2.x : 0.0f, (rowok1 && mlap[2 * P + 1]) ? lap2.y : 0.0f};
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const int e = 2 * P + q;
                        const float upd = leapfrog_prod(c1.v[e], ppt.v[e], q ? prod2.y : prod2.x);
                        u1.v[e] = (rowupd1 && mupd[e]) ? upd : ppt.v[e];
                    }
                });
            }
            if constexpr (INJ != 0) {
                if (inj_here) {
                    const bool hit = (INJ == 1) ? (s == a.inj_x) : ((s >= a.inj_x) && (s < a.inj_x + a.inj_n));
                    if (hit) {
                        const float v = (INJ == 1) ? inj0 : sload(a.inj, s - a.inj_x);
#pragma unroll
                        for (int e = 0; e < 4; ++e) u1.v[e] = ihit[e] ? u1.v[e] + v : u1.v[e];
                    }
                }
            }
            f4_store_rsrc(a.out1 + rowoff(s), row_bytes, soff1, u1);
            if (wave_tap) taper_row(u1, s);                              // as "p" of step 2 it is damped once
            ring2[U] = u1;                                               // row s of u^{n+1}
            stash[((m & 7) << 6) + lane] = qv2[Q];                       // v2(s) is needed again H rows later: park it in LDS
            __builtin_amdgcn_sched_barrier(0);                           // keep the two steps' temporaries apart (VGPRs)
            // ================= step 2: u^{n+2}(r), r = s - H =======================================
            {
                f4 pp2 = ring1[U];                                       // T(u^n(r)): oldest row of the u^n window
                const f4 v2r = stash[(((m - H) & 7) << 6) + lane];       // v2(r) parked H march steps ago (this lane's own slot)
                if (wave_tap) taper_row(pp2, r);                         // as "pp" of step 2 it is damped twice
                const f4 c2 = ring2[(U - H + R) % R];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    lft.v[e] = __shfl_up(c2.v[e], 1, 64);
                    rgt.v[e] = __shfl_down(c2.v[e], 1, 64);
                }
                const bool rowok2 = (r >= a.lap_x0) && (r < a.lap_x1);
                f4 u2;
                {
                    const ZPairs zp = zpairs(lft, c2, rgt);
                    static_for<2>([&](auto PP) {
                        constexpr int P = decltype(PP)::value;
                        const v2f lap2 = laplacian_pair<H, P>(zp, [&](auto IO) { return f4_pair(ring2[(U - 2 * H + decltype(IO)::value + R) % R], P); }, cpk);
                        const v2f prod2 = (f4_pair(v2r, P) * a.dt2) * v2f{(rowok2 && mlap[2 * P]) ? lap2.x : 0.0f, (rowok2 && mlap[2 * P + 1]) ? lap2.y : 0.0f};
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            const int e = 2 * P + q;
                            const float upd = leapfrog_prod(c2.v[e], pp2.v[e], q ? prod2.y : prod2.x);
                            u2.v[e] = mupd[e] ? upd : pp2.v[e];
                        }
                    });
                }
                if constexpr (INJ != 0) {
                    if (inj_here) {
                        const bool hit = (INJ == 1) ? (r == a.inj_x) : ((r >= a.inj_x) && (r < a.inj_x + a.inj_n));
                        if (hit) {
                            const float v = (INJ == 1) ? inj1 : sload(a.inj2, r - a.inj_x);
#pragma unroll
                            for (int e = 0; e < 4; ++e) u2.v[e] = ihit[e] ? u2.v[e] + v : u2.v[e];
                        }
                    }
                }
                f4_store_rsrc(a.out2 + rowoff(r), row_bytes, soff2, u2);
                if constexpr (IMG) {
                    // imaging condition of BOTH iterations at row r (kernel_img, R:133-144):  img += psrc_a * u^{n+1}, then
                    // img += psrc_b * u^{n+2}.  u^{n+1}(r) is ring2's centre row; where the image is extracted (interior) no
                    // damping applies, so the damped copy held there is the raw field.  Only owned cells are stored.
                    f4 im = qim[Q];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        im.v[e] = im.v[e] + qsa[Q].v[e] * c2.v[e];
                        im.v[e] = im.v[e] + qsb[Q].v[e] * u2.v[e];
                    }
                    f4_store_rsrc(a.img + rowoff(r), row_bytes, soff2, im);
                }
            }
            // ================= look-ahead loads into the slots this step freed =====================
            ring1[U] = load_p(b0 + m + R);
            qpp[Q] = load_pw(a.pp, s + PF);
            qv2[Q] = load_pw(a.v2, s + PF);
            if constexpr (IMG) {
                qsa[Q] = load_pw(a.psrc_a, r + PF);
                qsb[Q] = load_pw(a.psrc_b, r + PF);
                qim[Q] = load_pw(a.img, r + PF);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };

    for (int mb = 0; mb < M; mb += R)
        static_for<R>([&](auto UU) { row_step(mb, UU); });
}

template <int H, bool TAPER, int INJ, bool IMG, int PF>
__global__ __launch_bounds__(256, IMG ? 3 : (TAPER ? 4 : 2)) void fdw_step2_kernel(const Step2Args a)
{
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int bid = blockIdx.x;
    const int L = (bid & 7) * a.nper + (bid >> 3);
    if (L >= a.nblk) return;
    const int zb = L % a.nzblk;
    const int xb = L / a.nzblk;
    const int strip = zb * 4 + w;
    if (strip >= a.nstrip) return;
    const int xa = a.r0 + xb * a.xchunk;
    const int xe = min(xa + a.xchunk, a.r1);
    if (xa >= xe) return;
    // per-wave LDS slab: 8 rows x 64 lanes x 16 B for the v2 rows waiting between step 1 (row s) and step 2 (row s-H)
    __shared__ f4 v2_stash[4][8 * 64];
    march2<H, TAPER, INJ, IMG, PF>(a, lane, strip * 60 - 2, xa, xe, v2_stash[w]);
}

// ------------------------------------------------------------------------------------------------
// NS time steps per pass: a pipeline of NS waves per workgroup, one wave per time level, rows handed from
// wave to wave through LDS.  Wave k (k = 0..NS-1) computes u^{n+k+1}; only wave 0 reads global memory
// (u^n, u^{n-1}, v2) and only the last two waves write it (u^{n+NS-1} -> out1, u^{n+NS} -> out2), so a
// pass moves 12 B in + 8 B out per point for NS steps instead of per step.
//
// Every wave is the one-step march on a register ring of 2H+1 rows of "its" p field.  At march step m wave k
// works on row r_k(m) = xa - (NS-1)H - k(H+1) + m: a skew of H+1 rows per stage, so that what wave k-1
// produced during step m-1 (its result row r_{k-1}(m-1) = r_k(m)+H, which enters wave k's window, and the
// row its own window dropped, r_{k-1}(m-1)-H = r_k(m), which is wave k's "pp") is consumed during step m;
// one workgroup barrier per march step separates producer and consumer, link buffers alternate by the parity
// of m.  v2 rows ride a 16-row LDS FIFO filled by wave 0.  All waves run IDENTICAL code: the global loads of
// waves k > 0 are sent out of range through the buffer descriptor (no memory request, zeros returned) and the
// stores of waves < NS-2 likewise, so the s_waitcnt counting stays exact and nothing diverges.
// Validity: wave k's rows are good from march step k(2H+1) on (its window then holds only good rows of wave
// k-1); in z every step costs H = one lane per side, so NS lanes per side of a wave are halo and 64-2NS owned.
// Per point and step the arithmetic is the one-step kernel's (packed pairs as in the two-step kernel).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ f4 f4_load_rsrc(const float* row, unsigned row_bytes, unsigned voff_bytes, bool nt)
{
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(row), 0, row_bytes, 0x00020000);
    const v4u t = nt ? __builtin_amdgcn_raw_buffer_load_b128(rs, voff_bytes, 0, 2) : __builtin_amdgcn_raw_buffer_load_b128(rs, voff_bytes, 0, 0);
    const v4f r = __builtin_bit_cast(v4f, t);
    f4 o;
    o.v[0] = r.x; o.v[1] = r.y; o.v[2] = r.z; o.v[3] = r.w;
    return o;
}

// Whole-array descriptors: the row goes in as the scalar offset (one s_mul per row instead of a 64-bit address and a fresh
// descriptor per row and stream), the lane's column as the vector offset.  A lane is switched off with kLaneOff, which is out
// of range of any array the host admits (< 2 GiB) whether or not the hardware adds the scalar offset before its range check.
constexpr unsigned kLaneOff = 0x80000000u;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t array_rsrc(const float* base, unsigned bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, bytes, 0x00020000);
}
__device__ __forceinline__ f4 f4_load_arr(__amdgpu_buffer_rsrc_t rs, unsigned voff_bytes, unsigned row_off_bytes, bool nt)
{
    const v4u t = nt ? __builtin_amdgcn_raw_buffer_load_b128(rs, voff_bytes, row_off_bytes, 2) : __builtin_amdgcn_raw_buffer_load_b128(rs, voff_bytes, row_off_bytes, 0);
    const v4f r = __builtin_bit_cast(v4f, t);
    f4 o;
    o.v[0] = r.x; o.v[1] = r.y; o.v[2] = r.z; o.v[3] = r.w;
    return o;
}
__device__ __forceinline__ void f4_store_arr(__amdgpu_buffer_rsrc_t rs, unsigned voff_bytes, unsigned row_off_bytes, const f4& a)
{
    const v4f t = {a.v[0], a.v[1], a.v[2], a.v[3]};
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, t), rs, voff_bytes, row_off_bytes, (FDW_NT & 2) ? 2 : 0);
}

#ifndef FDW_PIPE_PF
#define FDW_PIPE_PF 2      // rows of global look-ahead of wave 0
#endif
#ifndef FDW_PIPE_ROWS
#define FDW_PIPE_ROWS 1    // march steps between two workgroup barriers of the pipeline kernel (1 or 2)
#endif
// v2 FIFO depth: the last wave reads row m - (NS-1)(H+ROWS) while wave 0 writes rows m .. m+ROWS-1
constexpr int pipe_fifo_rows(int ns, int h, int rows) { return rows == 1 ? 16 : (ns - 1) * (h + rows) + rows; }
template <int FD>
__device__ __forceinline__ int pipe_fifo_slot(int m)
{
    if constexpr ((FD & (FD - 1)) == 0) return m & (FD - 1);
    else return ((m % FD) + FD) % FD;
}

template <int H, int NS, bool TAPER, int INJ, int PF, bool DD = false, int ROWS = FDW_PIPE_ROWS>
__device__ __forceinline__ void marchn(const Step2Args& a, const int lane, const int k, const int cs, const int xa, const int xe,
                                       f4 (*link)[2][2][ROWS][64], f4 (*fifo)[64])
{
    // ROWS march steps between two workgroup barriers (1 or 2): a wave consumes what its predecessor produced during the previous
    // ROWS steps, so consecutive waves work H + ROWS rows apart and a wave's first good row comes ROWS later per stage.
    constexpr int R = ((2 * H + PF + PF - 1) / PF) * PF;
    constexpr int LOOK = R - 2 * H;
    constexpr int SK = H + ROWS;
    constexpr int FD = pipe_fifo_rows(NS, H, ROWS);
    static_assert(ROWS == 1 || (ROWS == 2 && R % 2 == 0), "one or two rows per barrier");
    const bool first = (k == 0);
    const int cell = cs + lane;
    const int z0 = cell * 4;
    const unsigned voff = (unsigned)min(max(z0, 0), a.pitch - 4) * 4u;
    const bool own = (lane >= NS) && (lane <= 63 - NS) && (z0 >= 0) && (z0 < a.pitch);
    const unsigned soff = (own && k >= NS - 2) ? voff : kLaneOff;         // only the last two waves store
    const unsigned loff = first ? voff : kLaneOff;                        // only wave 0 loads
    const unsigned row_bytes = (unsigned)a.pitch * 4u;
    const int rowmax = a.nxl - 1;
    const unsigned arr_bytes = (unsigned)a.nxl * row_bytes;               // < 2 GiB (checked by the host)
    const __amdgpu_buffer_rsrc_t rs_p = array_rsrc(a.p, arr_bytes), rs_pp = array_rsrc(a.pp, arr_bytes), rs_v2 = array_rsrc(a.v2, arr_bytes);
    const __amdgpu_buffer_rsrc_t rs_out = array_rsrc((k == NS - 1) ? a.out2 : a.out1, arr_bytes);

    const bool wave_tap = TAPER && (cs * 4 < a.ztap);
    const bool xtap = wave_tap && ((xa - NS * H < a.xt_lo) || (xe + NS * H > a.xt_hi));
    const CoefPairs<H> cpk = coef_pairs<H>(a.cx, a.cz);
    const int blob = (INJ == 3) ? 3 : 0;                                  // INJ 3: 7x7 Gaussian source of the CPU-serial sibling (ptsrc.c:49-55)
    const bool inj_here = (INJ != 0) && (a.inj_z + blob >= cs * 4) && (a.inj_z - blob < cs * 4 + 256) && (a.inj_x + blob >= xa - NS * H) && (a.inj_x - blob < xe + NS * H);
    const float injv = inj_here ? sload(a.inj, k) : 0.0f;                 // source sample of this wave's time step (R:119-122)
    const bool rec_here = DD && (a.rec != nullptr) && (a.rec_z >= cs * 4) && (a.rec_z < cs * 4 + 256);

    bool mlap[4], mupd[4], znc[4], ihit[4];
    float tzc[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int z = z0 + e;
        mlap[e] = (z >= a.lap_z0) && (z < a.lap_z1);
        mupd[e] = (z >= 0) && (z < a.upd_z1);
        ihit[e] = (z == a.inj_z);
        znc[e] = (z >= 0) && (z < a.ztap);
        tzc[e] = 1.0f;
    }
    if (wave_tap) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (znc[e]) tzc[e] = a.taperz[z0 + e];
    }
    auto taper_row = [&](f4& v, int row) {
        if (!xtap) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v.v[e] = v.v[e] * tzc[e];
        } else {
            const int rc = min(max(row, 0), a.nxl - 1);
            const float txr = sload(a.txfac, rc);
            const bool rowtz = row < a.tz_x1;
#pragma unroll
            for (int e = 0; e < 4; ++e) v.v[e] = taper1(v.v[e], tzc[e], znc[e], rowtz, txr);
        }
    };
    auto rowoff = [&](int row) -> unsigned { return (unsigned)min(max(row, 0), rowmax) * row_bytes; };
    auto load_p = [&](int row) -> f4 { return f4_load_arr(rs_p, loff, rowoff(row), (FDW_NT & 4) != 0); };
    auto load_pw = [&](__amdgpu_buffer_rsrc_t rs, int row) -> f4 { return f4_load_arr(rs, loff, rowoff(row), (FDW_NT & 1) != 0); };

    // rows: wave 0's centre row at march step m is s0 + m (what the global loads follow); this wave's is rk + m
    const int s0 = xa - (NS - 1) * H, b0 = s0 - H;
    const int rk = s0 - k * SK;
    const int M = (xe - xa) + (NS - 1) * (2 * H + ROWS);
    const int kp = max(k - 1, 0);
    f4 ring[R];
    f4 qpp[PF], qv2[PF];
    constexpr int NV = LOOK > PF ? LOOK : PF;
    static_for<2 * H>([&](auto K) {
        constexpr int kk = decltype(K)::value;
        ring[kk] = load_p(b0 + kk);
        __builtin_amdgcn_sched_barrier(0);
    });
    static_for<NV>([&](auto JJ) {
        constexpr int j = decltype(JJ)::value - NV;
        if constexpr (j >= -LOOK) ring[j + 2 * H + LOOK] = load_p(b0 + j + 2 * H + LOOK);
        if constexpr (j >= -PF) {
            constexpr int mm = j + PF;
            qpp[mm] = load_pw(rs_pp, s0 + mm);
            qv2[mm] = load_pw(rs_v2, s0 + mm);
        }
        __builtin_amdgcn_sched_barrier(0);
    });
    if (wave_tap) {
        static_for<2 * H>([&](auto K) {
            constexpr int kk = decltype(K)::value;
            taper_row(ring[kk], b0 + kk);
        });
    }

    auto row_step = [&](const int mb, auto UU) {
        constexpr int U = decltype(UU)::value;
        constexpr int Q = U % PF;
        constexpr int E = (U + 2 * H) % R;                      // slot of the row entering the window this step
        const int m = mb + U;
        const int r = rk + m;
        constexpr int SLOT = U % ROWS;                          // which of the ROWS rows between two barriers
        const int par = (m / ROWS) & 1;                          // link buffers alternate per barrier interval
        // ---- what the previous wave handed over during march step m-1 ----
#if FDW_ABL_BITS & 128
        const f4 nr = ring[U], ppl = ring[(U + 1) % R], v2l = ring[(U + 2) % R];
#else
        const f4 nr = link[kp][par ^ 1][0][SLOT][lane];
        const f4 ppl = link[kp][par ^ 1][1][SLOT][lane];
        const f4 v2l = fifo[pipe_fifo_slot<FD>(m - k * SK)][lane];
#endif
        f4 ppt, v2t;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            ring[E].v[e] = first ? ring[E].v[e] : nr.v[e];
            ppt.v[e] = first ? qpp[Q].v[e] : ppl.v[e];
            v2t.v[e] = first ? qv2[Q].v[e] : v2l.v[e];
        }
#if !(FDW_ABL_BITS & 128)
        if (first) fifo[pipe_fifo_slot<FD>(m)][lane] = qv2[Q];
#endif
        if (wave_tap) {
            taper_row(ring[E], r + H);                          // entering row: damped once as "p" of this step
            taper_row(ppt, r);                                  // "pp": from memory once (+ once owed), from LDS once more
            if (first && a.pp_twice) taper_row(ppt, r);
        }
        const f4 c1 = ring[(U + H) % R];
        f4 lft, rgt;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
#if FDW_ABL_BITS & 2048
            lft.v[e] = c1.v[(e + 1) & 3]; rgt.v[e] = c1.v[(e + 2) & 3];
#else
            lft.v[e] = __shfl_up(c1.v[e], 1, 64);
            rgt.v[e] = __shfl_down(c1.v[e], 1, 64);
#endif
        }
        const bool rowok = (r >= a.lap_x0) && (r < a.lap_x1);
        const bool rowupd = (r >= 0) && (r < a.upd_x1);
        f4 u;
        if constexpr (DD) {
            // this wave's p field is P of iteration it0 + k: its trace sample (mod_main.cpp:155-157); owned lanes and rows only
            if (rec_here && own && r >= xa && r < xe && r >= a.rec_x0 && r < a.rec_x0 + a.rec_n) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (z0 + e == a.rec_z) a.rec[k * a.rec_n + (r - a.rec_x0)] = c1.v[e];
            }
            float W[12];
#pragma unroll
            for (int e = 0; e < 4; ++e) { W[e] = lft.v[e]; W[4 + e] = c1.v[e]; W[8 + e] = rgt.v[e]; }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float col[2 * H + 1];
#pragma unroll
                for (int io = 0; io <= 2 * H; ++io) col[io] = ring[(U + io) % R].v[e];
                float lap = laplacian_dd_pt<H>(W, e, col, a.cz, a.dx2inv, a.dz2inv);
                lap = (rowok && mlap[e]) ? lap : 0.0f;
                const float upd = leapfrog_prod(c1.v[e], ppt.v[e], (v2t.v[e] * a.dt2) * lap);
                u.v[e] = (rowupd && mupd[e]) ? upd : ppt.v[e];
            }
        } else {
            const ZPairs zp = zpairs(lft, c1, rgt);
            v2f lapq[2];
            laplacian_quad<H>(zp, [&](auto IO) -> const f4& { return ring[(U + decltype(IO)::value) % R]; }, cpk, lapq[0], lapq[1]);
            static_for<2>([&](auto PP) {
                constexpr int P = decltype(PP)::value;
                const v2f lap2 = lapq[P];
                const v2f prod2 = (f4_pair(v2t, P) * a.dt2) * v2f{(rowok && mlap[2 * P]) ? lap2.x : 0.0f, (rowok && mlap[2 * P + 1]) ? lap2.y : 0.0f};
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int e = 2 * P + q;
                    const float upd = leapfrog_prod(c1.v[e], ppt.v[e], q ? prod2.y : prod2.x);
                    u.v[e] = (rowupd && mupd[e]) ? upd : ppt.v[e];
                }
            });
        }
        if constexpr (INJ == 1) {
            if (inj_here && r == a.inj_x) {
#pragma unroll
                for (int e = 0; e < 4; ++e) u.v[e] = ihit[e] ? u.v[e] + injv : u.v[e];
            }
        }
        if constexpr (INJ == 3) {
            if (inj_here && r >= a.inj_x - 3 && r <= a.inj_x + 3) {
                const int dxa = r > a.inj_x ? r - a.inj_x : a.inj_x - r;
                const float g0 = a.gw[dxa][0], g1 = a.gw[dxa][1], g2 = a.gw[dxa][2], g3 = a.gw[dxa][3];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int dz = z0 + e - a.inj_z, dza = dz < 0 ? -dz : dz;
                    const float g = dza == 0 ? g0 : (dza == 1 ? g1 : (dza == 2 ? g2 : g3));
                    if (dza <= 3) u.v[e] = u.v[e] + injv * g;
                }
            }
        }
        // ---- hand over to the next wave: the new row (raw) and the row leaving this window (damped once) ----
#if !(FDW_ABL_BITS & 128)
        link[k][par][0][SLOT][lane] = u;
        link[k][par][1][SLOT][lane] = ring[U];
#endif
        const unsigned so = ((r >= xa) && (r < xe) && (m < M)) ? soff : kLaneOff;
#if FDW_ABL_BITS & 1024
        ring[U] = u;
#else
        f4_store_arr(rs_out, so, rowoff(r), u);
        // ---- look-ahead loads of wave 0 into the slots this step freed ----
        ring[U] = load_p(b0 + m + R);
        qpp[Q] = load_pw(rs_pp, s0 + m + PF);
        qv2[Q] = load_pw(rs_v2, s0 + m + PF);
#endif
#if !(FDW_ABL_BITS & 64)
        if constexpr (SLOT == ROWS - 1) __syncthreads();
#endif
    };

    for (int mb = 0; mb < M; mb += R)
        static_for<R>([&](auto UU) { row_step(mb, UU); });
}

template <int H, int NS, bool TAPER, int INJ, int PF, bool DD = false>
__global__ __launch_bounds__(64 * NS, FDW_PIPE_ROWS != 1 ? 3 : (DD ? 4 : 5)) void fdw_stepn_kernel(const Step2Args a)
{
    const int lane = threadIdx.x & 63;
    const int k = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int bid = blockIdx.x;
    const int L = (bid & 7) * a.nper + (bid >> 3);
    if (L >= a.nblk) return;                    // whole workgroups only: every barrier below is reached by all NS waves
    const int zb = L % a.nstrip;
    const int xb = L / a.nstrip;
    const bool second = xb >= a.chunks_a;       // two row ranges in one launch (the two boundary strips of a slab)
    const int xa = second ? a.r0b + (xb - a.chunks_a) * a.xchunk : a.r0 + xb * a.xchunk;
    const int xe = min(xa + a.xchunk, second ? a.r1b : a.r1);
    if (xa >= xe) return;
    __shared__ f4 link[NS][2][2][FDW_PIPE_ROWS][64];       // [producer wave][parity][0 new row | 1 row leaving the window][row of the interval][lane]
    __shared__ f4 fifo[pipe_fifo_rows(NS, H, FDW_PIPE_ROWS)][64];
    marchn<H, NS, TAPER, INJ, PF, DD>(a, lane, k, zb * (64 - 2 * NS) - NS, xa, xe, link, fifo);
}

// ------------------------------------------------------------------------------------------------
// generic-order kernel: any even order up to FDW_MAX_ORDER, one thread per point, every tap from
// global memory (L1/L2 absorb the reuse).  Same arithmetic, same lazy-taper rules; used for orders
// the register-ring kernel is not instantiated for, and as an independent cross-check of it.
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
        const float upd = leapfrog_pt(pc, ppv, a.v2[k], a.dt2, lap);
        out = (z < a.upd_z1) ? upd : ppv;
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
// device self test of the two hardware behaviours the kernels rely on:
//   out[0..63]    = __shfl_up(src, 1)     (lane 0 keeps its own value)
//   out[64..127]  = __shfl_down(src, 1)   (lane 63 keeps its own value)
//   out[128..383] = a 64 x float4 row written with the range-predicated buffer store: lanes 2..61 store at
//                   their own offset, the others at 0xFFFFFFF0 and must be dropped by the descriptor check
// ------------------------------------------------------------------------------------------------
// Image post-processing: the second-order Laplacian filter of laplace.f90:25-29 on a dense [nx][nz] image, in the
// order the Fortran expression spells, frame left at zero.  HBM bound and tiny (one read, one write per point).
__global__ __launch_bounds__(256) void fdw_image_lap_kernel(const float* img, float* out, int nx, int nz, float dx, float dz)
{
    const int iz = blockIdx.x * 256 + threadIdx.x, ix = blockIdx.y;
    if (iz >= nz) return;
    const size_t k = (size_t)ix * nz + iz;
    float r = 0.0f;
    if (ix >= 1 && ix < nx - 1 && iz >= 1 && iz < nz - 1) {
        const float c = img[k];
        r = ((img[k + 1] - 2.0f * c) + img[k - 1]) / (dz * dz) + ((img[k + nz] - 2.0f * c) + img[k - nz]) / (dx * dx);
    }
    out[k] = r;
}

__global__ void fdw_selftest_kernel(const float* src, float* out)
{
    const int t = threadIdx.x;
    out[t] = __shfl_up(src[t], 1, 64);
    out[64 + t] = __shfl_down(src[t], 1, 64);
    f4 v;
    v.v[0] = v.v[1] = v.v[2] = v.v[3] = src[t];
    const unsigned off = (t >= 2 && t <= 61) ? (unsigned)t * 16u : 0xFFFFFFF0u;
    f4_store_rsrc(out + 128, 64u * 16u, off, v);
}

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
template <int H, int PF>
static hipError_t launch_fast_hp(const StepArgs& a, int mode, hipStream_t s)
{
    const dim3 grid(8 * a.nper), block(256);
    switch (mode) {
    case FDW_MODE_FWD:   hipLaunchKernelGGL((fdw_step_kernel<H, true, 1, false, false, PF>), grid, block, 0, s, a); break;
    case FDW_MODE_PLAIN: hipLaunchKernelGGL((fdw_step_kernel<H, false, 0, false, false, PF>), grid, block, 0, s, a); break;
    case FDW_MODE_RECV:  hipLaunchKernelGGL((fdw_step_kernel<H, true, 2, true, false, PF>), grid, block, 0, s, a); break;
    case FDW_MODE_LAP:   hipLaunchKernelGGL((fdw_step_kernel<H, false, 0, false, true, PF>), grid, block, 0, s, a); break;
    case FDW_MODE_MOD:   hipLaunchKernelGGL((fdw_step_kernel<H, true, 3, false, false, PF, true>), grid, block, 0, s, a); break;
    case FDW_MODE_DD_FWD:  hipLaunchKernelGGL((fdw_step_kernel<H, true, 1, false, false, PF, true>), grid, block, 0, s, a); break;
    case FDW_MODE_DD_RECV: hipLaunchKernelGGL((fdw_step_kernel<H, true, 2, true, false, PF, true>), grid, block, 0, s, a); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_step_fast(const StepArgs& a, int h, int mode, int pf, hipStream_t s)
{
    if (a.nper <= 0) return hipSuccess;
    if (h == 4) {
        switch (pf) {
        case 1: return launch_fast_hp<4, 1>(a, mode, s);
        case 3: return launch_fast_hp<4, 3>(a, mode, s);
        default: return launch_fast_hp<4, 2>(a, mode, s);
        }
    }
    switch (h) {
    case 1: return launch_fast_hp<1, 2>(a, mode, s);
    case 2: return launch_fast_hp<2, 2>(a, mode, s);
    case 3: return launch_fast_hp<3, 2>(a, mode, s);
    default: return hipErrorInvalidValue;
    }
}

hipError_t launch_step2(const Step2Args& a, int h, int mode, hipStream_t s)
{
    if (a.nper <= 0) return hipSuccess;
    if (h != 4) return hipErrorInvalidValue;
    const dim3 grid(8 * a.nper), block(256);
    switch (mode) {
    case FDW_MODE_FWD:   hipLaunchKernelGGL((fdw_step2_kernel<4, true, 1, false, 2>), grid, block, 0, s, a); break;
    case FDW_MODE_PLAIN: hipLaunchKernelGGL((fdw_step2_kernel<4, false, 0, false, 2>), grid, block, 0, s, a); break;
    case FDW_MODE_RECV:  hipLaunchKernelGGL((fdw_step2_kernel<4, true, 2, true, 2>), grid, block, 0, s, a); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_stepn(const Step2Args& a, int h, int mode, hipStream_t s)
{
    if (a.nper <= 0) return hipSuccess;
    if (h != 4) return hipErrorInvalidValue;
    const dim3 grid(8 * a.nper), block(64 * kPipeSteps);
    switch (mode) {
    case FDW_MODE_FWD:   hipLaunchKernelGGL((fdw_stepn_kernel<4, kPipeSteps, true, 1, FDW_PIPE_PF>), grid, block, 0, s, a); break;
    case FDW_MODE_PLAIN: hipLaunchKernelGGL((fdw_stepn_kernel<4, kPipeSteps, false, 0, FDW_PIPE_PF>), grid, block, 0, s, a); break;
    case FDW_MODE_MOD:   hipLaunchKernelGGL((fdw_stepn_kernel<4, kPipeSteps, true, 3, FDW_PIPE_PF, true>), grid, block, 0, s, a); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
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

hipError_t launch_image_laplacian(const float* d_img, float* d_out, int nx, int nz, float dx, float dz, hipStream_t s)
{
    hipLaunchKernelGGL(fdw_image_lap_kernel, dim3((nz + 255) / 256, nx), dim3(256), 0, s, d_img, d_out, nx, nz, dx, dz);
    return hipGetLastError();
}

hipError_t launch_selftest(const float* src, float* out, hipStream_t s)
{
    hipLaunchKernelGGL(fdw_selftest_kernel, dim3(1), dim3(64), 0, s, src, out);
    return hipGetLastError();
}

}  // namespace fdw
