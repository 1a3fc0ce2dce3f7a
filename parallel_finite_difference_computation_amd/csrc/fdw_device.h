// fdw_device.h -- device-side helpers shared by the gfx950 (CDNA4, wave64) kernels of the 2-D acoustic FD path
// (fdw_step1.hip one step per pass, fdw_step2.hip two, fdw_stepn.hip four through a pipeline of waves).
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
//   * z taps come from the two neighbouring lanes (DPP wave shifts, lane_up / lane_down below; the LDS crossbar that
//     __shfl_up / __shfl_down compile to cost the pipeline kernel 4 %); the 4+4 halo columns of a strip come from ONE
//     extra load issued by all lanes (lane 0 the left piece, every other lane the right piece);
//   * taper, Laplacian, leap-frog update, point-source / receiver injection and the imaging
//     condition are fused: p, pp, v2 are read once and pp written once = 16 B/point/step
//     (+12 B/point for the imaging epilogue).  No MFMA: the stencil is HBM-bound (about 2.5 flop/B),
//     the register ring is the cheapest tile there is;
//   * every global load of the march is unconditional (clamped addresses instead of predication) so
//     that hipcc's s_waitcnt bookkeeping stays exact (vmcnt(4..7), never 0); edge handling is
//     wave-uniform branches around VALU / scalar-cache loads only;
//   * fdw_step2_kernel (fdw_step2.hip) does TWO time steps per pass (temporal blocking, 10 B/point/step): overlapped
//     60-cell tiles, a second register ring for u^{n+1}, v2 rows parked in LDS, range-predicated stores;
//   * fdw_stepn_kernel (fdw_stepn.hip) does FOUR: a workgroup is a pipeline of four waves, one per time level, rows
//     handed from wave to wave through LDS, one barrier per march step (5 B/point/step; the kernel the headline bench runs).
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
#pragma once
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
#if FDW_ABL_BITS & 512
    return (2.0f * p - pp) + prod;      // timing experiment: what the fp64 chain (3 cvt + fma + add + cvt per cell) costs
#endif
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

// The neighbouring lane's value through the VALU's DPP path (v_mov_b32_dpp wave_shr:1 / wave_shl:1) instead of the LDS crossbar
// (ds_bpermute_b32, what __shfl_up / __shfl_down compile to).  Lane 0 (63) has no source and reads 0: both are halo lanes.
template <int CTRL>
__device__ __forceinline__ float lane_shift(float x)
{
    // bound_ctrl: the lane without a source reads 0, so the builtin's "old" operand is dead and costs no initialising v_mov
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, true));
}
#ifndef FDW_DPP
#define FDW_DPP 1          // 0: __shfl_up / __shfl_down (timing experiments)
#endif
__device__ __forceinline__ float lane_up(float x)
{
#if FDW_DPP
    return lane_shift<0x138>(x);        // wave_shr:1 -- lane i takes lane i-1's
#else
    return __shfl_up(x, 1, 64);
#endif
}
__device__ __forceinline__ float lane_down(float x)
{
#if FDW_DPP
    return lane_shift<0x130>(x);        // wave_shl:1
#else
    return __shfl_down(x, 1, 64);
#endif
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
#if FDW_ABL_BITS & 256
    {      // timing experiment: every input kept alive, almost no arithmetic
        const f4 r0 = row(std::integral_constant<int, 0>{}), r8 = row(std::integral_constant<int, 2 * H>{});
        lap01 = z.E[0] + z.E[5] + v2f{r0.v[0], r0.v[1]} + v2f{r8.v[0], r8.v[1]};
        lap23 = z.O[0] + z.O[4] + v2f{r0.v[2], r0.v[3]} + v2f{r8.v[2], r8.v[3]};
        return;
    }
#endif
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

// ---- FAST numerics (fdw_params.numerics = FDW_NUMERICS_FAST; include/fdwave.h) ---------------------------------------------------
// The EXACT Laplacian above is the reference's arithmetic operation for operation (nvcc --fmad=false): 17 products and 17 sums per axis pair,
// every one rounded -- 72 packed instructions per lane and row, which is what binds the pipeline kernel (VALU issue).  The
// north star asks for 1e-5, not for bits, and the reference's own FMA / no-FMA builds differ by 4e-6 over 1 700 steps.  FAST
// uses the symmetry of the weights and fused multiply-adds:
//     lap = c0 p(i,j) + sum_{k=1..H} [ cz_k (p(i,j-k) + p(i,j+k)) + cx_k (p(i-k,j) + p(i+k,j)) ],   c0 = cz_0 + cx_0 (formed on the host in fp32)
// as ONE chain per point: acc = c0 * p; for k = 1..H: acc = fma(p(j-k) + p(j+k), cz_k, acc); acc = fma(p(i-k) + p(i+k), cx_k, acc)
// = 1 + 4 H operations instead of 4 (2 H + 1) + 1: 34 packed instructions per lane and row at order 8 instead of 72.  Everything else (lazy
// taper, masks, truncated extents, the fp64 leap-frog with its single rounding, injection, imaging) is unchanged.  The formula is a
// specification of its own: the scalar and the packed form below and oracle/fdw_oracle.c's orc numerics = 1 agree bit for bit (fma and add
// are correctly rounded wherever they run), so decomposed == single-domain and kernel == kernel still hold bitwise in FAST mode.
template <int SEL>
__device__ __forceinline__ v2f pk_fma_sel(v2f w, v2f cpair, v2f acc)
{
    v2f r;
    if constexpr (SEL == 0)
        asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(r) : "v"(w), "s"(cpair), "v"(acc));
    else
        asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "=v"(r) : "v"(w), "s"(cpair), "v"(acc));
    return r;
}
// scalar form (generic-order kernel; cz / cx are the io-indexed weight vectors, tap k = index H - k = H + k)
__device__ __forceinline__ float laplacian_fast_tap(float acc, float zl, float zr, float xl, float xr, float czk, float cxk)
{
    acc = __builtin_fmaf(zl + zr, czk, acc);
    return __builtin_fmaf(xl + xr, cxk, acc);
}
// cells (2P, 2P+1) of the lane
template <int H, int P, class Col>
__device__ __forceinline__ v2f laplacian_fast_pair(const ZPairs& z, Col&& col, const CoefPairs<H>& c, v2f c0)
{
    auto zw = [&](auto KK) -> v2f {
        constexpr int k = decltype(KK)::value;
        return (k & 1) ? z.O[k >> 1] : z.E[k >> 1];
    };
    v2f acc = pk_mul_sel<0>(col(std::integral_constant<int, H>{}), c0);
    static_for<H>([&](auto KK) {
        constexpr int k = decltype(KK)::value + 1;
        constexpr int ic = H - k;
        const v2f sz = zw(std::integral_constant<int, 4 + 2 * P - k>{}) + zw(std::integral_constant<int, 4 + 2 * P + k>{});
        acc = pk_fma_sel<ic & 1>(sz, c.z[ic >> 1], acc);
        const v2f sx = col(std::integral_constant<int, H - k>{}) + col(std::integral_constant<int, H + k>{});
        acc = pk_fma_sel<ic & 1>(sx, c.x[ic >> 1], acc);
    });
    return acc;
}
// both pairs of a lane side by side (two independent chains)
template <int H, class Row>
__device__ __forceinline__ void laplacian_fast_quad(const ZPairs& z, Row&& row, const CoefPairs<H>& c, v2f c0, v2f& lap01, v2f& lap23)
{
    auto zw = [&](auto KK) -> v2f {
        constexpr int k = decltype(KK)::value;
        return (k & 1) ? z.O[k >> 1] : z.E[k >> 1];
    };
#if FDW_ABL_BITS & 256
    {      // timing experiment: every input kept alive, almost no arithmetic
        const f4 r0 = row(std::integral_constant<int, 0>{}), r8 = row(std::integral_constant<int, 2 * H>{});
        lap01 = z.E[0] + z.E[5] + v2f{r0.v[0], r0.v[1]} + v2f{r8.v[0], r8.v[1]};
        lap23 = z.O[0] + z.O[4] + v2f{r0.v[2], r0.v[3]} + v2f{r8.v[2], r8.v[3]};
        return;
    }
#endif
    const f4 rc = row(std::integral_constant<int, H>{});
    v2f a0 = pk_mul_sel<0>(v2f{rc.v[0], rc.v[1]}, c0), a1 = pk_mul_sel<0>(v2f{rc.v[2], rc.v[3]}, c0);
    static_for<H>([&](auto KK) {
        constexpr int k = decltype(KK)::value + 1;
        constexpr int ic = H - k;
        const v2f sz0 = zw(std::integral_constant<int, 4 - k>{}) + zw(std::integral_constant<int, 4 + k>{});
        const v2f sz1 = zw(std::integral_constant<int, 6 - k>{}) + zw(std::integral_constant<int, 6 + k>{});
        a0 = pk_fma_sel<ic & 1>(sz0, c.z[ic >> 1], a0);
        a1 = pk_fma_sel<ic & 1>(sz1, c.z[ic >> 1], a1);
        const f4 rm = row(std::integral_constant<int, H - k>{}), rp = row(std::integral_constant<int, H + k>{});
        const v2f sx0 = v2f{rm.v[0], rm.v[1]} + v2f{rp.v[0], rp.v[1]}, sx1 = v2f{rm.v[2], rm.v[3]} + v2f{rp.v[2], rp.v[3]};
        a0 = pk_fma_sel<ic & 1>(sx0, c.x[ic >> 1], a0);
        a1 = pk_fma_sel<ic & 1>(sx1, c.x[ic >> 1], a1);
    });
    lap01 = a0;
    lap23 = a1;
}
// dispatch on the numerics mode (NUM 0 = EXACT, 1 = FAST)
template <int NUM, int H, int P, class Col>
__device__ __forceinline__ v2f lap_pair(const ZPairs& z, Col&& col, const CoefPairs<H>& c, v2f c0)
{
    if constexpr (NUM == 0) return laplacian_pair<H, P>(z, col, c);
    else return laplacian_fast_pair<H, P>(z, col, c, c0);
}
template <int NUM, int H, class Row>
__device__ __forceinline__ void lap_quad(const ZPairs& z, Row&& row, const CoefPairs<H>& c, v2f c0, v2f& lap01, v2f& lap23)
{
    if constexpr (NUM == 0) laplacian_quad<H>(z, row, c, lap01, lap23);
    else laplacian_fast_quad<H>(z, row, c, c0, lap01, lap23);
}

// The CPU-serial sibling's Laplacian (laplacian_dd_pt above; fd.c:28-36) for the lane's two pairs: one accumulator chain per pair, per
// tap the z term then the x term, each (value * weight) * inverse spacing squared -- the same individually rounded operations in the
// same order, two cells per instruction.  c.z = the unscaled weights; inv = (dz2inv, dx2inv) in an SGPR pair.
template <int H, class Row>
__device__ __forceinline__ void laplacian_dd_quad(const ZPairs& z, Row&& row, const CoefPairs<H>& c, v2f inv, v2f& lap01, v2f& lap23)
{
    v2f a0 = {0.0f, 0.0f}, a1 = {0.0f, 0.0f};
    static_for<2 * H + 1>([&](auto IO) {
        constexpr int io = decltype(IO)::value;
        constexpr int k0 = 4 - H + io, k1 = 6 - H + io;
        constexpr int ic = io <= H ? io : 2 * H - io;
        const f4 r = row(IO);
        a0 = a0 + pk_mul_sel<0>(pk_mul_sel<ic & 1>((k0 & 1) ? z.O[k0 >> 1] : z.E[k0 >> 1], c.z[ic >> 1]), inv);
        a1 = a1 + pk_mul_sel<0>(pk_mul_sel<ic & 1>((k1 & 1) ? z.O[k1 >> 1] : z.E[k1 >> 1], c.z[ic >> 1]), inv);
        a0 = a0 + pk_mul_sel<1>(pk_mul_sel<ic & 1>(v2f{r.v[0], r.v[1]}, c.z[ic >> 1]), inv);
        a1 = a1 + pk_mul_sel<1>(pk_mul_sel<ic & 1>(v2f{r.v[2], r.v[3]}, c.z[ic >> 1]), inv);
    });
    lap01 = a0;
    lap23 = a1;
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

// ---- buffer-descriptor accesses: lanes switched off by an out-of-range offset (two-step and pipeline kernels) ----
typedef unsigned v4u __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void f4_store_rsrc(float* row, unsigned row_bytes, unsigned voff_bytes, const f4& a)
{
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(row, 0, row_bytes, 0x00020000);
    const v4f t = {a.v[0], a.v[1], a.v[2], a.v[3]};
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, t), rs, voff_bytes, 0, (FDW_NT & 2) ? 2 : 0);
}

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
// HARDWARE HAZARD (gfx950, found in round 2 by scripts/stress_rtmslab.py): a buffer_store_dwordx4 whose soffset is an SGPR still reads its
// data VGPRs during the next instruction slots; a VALU instruction that overwrites one of them straight after the store makes the store
// write the NEW value (seen as a lane's byte offset landing in the wavefield, on some launches only).  hipcc (ROCm 7.2) emitted the two
// back to back (it seems to pad that pair only for stores without a register soffset).  The wait below takes the four data registers as
// inputs, so every later write to them is at least two wait states behind the store.
__device__ __forceinline__ void f4_store_arr(__amdgpu_buffer_rsrc_t rs, unsigned voff_bytes, unsigned row_off_bytes, const f4& a)
{
    const v4f t = {a.v[0], a.v[1], a.v[2], a.v[3]};
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, t), rs, voff_bytes, row_off_bytes, (FDW_NT & 2) ? 2 : 0);
#ifndef FDW_NO_STORE_PAD       // (defined only to show that tests/test_slabs_gpu.py::test_full_size_shot_is_reproducible... catches the hazard)
    asm volatile("s_nop 1" : : "v"(t) : "memory");
#endif
}

}  // namespace fdw
