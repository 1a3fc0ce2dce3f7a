// fdw_step2.hip -- two time steps per pass (temporal blocking inside one wave).  Shared helpers and design notes: fdw_device.h.
#include "fdw_device.h"

#pragma clang fp contract(off)

namespace fdw {

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
template <int H, bool TAPER, int INJ, bool IMG, int PF, int NUM = 0>
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
    const v2f c0p = v2f{a.c0, a.c0};                          // FAST numerics: weight of the centre point
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
                lft.v[e] = lane_up(c1.v[e]);
                rgt.v[e] = lane_down(c1.v[e]);
            }
            const bool rowok1 = (s >= a.lap_x0) && (s < a.lap_x1);
            const bool rowupd1 = (s >= 0) && (s < a.upd_x1);
            f4 u1;
            {
                const ZPairs zp = zpairs(lft, c1, rgt);
                static_for<2>([&](auto PP) {
                    constexpr int P = decltype(PP)::value;
                    const v2f lap2 = lap_pair<NUM, H, P>(zp, [&](auto IO) { return f4_pair(ring1[(U + decltype(IO)::value) % R], P); }, cpk, c0p);
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
                    lft.v[e] = lane_up(c2.v[e]);
                    rgt.v[e] = lane_down(c2.v[e]);
                }
                const bool rowok2 = (r >= a.lap_x0) && (r < a.lap_x1);
                f4 u2;
                {
                    const ZPairs zp = zpairs(lft, c2, rgt);
                    static_for<2>([&](auto PP) {
                        constexpr int P = decltype(PP)::value;
                        const v2f lap2 = lap_pair<NUM, H, P>(zp, [&](auto IO) { return f4_pair(ring2[(U - 2 * H + decltype(IO)::value + R) % R], P); }, cpk, c0p);
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
                        im.v[e] = (z0 + e < a.img_z1) ? im.v[e] : qim[Q].v[e];      // kernel_img's launch covers interior columns j < zlim only
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

template <int H, bool TAPER, int INJ, bool IMG, int PF, int NUM = 0>
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
    march2<H, TAPER, INJ, IMG, PF, NUM>(a, lane, strip * 60 - 2, xa, xe, v2_stash[w]);
}

hipError_t launch_step2(const Step2Args& a, int h, int mode, hipStream_t s)
{
    if (a.nper <= 0) return hipSuccess;
    if (h != 4) return hipErrorInvalidValue;
    const dim3 grid(8 * a.nper), block(256);
    if (a.numerics) {      // FAST numerics (fdw_device.h)
        switch (mode) {
        case FDW_MODE_FWD:   hipLaunchKernelGGL((fdw_step2_kernel<4, true, 1, false, 2, 1>), grid, block, 0, s, a); break;
        case FDW_MODE_PLAIN: hipLaunchKernelGGL((fdw_step2_kernel<4, false, 0, false, 2, 1>), grid, block, 0, s, a); break;
        case FDW_MODE_RECV:  hipLaunchKernelGGL((fdw_step2_kernel<4, true, 2, true, 2, 1>), grid, block, 0, s, a); break;
        default: return hipErrorInvalidValue;
        }
        return hipGetLastError();
    }
    switch (mode) {
    case FDW_MODE_FWD:   hipLaunchKernelGGL((fdw_step2_kernel<4, true, 1, false, 2>), grid, block, 0, s, a); break;
    case FDW_MODE_PLAIN: hipLaunchKernelGGL((fdw_step2_kernel<4, false, 0, false, 2>), grid, block, 0, s, a); break;
    case FDW_MODE_RECV:  hipLaunchKernelGGL((fdw_step2_kernel<4, true, 2, true, 2>), grid, block, 0, s, a); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace fdw
