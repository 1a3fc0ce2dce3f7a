// fdw_stepn.hip -- four time steps per pass: a pipeline of waves through LDS.  Shared helpers and design notes: fdw_device.h.
#include "fdw_device.h"

#pragma clang fp contract(off)

namespace fdw {

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
// of m.  v2 dt2 rows (formed once, by wave 0) ride a 16-row LDS FIFO.  In the FULL body all waves run identical code: the global loads of
// waves k > 0 are sent out of range through the buffer descriptor (no memory request, zeros returned) and the stores of waves < NS-2
// likewise, so the s_waitcnt counting stays exact and nothing diverges.  Workgroups away from the frame of the grid, the damped strip and
// the sources -- nine in ten on a large grid -- run the LEAN body instead (pipe_lean: no masks, clamps, damping or injection code), which
// is compiled once for wave 0 and once for the other waves (WK: no selects between "from memory" and "from LDS", no switched-off loads).
// Validity: wave k's rows are good from march step k(2H+1) on (its window then holds only good rows of wave
// k-1); in z every step costs H = one lane per side, so NS lanes per side of a wave are halo and 64-2NS owned.
// Per point and step the arithmetic is the one-step kernel's (packed pairs as in the two-step kernel).
// ------------------------------------------------------------------------------------------------
#ifndef FDW_PIPE_PF
#define FDW_PIPE_PF 2      // rows of global look-ahead of wave 0
#endif
#ifndef FDW_DD_WG
#define FDW_DD_WG 4
#endif
#ifndef FDW_PIPE_WG
#define FDW_PIPE_WG 5      // workgroups per CU the forward kernels are held to by their launch bounds (5 x 4 waves: 96 VGPRs)
#endif
#ifndef FDW_PIPE_OPT
#define FDW_PIPE_OPT 993   // 1: waves skip the march steps outside their useful window; 4: the frame masks only in workgroups that touch the frame;
                           // 32: workgroups away from the frame, the damped strip and the sources run the lean body (pipe_lean);
                           // 64: neighbouring lanes' values through DPP (v_mov_b32_dpp wave_shr / wave_shl) instead of ds_bpermute_b32;
                           // 128 / 256 / 512: the lean body compiled once for wave 0 and once for the other waves (forward kernel / source-field role / receiver
                           // role of the fused backward kernel)
#endif                     //    (bit 4 measured slower: 581 vs 590 Gpoints/s at 8192^2; bit 16: idle-step skipping for the modelling dialect, 5 % slower)
#ifndef FDW_PIPE_ROWS
#define FDW_PIPE_ROWS 1    // march steps between two workgroup barriers of the pipeline kernel (1 or 2)
#endif
// v2 FIFO depth: the last wave reads row m - (NS-1)(H+ROWS) while wave 0 writes rows m .. m+ROWS-1
constexpr int kFusedFifoRows = 20;      // fused backward kernel: a v2 row is 1 + 3 (H + 1) = 16 march steps under way from the first wave to the last
constexpr int pipe_fifo_rows(int ns, int h, int rows) { return rows == 1 ? ((ns - 1) * (h + 1) + 1 <= 8 ? 8 : 16) : (ns - 1) * (h + rows) + rows; }
template <int FD>
__device__ __forceinline__ int pipe_fifo_slot(int m)
{
    if constexpr ((FD & (FD - 1)) == 0) return m & (FD - 1);
    else return ((m % FD) + FD) % FD;
}

// BK: 0 forward / modelling loops; 1 source field of the backward loop (every wave stores its level); 2 receiver field of the backward
// loop (INJ = 2: trace samples per level, imaging against plev[k], image rows chained through imf);
// 3 / 4: the two roles of the FUSED backward kernel (fdw_back4_kernel: eight waves, four per field): 3 = source field (PLAIN arithmetic,
// wave 0 fills the shared v2 FIFO, every wave's new row also serves the receiver wave of its level), 4 = receiver field (one march step
// behind: its rows, windows and FIFO slots are those of role 3 shifted by D = 1, so that the source-field row it images against was
// written to the link buffers during the step before; v2 for all four waves from the FIFO; image as in BK 2)
// LEAN: the body for workgroups that touch neither the frame of the grid (no Laplacian / update masks, no row clamps), nor the damped strip,
// nor the source (instantiate with TAPER = false, INJ = 0): the kernel picks it per workgroup (pipe_lean)
// WK: 0 = the wave finds out at run time whether it is wave 0 (full body); 1 / 2 = compiled for wave 0 / for the other waves (lean body)
// NUM: 0 = the reference's exact arithmetic, 1 = FAST numerics (symmetric sums + fused multiply-adds, fdw_device.h)
template <int H, int NS, bool TAPER, int INJ, int PF, bool DD = false, int BK = 0, int ROWS = FDW_PIPE_ROWS, bool LEAN = false, int WK = 0, int NUM = 0>
__device__ __forceinline__ void marchn(const Step2Args& a, const int lane, const int k, const int cs, const int xa, const int xe,
                                       f4 (*link)[2][2][ROWS][64], f4 (*fifo)[64], f4 (*imf)[64] = nullptr, f4 (*linkx)[2][2][ROWS][64] = nullptr)
{
    static_assert(!LEAN || (!TAPER && INJ == 0), "the lean body has no damping, no injection (and records no trace)");
    constexpr bool IMG = (BK == 2 || BK == 4);
    constexpr int D = (BK == 4) ? 1 : 0;                      // this role runs D march steps behind
    constexpr int DL = (BK >= 3) ? 1 : 0;                     // ... so both roles of the fused kernel loop one step longer
    // ROWS march steps between two workgroup barriers (1 or 2): a wave consumes what its predecessor produced during the previous
    // ROWS steps, so consecutive waves work H + ROWS rows apart and a wave's first good row comes ROWS later per stage.
    constexpr int R = ((2 * H + PF + PF - 1) / PF) * PF;
    constexpr int LOOK = R - 2 * H;
    constexpr int SK = H + ROWS;
    constexpr int FD = BK >= 3 ? kFusedFifoRows : pipe_fifo_rows(NS, H, ROWS);
    static_assert(ROWS == 1 || (ROWS == 2 && R % 2 == 0), "one or two rows per barrier");
    static_assert(BK < 3 || ROWS == 1, "the fused backward kernel assumes one row per barrier");
    const bool first = WK == 1 ? true : (WK == 2 ? false : (k == 0));      // WK 1 / 2: the body compiled for wave 0 / for the other waves
    const int cell = cs + lane;
    const int z0 = cell * 4;
    const unsigned voff = (unsigned)min(max(z0, 0), a.pitch - 4) * 4u;
    const bool own = (lane >= NS) && (lane <= 63 - NS) && (z0 >= 0) && (z0 < a.pitch);
#if FDW_ABL_BITS & 8192          // timing experiment: no global stores
    const unsigned soff = kLaneOff;
#else
    const unsigned soff = (own && (BK == 1 || k >= NS - 2)) ? voff : kLaneOff;      // only the last two waves store (BK 1: all four levels are kept)
#endif
#if FDW_ABL_BITS & 16384         // timing experiment: no global loads
    const unsigned loff = kLaneOff;
#else
    const unsigned loff = first ? voff : kLaneOff;                        // only wave 0 loads
#endif
    const unsigned row_bytes = (unsigned)a.pitch * 4u;
    const int rowmax = a.nxl - 1;
    const unsigned arr_bytes = (unsigned)a.nxl * row_bytes;               // < 2 GiB (checked by the host)
    const float *gp = BK == 4 ? a.rp : a.p, *gpp = BK == 4 ? a.rpp : a.pp;      // the fused kernel's receiver role has its own pair of fields
    float *go1 = BK == 4 ? a.rout1 : a.out1, *go2 = BK == 4 ? a.rout2 : a.out2;
    const __amdgpu_buffer_rsrc_t rs_p = array_rsrc(gp, arr_bytes), rs_pp = array_rsrc(gpp, arr_bytes), rs_v2 = array_rsrc(a.v2, arr_bytes);
    const __amdgpu_buffer_rsrc_t rs_out = array_rsrc((k == NS - 1) ? go2 : ((BK == 1 && k < NS - 2) ? (k == 0 ? a.lvl0 : a.lvl1) : go1), arr_bytes);
    // BK 2: this wave's source-field level and, for wave 0 / the last wave, the image
    const __amdgpu_buffer_rsrc_t rs_lev = array_rsrc(BK == 2 ? a.plev[k] : gp, arr_bytes), rs_img = array_rsrc(IMG ? a.img : go1, arr_bytes);
    const unsigned ioff = (IMG && own) ? voff : kLaneOff;                 // imaging: owned lanes only

    const bool wave_tap = TAPER && (cs * 4 < a.ztap);
    const bool xtap = wave_tap && ((xa - NS * H < a.xt_lo) || (xe + NS * H > a.xt_hi));
    const CoefPairs<H> cpk = (DD && NUM == 0) ? coef_pairs<H>(a.cz, a.cz) : coef_pairs<H>(a.cx, a.cz);      // DD, exact: the unscaled weights (the spacings enter per term)
    const v2f ddinv = v2f{a.dz2inv, a.dx2inv};
    const v2f c0p = v2f{a.c0, a.c0};                                      // FAST numerics: weight of the centre point
    const int blob = (INJ == 3) ? 3 : 0;                                  // INJ 3: 7x7 Gaussian source of the CPU-serial sibling (ptsrc.c:49-55)
    const bool inj_here = (INJ == 2) ? ((a.inj_z >= cs * 4) && (a.inj_z < cs * 4 + 256) && (a.inj_x < xe + NS * H) && (a.inj_x + a.inj_n > xa - NS * H))
                                     : ((INJ != 0) && (a.inj_z + blob >= cs * 4) && (a.inj_z - blob < cs * 4 + 256) && (a.inj_x + blob >= xa - NS * H) && (a.inj_x - blob < xe + NS * H));
    const float injv = (INJ != 2 && inj_here) ? sload(a.inj, k) : 0.0f;    // source sample of this wave's time step (R:119-122)
    const float* injk = a.inj + (INJ == 2 ? k * a.inj_stride : 0);        // INJ 2: the trace samples of iteration it + k (R:124-131)
    const bool rec_here = DD && !LEAN && (a.rec != nullptr) && (a.rec_z >= cs * 4) && (a.rec_z < cs * 4 + 256);

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
    auto rowoff = [&](int row) -> unsigned { return LEAN ? (unsigned)row * row_bytes : (unsigned)min(max(row, 0), rowmax) * row_bytes; };
#if FDW_ABL_BITS & 4096          // timing experiment: every load of the march reads the chunk's first rows again (cache hits: the cost of the instructions without the memory behind them)
    auto lrow = [&](int row) -> unsigned { return (unsigned)(max(xa, 0) + (row & 3)) * row_bytes; };
#else
    auto lrow = [&](int row) -> unsigned { return rowoff(row); };
#endif
    auto load_p = [&](int row) -> f4 { return f4_load_arr(rs_p, loff, lrow(row), (FDW_NT & 4) != 0); };
    auto load_pw = [&](__amdgpu_buffer_rsrc_t rs, int row) -> f4 { return f4_load_arr(rs, loff, lrow(row), (FDW_NT & 1) != 0); };

    // rows: wave 0's centre row at march step m is s0 + m (what the global loads follow); this wave's is rk + m
    const int s0 = xa - (NS - 1) * H - D, b0 = s0 - H;
    const int rk = s0 - k * SK;
    const int M = (xe - xa) + (NS - 1) * (2 * H + ROWS) + DL;
    const int kp = max(k - 1, 0);
    f4 ring[R];
    f4 qpp[PF], qv2[PF];
    f4 qlv[BK == 2 ? PF : 1], qim[IMG ? PF : 1];                        // BK 2: this wave's source-field rows; BK 2, 4: (wave 0) the image rows, PF steps ahead
    const unsigned imoff = (IMG && first) ? ioff : kLaneOff;              // only wave 0 reads the image
    bool zim[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) zim[e] = (z0 + e >= 0) && (z0 + e < a.img_z1);
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
            if constexpr (BK != 4) qv2[mm] = load_pw(rs_v2, s0 + mm);
            if constexpr (BK == 2) qlv[mm] = f4_load_arr(rs_lev, ioff, rowoff(rk + mm), (FDW_NT & 1) != 0);
            if constexpr (IMG) qim[mm] = f4_load_arr(rs_img, imoff, rowoff(rk + mm), (FDW_NT & 1) != 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    });
    if (wave_tap) {
        static_for<2 * H>([&](auto K) {
            constexpr int kk = decltype(K)::value;
            taper_row(ring[kk], b0 + kk);
        });
    }

    // A wave has nothing useful to compute before its window holds good rows of its predecessor (march steps < k (2H + ROWS); during the
    // last 2H of them it only collects the rows entering its window) nor after the last row a later level needs from it: outside
    // [m_lo, m_hi) it keeps the barriers and its memory instructions, which are predicated off through the buffer descriptor so that
    // the s_waitcnt counting stays exact.  That frees a fifth of the issue slots of a 43-row chunk (7 % at 173 rows) for the other
    // workgroups of the CU.
    const int m_lo = k * (2 * H + ROWS) + D, m_hi = (xe - xa) + 2 * (NS - 1) * H + k * ROWS + D;
    // The frame of the grid (rows / columns where the Laplacian or the update is masked) only concerns the workgroups that touch it;
    // all others take the wave-uniform branch around the mask selects.
    const bool edge = !LEAN && (!(FDW_PIPE_OPT & 4) || (cs * 4 < a.lap_z0) || (cs * 4 + 256 > min(a.lap_z1, a.upd_z1)) || (xa - (NS - 1) * H - NS * SK < max(a.lap_x0, 0)) ||
                      (xe + (NS - 1) * H + NS * SK > min(a.lap_x1, a.upd_x1)));

    auto row_step = [&](const int mb, auto UU) {
        constexpr int U = decltype(UU)::value;
        constexpr int Q = U % PF;
        constexpr int E = (U + 2 * H) % R;                      // slot of the row entering the window this step
        const int m = mb + U;
        const int r = rk + m;
        constexpr int SLOT = U % ROWS;                          // which of the ROWS rows between two barriers
        const int par = (m / ROWS) & 1;                          // link buffers alternate per barrier interval
        constexpr bool SKIP = (FDW_PIPE_OPT & 1) && (!DD || (FDW_PIPE_OPT & 16));   // (the modelling dialect measured 5 % slower with it: its scalar Laplacian leaves no slack)
        const bool act = !SKIP || ((m >= m_lo) && (m < m_hi));
        const bool fill = !SKIP || ((m >= m_lo - 2 * H) && (m < m_hi));                   // collecting the rows that enter the window
        // ---- what the previous wave handed over during march step m-1: the row entering this wave's window ----
        if (fill) {
#if !(FDW_ABL_BITS & 128)
            if (!first) ring[E] = link[kp][par ^ 1][0][SLOT][lane];
#endif
            if (wave_tap) taper_row(ring[E], r + H);            // damped once as "p" of this step
        }
        f4 u;
        float im0, im1, im2, im3;                               // the image row (BK 2, 4), as scalars: defined on every path without an instruction
        if constexpr (IMG) asm volatile("" : "=v"(im0), "=v"(im1), "=v"(im2), "=v"(im3));
        if (act) {
        // pp and v2 of this row live in the look-ahead queue's registers: wave 0 finds them there (loaded PF steps ago), the other waves
        // read theirs from LDS into the same registers (their own look-ahead loads are switched off and return nothing they need)
        f4 &ppt = qpp[Q], &v2t = qv2[Q];                        // v2t: v2 dt2 once past the block below
        if (first) {
            if constexpr (BK == 4) {
                v2t = fifo[pipe_fifo_slot<FD>(m - D)][lane];     // the source-field role's wave 0 parked it one step ago
            } else {
                // v2 dt2 (R:89's first product) is formed once, here, and travels through the FIFO in place of v2
                const v2f w01 = v2f{qv2[Q].v[0], qv2[Q].v[1]} * a.dt2, w23 = v2f{qv2[Q].v[2], qv2[Q].v[3]} * a.dt2;
                qv2[Q].v[0] = w01.x; qv2[Q].v[1] = w01.y; qv2[Q].v[2] = w23.x; qv2[Q].v[3] = w23.y;
#if !(FDW_ABL_BITS & 128)
                fifo[pipe_fifo_slot<FD>(m)][lane] = qv2[Q];
#endif
            }
        } else {
#if FDW_ABL_BITS & 128
            ppt = ring[(U + 1) % R]; v2t = ring[(U + 2) % R];
#else
            ppt = link[kp][par ^ 1][1][SLOT][lane];
            v2t = fifo[pipe_fifo_slot<FD>(m - D - k * SK)][lane];
#endif
        }
        if (wave_tap) {
            taper_row(ppt, r);                                  // "pp": from memory once (+ once owed), from LDS once more
            if (first && a.pp_twice) taper_row(ppt, r);
        }
        const f4 c1 = ring[(U + H) % R];
        f4 lft, rgt;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
#if FDW_ABL_BITS & 2048
            lft.v[e] = c1.v[(e + 1) & 3]; rgt.v[e] = c1.v[(e + 2) & 3];
#elif FDW_PIPE_OPT & 64
            lft.v[e] = lane_shift<0x138>(c1.v[e]);              // wave_shr:1 -- lane i takes lane i-1's
            rgt.v[e] = lane_shift<0x130>(c1.v[e]);              // wave_shl:1
#else
            lft.v[e] = __shfl_up(c1.v[e], 1, 64);
            rgt.v[e] = __shfl_down(c1.v[e], 1, 64);
#endif
        }
        const bool rowok = LEAN || ((r >= a.lap_x0) && (r < a.lap_x1));
        const bool rowupd = LEAN || ((r >= 0) && (r < a.upd_x1));
        if constexpr (DD) {
            // this wave's p field is P of iteration it0 + k: its trace sample (mod_main.cpp:155-157); owned lanes and rows only
            if (rec_here && own && r >= xa && r < xe && r >= a.rec_x0 && r < a.rec_x0 + a.rec_n) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (z0 + e == a.rec_z) a.rec[k * a.rec_n + (r - a.rec_x0)] = c1.v[e];
            }
            // the sibling's single-accumulator Laplacian, two cells per instruction (laplacian_dd_quad)
            const ZPairs zp = zpairs(lft, c1, rgt);
            v2f lapq[2];
            if constexpr (NUM == 0) laplacian_dd_quad<H>(zp, [&](auto IO) -> const f4& { return ring[(U + decltype(IO)::value) % R]; }, cpk, ddinv, lapq[0], lapq[1]);
            else lap_quad<1, H>(zp, [&](auto IO) -> const f4& { return ring[(U + decltype(IO)::value) % R]; }, cpk, c0p, lapq[0], lapq[1]);      // FAST: weights carry their spacing
            static_for<2>([&](auto PP) {
                constexpr int P = decltype(PP)::value;
                v2f lap2 = lapq[P];
                if constexpr (!LEAN) lap2 = v2f{(rowok && mlap[2 * P]) ? lap2.x : 0.0f, (rowok && mlap[2 * P + 1]) ? lap2.y : 0.0f};
                const v2f prod2 = f4_pair(v2t, P) * lap2;          // v2t holds v2 dt2 (see above)
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int e = 2 * P + q;
                    const float upd = leapfrog_prod(c1.v[e], ppt.v[e], q ? prod2.y : prod2.x);
                    u.v[e] = (LEAN || (rowupd && mupd[e])) ? upd : ppt.v[e];
                }
            });
        } else {
            const ZPairs zp = zpairs(lft, c1, rgt);
            v2f lapq[2];
            lap_quad<NUM, H>(zp, [&](auto IO) -> const f4& { return ring[(U + decltype(IO)::value) % R]; }, cpk, c0p, lapq[0], lapq[1]);
            static_for<2>([&](auto PP) {
                constexpr int P = decltype(PP)::value;
                v2f lap2 = lapq[P];
                if constexpr (!LEAN)
                    if (edge) lap2 = v2f{(rowok && mlap[2 * P]) ? lap2.x : 0.0f, (rowok && mlap[2 * P + 1]) ? lap2.y : 0.0f};
                const v2f prod2 = f4_pair(v2t, P) * lap2;          // v2t holds v2 dt2 (see above)
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int e = 2 * P + q;
                    const float upd = leapfrog_prod(c1.v[e], ppt.v[e], q ? prod2.y : prod2.x);
                    u.v[e] = (LEAN || !edge || (rowupd && mupd[e])) ? upd : ppt.v[e];
                }
            });
        }
        if constexpr (INJ == 1) {
            if (inj_here && r == a.inj_x) {
#pragma unroll
                for (int e = 0; e < 4; ++e) u.v[e] = ihit[e] ? u.v[e] + injv : u.v[e];
            }
        }
        if constexpr (INJ == 2) {
            if (inj_here && r >= a.inj_x && r < a.inj_x + a.inj_n) {      // kernel_sism: one add per receiver row (R:124-131)
                const float v = sload(injk, r - a.inj_x);
#pragma unroll
                for (int e = 0; e < 4; ++e) u.v[e] = ihit[e] ? u.v[e] + v : u.v[e];
            }
        }
        if constexpr (IMG) {
            // kernel_img (R:133-144) for iteration it + k: img += F_{it+k} * (the receiver field just formed).  The image row enters at wave 0
            // from memory, collects the four products in iteration order on its way through the LDS FIFO and leaves from the last wave.
            f4 im;
            if (first) im = qim[Q];
            else im = imf[r & 15][lane];
            if (r >= xa && r < xe) {
                f4 lv;
                if constexpr (BK == 4) lv = linkx[k][par ^ 1][0][SLOT][lane];      // F_{it+k}(r): the source-field wave of this level formed it one step ago
                else lv = qlv[Q];
#pragma unroll
                for (int e = 0; e < 4; ++e) im.v[e] = zim[e] ? im.v[e] + lv.v[e] * u.v[e] : im.v[e];
            }
            if (k < NS - 1) imf[r & 15][lane] = im;
            im0 = im.v[0]; im1 = im.v[1]; im2 = im.v[2]; im3 = im.v[3];
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
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) asm volatile("" : "=v"(u.v[e]));      // defined (no instruction) on the path that skips the row
        }
        const unsigned so = (act && (r >= xa) && (r < xe) && (m < M)) ? soff : kLaneOff;
#if FDW_ABL_BITS & 1024
        ring[U] = u;
#else
        f4_store_arr(rs_out, so, rowoff(r), u);
        if constexpr (IMG) {
            const unsigned sim = (k == NS - 1 && act && (r >= xa) && (r < xe)) ? ioff : kLaneOff;
            f4 im;
            im.v[0] = im0; im.v[1] = im1; im.v[2] = im2; im.v[3] = im3;
            f4_store_arr(rs_img, sim, rowoff(r), im);
        }
        // ---- look-ahead loads of wave 0 into the slots this step freed ----
        if constexpr (WK != 2) ring[U] = load_p(b0 + m + R);
        qpp[Q] = load_pw(rs_pp, s0 + m + PF);
        if constexpr (BK != 4) qv2[Q] = load_pw(rs_v2, s0 + m + PF);
        if constexpr (BK == 2) qlv[Q] = f4_load_arr(rs_lev, ioff, rowoff(r + PF), (FDW_NT & 1) != 0);
        if constexpr (IMG) qim[Q] = f4_load_arr(rs_img, imoff, rowoff(r + PF), (FDW_NT & 1) != 0);
#endif
#if !(FDW_ABL_BITS & 64)
        if constexpr (SLOT == ROWS - 1) __syncthreads();
#endif
    };

    for (int mb = 0; mb < M; mb += R)
        static_for<R>([&](auto UU) { row_step(mb, UU); });
}

// Workgroup-uniform: may this tile run the lean body?  Every row it touches -- stencil taps, look-ahead loads (PF + ring rows beyond the
// chunk) -- lies inside the rows where the Laplacian and the update are unmasked, its columns likewise, it is outside the damped strip, and
// neither a source nor (modelling) the receiver line is in it.
template <int H, int NS, bool TAPER, int INJ, bool DD = false>
__device__ __forceinline__ bool pipe_lean(const Step2Args& a, int cs, int xa, int xe)
{
    const int lo = xa - (NS - 1) * H - H - NS * (H + FDW_PIPE_ROWS), hi = xe + (NS - 1) * (2 * H + FDW_PIPE_ROWS) + 2 * H + 16;
    const int c0 = cs * 4, c1 = cs * 4 + 256;
    bool ok = (c0 >= a.lap_z0) && (c1 <= min(a.lap_z1, a.upd_z1)) && (lo >= max(a.lap_x0, 0)) && (hi <= min(min(a.lap_x1, a.upd_x1), a.nxl));
    if (TAPER) {
        ok = ok && (c0 >= a.zt_lo);
        if (a.zt_hi >= 0) ok = ok && (c1 <= a.zt_hi) && (lo >= a.xt_lo) && (hi <= a.xt_hi);      // four-sided damping (taper_apply)
    }
    if (INJ == 3) ok = ok && !((a.inj_z + 3 >= c0) && (a.inj_z - 3 < c1) && (a.inj_x + 3 >= xa - NS * H) && (a.inj_x - 3 < xe + NS * H));
    if (INJ == 2) ok = ok && !((a.inj_z >= c0) && (a.inj_z < c1) && (a.inj_x < xe + NS * H) && (a.inj_x + a.inj_n > xa - NS * H));
    if (INJ == 1) ok = ok && !((a.inj_z >= c0) && (a.inj_z < c1) && (a.inj_x >= xa - NS * H) && (a.inj_x < xe + NS * H));
    if (DD) ok = ok && !((a.rec != nullptr) && (a.rec_z >= c0) && (a.rec_z < c1));                  // trace recording
    return ok;
}

template <int H, int NS, bool TAPER, int INJ, int PF, bool DD = false, int BK = 0, int NUM = 0>
__global__ __launch_bounds__(64 * NS, (FDW_PIPE_ROWS != 1 || BK == 2) ? 3 : (DD ? FDW_DD_WG : FDW_PIPE_WG)) void fdw_stepn_kernel(const Step2Args a)
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
    if constexpr (BK == 2) {
        static_assert(FDW_PIPE_ROWS == 1, "the image FIFO assumes one row per barrier");
        __shared__ f4 imf[16][64];                         // image rows on their way from wave to wave (a row is 3 (H + 1) = 15 steps under way)
        marchn<H, NS, TAPER, INJ, PF, DD, BK, FDW_PIPE_ROWS, false, 0, NUM>(a, lane, k, zb * (64 - 2 * NS) - NS, xa, xe, link, fifo, imf);
    } else if constexpr (BK == 0 && (FDW_PIPE_OPT & 32)) {
        // nine workgroups in ten of a large grid touch neither the frame, nor the damped strip, nor the source: they take the lean body
        const int cs = zb * (64 - 2 * NS) - NS;
        if (pipe_lean<H, NS, TAPER, INJ, DD>(a, cs, xa, xe)) {
            if constexpr ((FDW_PIPE_OPT & 128) != 0 && !DD) {
                if (k == 0) marchn<H, NS, false, 0, PF, DD, 0, FDW_PIPE_ROWS, true, 1, NUM>(a, lane, k, cs, xa, xe, link, fifo);
                else marchn<H, NS, false, 0, PF, DD, 0, FDW_PIPE_ROWS, true, 2, NUM>(a, lane, k, cs, xa, xe, link, fifo);
            } else {
                marchn<H, NS, false, 0, PF, DD, 0, FDW_PIPE_ROWS, true, 0, NUM>(a, lane, k, cs, xa, xe, link, fifo);
            }
        }
        else marchn<H, NS, TAPER, INJ, PF, DD, BK, FDW_PIPE_ROWS, false, 0, NUM>(a, lane, k, cs, xa, xe, link, fifo);
    } else {
        marchn<H, NS, TAPER, INJ, PF, DD, BK, FDW_PIPE_ROWS, false, 0, NUM>(a, lane, k, zb * (64 - 2 * NS) - NS, xa, xe, link, fifo);
    }
}

// Four iterations of the backward loop in ONE pass: a workgroup of eight waves, waves 0-3 the pipeline of the source field (role 3), waves 4-7
// the pipeline of the receiver field one march step behind (role 4).  The source-field levels never leave the chip: the receiver wave of
// level k reads F_{it+k}(row) from the link buffer the source-field wave k wrote it to for its own successor.  6 fields in + 5 out per
// four iterations = 44 B/point (the two-pass form moves 92).  Both roles run the same number of march steps and reach one barrier per step.
template <int H, int NS, int PF, int NUM = 0>
__global__ __launch_bounds__(128 * NS, 2) void fdw_back4_kernel(const Step2Args a)
{
    const int lane = threadIdx.x & 63;
    const int k8 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int bid = blockIdx.x;
    const int L = (bid & 7) * a.nper + (bid >> 3);
    if (L >= a.nblk) return;
    const int zb = L % a.nstrip;
    const int xb = L / a.nstrip;
    const bool second = xb >= a.chunks_a;
    const int xa = second ? a.r0b + (xb - a.chunks_a) * a.xchunk : a.r0 + xb * a.xchunk;
    const int xe = min(xa + a.xchunk, second ? a.r1b : a.r1);
    if (xa >= xe) return;
    static_assert(FDW_PIPE_ROWS == 1, "one row per barrier");
    __shared__ f4 linkF[NS][2][2][1][64];
    __shared__ f4 linkR[NS][2][2][1][64];
    __shared__ f4 fifo[kFusedFifoRows][64];
    __shared__ f4 imf[16][64];
    const int cs = zb * (64 - 2 * NS) - NS;
    constexpr bool kLean = (FDW_PIPE_OPT & 32) != 0, kSplit = (FDW_PIPE_OPT & 256) != 0, kSplitR = (FDW_PIPE_OPT & 512) != 0;
    if (k8 < NS) {
        if (kLean && pipe_lean<H, NS, false, 0>(a, cs, xa, xe)) {
            if constexpr (kSplit) {
                if (k8 == 0) marchn<H, NS, false, 0, PF, false, 3, 1, true, 1, NUM>(a, lane, k8, cs, xa, xe, linkF, fifo);
                else marchn<H, NS, false, 0, PF, false, 3, 1, true, 2, NUM>(a, lane, k8, cs, xa, xe, linkF, fifo);
            } else {
                marchn<H, NS, false, 0, PF, false, 3, 1, true, 0, NUM>(a, lane, k8, cs, xa, xe, linkF, fifo);
            }
        } else {
            marchn<H, NS, false, 0, PF, false, 3, 1, false, 0, NUM>(a, lane, k8, cs, xa, xe, linkF, fifo);
        }
    } else {
        // receiver role: lean where the tile holds neither the damped strip nor the receiver line
        if (kLean && pipe_lean<H, NS, true, 2>(a, cs, xa, xe)) {
            if constexpr (kSplitR) {
                if (k8 == NS) marchn<H, NS, false, 0, PF, false, 4, 1, true, 1, NUM>(a, lane, 0, cs, xa, xe, linkR, fifo, imf, linkF);
                else marchn<H, NS, false, 0, PF, false, 4, 1, true, 2, NUM>(a, lane, k8 - NS, cs, xa, xe, linkR, fifo, imf, linkF);
            } else {
                marchn<H, NS, false, 0, PF, false, 4, 1, true, 0, NUM>(a, lane, k8 - NS, cs, xa, xe, linkR, fifo, imf, linkF);
            }
        } else {
            marchn<H, NS, true, 2, PF, false, 4, 1, false, 0, NUM>(a, lane, k8 - NS, cs, xa, xe, linkR, fifo, imf, linkF);
        }
    }
}

hipError_t launch_stepn(const Step2Args& a, int h, int mode, hipStream_t s)
{
    if (a.nper <= 0) return hipSuccess;
    if (h != 4) return hipErrorInvalidValue;
    const dim3 grid(8 * a.nper), block(64 * kPipeSteps);
    if (a.numerics) {      // FAST numerics (fdw_device.h): the same kernels with NUM = 1
        switch (mode) {
        case FDW_MODE_FWD:   hipLaunchKernelGGL((fdw_stepn_kernel<4, kPipeSteps, true, 1, FDW_PIPE_PF, false, 0, 1>), grid, block, 0, s, a); break;
        case FDW_MODE_PLAIN: hipLaunchKernelGGL((fdw_stepn_kernel<4, kPipeSteps, false, 0, FDW_PIPE_PF, false, 0, 1>), grid, block, 0, s, a); break;
        case FDW_MODE_MOD:   hipLaunchKernelGGL((fdw_stepn_kernel<4, kPipeSteps, true, 3, FDW_PIPE_PF, true, 0, 1>), grid, block, 0, s, a); break;
        case FDW_MODE_PLAIN_ALL: hipLaunchKernelGGL((fdw_stepn_kernel<4, kPipeSteps, false, 0, FDW_PIPE_PF, false, 1, 1>), grid, block, 0, s, a); break;
        case FDW_MODE_RECV:  hipLaunchKernelGGL((fdw_stepn_kernel<4, kPipeSteps, true, 2, FDW_PIPE_PF, false, 2, 1>), grid, block, 0, s, a); break;
        case FDW_MODE_BACK4: hipLaunchKernelGGL((fdw_back4_kernel<4, kPipeSteps, FDW_PIPE_PF, 1>), grid, dim3(128 * kPipeSteps), 0, s, a); break;
        default: return hipErrorInvalidValue;
        }
        return hipGetLastError();
    }
    switch (mode) {
    case FDW_MODE_FWD:   hipLaunchKernelGGL((fdw_stepn_kernel<4, kPipeSteps, true, 1, FDW_PIPE_PF>), grid, block, 0, s, a); break;
    case FDW_MODE_PLAIN: hipLaunchKernelGGL((fdw_stepn_kernel<4, kPipeSteps, false, 0, FDW_PIPE_PF>), grid, block, 0, s, a); break;
    case FDW_MODE_MOD:   hipLaunchKernelGGL((fdw_stepn_kernel<4, kPipeSteps, true, 3, FDW_PIPE_PF, true>), grid, block, 0, s, a); break;
    case FDW_MODE_PLAIN_ALL: hipLaunchKernelGGL((fdw_stepn_kernel<4, kPipeSteps, false, 0, FDW_PIPE_PF, false, 1>), grid, block, 0, s, a); break;
    case FDW_MODE_RECV:  hipLaunchKernelGGL((fdw_stepn_kernel<4, kPipeSteps, true, 2, FDW_PIPE_PF, false, 2>), grid, block, 0, s, a); break;
    case FDW_MODE_BACK4: hipLaunchKernelGGL((fdw_back4_kernel<4, kPipeSteps, FDW_PIPE_PF>), grid, dim3(128 * kPipeSteps), 0, s, a); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace fdw
