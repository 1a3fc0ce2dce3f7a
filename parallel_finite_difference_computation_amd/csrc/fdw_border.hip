// Random-border velocity model on the device: extendvel_linear (cuda_reference_RTM/src/functions.c:336-394)
// and vel2 = vpe * vpe (fd-code.cu:488-494) from an interior model that stays resident in HBM, so a shot uploads nothing but its gather.
//
// The reference draws its border from ONE sequential glibc rand() stream, never seeded, consumed in a fixed loop order (fdw_host.c keeps
// that order for the host array).  Two facts make it a data-parallel job:
//   * the generator is glibc's TYPE_3 additive feedback  y[t] = y[t-31] + y[t-3]  over 32-bit words, a LINEAR recurrence: the window
//     with g = x^K mod (x^31 - x^28 - 1) over Z/2^32,  y[K + s] = sum_i g_i y[i + s]  for every s, so any thread can jump to its own place
//     in the stream with two 31 x 31 products (small tables of x^(31 l) and x^(31 64 h), built once on the host) and then run the
//     recurrence for one ring turn of 31 draws;
//   * which draw a border cell receives, and whether a later phase of the reference's loops overwrites it, is a closed form of its
//     coordinates (the phases below), so the fill is one thread per cell with no ordering between cells.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "fdw_kernels.h"

namespace fdw {

namespace {

template <int I, int N, class F>
__device__ __forceinline__ void unroll(F&& f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        unroll<I + 1, N>(f);
    }
}

// thread t = 64 h + l: draws [31 t, 31 t + 31) of the n asked for.  x^(31 t) = x^(31 l) * x^(31 64 h) mod P: one polynomial product (the block's
// factor is wave-uniform: scalar loads), reduced through x^k = x^(k-3) + x^(k-31), gives g with  y[31 t + s] = sum_i g_i y[i + s]  for every s --
// a second 31 x 31 product against the base words yields the thread's window, then the recurrence runs one ring turn in registers.
__global__ __launch_bounds__(64) void fdw_rand_stream_kernel(RandBase base, const unsigned* __restrict__ tab, long long n, int* __restrict__ out)
{
    const int lane = threadIdx.x;
    const long long t = (long long)blockIdx.x * 64 + lane;
    if (t * kRandLag >= n) return;
    const unsigned* hi = tab + kRandLag * 64 + (size_t)blockIdx.x * kRandLag;
    unsigned a[kRandLag], c[2 * kRandLag - 1];
    unroll<0, kRandLag>([&](auto i) { a[i] = tab[i * 64 + lane]; });
    unroll<0, 2 * kRandLag - 1>([&](auto k) { c[k] = 0; });
    unroll<0, kRandLag>([&](auto j) {
        const unsigned b = hi[j];
        unroll<0, kRandLag>([&](auto i) { c[i + j] += a[i] * b; });
    });
    unroll<0, kRandLag - 1>([&](auto d) {
        constexpr int k = 2 * kRandLag - 2 - d;      // 60 .. 31
        c[k - 3] += c[k];
        c[k - kRandLag] += c[k];
    });
    unsigned w[kRandLag];
    unroll<0, kRandLag>([&](auto m) {
        unsigned acc = 0;
        unroll<0, kRandLag>([&](auto i) { acc += c[i] * base.y[i + m]; });
        w[m] = acc;
    });
    // one ring turn: slot s holds y[K-31+s]; y[K+s] = y[K+s-31] + y[K+s-3], and slot (s+28) mod 31 holds y[K+s-3] by then
    const long long first = t * kRandLag;
    unroll<0, kRandLag>([&](auto s) {
        w[s] += w[(s + 28) % kRandLag];
        if (first + s < n) out[first + s] = (int)(w[s] >> 1);
    });
}

__device__ __forceinline__ float ramp_to_floor(float v, int k, int nb)
{
    const float floor_v = 300.f;
    return v - (v - floor_v) * (float)k / (float)(nb - 1);
}
__device__ __forceinline__ float draw_near(int r, float v, float centre)
{
    const float half = 200.f;
    const int span = (int)(v + half - (centre - half) + 1.f);
    return (float)(r % span) + centre - half;
}

// one thread per cell of the extended grid; `vp` = interior model [nx][nz]; writes vel (may be null) and vel2, both [nxe][pitch]
__global__ __launch_bounds__(256) void fdw_extendvel_kernel(BorderArgs a)
{
    const int z = blockIdx.x * 256 + threadIdx.x;
    const int x = blockIdx.y;
    const int nxe = a.nx + 2 * a.nxb, nze = a.nz + 2 * a.nzb;
    if (z >= nze) return;
    const int x_first = a.nxb, x_last = a.nxb + a.nx - 1, z_first = a.nzb, z_last = a.nzb + a.nz - 1, x_end = nxe - 1, z_end = nze - 1;
    auto interior = [&](int i, int j) { return a.vp[(size_t)(i - x_first) * a.nz + (j - z_first)]; };
    const long long b2 = (long long)a.nx * a.nzb, b4 = b2 + 2ll * a.nz * a.nxb, corner = (long long)a.nzb * (a.nzb + 1);
    const int bz = z_end - z;                    // depth counted from the bottom edge
    float v = 0.f;                               // cells no phase writes (bottom corners wider than deep) keep the caller's zero
    // the reference's phases, latest writer first
    int side = -1, ca = 0;
    if (bz < a.nzb) {
        if (x_end - x < a.nzb) side = 1, ca = x_end - x;          // bottom-right corner triangle pair (done after the left one)
        else if (x < a.nzb) side = 0, ca = x;
    }
    if (side >= 0) {
        // visit (m, n <= m) writes (n, depth m) with its first draw and (m, depth n) with its second
        const bool second = ca >= bz;
        const int m = second ? ca : bz, n = second ? bz : ca;
        const long long idx = b4 + side * corner + 2 * ((long long)m * (m + 1) / 2 + n) + (second ? 1 : 0);
        const float edge = interior(side ? x_last : x_first, z_last);
        v = draw_near(a.draws[idx], edge, ramp_to_floor(edge, a.nxb - 1 - n, a.nzb));
    } else if (z < z_first && (x < x_first || x > x_last)) {
        v = interior(x < x_first ? x_first : x_last, z_first);    // top corners: the replicated top of the first / last interior column
    } else if (x < x_first || x > x_last) {
        if (z <= z_last) {                                        // left / right borders at interior depths
            const bool right = x > x_last;
            const int d = right ? x - x_last - 1 : x_first - 1 - x;
            const long long idx = b2 + 2 * ((long long)(z - z_first) * a.nxb + d) + (right ? 1 : 0);
            const float edge = interior(right ? x_last : x_first, z);
            v = draw_near(a.draws[idx], edge, ramp_to_floor(edge, d, a.nxb));
        }
    } else if (z < z_first) {
        v = interior(x, z_first);                                 // top border: first interior sample replicated
    } else if (z > z_last) {
        const int d = z - z_last - 1;                             // bottom border
        const float edge = interior(x, z_last);
        v = draw_near(a.draws[(long long)(x - x_first) * a.nzb + d], edge, ramp_to_floor(edge, d, a.nzb));
    } else {
        v = interior(x, z);
    }
    const size_t o = (size_t)x * a.pitch + z;
    if (a.vel) a.vel[o] = v;
    a.vel2[o] = v * v;
}

// gathers d_obs[shot][nx][nt] (R:426-435) -> [shot][nt][nx], so that one time step's receiver samples are contiguous: 32 x 32 tiles through LDS
__global__ __launch_bounds__(256) void fdw_gather_transpose_kernel(const float* __restrict__ in, float* __restrict__ out, int rows, int cols)
{
    __shared__ float tile[32][33];
    const size_t shot = (size_t)blockIdx.z * rows * cols;
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int j = ty; j < 32; j += 8)
        if (r0 + j < rows && c0 + tx < cols) tile[j][tx] = in[shot + (size_t)(r0 + j) * cols + c0 + tx];
    __syncthreads();
    for (int j = ty; j < 32; j += 8)
        if (c0 + j < cols && r0 + tx < rows) out[shot + (size_t)(c0 + j) * rows + r0 + tx] = tile[tx][j];
}

}  // namespace

hipError_t launch_gather_transpose(const float* d_in, float* d_out, int nx, int nt, int nshots, hipStream_t s)
{
    if (nx <= 0 || nt <= 0 || nshots <= 0) return hipSuccess;
    hipLaunchKernelGGL(fdw_gather_transpose_kernel, dim3((nt + 31) / 32, (nx + 31) / 32, nshots), dim3(256), 0, s, d_in, d_out, nx, nt);
    return hipGetLastError();
}

hipError_t launch_rand_stream(const RandBase& base, const unsigned* d_tab, long long n, int* d_out, hipStream_t s)
{
    if (n <= 0) return hipSuccess;
    const long long threads = (n + kRandLag - 1) / kRandLag;
    hipLaunchKernelGGL(fdw_rand_stream_kernel, dim3((unsigned)((threads + 63) / 64)), dim3(64), 0, s, base, d_tab, n, d_out);
    return hipGetLastError();
}

hipError_t launch_extendvel(const BorderArgs& a, hipStream_t s)
{
    const int nxe = a.nx + 2 * a.nxb, nze = a.nz + 2 * a.nzb;
    hipLaunchKernelGGL(fdw_extendvel_kernel, dim3((nze + 255) / 256, nxe), dim3(256), 0, s, a);
    return hipGetLastError();
}

// ---- the precondition of the lazy damping, checked on the device (include/fdwave.h, "PRECONDITION") -------------------------------------
// With the reference's truncated launch extents the rows >= xlim are never time-stepped, yet the reference damps them inside the top strip
// every step; the kernels damp on load and cannot reproduce that, so those cells must be zero (as at every call site of the reference).
// The host-array entry points check it on the host; this kernel lets a caller of the fdw_dev_* entry points check a DEVICE array:
// *flag += number of non-zero cells among rows [row0, nxl) x columns [0, ztap).
__global__ __launch_bounds__(256) void fdw_static_strip_check_kernel(const float* __restrict__ f, int pitch, int row0, int ztap, unsigned* __restrict__ flag)
{
    const int z = blockIdx.x * 256 + threadIdx.x, r = row0 + blockIdx.y;
    if (z >= ztap) return;
    if (f[(size_t)r * pitch + z] != 0.0f) atomicAdd(flag, 1u);
}
hipError_t launch_static_strip_check(const float* f, int pitch, int row0, int nrows, int ztap, unsigned* flag, hipStream_t s)
{
    if (nrows <= 0 || ztap <= 0) return hipSuccess;
    hipLaunchKernelGGL(fdw_static_strip_check_kernel, dim3((ztap + 255) / 256, nrows), dim3(256), 0, s, f, pitch, row0, ztap, flag);
    return hipGetLastError();
}

}  // namespace fdw
