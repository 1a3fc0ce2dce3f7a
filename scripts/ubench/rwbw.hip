// Strided read/write microbenchmark (development tool): the memory skeleton of the fused step:
// per row read a (window array, +AHEAD rows ahead), b, c and write b in place.  No stencil math.
//   MODE 0: batch (issue DEPTH rows of loads, wait all, combine, store)
//   MODE 1: software pipeline with counted waits (loads for row r+DEPTH issued while row r is consumed)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int DEPTH, int MODE, int NSTORE, int AHEAD>
__global__ __launch_bounds__(256) void k_rw(const float* __restrict__ a, float* __restrict__ b, const float* __restrict__ c,
                                            int pitch, int rows, int xchunk, int nzblk, int nblk, int nper)
{
    const int lane = threadIdx.x & 63;
    const int w = threadIdx.x >> 6;
    const int bid = blockIdx.x;
    const int L = (bid & 7) * nper + (bid >> 3);
    if (L >= nblk) return;
    const int zb = L % nzblk, xb = L / nzblk;
    const int strip = zb * 4 + w;
    const int z0 = strip * 256 + lane * 4;
    if (z0 >= pitch) return;
    const int xa = xb * xchunk, xe = min(xa + xchunk, rows);
    if (MODE == 0) {
        for (int r = xa; r < xe; r += DEPTH) {
            float4 va[DEPTH], vb[DEPTH], vc[DEPTH];
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                const size_t o = (size_t)min(r + d, xe - 1) * pitch + z0;
                va[d] = *reinterpret_cast<const float4*>(a + o);
                vb[d] = *reinterpret_cast<const float4*>(b + o);
                vc[d] = *reinterpret_cast<const float4*>(c + o);
            }
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                float4 o4 = make_float4(va[d].x + vb[d].x * vc[d].x, va[d].y + vb[d].y * vc[d].y, va[d].z + vb[d].z * vc[d].z, va[d].w + vb[d].w * vc[d].w);
                if (NSTORE && r + d < xe) *reinterpret_cast<float4*>(b + (size_t)(r + d) * pitch + z0) = o4;
                if (!NSTORE && o4.x == 123.456f) b[0] = o4.y;
            }
        }
    } else {
        float4 qa[DEPTH], qb[DEPTH], qc[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const size_t o = (size_t)min(xa + d, xe - 1) * pitch + z0;
            qa[d] = *reinterpret_cast<const float4*>(a + (size_t)min(xa + d + AHEAD, rows - 1) * pitch + z0);
            qb[d] = *reinterpret_cast<const float4*>(b + o);
            qc[d] = *reinterpret_cast<const float4*>(c + o);
            __builtin_amdgcn_sched_barrier(0);
        }
        float4 extra = make_float4(0, 0, 0, 0);
        if (AHEAD) {   // ring prologue: AHEAD + 4 more rows of a, consumed immediately
#pragma unroll
            for (int k = -4; k < AHEAD; ++k) {
                const float4 t = *reinterpret_cast<const float4*>(a + (size_t)max(xa + k, 0) * pitch + z0);
                extra.x += t.x; extra.y += t.y; extra.z += t.z; extra.w += t.w;
            }
        }
        for (int rb = xa; rb < xe; rb += DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                const int r = rb + d;
                float4 o4 = make_float4(qa[d].x + qb[d].x * qc[d].x + extra.x, qa[d].y + qb[d].y * qc[d].y + extra.y, qa[d].z + qb[d].z * qc[d].z + extra.z, qa[d].w + qb[d].w * qc[d].w + extra.w);
                if (NSTORE) *reinterpret_cast<float4*>(b + (size_t)min(r, xe - 1) * pitch + z0) = o4;
                if (!NSTORE && o4.x == 123.456f) b[0] = o4.y;
                const size_t o = (size_t)min(r + DEPTH, xe - 1) * pitch + z0;
                qa[d] = *reinterpret_cast<const float4*>(a + (size_t)min(r + DEPTH + AHEAD, min(xe + 3, rows - 1)) * pitch + z0);
                qb[d] = *reinterpret_cast<const float4*>(b + o);
                qc[d] = *reinterpret_cast<const float4*>(c + o);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
}

template <int DEPTH, int MODE, int NSTORE, int AHEAD>
static void run(const float* a, float* b, const float* c, int n, int xchunk)
{
    const int nstrips = n / 256, nzblk = nstrips / 4;
    const int nxblk = (n + xchunk - 1) / xchunk;
    const int nblk = nzblk * nxblk, nper = (nblk + 7) / 8;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep)
        hipLaunchKernelGGL((k_rw<DEPTH, MODE, NSTORE, AHEAD>), dim3(8 * nper), dim3(256), 0, 0, a, b, c, n, n, xchunk, nzblk, nblk, nper);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    const int reps = 10;
    for (int rep = 0; rep < reps; ++rep)
        hipLaunchKernelGGL((k_rw<DEPTH, MODE, NSTORE, AHEAD>), dim3(8 * nper), dim3(256), 0, 0, a, b, c, n, n, xchunk, nzblk, nblk, nper);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = (double)n * n * 4 * (3 + NSTORE);
    printf("n=%d mode=%s depth=%d store=%d ahead=%d xchunk=%3d waves=%6d: %8.1f us  %6.2f TB/s  (%.1f Gpt/s)\n", n, MODE ? "pipe " : "batch", DEPTH, NSTORE, AHEAD,
           xchunk, nstrips * nxblk, ms / reps * 1e3, bytes / (ms / reps * 1e-3) / 1e12, (double)n * n / (ms / reps * 1e-3) / 1e9);
}

int main(int argc, char** argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 8192;
    float *a, *b, *c;
    CK(hipMalloc(&a, (size_t)n * n * 4)); CK(hipMalloc(&b, (size_t)n * n * 4)); CK(hipMalloc(&c, (size_t)n * n * 4));
    if (argc > 2) {   // random bit patterns (finite floats) instead of zeros
        std::vector<unsigned> h((size_t)n * n);
        unsigned x = 12345u;
        for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (x & 0x807fffffu) | 0x3f000000u; }
        CK(hipMemcpy(a, h.data(), h.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(b, h.data(), h.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(c, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    } else {
        CK(hipMemset(a, 0, (size_t)n * n * 4)); CK(hipMemset(b, 0, (size_t)n * n * 4)); CK(hipMemset(c, 0, (size_t)n * n * 4));
    }
    for (int xchunk : {16, 64, 128}) {
        run<4, 0, 1, 0>(a, b, c, n, xchunk);
        run<2, 1, 1, 0>(a, b, c, n, xchunk);
        run<2, 1, 1, 4>(a, b, c, n, xchunk);
        run<3, 1, 1, 4>(a, b, c, n, xchunk);
    }
    return 0;
}
