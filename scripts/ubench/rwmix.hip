// Read/write mix microbenchmark (development tool): what limits a 3-read + 1-write stream?
//   variant 0: b read and written in place      (the leap-frog update as implemented)
//   variant 1: write goes to a 4th array d      (three-buffer rotation)
//   variant 2: in place, nontemporal store
//   variant 3: out of place, nontemporal store
//   variant 4: in place, nontemporal loads + store
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));

template <int VAR>
__global__ __launch_bounds__(256) void k(const v4f* __restrict__ a, v4f* b, const v4f* __restrict__ c, v4f* d, int pitch4, int rows, int xchunk, int nzblk, int nblk, int nper)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int bid = blockIdx.x;
    const int L = (bid & 7) * nper + (bid >> 3);
    if (L >= nblk) return;
    const int zb = L % nzblk, xb = L / nzblk;
    const int z4 = (zb * 4 + w) * 64 + lane;
    if (z4 >= pitch4) return;
    const int xa = xb * xchunk, xe = min(xa + xchunk, rows);
    constexpr int D = 2;
    v4f qa[D], qb[D], qc[D];
#pragma unroll
    for (int t = 0; t < D; ++t) {
        const size_t o = (size_t)min(xa + t, xe - 1) * pitch4 + z4;
        if (VAR == 4) { qa[t] = __builtin_nontemporal_load(a + o); qb[t] = __builtin_nontemporal_load(b + o); qc[t] = __builtin_nontemporal_load(c + o); }
        else { qa[t] = a[o]; qb[t] = b[o]; qc[t] = c[o]; }
        __builtin_amdgcn_sched_barrier(0);
    }
    for (int rb = xa; rb < xe; rb += D) {
#pragma unroll
        for (int t = 0; t < D; ++t) {
            const int r = min(rb + t, xe - 1);
            const v4f o4 = qa[t] + qb[t] * qc[t];
            v4f* dst = ((VAR == 1 || VAR == 3) ? d : b) + (size_t)r * pitch4 + z4;
            if (VAR >= 2) __builtin_nontemporal_store(o4, dst); else *dst = o4;
            const size_t o = (size_t)min(rb + t + D, xe - 1) * pitch4 + z4;
            if (VAR == 4) { qa[t] = __builtin_nontemporal_load(a + o); qb[t] = __builtin_nontemporal_load(b + o); qc[t] = __builtin_nontemporal_load(c + o); }
            else { qa[t] = a[o]; qb[t] = b[o]; qc[t] = c[o]; }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

template <int VAR>
static void run(const v4f* a, v4f* b, const v4f* c, v4f* d, int n, int xchunk)
{
    const int nzblk = n / 1024, nxblk = (n + xchunk - 1) / xchunk, nblk = nzblk * nxblk, nper = (nblk + 7) / 8;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k<VAR>, dim3(8 * nper), dim3(256), 0, 0, a, b, c, d, n / 4, n, xchunk, nzblk, nblk, nper);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    const int reps = 10;
    for (int rep = 0; rep < reps; ++rep) hipLaunchKernelGGL(k<VAR>, dim3(8 * nper), dim3(256), 0, 0, a, b, c, d, n / 4, n, xchunk, nzblk, nblk, nper);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("n=%d variant=%d xchunk=%3d: %8.1f us  %6.2f TB/s  (%.1f Gpt/s)\n", n, VAR, xchunk, ms / reps * 1e3, (double)n * n * 16 / (ms / reps * 1e-3) / 1e12,
           (double)n * n / (ms / reps * 1e-3) / 1e9);
}

int main(int argc, char** argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 8192;
    v4f *a, *b, *c, *d;
    const size_t bytes = (size_t)n * n * 4;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&c, bytes)); CK(hipMalloc(&d, bytes));
    CK(hipMemset(a, 0, bytes)); CK(hipMemset(b, 0, bytes)); CK(hipMemset(c, 0, bytes)); CK(hipMemset(d, 0, bytes));
    for (int xchunk : {16, 64}) {
        run<0>(a, b, c, d, n, xchunk); run<1>(a, b, c, d, n, xchunk); run<2>(a, b, c, d, n, xchunk);
        run<3>(a, b, c, d, n, xchunk); run<4>(a, b, c, d, n, xchunk);
    }
    return 0;
}
