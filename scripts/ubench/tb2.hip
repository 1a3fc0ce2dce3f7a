// Traffic skeleton of a 2-steps-per-launch (temporal blocking) kernel (development tool):
// per wave tile of 62 output cells (248 columns) x xchunk rows: reads a on xchunk+16 rows, b and c on
// xchunk+8 rows (64 cells wide), writes d and e on xchunk rows (62 cells).  No stencil arithmetic.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));

template <int NT>
__global__ __launch_bounds__(256) void k(const v4f* __restrict__ a, const v4f* __restrict__ b, const v4f* __restrict__ c, v4f* d, v4f* e,
                                         int pitch4, int rows, int xchunk, int nstrip, int nblk, int nper)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int bid = blockIdx.x;
    const int L = (bid & 7) * nper + (bid >> 3);
    if (L >= nblk) return;
    const int nzblk = (nstrip + 3) / 4;
    const int zb = L % nzblk, xb = L / nzblk;
    const int strip = zb * 4 + w;
    if (strip >= nstrip) return;
    const int c4 = min(max(strip * 62 - 1 + lane, 0), pitch4 - 1);   // cell index (float4 units), clamped
    const bool own = lane >= 1 && lane <= 62 && (strip * 62 - 1 + lane) < pitch4;
    const int xa = xb * xchunk, xe = min(xa + xchunk, rows);
    v4f acc = {0, 0, 0, 0};
    // prologue-ish: 16 extra rows of a, 8 extra rows of b and c
    for (int r = xa - 8; r < xa + 8; ++r) { const int rr = min(max(r, 0), rows - 1); acc += a[(size_t)rr * pitch4 + c4]; }
    for (int r = xa - 4; r < xa + 4; ++r) { const int rr = min(max(r, 0), rows - 1); acc += b[(size_t)rr * pitch4 + c4] + c[(size_t)rr * pitch4 + c4]; }
    constexpr int D = 2;
    v4f qa[D], qb[D], qc[D];
#pragma unroll
    for (int t = 0; t < D; ++t) {
        const size_t o = (size_t)min(xa + 8 + t, rows - 1) * pitch4 + c4, o2 = (size_t)min(xa + 4 + t, rows - 1) * pitch4 + c4;
        qa[t] = a[o];
        qb[t] = NT ? __builtin_nontemporal_load(b + o2) : b[o2];
        qc[t] = NT ? __builtin_nontemporal_load(c + o2) : c[o2];
        __builtin_amdgcn_sched_barrier(0);
    }
    for (int rb = xa; rb < xe; rb += D) {
#pragma unroll
        for (int t = 0; t < D; ++t) {
            const int r = min(rb + t, xe - 1);
            const v4f o1 = qa[t] + qb[t] * qc[t] + acc, o2v = qa[t] - qb[t] * qc[t];
            if (own) {
                if (NT) { __builtin_nontemporal_store(o1, d + (size_t)r * pitch4 + c4); __builtin_nontemporal_store(o2v, e + (size_t)r * pitch4 + c4); }
                else { d[(size_t)r * pitch4 + c4] = o1; e[(size_t)r * pitch4 + c4] = o2v; }
            }
            const size_t o = (size_t)min(rb + t + D + 8, rows - 1) * pitch4 + c4, oo = (size_t)min(rb + t + D + 4, rows - 1) * pitch4 + c4;
            qa[t] = a[o];
            qb[t] = NT ? __builtin_nontemporal_load(b + oo) : b[oo];
            qc[t] = NT ? __builtin_nontemporal_load(c + oo) : c[oo];
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

template <int NT>
static void run(const v4f* a, const v4f* b, const v4f* c, v4f* d, v4f* e, int n, int xchunk)
{
    const int pitch4 = n / 4, nstrip = (pitch4 + 61) / 62, nzblk = (nstrip + 3) / 4;
    const int nxblk = (n + xchunk - 1) / xchunk, nblk = nzblk * nxblk, nper = (nblk + 7) / 8;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k<NT>, dim3(8 * nper), dim3(256), 0, 0, a, b, c, d, e, pitch4, n, xchunk, nstrip, nblk, nper);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    const int reps = 10;
    for (int rep = 0; rep < reps; ++rep) hipLaunchKernelGGL(k<NT>, dim3(8 * nper), dim3(256), 0, 0, a, b, c, d, e, pitch4, n, xchunk, nstrip, nblk, nper);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms / reps * 1e3;
    printf("n=%d nt=%d xchunk=%3d: %8.1f us per launch = %6.1f us/step  -> %6.1f Gpt/s/step-equivalent  (%.2f TB/s of 20 B/pt)\n", n, NT, xchunk, us, us / 2,
           2.0 * n * n / (us * 1e-6) / 1e9, 20.0 * n * n / (us * 1e-6) / 1e12);
}

int main(int argc, char** argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 8192;
    v4f *a, *b, *c, *d, *ee;
    const size_t bytes = (size_t)n * n * 4;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&c, bytes)); CK(hipMalloc(&d, bytes)); CK(hipMalloc(&ee, bytes));
    CK(hipMemset(a, 0, bytes)); CK(hipMemset(b, 0, bytes)); CK(hipMemset(c, 0, bytes));
    for (int xchunk : {16, 32, 64, 128}) { run<0>(a, b, c, d, ee, n, xchunk); run<1>(a, b, c, d, ee, n, xchunk); }
    return 0;
}
