// Strided-read microbenchmark (development tool): a wave reads a 1 KiB-wide column strip of a
// [rows][pitch] fp32 grid, marching down rows (the access pattern of the register-ring stencil),
// DEPTH rows in flight.  Variants: waves of a block side by side in z (wz=4) or stacked in x (wz=1);
// WIDTH = KiB per wave per row (1, 2, 4).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int DEPTH, int WIDTH>
__global__ __launch_bounds__(256) void k_strided(const float* __restrict__ src, float* __restrict__ out, int pitch, int rows,
                                                 int xchunk, int wz, int nzblk, int nblk, int nper, int remap)
{
    const int lane = threadIdx.x & 63;
    const int w = threadIdx.x >> 6;
    const int bid = blockIdx.x;
    const int L = remap ? (bid & 7) * nper + (bid >> 3) : bid;
    if (L >= nblk) return;
    const int zb = L % nzblk, xb = L / nzblk;
    const int strip = zb * wz + (w & (wz - 1));
    const int chunk = xb * (4 / wz) + (w / wz);
    const int z0 = strip * 256 * WIDTH + lane * 4;
    if (z0 >= pitch) return;
    const int xa = chunk * xchunk, xe = min(xa + xchunk, rows);
    float acc = 0.f;
    for (int r = xa; r < xe; r += DEPTH) {
        float4 v[DEPTH][WIDTH];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d)
#pragma unroll
            for (int k = 0; k < WIDTH; ++k)
                v[d][k] = *reinterpret_cast<const float4*>(src + (size_t)min(r + d, xe - 1) * pitch + z0 + k * 256);
#pragma unroll
        for (int d = 0; d < DEPTH; ++d)
#pragma unroll
            for (int k = 0; k < WIDTH; ++k) acc += v[d][k].x + v[d][k].y + v[d][k].z + v[d][k].w;
    }
    if (acc == 123.4567f) out[bid] = acc;
}

template <int DEPTH, int WIDTH>
static void run(const float* src, float* out, int n, int xchunk, int wz, int remap)
{
    const int nstrips = n / (256 * WIDTH);
    const int nzblk = (nstrips + wz - 1) / wz;
    const int chunks = (n + xchunk - 1) / xchunk;
    const int wpx = 4 / wz;
    const int nxblk = (chunks + wpx - 1) / wpx;
    const int nblk = nzblk * nxblk, nper = (nblk + 7) / 8;
    const int grid = remap ? 8 * nper : nblk;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep)
        hipLaunchKernelGGL((k_strided<DEPTH, WIDTH>), dim3(grid), dim3(256), 0, 0, src, out, n, n, xchunk, wz, nzblk, nblk, nper, remap);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    const int reps = 10;
    for (int rep = 0; rep < reps; ++rep)
        hipLaunchKernelGGL((k_strided<DEPTH, WIDTH>), dim3(grid), dim3(256), 0, 0, src, out, n, n, xchunk, wz, nzblk, nblk, nper, remap);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("n=%d depth=%d width=%dKiB xchunk=%3d wz=%d remap=%d waves=%6d: %8.1f us  %6.2f TB/s\n", n, DEPTH, WIDTH, xchunk, wz, remap,
           nstrips * chunks, ms / reps * 1e3, (double)n * n * 4 / (ms / reps * 1e-3) / 1e12);
}

int main(int argc, char** argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 16384;   // 16384^2 fp32 = 1 GiB
    float* src; float* out;
    CK(hipMalloc(&src, (size_t)n * n * 4)); CK(hipMalloc(&out, 1 << 22));
    CK(hipMemset(src, 1, (size_t)n * n * 4));
    for (int xchunk : {16, 64, 256}) {
        for (int wz : {4, 1}) {
            run<4, 1>(src, out, n, xchunk, wz, 1);
            run<8, 1>(src, out, n, xchunk, wz, 1);
        }
        run<4, 1>(src, out, n, xchunk, 4, 0);
        run<4, 2>(src, out, n, xchunk, 4, 1);
        run<2, 4>(src, out, n, xchunk, 4, 1);
        run<4, 4>(src, out, n, xchunk, 1, 1);
    }
    return 0;
}
