// Read-bandwidth microbenchmarks (development tool): which load path sustains what on gfx950?
//   k_vgpr  : global_load_dwordx4 into registers, DEPTH loads in flight per wave
//   k_lds   : global_load_lds_dwordx4 (LDS-DMA), DEPTH KiB in flight per wave, data consumed from LDS
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int DEPTH>
__global__ __launch_bounds__(256) void k_vgpr(const float4* __restrict__ src, float* __restrict__ out, size_t n4_per_wave, int iters)
{
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const float4* p = src + wave * n4_per_wave + lane;
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        float4 v[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) v[d] = p[(size_t)(it * DEPTH + d) * 64];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) acc += v[d].x + v[d].y + v[d].z + v[d].w;
    }
    if (acc == 123.4567f) out[wave] = acc;
}

template <int DEPTH>
__global__ __launch_bounds__(256) void k_lds(const float4* __restrict__ src, float* __restrict__ out, size_t n4_per_wave, int iters)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int w = threadIdx.x >> 6;
    const size_t wave = (size_t)blockIdx.x * 4 + w;
    const float4* p = src + wave * n4_per_wave + lane;
    char* my = smem + (size_t)w * DEPTH * 1024;
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d)
            __builtin_amdgcn_global_load_lds((const void*)(p + (size_t)(it * DEPTH + d) * 64),
                                             (__attribute__((address_space(3))) void*)(my + d * 1024), 16, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const float4 v = *reinterpret_cast<const float4*>(my + d * 1024 + lane * 16);
            acc += v.x + v.y + v.z + v.w;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    if (acc == 123.4567f) out[wave] = acc;
}

template <class K>
static void run(const char* name, K kern, int depth, size_t lds, const float4* src, float* out, size_t total4, int blocks)
{
    const size_t waves = (size_t)blocks * 4;
    const size_t n4_per_wave = total4 / waves;
    const int iters = (int)(n4_per_wave / 64 / depth);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, 0, src, out, n4_per_wave, iters);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    const int reps = 10;
    for (int rep = 0; rep < reps; ++rep) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, 0, src, out, n4_per_wave, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = (double)waves * iters * depth * 1024;
    printf("%-8s depth=%2d blocks=%5d: %8.1f us  %6.2f TB/s\n", name, depth, blocks, ms / reps * 1e3, bytes / (ms / reps * 1e-3) / 1e12);
}

int main()
{
    const size_t bytes = (size_t)2 << 30;   // 2 GiB: well past the 256 MiB Infinity Cache
    float4* src; float* out;
    CK(hipMalloc(&src, bytes)); CK(hipMalloc(&out, 1 << 20));
    CK(hipMemset(src, 1, bytes));
    const size_t total4 = bytes / 16;
    for (int blocks : {1024, 2048, 4096}) {
        run("vgpr", k_vgpr<4>, 4, 0, src, out, total4, blocks);
        run("vgpr", k_vgpr<8>, 8, 0, src, out, total4, blocks);
        run("vgpr", k_vgpr<16>, 16, 0, src, out, total4, blocks);
        run("lds", k_lds<4>, 4, 4 * 4 * 1024, src, out, total4, blocks);
        run("lds", k_lds<8>, 8, 4 * 8 * 1024, src, out, total4, blocks);
    }
    run("lds", k_lds<16>, 16, 4 * 16 * 1024, src, out, total4, 512);
    run("lds", k_lds<16>, 16, 4 * 16 * 1024, src, out, total4, 1024);
    return 0;
}
