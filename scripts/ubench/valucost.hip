// Issue cost of the VALU instructions the pipeline kernel's march step is made of, at the kernel's occupancy (5 waves per SIMD): independent
// streams of one instruction kind over 16 registers per lane, 20 waves per CU (development tool).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float c0, float c1)
{
    float a[16];
    double d[8];
    for (int i = 0; i < 16; i++) a[i] = (float)threadIdx.x * 0.001f + i;
    for (int i = 0; i < 8; i++) d[i] = (double)a[i];
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
#pragma unroll
            for (int i = 0; i < 16; i++) {
                if (MODE == 0) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c0));
                if (MODE == 1) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c1));
                if (MODE == 2) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c0), "v"(c1));
                if (MODE == 3 && (i & 1) == 0) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(*(v2f*)&a[i]) : "v"(v2f{c0, c0}));
                if (MODE == 4 && (i & 1) == 0) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(*(v2f*)&a[i]) : "v"(v2f{c1, c1}));
                if (MODE == 5 && i < 8) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"((double)c0), "v"((double)c1));
                if (MODE == 6 && i < 8) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"((double)c1));
                if (MODE == 7 && i < 8) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(a[i]));
                if (MODE == 8 && i < 8) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(a[i]) : "v"(d[i]));
                if (MODE == 9) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(c0));
                if (MODE == 10) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(a[(i + 1) & 15]));
                if (MODE == 11) asm volatile("ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)" : "+v"(a[i]) : "v"((int)(threadIdx.x & 63) * 4));
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 16; i++) s += a[i];
    for (int i = 0; i < 8; i++) s += (float)d[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
int run(const char* name, int per_iter)
{
    float* out;
    const int blocks = 256 * 5, iters = 1500;
    CHK(hipMalloc(&out, blocks * 256 * sizeof(float)));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 10, 1.0001f, 0.5f);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
    CHK(hipEventRecord(e1));
    CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    const double winstr = (double)blocks * 4 * iters * 4 * per_iter;   // wave-instructions
    printf("%-16s %8.3f ms  %6.2f cycles per wave-instruction per SIMD (5 waves per SIMD, 2.4 GHz assumed)\n", name, ms, 1024 * 2.4e6 * ms / winstr);
    hipFree(out);
    return 0;
}
int main()
{
    run<0>("v_mul_f32", 16); run<1>("v_add_f32", 16); run<2>("v_fma_f32", 16); run<3>("v_pk_mul_f32", 8); run<4>("v_pk_add_f32", 8);
    run<5>("v_fma_f64", 8); run<6>("v_add_f64", 8); run<7>("v_cvt_f64_f32", 8); run<8>("v_cvt_f32_f64", 8); run<9>("v_cndmask_b32", 16);
    run<10>("v_mov_b32", 16); run<11>("ds_bpermute_b32", 16);
    return 0;
}
