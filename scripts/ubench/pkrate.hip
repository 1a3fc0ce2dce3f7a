// VALU issue-rate microbenchmark: scalar vs packed fp32 mul/add (development tool; prints wave-instructions per ns per SIMD-equivalent)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float c0, float c1)
{
    v2f a[8];
    for (int i = 0; i < 8; i++) a[i] = v2f{(float)threadIdx.x + i, (float)i};
    const v2f m = {c0, c0}, ad = {c1, c1};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (MODE == 0) { a[i].x = a[i].x * c0; a[i].x = a[i].x + c1; }                                  // 1 v_mul_f32 + 1 v_add_f32
                if (MODE == 1) { a[i] = a[i] * m; a[i] = a[i] + ad; }                                           // 1 v_pk_mul_f32 + 1 v_pk_add_f32
                if (MODE == 2) { a[i].x = __builtin_fmaf(a[i].x, c0, c1); a[i].y = __builtin_fmaf(a[i].y, c0, c1); }   // 2 v_fma_f32
                if (MODE == 3) { a[i] = __builtin_elementwise_fma(a[i], m, ad); }                               // 1 v_pk_fma_f32
                if (MODE == 4) { double d = (double)a[i].x; d = d + d; a[i].x = (float)d; }                      // cvt, add_f64, cvt
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 8; i++) s += a[i].x + a[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
int run(const char* name, int per_iter)
{
    float* out;
    const int blocks = 256 * 8, iters = 2000;
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
    const double winstr = (double)blocks * 4 * iters * 32 * per_iter;   // wave-instructions
    printf("%-28s %8.3f ms  %7.2f wave-instr/ns total = %5.2f cycles per wave-instr per SIMD at 2.4 GHz\n", name, ms, winstr / (ms * 1e6),
           1024 * 2.4 / (winstr / (ms * 1e6)));
    hipFree(out);
    return 0;
}
int main()
{
    run<0>("v_mul_f32 + v_add_f32", 2);
    run<1>("v_pk_mul_f32 + v_pk_add_f32", 2);
    run<2>("2 x v_fma_f32", 2);
    run<3>("v_pk_fma_f32", 1);
    run<4>("cvt + add_f64 + cvt", 3);
    return 0;
}
