// VALU fp32 rate on gfx950: v_fma_f32 vs v_pk_fma_f32, per-CU flop/clk (tools/ubench_src; build: hipcc --offload-arch=gfx950 -O3)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void k_fma(float* out, int iters, float a, float b) {
    float x[16];
    for (int i = 0; i < 16; ++i) x[i] = threadIdx.x * 0.001f + i;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < 16; ++i) x[i] = __builtin_fmaf(x[i], a, b);
    float s = 0; for (int i = 0; i < 16; ++i) s += x[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_pk(float* out, int iters, float a, float b) {
    f2 x[16];
    for (int i = 0; i < 16; ++i) x[i] = f2{threadIdx.x * 0.001f + i, threadIdx.x * 0.002f + i};
    const f2 av = {a, a}, bv = {b, b};
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(x[i]) : "v"(x[i]), "v"(av), "v"(bv));
    float s = 0; for (int i = 0; i < 16; ++i) s += x[i].x + x[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    float* out; hipMalloc(&out, 256 * 2048 * 4 * sizeof(float));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4096, blocks = 256 * 8;
    for (int which = 0; which < 2; ++which) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (which == 0) hipLaunchKernelGGL(k_fma, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
            else hipLaunchKernelGGL(k_pk, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double fl = (double)blocks * 256 * iters * 16 * 2 * (which ? 2 : 1);
            if (rep) printf("%s: %.3f ms  %.1f TFLOP/s\n", which ? "v_pk_fma_f32" : "v_fma_f32   ", ms, fl / ms / 1e9);
        }
    }
    return 0;
}
