// fp32 MFMA rate with the GEMM's exact inner structure (tools/ubench_src): 4 waves per workgroup, 2x2 32x32x2 accumulators per wave,
// fragments (a) from registers, (b) from LDS with the GEMM's ds_read pattern, no barrier, (c) LDS + one __syncthreads per 8 k-steps.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f16v __attribute__((ext_vector_type(16)));
template <int MODE>
__global__ __launch_bounds__(256, 2) void k(float* out, int iters) {
    __shared__ float As[2][16][132], Bs[2][16][132];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lr = lane & 31, lk = lane >> 5;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    for (int i = threadIdx.x; i < 2 * 16 * 132; i += 256) { (&As[0][0][0])[i] = i * 0.001f; (&Bs[0][0][0])[i] = i * 0.002f; }
    __syncthreads();
    f16v acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float a[2] = {lane * 0.5f, lane * 0.25f}, b[2] = {lane * 0.125f, lane * 1.5f};
    for (int it = 0; it < iters; ++it) {
        const int cur = it & 1;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            if (MODE >= 1) {
                a[0] = As[cur][kk * 2 + lk][wm + lr]; a[1] = As[cur][kk * 2 + lk][wm + 32 + lr];
                b[0] = Bs[cur][kk * 2 + lk][wn + lr]; b[1] = Bs[cur][kk * 2 + lk][wn + 32 + lr];
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (MODE == 2) __syncthreads();
    }
    float s = 0;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE> void run(const char* name, int blocks) {
    float* out; (void)hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 2000;
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        const double fl = (double)blocks * 4 * iters * 8 * 4 * 4096.0;
        if (rep) printf("%-44s blocks=%4d: %8.3f ms  %6.1f TFLOP/s\n", name, blocks, ms, fl / ms / 1e9);
    }
    (void)hipFree(out);
}
int main() {
    for (int blocks : {256, 512, 1024}) {
        run<0>("register operands", blocks);
        run<1>("LDS fragments, no barrier", blocks);
        run<2>("LDS fragments + barrier per 8 k-steps", blocks);
    }
    return 0;
}
