// Cost of the GEMM epilogue patterns on a 6400x1024 fp32 C (tools/ubench_src): 400 tiles x 128x128, 4 waves per tile, 64 values per
// lane; (a) plain stores, (b) atomicAdd, (c) read-add-write, each tile touched by `contrib` workgroups at once.
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int MODE>
__global__ __launch_bounds__(256) void k(float* C, int ldc, int gx) {
    const int tile = blockIdx.x % (gx * 50), lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m0 = (tile / gx) * 128 + (wave >> 1) * 64, n0 = (tile % gx) * 128 + (wave & 1) * 64;
    const int lr = lane & 31, lk = lane >> 5;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float* c = C + (long)(m0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk) * ldc + n0 + j * 32 + lr;
                const float v = r * 0.5f + lane;
                if (MODE == 0) *c = v; else if (MODE == 1) atomicAdd(c, v); else *c += v;
            }
}
int main() {
    float* C; (void)hipMalloc(&C, 6400L * 1024 * 4); (void)hipMemset(C, 0, 6400L * 1024 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const char* names[3] = {"plain store", "atomicAdd", "read-add-write"};
    for (int contrib = 1; contrib <= 2; ++contrib)
        for (int mode = 0; mode < 3; ++mode) {
            float ms = 0;
            for (int rep = 0; rep < 3; ++rep) {
                (void)hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(400 * contrib), dim3(256), 0, 0, C, 1024, 8);
                else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(400 * contrib), dim3(256), 0, 0, C, 1024, 8);
                else hipLaunchKernelGGL(k<2>, dim3(400 * contrib), dim3(256), 0, 0, C, 1024, 8);
                (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                (void)hipEventElapsedTime(&ms, e0, e1);
            }
            printf("%-16s %d contributor(s) per tile: %7.1f us\n", names[mode], contrib, ms * 1e3);
        }
    return 0;
}
