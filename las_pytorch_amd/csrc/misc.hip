// Small memory-bound helpers around the hot path (bias gradients, masks, label conversion).
#include "las_common.h"
#include "las_kernels.h"

namespace las {

thread_local char g_err[512] = {0};

__global__ void colsum_kernel(const float* __restrict__ src, long ld, int rows, int cols, float* __restrict__ dst,
                              int rows_per_block) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= cols) return;
    const int r0 = blockIdx.y * rows_per_block, r1 = min(rows, r0 + rows_per_block);
    float a0 = 0.f, a1 = 0.f;
    int r = r0;
    for (; r + 1 < r1; r += 2) { a0 += src[(long)r * ld + c]; a1 += src[(long)(r + 1) * ld + c]; }
    if (r < r1) a0 += src[(long)r * ld + c];
    atomicAdd(dst + c, a0 + a1);
}

int colsum(const float* src, long ld, int rows, int cols, float* dst, int accumulate, hipStream_t stream) {
    if (!accumulate) LAS_HIP_CHECK(hipMemsetAsync(dst, 0, sizeof(float) * cols, stream));
    const int rpb = 128;
    dim3 grid(cdiv(cols, 256), cdiv(rows, rpb)), block(256);
    hipLaunchKernelGGL(colsum_kernel, grid, block, 0, stream, src, ld, rows, cols, dst, rpb);
    LAS_LAUNCH_CHECK();
    return LAS_OK;
}

__global__ void relu_mask_kernel(float* __restrict__ grad, const float* __restrict__ act, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && !(act[i] > 0.f)) grad[i] = 0.f;
}
int relu_mask_inplace(float* grad, const float* act, long n, hipStream_t stream) {
    hipLaunchKernelGGL(relu_mask_kernel, dim3(cdiv(n, 256)), dim3(256), 0, stream, grad, act, n);
    LAS_LAUNCH_CHECK();
    return LAS_OK;
}

__global__ void add_kernel(float* __restrict__ dst, const float* __restrict__ src, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] += src[i];
}
int add_inplace(float* dst, const float* src, long n, hipStream_t stream) {
    hipLaunchKernelGGL(add_kernel, dim3(cdiv(n, 256)), dim3(256), 0, stream, dst, src, n);
    LAS_LAUNCH_CHECK();
    return LAS_OK;
}

// dst[r][c] (+)= src[r][c] for a rows x cols window with independent leading dimensions
__global__ void copy2d_kernel(const float* __restrict__ src, long lds, float* __restrict__ dst, long ldd, int rows, int cols,
                              int accumulate) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)rows * cols) return;
    const int r = i / cols, c = i % cols;
    const float v = src[(long)r * lds + c];
    if (accumulate) dst[(long)r * ldd + c] += v; else dst[(long)r * ldd + c] = v;
}
int copy2d(const float* src, long lds, float* dst, long ldd, int rows, int cols, int accumulate, hipStream_t stream) {
    const long n = (long)rows * cols;
    hipLaunchKernelGGL(copy2d_kernel, dim3(cdiv(n, 256)), dim3(256), 0, stream, src, lds, dst, ldd, rows, cols, accumulate);
    LAS_LAUNCH_CHECK();
    return LAS_OK;
}

// Teacher-forcing inputs: y_all[0] = onehot(<sos>=0) (reference las_model.py:193-195),
// y_all[1+s][b][:] = float(ground_truth[b][s][:]) (las_model.py:216-217; any label rows, incl. all-zero padding)
__global__ void labels_to_y_kernel(const long long* __restrict__ labels, float* __restrict__ y_all, int B, int U, int V,
                                   int Vp, int u_lab) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long n = (long)(U + 1) * B * Vp;
    if (i >= n) return;
    const int v = i % Vp;                       // rows are padded to Vp (multiple of 4) with zeros
    const int b = (i / Vp) % B;
    const int s = i / ((long)Vp * B);
    float val = 0.f;
    if (v < V) {
        if (s == 0) val = (v == 0) ? 1.f : 0.f;
        else if (labels && s - 1 < u_lab) val = (float)labels[((long)b * u_lab + (s - 1)) * V + v];
    }
    y_all[i] = val;
}
int labels_to_y(const long long* labels, float* y_all, int B, int U, int V, int Vp, int u_lab, hipStream_t stream) {
    const long n = (long)(U + 1) * B * Vp;
    hipLaunchKernelGGL(labels_to_y_kernel, dim3(cdiv(n, 256)), dim3(256), 0, stream, labels, y_all, B, U, V, Vp, u_lab);
    LAS_LAUNCH_CHECK();
    return LAS_OK;
}

// In-place log-softmax over rows of width V (teacher-forced decode: the character distribution of ALL steps is one
// GEMM after the loop instead of a per-step phase; reference las_model.py:182)
__global__ void log_softmax_rows_kernel(float* __restrict__ x, long rows, int V) {
    const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    float* p = x + r * V;
    float m = -INFINITY;
    for (int v = 0; v < V; ++v) m = fmaxf(m, p[v]);
    float s = 0.f;
    for (int v = 0; v < V; ++v) s += expf(p[v] - m);
    const float lse = m + logf(s);
    for (int v = 0; v < V; ++v) p[v] -= lse;
}
int log_softmax_rows(float* x, long rows, int V, hipStream_t stream) {
    hipLaunchKernelGGL(log_softmax_rows_kernel, dim3(cdiv(rows, 128)), dim3(128), 0, stream, x, rows, V);
    LAS_LAUNCH_CHECK();
    return LAS_OK;
}
// dz = dlogp - exp(logp) * sum(dlogp) per row
__global__ void log_softmax_bwd_rows_kernel(const float* __restrict__ dlogp, const float* __restrict__ logp,
                                            float* __restrict__ dz, long rows, int V) {
    const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    float s = 0.f;
    for (int v = 0; v < V; ++v) s += dlogp[r * V + v];
    for (int v = 0; v < V; ++v) dz[r * V + v] = dlogp[r * V + v] - expf(logp[r * V + v]) * s;
}
int log_softmax_bwd_rows(const float* dlogp, const float* logp, float* dz, long rows, int V, hipStream_t stream) {
    hipLaunchKernelGGL(log_softmax_bwd_rows_kernel, dim3(cdiv(rows, 128)), dim3(128), 0, stream, dlogp, logp, dz, rows, V);
    LAS_LAUNCH_CHECK();
    return LAS_OK;
}

}  // namespace las
