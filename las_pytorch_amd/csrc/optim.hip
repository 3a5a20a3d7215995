// Global-norm gradient clip + Adam step over the FLAT gradient buffer in two launches (las_clip_adam, include/las_hip.h).
//
// Replaces torch.nn.utils.clip_grad_norm_(las_model.parameters(), 1) + optimizer.step() of the reference's batch_iterator
// (solver/solver.py:96-97; Adam from train.py:82) — ~12 elementwise ATen launches per step — for callers that keep every
// parameter gradient in one contiguous fp32 buffer (las_pytorch_amd/dp.py::FlatGradAllReducer does).  HBM-bound by
// construction: reads g, m, v, p and writes m, v, p (7 x 4 B per element; g is rewritten only when the clip is active).
//
//   launch 1  sumsq_partials : grid-stride float4 sum of squares -> one partial per workgroup
//   launch 2  clip_adam      : every workgroup re-reduces the partials (1 KB, L2-resident) -> total norm and clip factor;
//                              then, for its slice of ONE parameter tensor, Adam exactly as torch.optim.Adam evaluates it
//                              (amsgrad off, weight_decay 0, maximize off):
//                                  m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2
//                                  p -= (lr / (1-b1^t)) * m / (sqrt(v) / sqrt(1-b2^t) + eps)
//                              If *err_word != 0 (a persistent kernel of this step reported a hand-off timeout: its gradients are
//                              invalid) the update is SKIPPED on the device, so that the host can re-run the step on the generic
//                              kernels from unchanged parameters (solver.batch_iterator does).
#include "../../include/las_hip.h"
#include "las_common.h"
#include "las_kernels.h"
#include <algorithm>
#include <math.h>

namespace las {

constexpr int OPT_THREADS = 256;
constexpr int NORM_BLOCKS = 1024;                 // partial sums (workspace floats)
constexpr int CHUNK = OPT_THREADS * 4 * 4;        // elements per workgroup of the update: 4 float4 per thread
constexpr int MAX_T = 96;                         // parameter tensors per launch (kernel-argument table)

__device__ __forceinline__ float block_sum(float v, float* sh) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    v = (sh[0] + sh[1]) + (sh[2] + sh[3]);
    __syncthreads();
    return v;
}

__global__ __launch_bounds__(OPT_THREADS) void sumsq_partials_kernel(const float* __restrict__ g, long n, float* __restrict__ part) {
    __shared__ float sh[4];
    float a0 = 0.f, a1 = 0.f;
    const long n4 = ((uintptr_t)g % 16 == 0) ? n / 4 : 0;
    const f32x4* g4 = reinterpret_cast<const f32x4*>(g);
    const long stride = (long)gridDim.x * OPT_THREADS;
    long i = (long)blockIdx.x * OPT_THREADS + threadIdx.x;
    for (; i + stride < n4; i += 2 * stride) {
        const f32x4 x = g4[i], y = g4[i + stride];
        a0 += (x[0] * x[0] + x[1] * x[1]) + (x[2] * x[2] + x[3] * x[3]);
        a1 += (y[0] * y[0] + y[1] * y[1]) + (y[2] * y[2] + y[3] * y[3]);
    }
    for (; i < n4; i += stride) {
        const f32x4 x = g4[i];
        a0 += (x[0] * x[0] + x[1] * x[1]) + (x[2] * x[2] + x[3] * x[3]);
    }
    for (long j = n4 * 4 + (long)blockIdx.x * OPT_THREADS + threadIdx.x; j < n; j += stride) a1 += g[j] * g[j];
    const float s = block_sum(a0 + a1, sh);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}

struct AdamTable {
    float* param[MAX_T];
    long off[MAX_T + 1];          // element offset of tensor t inside the flat buffers; off[n] = end
    int blk[MAX_T + 1];           // first workgroup of tensor t; blk[n] = grid size
    int n;
};

struct AdamScalars { float max_norm, lr_c1, sqrt_c2, w1, b2, w2, eps; };

__device__ __forceinline__ void adam_one(float& p, float& m, float& v, float g, const AdamScalars& h) {
    // the operation order of torch's Adam: exp_avg.lerp_(grad, 1 - beta1); exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2);
    // denom = sqrt(exp_avg_sq) / sqrt(bias_correction2) + eps; param.addcdiv_(exp_avg, denom, value=-lr / bias_correction1)
    m = m + (g - m) * h.w1;
    v = h.b2 * v + h.w2 * (g * g);
    const float denom = sqrtf(v) / h.sqrt_c2 + h.eps;
    p -= h.lr_c1 * (m / denom);
}

__global__ __launch_bounds__(OPT_THREADS) void clip_adam_kernel(AdamTable tb, float* __restrict__ grad, float* __restrict__ ea,
                                                                float* __restrict__ es, const float* __restrict__ part,
                                                                AdamScalars h, float* __restrict__ norm_out,
                                                                const unsigned* __restrict__ err_word) {
    __shared__ float sh[4];
    // total norm from the partials (fp32 sum of 1024 fp32 partials of up to ~10^4 squares each)
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < NORM_BLOCKS / OPT_THREADS; ++k) s += part[threadIdx.x + k * OPT_THREADS];
    const float total = sqrtf(block_sum(s, sh));
    if (blockIdx.x == 0 && threadIdx.x == 0 && norm_out) *norm_out = total;
    if (err_word && __hip_atomic_load(err_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;    // invalid gradients: no update
    const float coef = fminf(h.max_norm / (total + 1e-6f), 1.0f);       // torch: clip_coef = max_norm / (norm + 1e-6), clamped to 1
    const bool clip = h.max_norm > 0.f && coef < 1.0f;
    // which tensor: blk[] is ascending, <= MAX_T entries (wave-uniform search)
    int t = 0;
    {
        int lo = 0, hi = tb.n;
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if ((int)blockIdx.x >= tb.blk[mid]) lo = mid; else hi = mid; }
        t = lo;
    }
    const long cnt = tb.off[t + 1] - tb.off[t];
    const long e0 = (long)((int)blockIdx.x - tb.blk[t]) * CHUNK;
    const long e1 = e0 + CHUNK < cnt ? e0 + CHUNK : cnt;
    float* __restrict__ p = tb.param[t];
    float* __restrict__ g = grad + tb.off[t];
    float* __restrict__ m = ea + tb.off[t];
    float* __restrict__ v = es + tb.off[t];
    const bool vec = ((uintptr_t)p % 16 == 0) && ((uintptr_t)g % 16 == 0) && ((uintptr_t)m % 16 == 0) && ((uintptr_t)v % 16 == 0);
    long i = e0;
    if (vec) {
        for (long j = e0 + 4 * threadIdx.x; j + 3 < e1; j += 4 * OPT_THREADS) {
            f32x4 gv = *reinterpret_cast<const f32x4*>(g + j);
            f32x4 pv = *reinterpret_cast<const f32x4*>(p + j), mv = *reinterpret_cast<const f32x4*>(m + j), vv = *reinterpret_cast<const f32x4*>(v + j);
            if (clip) { gv[0] *= coef; gv[1] *= coef; gv[2] *= coef; gv[3] *= coef; *reinterpret_cast<f32x4*>(g + j) = gv; }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float pk = pv[k], mk = mv[k], vk = vv[k];
                adam_one(pk, mk, vk, gv[k], h);
                pv[k] = pk; mv[k] = mk; vv[k] = vk;
            }
            *reinterpret_cast<f32x4*>(p + j) = pv; *reinterpret_cast<f32x4*>(m + j) = mv; *reinterpret_cast<f32x4*>(v + j) = vv;
        }
        i = e0 + ((e1 - e0) & ~3L);
    }
    for (long j = i + threadIdx.x; j < e1; j += OPT_THREADS) {
        float gv = g[j];
        if (clip) { gv *= coef; g[j] = gv; }
        float pv = p[j], mv = m[j], vv = v[j];
        adam_one(pv, mv, vv, gv, h);
        p[j] = pv; m[j] = mv; v[j] = vv;
    }
}

}  // namespace las

using namespace las;

extern "C" {

size_t las_clip_adam_workspace_floats(void) { return NORM_BLOCKS; }

int las_clip_adam(float* const* params, const int64_t* offsets, int n_tensors, float* grad_flat, float* exp_avg, float* exp_avg_sq,
                  float max_norm, double lr, double beta1, double beta2, double eps, int step, float* norm_out, float* workspace,
                  const uint32_t* err_word, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    LAS_REQUIRE(params && offsets && n_tensors > 0 && grad_flat && exp_avg && exp_avg_sq && workspace, "clip_adam pointers");
    LAS_REQUIRE(step >= 1 && lr >= 0. && beta1 >= 0. && beta1 < 1. && beta2 >= 0. && beta2 < 1. && eps >= 0., "Adam hyper-parameters");
    LAS_REQUIRE(offsets[0] == 0, "offsets start at 0");
    for (int t = 0; t < n_tensors; ++t) LAS_REQUIRE(params[t] && offsets[t + 1] > offsets[t], "parameter table");
    const long total = (long)offsets[n_tensors];
    hipLaunchKernelGGL(sumsq_partials_kernel, dim3(NORM_BLOCKS), dim3(OPT_THREADS), 0, stream, grad_flat, total, workspace);
    LAS_LAUNCH_CHECK();
    AdamScalars h;
    // torch.optim.Adam (single-tensor / fused): step_size = lr / bias_correction1 ; denom = sqrt(v) / sqrt(bias_correction2) + eps
    const double c1 = 1.0 - pow(beta1, (double)step), c2 = 1.0 - pow(beta2, (double)step);
    h.max_norm = max_norm; h.lr_c1 = (float)(lr / c1); h.sqrt_c2 = (float)sqrt(c2);
    h.w1 = (float)(1.0 - beta1); h.b2 = (float)beta2; h.w2 = (float)(1.0 - beta2); h.eps = (float)eps;
    for (int t0 = 0; t0 < n_tensors; t0 += MAX_T) {
        AdamTable tb;
        tb.n = std::min(MAX_T, n_tensors - t0);
        int blk = 0;
        for (int t = 0; t < tb.n; ++t) {
            tb.param[t] = params[t0 + t];
            tb.off[t] = (long)offsets[t0 + t];
            tb.blk[t] = blk;
            blk += cdiv((long)(offsets[t0 + t + 1] - offsets[t0 + t]), CHUNK);
        }
        tb.off[tb.n] = (long)offsets[t0 + tb.n];
        tb.blk[tb.n] = blk;
        hipLaunchKernelGGL(clip_adam_kernel, dim3(blk), dim3(OPT_THREADS), 0, stream, tb, grad_flat, exp_avg, exp_avg_sq, workspace, h,
                           t0 == 0 ? norm_out : nullptr, err_word);
        LAS_LAUNCH_CHECK();
    }
    return LAS_OK;
}

}  // extern "C"
