// Small memory-bound helpers around the hot path (bias gradients, masks, label conversion).
#include "las_common.h"
#include "las_kernels.h"
#include <algorithm>

namespace las {

thread_local char g_err[512] = {0};

// dst[c] (+)= sum_r src[r][c]: a workgroup covers 64 columns x a slab of rows; 4 row-lanes per column accumulate
// with 4 independent loads in flight each, LDS-combine, one atomic per column per workgroup
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ src, long ld, int rows, int cols,
                                                     float* __restrict__ dst, float* __restrict__ dst2, int rows_per_block) {
    __shared__ float part[4][64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    const int r0 = blockIdx.y * rows_per_block, r1 = min(rows, r0 + rows_per_block);
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (c < cols) {
        int r = r0 + rl;
        for (; r + 12 < r1; r += 16) {
            a0 += src[(long)r * ld + c]; a1 += src[(long)(r + 4) * ld + c];
            a2 += src[(long)(r + 8) * ld + c]; a3 += src[(long)(r + 12) * ld + c];
        }
        for (; r < r1; r += 4) a0 += src[(long)r * ld + c];
    }
    part[rl][cl] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (rl == 0 && c < cols) {
        const float v = (part[0][cl] + part[1][cl]) + (part[2][cl] + part[3][cl]);
        atomicAdd(dst + c, v);
        if (dst2) atomicAdd(dst2 + c, v);           // b_ih and b_hh of an LSTM receive the same gradient
    }
}

int colsum(const float* src, long ld, int rows, int cols, float* dst, int accumulate, hipStream_t stream, float* dst2) {
    if (!accumulate) {
        LAS_HIP_CHECK(hipMemsetAsync(dst, 0, sizeof(float) * cols, stream));
        if (dst2) LAS_HIP_CHECK(hipMemsetAsync(dst2, 0, sizeof(float) * cols, stream));
    }
    const int rpb = 256;
    dim3 grid(cdiv(cols, 64), cdiv(rows, rpb)), block(256);
    hipLaunchKernelGGL(colsum_kernel, grid, block, 0, stream, src, ld, rows, cols, dst, dst2, rpb);
    LAS_LAUNCH_CHECK();
    return LAS_OK;
}

// Several independent column sums in ONE launch (the bias gradients of the Speller backward: dz, dG of each layer, dqpre,
// dK): blockIdx.z selects the job, the grid covers the largest one.
struct ColsumJobs { const float* src[COLSUM_MAX_JOBS]; long ld[COLSUM_MAX_JOBS]; int rows[COLSUM_MAX_JOBS]; int cols[COLSUM_MAX_JOBS];
                    float* dst[COLSUM_MAX_JOBS]; float* dst2[COLSUM_MAX_JOBS]; };
__global__ __launch_bounds__(256) void colsum_multi_kernel(ColsumJobs j, int rows_per_block) {
    __shared__ float part[4][64];
    const int job = blockIdx.z;
    const float* __restrict__ src = j.src[job];
    const long ld = j.ld[job];
    const int rows = j.rows[job], cols = j.cols[job];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    const int r0 = blockIdx.y * rows_per_block, r1 = min(rows, r0 + rows_per_block);
    if (blockIdx.x * 64 >= cols || r0 >= rows) return;        // workgroup-uniform
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (c < cols) {
        int r = r0 + rl;
        for (; r + 12 < r1; r += 16) {
            a0 += src[(long)r * ld + c]; a1 += src[(long)(r + 4) * ld + c];
            a2 += src[(long)(r + 8) * ld + c]; a3 += src[(long)(r + 12) * ld + c];
        }
        for (; r < r1; r += 4) a0 += src[(long)r * ld + c];
    }
    part[rl][cl] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (rl == 0 && c < cols) {
        const float v = (part[0][cl] + part[1][cl]) + (part[2][cl] + part[3][cl]);
        atomicAdd(j.dst[job] + c, v);
        if (j.dst2[job]) atomicAdd(j.dst2[job] + c, v);
    }
}
int colsum_multi(const ColsumJob* jobs, int n, int zeroed, hipStream_t stream) {
    if (n <= 0) return LAS_OK;
    LAS_REQUIRE(n <= COLSUM_MAX_JOBS, "too many column-sum jobs");
    ColsumJobs j;
    int max_cols = 0, max_rows = 0;
    for (int i = 0; i < n; ++i) {
        j.src[i] = jobs[i].src; j.ld[i] = jobs[i].ld; j.rows[i] = jobs[i].rows; j.cols[i] = jobs[i].cols;
        j.dst[i] = jobs[i].dst; j.dst2[i] = jobs[i].dst2;
        max_cols = std::max(max_cols, jobs[i].cols); max_rows = std::max(max_rows, jobs[i].rows);
        if (!zeroed) {
            LAS_HIP_CHECK(hipMemsetAsync(jobs[i].dst, 0, sizeof(float) * jobs[i].cols, stream));
            if (jobs[i].dst2) LAS_HIP_CHECK(hipMemsetAsync(jobs[i].dst2, 0, sizeof(float) * jobs[i].cols, stream));
        }
    }
    const int rpb = 256;
    dim3 grid(cdiv(max_cols, 64), cdiv(max_rows, rpb), n), block(256);
    hipLaunchKernelGGL(colsum_multi_kernel, grid, block, 0, stream, j, rpb);
    LAS_LAUNCH_CHECK();
    return LAS_OK;
}

// grad *= act'(x) from the post-activation values (relu: mask)
__global__ void act_bwd_kernel(float* __restrict__ grad, const float* __restrict__ act, long n, int code) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) grad[i] *= act_grad(act[i], code);
}
int act_bwd_inplace(float* grad, const float* act, long n, int code, hipStream_t stream) {
    hipLaunchKernelGGL(act_bwd_kernel, dim3(cdiv(n, 256)), dim3(256), 0, stream, grad, act, n, code);
    LAS_LAUNCH_CHECK();
    return LAS_OK;
}

__global__ void act_kernel(float* __restrict__ x, long n, int code) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] = act_apply(x[i], code);
}
int act_inplace(float* x, long n, int code, hipStream_t stream) {
    hipLaunchKernelGGL(act_kernel, dim3(cdiv(n, 256)), dim3(256), 0, stream, x, n, code);
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

// dst[i] = sum_k src[k*stride + i]: the parts several workgroups of one utterance produced, summed in a fixed order
__global__ void sum_parts_kernel(float* __restrict__ dst, const float* __restrict__ src, long n, long stride, int parts) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float acc = src[i];
    for (int k = 1; k < parts; ++k) acc += src[(long)k * stride + i];
    dst[i] = acc;
}
int sum_parts(float* dst, const float* src, long n, long stride, int parts, hipStream_t stream) {
    hipLaunchKernelGGL(sum_parts_kernel, dim3(cdiv(n, 256)), dim3(256), 0, stream, dst, src, n, stride, parts);
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

// 16-byte aligned, tail-free shadow of the Speller's W_ih0 (4Hs, V+Hs) as (4Hs, Vp+Hs): columns [0,V) = label part,
// [V,Vp) = 0, [Vp,Vp+Hs) = context part.  One launch (it used to be a memset and two strided copies per forward).
// wperm (optional): the context part again as (4Hs, Hs) with the rows in the order the persistent decode kernel's cell
// workgroups consume them: row (j*16 + unit*4 + gate) = W_ih0 row (gate*Hs + 4j + unit) — the B operand of the
// `feat . W_ctx^T` product whose attention-weighted sum replaces `W_ctx . context` on the decode chain.
__global__ void build_w0p_kernel(const float* __restrict__ w, float* __restrict__ w0p, float* __restrict__ wperm,
                                 float* __restrict__ wyperm, float* __restrict__ bperm, const float* __restrict__ b_ih0,
                                 const float* __restrict__ b_hh0, int rows, int V, int Vp, int Hs) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int ld = Vp + Hs;
    if (i >= (long)rows * ld) return;
    const int r = i / ld, c = i % ld;
    const int gate = r / Hs, u = r % Hs;
    const long pr = (u >> 2) * 16 + (u & 3) * 4 + gate;       // row in the cell workgroups' order
    float v = 0.f;
    if (c < V) v = w[(long)r * (V + Hs) + c];
    else if (c >= Vp) {
        v = w[(long)r * (V + Hs) + V + (c - Vp)];
        if (wperm) wperm[pr * Hs + (c - Vp)] = v;
    }
    if (wyperm && c < Vp) wyperm[pr * Vp + c] = v;
    if (bperm && c == 0) bperm[pr] = b_ih0[r] + b_hh0[r];
    w0p[i] = v;
}
int build_w0p(const float* w_ih0, float* w0p, int Hs, int V, int Vp, hipStream_t stream, float* wperm, float* wyperm, float* bperm,
              const float* b_ih0, const float* b_hh0) {
    const long n = (long)4 * Hs * (Vp + Hs);
    hipLaunchKernelGGL(build_w0p_kernel, dim3(cdiv(n, 256)), dim3(256), 0, stream, w_ih0, w0p, wperm, wyperm, bperm, b_ih0, b_hh0, 4 * Hs, V,
                       Vp, Hs);
    LAS_LAUNCH_CHECK();
    return LAS_OK;
}

// The three element-wise preparations of a Speller forward in ONE launch (each is launch-bound at ~5 us on its own): the W_ih0 shadow
// (build_w0p), the teacher-forcing inputs (labels_to_y) and the initial context ctx_{-1} = feat[:,0,:] (reference las_model.py:198).
// Block ranges select the job; the bodies are the kernels above.
__global__ __launch_bounds__(256) void speller_prologue_kernel(const float* __restrict__ w, float* __restrict__ w0p, float* __restrict__ wperm,
                                                               float* __restrict__ wyperm, float* __restrict__ bperm,
                                                               const float* __restrict__ b_ih0, const float* __restrict__ b_hh0, int Hs, int V, int Vp,
                                                               const long long* __restrict__ labels, float* __restrict__ y_all, int B, int U, int u_lab,
                                                               const float* __restrict__ feat, long ldfeat, float* __restrict__ ctx0, int D,
                                                               int nb_w, int nb_y) {
    int blk = blockIdx.x;
    if (blk < nb_w) {
        const long i = (long)blk * 256 + threadIdx.x;
        const int ld = Vp + Hs, rows = 4 * Hs;
        if (i >= (long)rows * ld) return;
        const int r = i / ld, c = i % ld;
        const int gate = r / Hs, u = r % Hs;
        const long pr = (u >> 2) * 16 + (u & 3) * 4 + gate;
        float v = 0.f;
        if (c < V) v = w[(long)r * (V + Hs) + c];
        else if (c >= Vp) {
            v = w[(long)r * (V + Hs) + V + (c - Vp)];
            if (wperm) wperm[pr * Hs + (c - Vp)] = v;
        }
        if (wyperm && c < Vp) wyperm[pr * Vp + c] = v;
        if (bperm && c == 0) bperm[pr] = b_ih0[r] + b_hh0[r];
        w0p[i] = v;
        return;
    }
    blk -= nb_w;
    if (blk < nb_y) {
        const long i = (long)blk * 256 + threadIdx.x;
        const long n = (long)(U + 1) * B * Vp;
        if (i >= n) return;
        const int v = i % Vp;
        const int b = (i / Vp) % B;
        const int s = i / ((long)Vp * B);
        float val = 0.f;
        if (v < V) {
            if (s == 0) val = (v == 0) ? 1.f : 0.f;
            else if (labels && s - 1 < u_lab) val = (float)labels[((long)b * u_lab + (s - 1)) * V + v];
        }
        y_all[i] = val;
        return;
    }
    blk -= nb_y;
    const long i = (long)blk * 256 + threadIdx.x;
    if (i >= (long)B * D) return;
    const int r = i / D, c = i % D;
    ctx0[(long)r * D + c] = feat[(long)r * ldfeat + c];
}
int speller_prologue(const float* w_ih0, float* w0p, int Hs, int V, int Vp, float* wperm, float* wyperm, float* bperm, const float* b_ih0,
                     const float* b_hh0, const long long* labels, float* y_all, int B, int U, int u_lab, const float* feat, long ldfeat,
                     float* ctx0, int D, hipStream_t stream) {
    const int nb_w = cdiv((long)4 * Hs * (Vp + Hs), 256), nb_y = cdiv((long)(U + 1) * B * Vp, 256), nb_c = cdiv((long)B * D, 256);
    hipLaunchKernelGGL(speller_prologue_kernel, dim3(nb_w + nb_y + nb_c), dim3(256), 0, stream, w_ih0, w0p, wperm, wyperm, bperm, b_ih0, b_hh0, Hs, V,
                       Vp, labels, y_all, B, U, u_lab, feat, ldfeat, ctx0, D, nb_w, nb_y);
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

// Collate contract on device (reference utils/data.py:116-149): ragged frames -> zero-padded (B,T,F); character indices ->
// int64 one-hot (B,U,V) whose padding rows are onehot(PAD=0).  Pure HBM streaming: every output element is written once,
// 16 bytes per lane when the row length allows it.
__global__ __launch_bounds__(256) void collate_feat_kernel(const float* __restrict__ packed, const long long* __restrict__ foff,
                                                           float* __restrict__ inputs, int T, int F, int vec) {
    const int b = blockIdx.y;
    const long long off = foff[b], len = foff[b + 1] - off;
    const long n = (long)T * F, valid = (long)len * F;
    const float* src = packed + off * F;
    float* dst = inputs + (long)b * n;
    if (vec) {       // F % 4 == 0 and 16-byte aligned bases: float4 lanes
        const long n4 = n / 4, v4 = valid / 4;
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (i < v4) v = *reinterpret_cast<const f32x4*>(src + i * 4);
            *reinterpret_cast<f32x4*>(dst + i * 4) = v;
        }
    } else {
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) dst[i] = i < valid ? src[i] : 0.f;
    }
}
__global__ void collate_label_kernel(const long long* __restrict__ plab, const long long* __restrict__ loff,
                                     long long* __restrict__ targets, int B, int U, int V) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)B * U * V) return;
    const int v = i % V, u = (i / V) % U, b = i / ((long)V * U);
    const long long off = loff[b], len = loff[b + 1] - off;
    const long long sym = u < len ? plab[off + u] : 0;      // PAD = 0
    targets[i] = (sym == v) ? 1 : 0;
}
int collate_pad(const float* packed, const long long* foff, const long long* plab, const long long* loff, int B, int T, int F, int U,
                int V, float* inputs, long long* targets, hipStream_t stream) {
    const int vec = (F % 4 == 0) && ((uintptr_t)packed % 16 == 0) && ((uintptr_t)inputs % 16 == 0);
    const long per = (long)T * F / (vec ? 4 : 1);
    dim3 grid((unsigned)std::min<long>(cdiv(per, 256), 64), B);
    hipLaunchKernelGGL(collate_feat_kernel, grid, dim3(256), 0, stream, packed, foff, inputs, T, F, vec);
    const long n = (long)B * U * V;
    hipLaunchKernelGGL(collate_label_kernel, dim3(cdiv(n, 256)), dim3(256), 0, stream, plab, loff, targets, B, U, V);
    LAS_LAUNCH_CHECK();
    return LAS_OK;
}

// In-place log-softmax over rows of width V (teacher-forced decode: the character distribution of ALL steps is one
// GEMM after the loop instead of a per-step phase; reference las_model.py:182).  V <= 32 (the reference's 30 characters):
// one 32-lane group per row, coalesced row reads, shuffle reductions; wider rows: one thread per row.
__device__ __forceinline__ float group32_max(float v) {
#pragma unroll
    for (int m = 16; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m, 32));
    return v;
}
__device__ __forceinline__ float group32_sum(float v) {
#pragma unroll
    for (int m = 16; m >= 1; m >>= 1) v += __shfl_xor(v, m, 32);
    return v;
}
__global__ __launch_bounds__(256) void log_softmax_rows32_kernel(float* __restrict__ x, long rows, int V) {
    const long r = (long)blockIdx.x * 8 + (threadIdx.x >> 5);
    const int v = threadIdx.x & 31;
    if (r >= rows) return;
    const float val = v < V ? x[r * V + v] : -INFINITY;
    const float m = group32_max(val);
    const float s = group32_sum(v < V ? expf(val - m) : 0.f);
    if (v < V) x[r * V + v] = val - (m + logf(s));
}
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
    if (V <= 32) hipLaunchKernelGGL(log_softmax_rows32_kernel, dim3(cdiv(rows, 8)), dim3(256), 0, stream, x, rows, V);
    else hipLaunchKernelGGL(log_softmax_rows_kernel, dim3(cdiv(rows, 128)), dim3(128), 0, stream, x, rows, V);
    LAS_LAUNCH_CHECK();
    return LAS_OK;
}
// dz = dlogp - exp(logp) * sum(dlogp) per row
__global__ __launch_bounds__(256) void log_softmax_bwd_rows32_kernel(const float* __restrict__ dlogp, const float* __restrict__ logp,
                                                                     float* __restrict__ dz, long rows, int V) {
    const long r = (long)blockIdx.x * 8 + (threadIdx.x >> 5);
    const int v = threadIdx.x & 31;
    if (r >= rows) return;
    const float g = v < V ? dlogp[r * V + v] : 0.f;
    const float s = group32_sum(g);
    if (v < V) dz[r * V + v] = g - expf(logp[r * V + v]) * s;
}
__global__ void log_softmax_bwd_rows_kernel(const float* __restrict__ dlogp, const float* __restrict__ logp,
                                            float* __restrict__ dz, long rows, int V) {
    const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    float s = 0.f;
    for (int v = 0; v < V; ++v) s += dlogp[r * V + v];
    for (int v = 0; v < V; ++v) dz[r * V + v] = dlogp[r * V + v] - expf(logp[r * V + v]) * s;
}
int log_softmax_bwd_rows(const float* dlogp, const float* logp, float* dz, long rows, int V, hipStream_t stream) {
    if (V <= 32) hipLaunchKernelGGL(log_softmax_bwd_rows32_kernel, dim3(cdiv(rows, 8)), dim3(256), 0, stream, dlogp, logp, dz, rows, V);
    else hipLaunchKernelGGL(log_softmax_bwd_rows_kernel, dim3(cdiv(rows, 128)), dim3(128), 0, stream, dlogp, logp, dz, rows, V);
    LAS_LAUNCH_CHECK();
    return LAS_OK;
}

// ------------------------------------------------------------------------------------------------
// Caller-side contract on device (reference solver/solver.py): label-smoothing loss with its gradient, and the
// letter error rate, so a training step needs no host round trip per batch.
// ------------------------------------------------------------------------------------------------
// loss = -(1/B) sum_b sum_s ( sum_c smooth[b,s,c] logp[b,s,c] ) / len_b ,  smooth = ((1-eps) y + eps/V) * sum_c y,
// len_b = sum_{s,c} y   (solver.py:33-45).  One workgroup per utterance; per-utterance partial losses are summed in a
// fixed order by the last-launched tiny kernel (deterministic).  dlogp = -smooth / (B len_b) * gscale.
constexpr int LS_THREADS = 1024;      // (four elements per thread at U = 128, V = 30: the kernel is a chain of dependent memory round trips, not work)
__global__ __launch_bounds__(LS_THREADS) void ls_loss_kernel(const float* __restrict__ logp, long sU, long sB,
                                                             const long long* __restrict__ labels, int U, int U_lab, int B, int V,
                                                             float eps, float* __restrict__ part, float* __restrict__ dlogp,
                                                             long dU, long dB) {
    // one workgroup per utterance, one element (step, class) per thread and pass: the label slab is read coalesced; the row
    // sums the smoothing needs (1 on labelled steps, 0 on padding) are built in LDS first
    extern __shared__ float rowsum[];                 // U floats
    __shared__ float red[LS_THREADS];
    const int b = blockIdx.x, tid = threadIdx.x;
    const long long* yb = labels + (long)b * U_lab * V;
    const int n = U * V;
    for (int s = tid; s < U; s += LS_THREADS) rowsum[s] = 0.f;
    __syncthreads();
    float len = 0.f;
    for (int i = tid; i < n; i += LS_THREADS) {
        const float y = (float)yb[i];                 // rows s < U of a (U_lab, V) slab are contiguous
        if (y != 0.f) atomicAdd(&rowsum[i / V], y);
        len += y;
    }
    red[tid] = len; __syncthreads();
    for (int o = LS_THREADS / 2; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
    len = red[0]; __syncthreads();
    float acc = 0.f;
    const float scale = 1.f / (B * len);
    for (int i = tid; i < n; i += LS_THREADS) {
        const int s = i / V, c = i - s * V;
        const float sm = ((1.f - eps) * (float)yb[i] + eps / V) * rowsum[s];
        acc += sm * logp[(long)s * sU + (long)b * sB + c];
        if (dlogp) dlogp[(long)s * dU + (long)b * dB + c] = -sm * scale;
    }
    red[tid] = acc; __syncthreads();
    for (int o = LS_THREADS / 2; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
    if (tid == 0) part[b] = red[0] / len;
}
__global__ void ls_loss_finish_kernel(const float* __restrict__ part, int B, float* __restrict__ loss) {
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += part[b];
    loss[0] = -s / B;
}
int ls_loss(const float* logp, long sU, long sB, const long long* labels, int U, int U_lab, int B, int V, float eps, float* part,
            float* loss, float* dlogp, long dU, long dB, hipStream_t stream) {
    hipLaunchKernelGGL(ls_loss_kernel, dim3(B), dim3(LS_THREADS), sizeof(float) * U, stream, logp, sU, sB, labels, U, U_lab, B, V, eps, part, dlogp,
                       dU, dB);
    hipLaunchKernelGGL(ls_loss_finish_kernel, dim3(1), dim3(1), 0, stream, part, B, loss);
    LAS_LAUNCH_CHECK();
    return LAS_OK;
}

// Letter error rate (solver.py:11-24): prediction = argmax over V per step, zeros skipped, stop at the first 1; truth =
// argmax of the label rows with 0 and 1 dropped; Levenshtein distance / len(truth).  len(truth) == 0 gives NaN / +inf where the
// reference raises ZeroDivisionError.
// One WAVE per utterance (round 5; the first version ran one thread per utterance with its DP rows in global memory: 2.2 ms per call
// at B = 32, U = 128 — a third of a training step of the solver path).  Phase 1: every lane takes steps lane, lane + 64, ...: arg-max of the
// log-prob row and of the label row, then an order-preserving compaction with wave ballots.  Phase 2: the DP table by anti-diagonals —
// a lane owns CPL consecutive truth columns, cell (i, j) on diagonal d = i + j needs (i-1, j) and (i, j-1) from diagonal d-1 and
// (i-1, j-1) from d-2: its own registers plus ONE value pair from the lane below per diagonal.  np + nt diagonals of CPL cells each.
constexpr int LER_CPL_MAX = 64;          // truth symbols per lane: U <= 64 * 64 - 1 (one wave per SIMD: the 5 C registers per lane fit)
constexpr int LER_THREADS = 256;         // phase 1 (arg-max per step) uses every thread, the DP the first wave
// D[np][nt] with every lane of the wave owning columns lane * C .. lane * C + C - 1 (wave-uniform result).  State per owned column: its cell on
// diagonal d-1 (a) and d-2 (p); per diagonal the lane fetches the lane below's LAST column of both (two DPP wave shifts) and updates its C
// cells.  The prediction symbols a lane's cells compare against slide by one per diagonal: a register window, one (prefetched) LDS read each.
template <int C>
static __device__ __forceinline__ int ler_dp(const int* seqp, const int* seqt, int np, int nt) {
    const int lane = threadIdx.x;
    int a[C], p[C], tj[C], w[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const int j = lane * C + c;
        a[c] = 0; p[c] = 0; w[c] = -2;
        tj[c] = (j >= 1 && j <= nt) ? seqt[j - 1] : -1;
    }
    // w[c] = pred[i - 1] of cell (i, j = lane C + c) on the current diagonal, i.e. pred[d - lane C - c - 1]; entering diagonal d the window
    // shifts up by one and w[0] takes pred[d - lane C - 1] (fetched one diagonal ahead)
    auto pred_at = [&](int k) { return (k >= 0 && k < np) ? seqp[k] : -2; };
    int nxt = pred_at(0 - lane * C - 1);
    for (int d = 0; d <= np + nt; ++d) {
#pragma unroll
        for (int c = C - 1; c >= 1; --c) w[c] = w[c - 1];
        w[0] = nxt;
        nxt = pred_at(d + 1 - lane * C - 1);
        // lane - 1's last column as DPP wave_shr:1 (one VALU move each instead of an LDS round trip; lane 0's column 0 is the border: unused there)
        const int left_a = __builtin_amdgcn_mov_dpp(a[C - 1], 0x138, 0xF, 0xF, true), left_p = __builtin_amdgcn_mov_dpp(p[C - 1], 0x138, 0xF, 0xF, true);
        int n[C];
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const int j = lane * C + c, i = d - j;
            n[c] = a[c];
            if (j <= nt && i >= 0 && i <= np) {
                if (j == 0) n[c] = i;
                else if (i == 0) n[c] = j;
                else {
                    const int la = c == 0 ? left_a : a[c - 1], lp = c == 0 ? left_p : p[c - 1];
                    n[c] = min(min(a[c] + 1, la + 1), lp + (w[c] != tj[c] ? 1 : 0));
                }
            }
        }
#pragma unroll
        for (int c = 0; c < C; ++c) { p[c] = a[c]; a[c] = n[c]; }
    }
    int mine = 0;      // the owner of column nt holds D[np][nt]
#pragma unroll
    for (int c = 0; c < C; ++c) if (lane * C + c == nt) mine = a[c];
    return __shfl(mine, nt / C);
}
__global__ __launch_bounds__(LER_THREADS) void ler_kernel(const float* __restrict__ logp, long sU, long sB, const long long* __restrict__ labels,
                                                          int U, int U_lab, int B, int V, float* __restrict__ out) {
    extern __shared__ int ler_lds[];      // [pred U][truth U][arg-max per step U][label per step U]
    int* seqp = ler_lds;
    int* seqt = ler_lds + U;
    int* amv = ler_lds + 2 * U;
    int* tmv = ler_lds + 3 * U;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    // ---- phase 1a (all threads): the symbol of every step
    for (int s = tid; s < U; s += LER_THREADS) {
        const float* lp = logp + (long)s * sU + (long)b * sB;
        int am = 0; float best = lp[0];
        for (int c = 1; c < V; ++c) { const float v = lp[c]; if (v > best) { best = v; am = c; } }
        const long long* y = labels + ((long)b * U_lab + s) * V;
        int tm = 0; long long tb = y[0];
        for (int c = 1; c < V; ++c) { const long long v = y[c]; if (v > tb) { tb = v; tm = c; } }
        amv[s] = am; tmv[s] = tm;
    }
    __syncthreads();
    if (tid >= 64) return;
    // ---- phase 1b (first wave): the first <eos> of the prediction, order-preserving compaction with wave ballots
    const unsigned long long below = (1ull << lane) - 1ull;
    int stop = U;
    for (int s0 = 0; s0 < U && stop == U; s0 += 64) {
        const unsigned long long eos = __ballot(s0 + lane < U && amv[s0 + lane] == 1);
        if (eos != 0ull) stop = s0 + __builtin_ctzll(eos);
    }
    int np = 0, nt = 0;
    for (int s0 = 0; s0 < U; s0 += 64) {
        const int s = s0 + lane;
        const int am = s < U ? amv[s] : 0, tm = s < U ? tmv[s] : 0;
        const bool kp = s < U && s < stop && am != 0;              // (am == 1 cannot occur before `stop`)
        const bool kt = s < U && tm != 0 && tm != 1;
        const unsigned long long mp = __ballot(kp), mt = __ballot(kt);
        if (kp) seqp[np + __builtin_popcountll(mp & below)] = am;
        if (kt) seqt[nt + __builtin_popcountll(mt & below)] = tm;
        np += __builtin_popcountll(mp); nt += __builtin_popcountll(mt);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    // ---- phase 2: Levenshtein distance, rows = prediction (np), columns = truth (nt); a lane owns C consecutive columns of 0..nt
    const int cpl = (nt + 1 + 63) / 64;
    int dist;
    if (cpl <= 1) dist = ler_dp<1>(seqp, seqt, np, nt);
    else if (cpl <= 2) dist = ler_dp<2>(seqp, seqt, np, nt);
    else if (cpl <= 4) dist = ler_dp<4>(seqp, seqt, np, nt);
    else if (cpl <= 8) dist = ler_dp<8>(seqp, seqt, np, nt);
    else if (cpl <= 16) dist = ler_dp<16>(seqp, seqt, np, nt);
    else if (cpl <= 32) dist = ler_dp<32>(seqp, seqt, np, nt);
    else dist = ler_dp<LER_CPL_MAX>(seqp, seqt, np, nt);
    if (lane == 0) out[b] = (float)dist / (float)nt;
}
// out[r] = sum_k w[r * ld + k] x[k] for r < rows: one wave per row (the bias W_ctx b_dr of the multi-head decode kernel: 4Hs rows, once per call)
__global__ __launch_bounds__(256) void matvec_rows_kernel(const float* __restrict__ w, long ld, const float* __restrict__ x, float* __restrict__ out, int rows, int K,
                                                          const float* __restrict__ addend) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= rows) return;
    float acc = 0.f;
    for (int k = lane; k < K; k += 64) acc = fmaf(w[(long)r * ld + k], x[k], acc);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) out[r] = acc + (addend ? addend[r] : 0.f);
}
int matvec_rows(const float* w, long ld, const float* x, float* out, int rows, int K, hipStream_t stream, const float* addend) {
    hipLaunchKernelGGL(matvec_rows_kernel, dim3((rows + 3) / 4), dim3(256), 0, stream, w, ld, x, out, rows, K, addend);
    LAS_LAUNCH_CHECK();
    return LAS_OK;
}

int ler(const float* logp, long sU, long sB, const long long* labels, int U, int U_lab, int B, int V, float* out, int* work,
        hipStream_t stream) {
    (void)work;      // (the one-thread-per-utterance kernel of rounds 1-4 kept its DP rows there; the ABI keeps the argument)
    LAS_REQUIRE(U >= 1 && U <= 64 * LER_CPL_MAX - 1, "letter error rate: 1 .. 4095 decode steps");
    hipLaunchKernelGGL(ler_kernel, dim3(B), dim3(LER_THREADS), sizeof(int) * 4 * (size_t)U, stream, logp, sU, sB, labels, U, U_lab, B, V, out);
    LAS_LAUNCH_CHECK();
    return LAS_OK;
}

}  // namespace las
