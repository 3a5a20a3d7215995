// Speller decode-step kernels (forward and backward) for gfx950.
//
// Replaces, per decode step, the op sequence of the reference's Speller.forward_step / Attention.forward
// (reference model/las_model.py:178-184, 275-297): 2-layer nn.LSTM on a length-1 sequence, phi + relu,
// bmm energy, softmax over ALL T' (no mask), context = sum_t a_t * feat_t, cat, Linear, LogSoftmax,
// argmax feedback (model/las_model.py:223-227) — and their autograd (solver/solver.py:95).
//
// Design (CDNA4):
//   * lstm_cell_*: all B utterances of a step form the M dimension of v_mfma_f32_16x16x4_f32 tiles (exact fp32);
//     a workgroup owns 4 hidden units (= one 16-column tile of gate rows i,f,g,o), its 8 waves split K and
//     reduce through LDS, and the cell non-linearity runs in the same kernel.  Weights are read in their
//     PyTorch (4Hs, K) layout straight from L2 (17 MB at paper size: resident in the 32 MB aggregate L2 /
//     256 MB Infinity Cache across steps); no LDS staging for operands that are used once per block.
//   * attn_step_*: one workgroup per utterance; query, energies, softmax, context and the character
//     distribution never leave LDS/registers between phases (the reference materialises a (B,T',2H)
//     temporary per step, las_model.py:293-297).
#include "las_common.h"
#include "las_kernels.h"

namespace las {

constexpr int CELL_THREADS = 512, CELL_NW = 8;

template <int VEC>
__device__ __forceinline__ void load_vec(const float* __restrict__ p, int remain, float (&v)[VEC]) {
    if (remain >= VEC) {
        if (VEC == 4) { const f32x4 t = *reinterpret_cast<const f32x4*>(p); v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3]; }
        else if (VEC == 2) { const float2 t = *reinterpret_cast<const float2*>(p); v[0] = t.x; v[1] = t.y; }
        else v[0] = p[0];
    } else {
#pragma unroll
        for (int e = 0; e < VEC; ++e) v[e] = (e < remain) ? p[e] : 0.f;
    }
}

struct CellParams {
    CellSeg seg[3];
    int nseg;
    const float* b_ih; const float* b_hh; const float* c_prev;
    float* h_out; float* c_out; float* gates_out;
    int B, Hs;
};

template <int VEC, int MT>
__global__ __launch_bounds__(CELL_THREADS) void lstm_cell_fwd_kernel(CellParams p) {
    __shared__ float red[CELL_NW][MT][16][17];
    const int j0 = blockIdx.x * 4, b0 = blockIdx.y * (16 * MT);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, kq = lane >> 4;
    const long wrow = (long)(r >> 2) * p.Hs + j0 + (r & 3);      // tile column n = gate*4 + unit

    f32x4 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};

    constexpr int KB = 4 * VEC;
    for (int s = 0; s < p.nseg; ++s) {
        const CellSeg sg = p.seg[s];
        const float* __restrict__ wp = sg.w + wrow * sg.ldw;
        const int nkb = (sg.K + KB - 1) / KB;
#pragma unroll 2
        for (int kb = wave; kb < nkb; kb += CELL_NW) {
            const int k = kb * KB + kq * VEC;
            float bw[VEC];
            load_vec<VEC>(wp + k, sg.K - k, bw);
            float ax[MT][VEC];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int b = b0 + mt * 16 + r;
                if (b < p.B) load_vec<VEC>(sg.x + (long)b * sg.ldx + k, sg.K - k, ax[mt]);
                else {
#pragma unroll
                    for (int e = 0; e < VEC; ++e) ax[mt][e] = 0.f;
                }
            }
#pragma unroll
            for (int e = 0; e < VEC; ++e)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ax[mt][e], bw[e], acc[mt], 0, 0, 0);
        }
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) red[wave][mt][kq * 4 + rr][r] = acc[mt][rr];
    __syncthreads();

    if (tid < MT * 64) {
        const int bl = tid >> 2, u = tid & 3;
        const int b = b0 + bl, j = j0 + u;
        if (b < p.B) {
            float g4[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float s = 0.f;
#pragma unroll
                for (int w = 0; w < CELL_NW; ++w) s += red[w][bl >> 4][bl & 15][g * 4 + u];
                const int row = g * p.Hs + j;
                g4[g] = s + p.b_ih[row] + p.b_hh[row];
            }
            const float ig = sigmoidf_acc(g4[0]), fg = sigmoidf_acc(g4[1]), gg = tanhf_acc(g4[2]), og = sigmoidf_acc(g4[3]);
            const float cp = p.c_prev ? p.c_prev[(long)b * p.Hs + j] : 0.f;
            const float c = fg * cp + ig * gg;
            const float h = og * tanhf_acc(c);
            p.h_out[(long)b * p.Hs + j] = h;
            p.c_out[(long)b * p.Hs + j] = c;
            if (p.gates_out) {
                float* go = p.gates_out + (long)b * 4 * p.Hs + j;
                go[0] = ig; go[p.Hs] = fg; go[2 * p.Hs] = gg; go[3 * p.Hs] = og;
            }
        }
    }
}

static int seg_vec(const CellSeg& s) {
    auto al = [&](int bytes) {
        return ((uintptr_t)s.x % bytes == 0) && ((uintptr_t)s.w % bytes == 0) && ((s.ldx * 4) % bytes == 0) &&
               ((s.ldw * 4) % bytes == 0);
    };
    if (al(16)) return 4;
    if (al(8)) return 2;
    return 1;
}

int lstm_cell_fwd(const CellSeg* segs, int nseg, const float* b_ih, const float* b_hh, const float* c_prev, float* h_out,
                  float* c_out, float* gates_out, int B, int Hs, hipStream_t stream) {
    LAS_REQUIRE(nseg >= 1 && nseg <= 3, "cell segments");
    LAS_REQUIRE(Hs % 4 == 0, "speller hidden size must be a multiple of 4");
    CellParams p;
    int vec = 4;
    for (int i = 0; i < nseg; ++i) { p.seg[i] = segs[i]; vec = min(vec, seg_vec(segs[i])); }
    p.nseg = nseg; p.b_ih = b_ih; p.b_hh = b_hh; p.c_prev = c_prev; p.h_out = h_out; p.c_out = c_out; p.gates_out = gates_out;
    p.B = B; p.Hs = Hs;
    const int mt = B <= 16 ? 1 : (B <= 32 ? 2 : 4);
    dim3 grid(Hs / 4, cdiv(B, 16 * mt)), block(CELL_THREADS);
#define CELL_LAUNCH(V, M) hipLaunchKernelGGL((lstm_cell_fwd_kernel<V, M>), grid, block, 0, stream, p)
#define CELL_DISPATCH(V) { if (mt == 1) CELL_LAUNCH(V, 1); else if (mt == 2) CELL_LAUNCH(V, 2); else CELL_LAUNCH(V, 4); }
    if (vec == 4) CELL_DISPATCH(4) else if (vec == 2) CELL_DISPATCH(2) else CELL_DISPATCH(1)
#undef CELL_DISPATCH
#undef CELL_LAUNCH
    LAS_LAUNCH_CHECK();
    return LAS_OK;
}

// ------------------------------------------------------------------------------------------------
// cell backward: pointwise part
// ------------------------------------------------------------------------------------------------
__global__ void lstm_cell_bwd_pointwise_kernel(const float* __restrict__ dh_a, const float* __restrict__ dh_b,
                                               const float* __restrict__ dc_in, const float* __restrict__ gates,
                                               const float* __restrict__ c, const float* __restrict__ c_prev,
                                               float* __restrict__ dG, float* __restrict__ dc_prev, int B, int Hs) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)B * Hs) return;
    const int b = idx / Hs, j = idx % Hs;
    float dh = dh_a ? dh_a[idx] : 0.f;
    if (dh_b) dh += dh_b[idx];
    const float* gp = gates + (long)b * 4 * Hs + j;
    const float ig = gp[0], fg = gp[Hs], gg = gp[2 * Hs], og = gp[3 * Hs];
    const float tc = tanhf_acc(c[idx]);
    const float cp = c_prev ? c_prev[idx] : 0.f;
    const float dct = (dc_in ? dc_in[idx] : 0.f) + dh * og * (1.f - tc * tc);
    float* dp = dG + (long)b * 4 * Hs + j;
    dp[0] = dct * gg * ig * (1.f - ig);
    dp[Hs] = dct * cp * fg * (1.f - fg);
    dp[2 * Hs] = dct * ig * (1.f - gg * gg);
    dp[3 * Hs] = dh * tc * og * (1.f - og);
    dc_prev[idx] = dct * fg;
}

int lstm_cell_bwd_pointwise(const float* dh_a, const float* dh_b, const float* dc_in, const float* gates, const float* c,
                            const float* c_prev, float* dG, float* dc_prev, int B, int Hs, hipStream_t stream) {
    const long n = (long)B * Hs;
    hipLaunchKernelGGL(lstm_cell_bwd_pointwise_kernel, dim3(cdiv(n, 256)), dim3(256), 0, stream, dh_a, dh_b, dc_in, gates, c,
                       c_prev, dG, dc_prev, B, Hs);
    LAS_LAUNCH_CHECK();
    return LAS_OK;
}

// ------------------------------------------------------------------------------------------------
// small-M product  out(B,N) = a(B,K) * W(K,N)   (dX = dG * W of the cell backward), up to two weight sets
// ------------------------------------------------------------------------------------------------
struct SmallMParams {
    const float* a; long lda; int B, K;
    const float* w[2]; long ldw[2]; float* out[2]; long ldo[2]; int N[2]; int tiles0;
};

template <int MT>
__global__ __launch_bounds__(CELL_THREADS) void smallm_gemm_nn_kernel(SmallMParams p) {
    __shared__ float red[CELL_NW][MT][16][17];
    const int set = blockIdx.x >= p.tiles0 ? 1 : 0;
    const int n0 = (set ? blockIdx.x - p.tiles0 : blockIdx.x) * 16;
    const int b0 = blockIdx.y * (16 * MT);
    const float* __restrict__ W = p.w[set];
    const long ldw = p.ldw[set];
    const int N = p.N[set];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, kq = lane >> 4;
    const int n = n0 + r;
    const bool nok = n < N;

    f32x4 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nkb = (p.K + 15) / 16;
#pragma unroll 2
    for (int kb = wave; kb < nkb; kb += CELL_NW) {
        const int k = kb * 16 + kq * 4;
        float bw[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) bw[e] = (nok && k + e < p.K) ? W[(long)(k + e) * ldw + n] : 0.f;
        float ax[MT][4];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int b = b0 + mt * 16 + r;
            if (b < p.B) load_vec<4>(p.a + (long)b * p.lda + k, p.K - k, ax[mt]);
            else { ax[mt][0] = ax[mt][1] = ax[mt][2] = ax[mt][3] = 0.f; }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ax[mt][e], bw[e], acc[mt], 0, 0, 0);
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) red[wave][mt][kq * 4 + rr][r] = acc[mt][rr];
    __syncthreads();
    for (int idx = tid; idx < MT * 256; idx += CELL_THREADS) {
        const int bl = idx >> 4, c = idx & 15;
        const int b = b0 + bl;
        if (b < p.B && n0 + c < N) {
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < CELL_NW; ++w) s += red[w][bl >> 4][bl & 15][c];
            p.out[set][(long)b * p.ldo[set] + n0 + c] = s;
        }
    }
}

int smallm_gemm_nn2(const float* a, long lda, int B, int K, const float* w0, long ldw0, float* out0, long ldo0, int N0,
                    const float* w1, long ldw1, float* out1, long ldo1, int N1, hipStream_t stream) {
    LAS_REQUIRE(((uintptr_t)a % 16 == 0) && (lda % 4 == 0), "smallm A alignment");
    SmallMParams p;
    p.a = a; p.lda = lda; p.B = B; p.K = K;
    p.w[0] = w0; p.ldw[0] = ldw0; p.out[0] = out0; p.ldo[0] = ldo0; p.N[0] = N0;
    p.w[1] = w1; p.ldw[1] = ldw1; p.out[1] = out1; p.ldo[1] = ldo1; p.N[1] = w1 ? N1 : 0;
    p.tiles0 = cdiv(N0, 16);
    const int tiles = p.tiles0 + (w1 ? cdiv(N1, 16) : 0);
    const int mt = B <= 16 ? 1 : (B <= 32 ? 2 : 4);
    dim3 grid(tiles, cdiv(B, 16 * mt)), block(CELL_THREADS);
    if (mt == 1) hipLaunchKernelGGL((smallm_gemm_nn_kernel<1>), grid, block, 0, stream, p);
    else if (mt == 2) hipLaunchKernelGGL((smallm_gemm_nn_kernel<2>), grid, block, 0, stream, p);
    else hipLaunchKernelGGL((smallm_gemm_nn_kernel<4>), grid, block, 0, stream, p);
    LAS_LAUNCH_CHECK();
    return LAS_OK;
}

// ------------------------------------------------------------------------------------------------
// attention + character distribution, forward.  One workgroup per utterance.
// ------------------------------------------------------------------------------------------------
constexpr int ATT_THREADS = 256, ATT_NW = 4, ATT_MAX_TP = 4096, ATT_MAX_V = 128;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m));
    return v;
}
__device__ __forceinline__ float block_sum(float v, float* scratch) {   // scratch: ATT_NW floats
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < ATT_NW; ++w) s += scratch[w];
    return s;
}
__device__ __forceinline__ float block_max(float v, float* scratch) {
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    float s = scratch[0];
#pragma unroll
    for (int w = 1; w < ATT_NW; ++w) s = fmaxf(s, scratch[w]);
    return s;
}

// dot of a global row (len n) with an LDS vector, lanes of one wave striding by 64 (coalesced), full-wave reduce
__device__ __forceinline__ float wave_dot(const float* __restrict__ row, const float* vec, int n, int lane) {
    float acc = 0.f;
    for (int k = lane; k < n; k += 64) acc = fmaf(row[k], vec[k], acc);
    return wave_sum(acc);
}

__global__ __launch_bounds__(ATT_THREADS) void attn_step_fwd_kernel(AttnFwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // layout: cat[Hs + D] | q[Mq] | e[Tp] | logit[V] | scratch[8]
    const int Mq = a.use_mlp ? a.M : a.Hs;
    float* cat = smem;                       // [h_top | ctx]
    float* qs = cat + a.Hs + a.D;
    float* es = qs + Mq;
    float* lg = es + a.Tp;
    float* scratch = lg + a.V;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    for (int k = tid; k < a.Hs; k += ATT_THREADS) cat[k] = a.h_top[(long)b * a.Hs + k];
    __syncthreads();
    // 1. query
    if (a.use_mlp) {
        for (int m = wave; m < a.M; m += ATT_NW) {
            float v = wave_dot(a.w_phi + (long)m * a.Hs, cat, a.Hs, lane);
            if (lane == 0) {
                v += a.b_phi[m];
                if (a.relu) v = fmaxf(v, 0.f);
                qs[m] = v;
                if (a.q_out) a.q_out[(long)b * a.M + m] = v;
            }
        }
    } else {
        for (int k = tid; k < a.Hs; k += ATT_THREADS) qs[k] = cat[k];
    }
    __syncthreads();
    // 2. energies: 16 lanes per frame
    {
        const float* kb = a.keys + (long)b * a.Tp * Mq;
        const int sub = lane >> 4, sl = lane & 15;
        for (int t = wave * 4 + sub; t < a.Tp; t += ATT_NW * 4) {
            const float* kr = kb + (long)t * Mq;
            float acc = 0.f;
            for (int m = sl; m < Mq; m += 16) acc = fmaf(qs[m], kr[m], acc);
#pragma unroll
            for (int mm = 8; mm >= 1; mm >>= 1) acc += __shfl_xor(acc, mm);
            if (sl == 0) es[t] = acc;
        }
    }
    __syncthreads();
    // 3. softmax over all Tp frames (no mask, reference las_model.py:292)
    float lmax = -INFINITY;
    for (int t = tid; t < a.Tp; t += ATT_THREADS) lmax = fmaxf(lmax, es[t]);
    const float mx = block_max(lmax, scratch);
    float lsum = 0.f;
    for (int t = tid; t < a.Tp; t += ATT_THREADS) { const float ex = expf(es[t] - mx); es[t] = ex; lsum += ex; }
    const float inv = 1.0f / block_sum(lsum, scratch);
    for (int t = tid; t < a.Tp; t += ATT_THREADS) {
        const float w = es[t] * inv;
        es[t] = w;
        a.att_out[(long)b * a.Tp + t] = w;
    }
    __syncthreads();
    // 4. context
    {
        const float* fb = a.feat + (long)b * a.Tp * a.D;
        for (int d = tid; d < a.D; d += ATT_THREADS) {
            float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
            int t = 0;
            for (; t + 3 < a.Tp; t += 4) {
                acc0 = fmaf(es[t], fb[(long)t * a.D + d], acc0);
                acc1 = fmaf(es[t + 1], fb[(long)(t + 1) * a.D + d], acc1);
                acc2 = fmaf(es[t + 2], fb[(long)(t + 2) * a.D + d], acc2);
                acc3 = fmaf(es[t + 3], fb[(long)(t + 3) * a.D + d], acc3);
            }
            for (; t < a.Tp; ++t) acc0 = fmaf(es[t], fb[(long)t * a.D + d], acc0);
            const float c = (acc0 + acc1) + (acc2 + acc3);
            cat[a.Hs + d] = c;
            a.ctx_out[(long)b * a.D + d] = c;
        }
    }
    __syncthreads();
    // 5. character distribution
    for (int v = wave; v < a.V; v += ATT_NW) {
        const float s = wave_dot(a.w_c + (long)v * (a.Hs + a.D), cat, a.Hs + a.D, lane);
        if (lane == 0) lg[v] = s + a.b_c[v];
    }
    __syncthreads();
    // 6. log-softmax + argmax (first maximal index, as torch.topk/argmax)
    if (wave == 0) {
        float m = -INFINITY;
        for (int v = lane; v < a.V; v += 64) m = fmaxf(m, lg[v]);
        m = wave_max(m);
        float s = 0.f;
        for (int v = lane; v < a.V; v += 64) s += expf(lg[v] - m);
        s = wave_sum(s);
        const float lse = m + logf(s);
        int best = 0x7fffffff;
        for (int v = lane; v < a.V; v += 64) {
            a.logp_out[(long)b * a.V + v] = lg[v] - lse;
            if (lg[v] == m) best = min(best, v);
        }
#pragma unroll
        for (int mm = 32; mm >= 1; mm >>= 1) best = min(best, __shfl_xor(best, mm));
        if (lane == 0 && a.argmax_out) a.argmax_out[b] = best;
        if (a.y_next) {
            // next-step input: log-probs (decode_mode 0, las_model.py:220-221) or one-hot argmax (mode 1, :223-227)
            for (int v = lane; v < a.V; v += 64)
                a.y_next[(long)b * a.V + v] = (a.y_mode == 0) ? (lg[v] - lse) : (v == best ? 1.0f : 0.f);
        }
    }
}

int attn_step_fwd(const AttnFwdArgs& a, hipStream_t stream) {
    LAS_REQUIRE(a.Tp <= ATT_MAX_TP, "attention length above kernel limit (4096 encoder frames)");
    LAS_REQUIRE(a.V <= ATT_MAX_V, "vocab above kernel limit");
    const int Mq = a.use_mlp ? a.M : a.Hs;
    LAS_REQUIRE(a.use_mlp || a.D == a.Hs, "attention without MLP needs decoder dim == feature dim");
    const size_t smem = sizeof(float) * (size_t)(a.Hs + a.D + Mq + a.Tp + a.V + 8);
    hipLaunchKernelGGL(attn_step_fwd_kernel, dim3(a.B), dim3(ATT_THREADS), smem, stream, a);
    LAS_LAUNCH_CHECK();
    return LAS_OK;
}

// ------------------------------------------------------------------------------------------------
// attention + character distribution, backward (one decode step, one workgroup per utterance)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(ATT_THREADS) void attn_step_bwd_kernel(AttnBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int Mq = a.use_mlp ? a.M : a.Hs;
    // layout: dz[V] | dcat[Hs + D] | da[Tp] | dq[Mq] | part[4*Mq] | scratch[8]
    float* dz = smem;
    float* dcat = dz + a.V;
    float* da = dcat + a.Hs + a.D;
    float* dq = da + a.Tp;
    float* part = dq + Mq;
    float* scratch = part + 4 * Mq;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    // 1. log-softmax backward: dz = dlogp - exp(logp) * sum(dlogp)
    float part_sum = 0.f;
    for (int v = tid; v < a.V; v += ATT_THREADS) {
        float g = a.dlogp[(long)b * a.V + v];
        if (a.dy_carry) g += a.dy_carry[(long)b * a.ldy + v];
        dz[v] = g;                      // stage the total upstream gradient
        part_sum += g;
    }
    const float gsum = block_sum(part_sum, scratch);
    for (int v = tid; v < a.V; v += ATT_THREADS) {
        const float z = dz[v] - expf(a.logp[(long)b * a.V + v]) * gsum;
        dz[v] = z;
        a.dz_out[(long)b * a.V + v] = z;
    }
    __syncthreads();
    // 2. [dh_top | dctx] = dz * W_c ; dctx += carry
    const int K2 = a.Hs + a.D;
    for (int k = tid; k < K2; k += ATT_THREADS) {
        float acc = 0.f;
        for (int v = 0; v < a.V; ++v) acc = fmaf(dz[v], a.w_c[(long)v * K2 + k], acc);
        if (k >= a.Hs) {
            if (a.dctx_carry) acc += a.dctx_carry[(long)b * a.ldc + (k - a.Hs)];
            a.dctx_out[(long)b * a.D + (k - a.Hs)] = acc;
        }
        dcat[k] = acc;
    }
    __syncthreads();
    // 3. da[t] = dctx . feat_t   (16 lanes per frame)
    {
        const float* fb = a.feat + (long)b * a.Tp * a.D;
        const float* dctx = dcat + a.Hs;
        const int sub = lane >> 4, sl = lane & 15;
        for (int t = wave * 4 + sub; t < a.Tp; t += ATT_NW * 4) {
            const float* fr = fb + (long)t * a.D;
            float acc = 0.f;
            for (int d = sl; d < a.D; d += 16) acc = fmaf(dctx[d], fr[d], acc);
#pragma unroll
            for (int mm = 8; mm >= 1; mm >>= 1) acc += __shfl_xor(acc, mm);
            if (sl == 0) da[t] = acc;
        }
    }
    __syncthreads();
    // 4. softmax backward: de = a * (da - sum_t a_t da_t)
    float ls = 0.f;
    for (int t = tid; t < a.Tp; t += ATT_THREADS) ls = fmaf(a.att[(long)b * a.Tp + t], da[t], ls);
    const float sdot = block_sum(ls, scratch);
    for (int t = tid; t < a.Tp; t += ATT_THREADS) {
        const float de = a.att[(long)b * a.Tp + t] * (da[t] - sdot);
        da[t] = de;
        a.de_out[(long)b * a.Tp + t] = de;
    }
    __syncthreads();
    // 5. dq[m] = sum_t de[t] * keys[t][m]   (4 time-quarters per m, then LDS reduce)
    {
        const float* kb = a.keys + (long)b * a.Tp * Mq;
        for (int idx = tid; idx < 4 * Mq; idx += ATT_THREADS) {
            const int m = idx % Mq, tq = idx / Mq;
            float acc = 0.f;
            for (int t = tq; t < a.Tp; t += 4) acc = fmaf(da[t], kb[(long)t * Mq + m], acc);
            part[idx] = acc;
        }
    }
    __syncthreads();
    for (int m = tid; m < Mq; m += ATT_THREADS) {
        float v = (part[m] + part[Mq + m]) + (part[2 * Mq + m] + part[3 * Mq + m]);
        if (a.use_mlp) {
            if (a.relu && !(a.q[(long)b * a.M + m] > 0.f)) v = 0.f;
            a.dqpre_out[(long)b * a.M + m] = v;
        }
        dq[m] = v;
    }
    __syncthreads();
    // 6. dh_top += dqpre * W_phi  (or += dq when there is no MLP)
    for (int k = tid; k < a.Hs; k += ATT_THREADS) {
        float acc = dcat[k];
        if (a.use_mlp) {
            for (int m = 0; m < a.M; ++m) acc = fmaf(dq[m], a.w_phi[(long)m * a.Hs + k], acc);
        } else {
            acc += dq[k];
        }
        a.dh_top_out[(long)b * a.Hs + k] = acc;
    }
}

int attn_step_bwd(const AttnBwdArgs& a, hipStream_t stream) {
    LAS_REQUIRE(a.Tp <= ATT_MAX_TP && a.V <= ATT_MAX_V, "attention backward limits");
    const int Mq = a.use_mlp ? a.M : a.Hs;
    const size_t smem = sizeof(float) * (size_t)(a.V + a.Hs + a.D + a.Tp + 5 * Mq + 8);
    hipLaunchKernelGGL(attn_step_bwd_kernel, dim3(a.B), dim3(ATT_THREADS), smem, stream, a);
    LAS_LAUNCH_CHECK();
    return LAS_OK;
}

}  // namespace las
