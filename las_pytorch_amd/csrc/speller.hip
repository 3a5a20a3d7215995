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
#include "options.h"

namespace las {

constexpr int CELL_THREADS = 512, CELL_NW = 8, CELL_UNR = 4;

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

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
// Branch-free guarded loads: the load itself is unconditional (from element 0 when !ok) and the result is selected
// afterwards.  A per-load branch makes hipcc wait for outstanding loads at every join, which serialises a prefetch
// batch into one memory round trip per load.
__device__ __forceinline__ f32x4 ld4c(const float* base, long off, bool ok) {
    const f32x4 v = ld4(base + (ok ? off : 0));
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    return ok ? v : z;
}
__device__ __forceinline__ float ld1c(const float* base, long off, bool ok) {
    const float v = base[ok ? off : 0];
    return ok ? v : 0.f;
}
__device__ __forceinline__ float dot4(const f32x4 a, const f32x4 b, float acc) {
    acc = fmaf(a[0], b[0], acc); acc = fmaf(a[1], b[1], acc); acc = fmaf(a[2], b[2], acc); acc = fmaf(a[3], b[3], acc);
    return acc;
}

constexpr int CELL_MAXB = 9;     // k-blocks (16 k each) whose loads one wave keeps in flight at once

struct CellParams {
    CellSeg seg[3];
    int nseg;
    const float* b_ih; const float* b_hh; const float* c_prev;
    float* h_out; float* c_out; float* gates_out;
    int B, Hs;
    float* linear_out; long ldo; const float* linear_bias;   // plain linear layer instead of the LSTM cell
};

// One LSTM cell step.  The decode step is a chain of dependent kernels and each kernel pays ~2-3 us for every
// dependent round trip to data another kernel just produced, so ALL global loads of a wave (its share of the
// concatenated K axis of every input segment, plus the cell's bias / c_prev operands) are issued before the
// first MFMA: one memory round trip per kernel.
template <int MT>
__global__ __launch_bounds__(CELL_THREADS) void lstm_cell_fwd_kernel(CellParams p) {
    __shared__ float red[CELL_NW][MT][16][17];
    const int j0 = blockIdx.x * 4, b0 = blockIdx.y * (16 * MT);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, kq = lane >> 4;
    const bool lin = p.linear_out != nullptr;
    const long wrow = lin ? (long)blockIdx.x * 16 + r : (long)(r >> 2) * p.Hs + j0 + (r & 3);   // tile column n = gate*4 + unit

    // operands of the cell non-linearity (threads that will apply it)
    const int pbl = tid >> 2, pu = tid & 3;
    const int pb = b0 + pbl, pj = j0 + pu;
    const bool pw = tid < MT * 64 && pb < p.B;
    // raw loads only (no arithmetic on them here: a use would force a wait before the main load batch is issued)
    float cp = 0.f, bi[4] = {0.f, 0.f, 0.f, 0.f}, bh[4] = {0.f, 0.f, 0.f, 0.f};
    if (!lin) {
        const long po = (long)min(pb, p.B - 1) * p.Hs + pj;
        cp = p.c_prev ? p.c_prev[po] : 0.f;
#pragma unroll
        for (int g = 0; g < 4; ++g) { bi[g] = p.b_ih[g * p.Hs + pj]; bh[g] = p.b_hh[g * p.Hs + pj]; }
    }

    const int nb0 = (p.seg[0].K + 15) >> 4;
    const int nb1 = p.nseg > 1 ? (p.seg[1].K + 15) >> 4 : 0;
    const int nb2 = p.nseg > 2 ? (p.seg[2].K + 15) >> 4 : 0;
    const int tot = nb0 + nb1 + nb2;
    const int per = (tot + CELL_NW - 1) / CELL_NW;
    const int beg = wave * per, end = min(tot, beg + per);

    f32x4 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int base = beg; base < end; base += CELL_MAXB) {
        float bw[CELL_MAXB][4];
        float ax[CELL_MAXB][MT][4];
#pragma unroll
        for (int u = 0; u < CELL_MAXB; ++u) {
            const int blk = base + u;
            int kb = blk, si = 0;
            if (kb >= nb0) { kb -= nb0; si = 1; if (kb >= nb1) { kb -= nb1; si = 2; } }
            const float* __restrict__ sx = p.seg[si].x; const float* __restrict__ sw = p.seg[si].w;
            const long ldx = p.seg[si].ldx, ldw = p.seg[si].ldw;
            const bool live = blk < end;                          // wave-uniform
            const int k = (live ? kb : 0) * 16 + kq * 4;           // every segment K is a multiple of 16 (host-checked)
            const f32x4 wv = ld4c(sw, wrow * ldw + k, live);
            bw[u][0] = wv[0]; bw[u][1] = wv[1]; bw[u][2] = wv[2]; bw[u][3] = wv[3];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int b = b0 + mt * 16 + r;
                const f32x4 xv = ld4c(sx, (long)b * ldx + k, live && b < p.B);
                ax[u][mt][0] = xv[0]; ax[u][mt][1] = xv[1]; ax[u][mt][2] = xv[2]; ax[u][mt][3] = xv[3];
            }
        }
        __builtin_amdgcn_sched_barrier(0);       // every load of the batch is issued before the first MFMA
#pragma unroll
        for (int u = 0; u < CELL_MAXB; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ax[u][mt][e], bw[u][e], acc[mt], 0, 0, 0);
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) red[wave][mt][kq * 4 + rr][r] = acc[mt][rr];
    __syncthreads();

    if (lin) {
        const int n0 = blockIdx.x * 16;
        for (int idx = tid; idx < MT * 256; idx += CELL_THREADS) {
            const int bl = idx >> 4, c = idx & 15;
            const int b = b0 + bl;
            if (b < p.B) {
                float s = p.linear_bias ? p.linear_bias[n0 + c] : 0.f;
#pragma unroll
                for (int w = 0; w < CELL_NW; ++w) s += red[w][bl >> 4][bl & 15][c];
                p.linear_out[(long)b * p.ldo + n0 + c] = s;
            }
        }
        return;
    }
    if (pw) {
        float g4[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < CELL_NW; ++w) s += red[w][pbl >> 4][pbl & 15][g * 4 + pu];
            g4[g] = s + (bi[g] + bh[g]);
        }
        const float ig = sigmoidf_acc(g4[0]), fg = sigmoidf_acc(g4[1]), gg = tanhf_acc(g4[2]), og = sigmoidf_acc(g4[3]);
        const float c = fg * cp + ig * gg;
        const float h = og * tanhf_acc(c);
        p.h_out[(long)pb * p.Hs + pj] = h;
        p.c_out[(long)pb * p.Hs + pj] = c;
        if (p.gates_out) {
            float* go = p.gates_out + (long)pb * 4 * p.Hs + pj;
            go[0] = ig; go[p.Hs] = fg; go[2 * p.Hs] = gg; go[3 * p.Hs] = og;
        }
    }
}

static bool al16(const void* p, long ld) { return ((uintptr_t)p % 16 == 0) && (ld % 4 == 0); }

int lstm_cell_fwd(const CellSeg* segs, int nseg, const float* b_ih, const float* b_hh, const float* c_prev, float* h_out,
                  float* c_out, float* gates_out, int B, int Hs, hipStream_t stream) {
    LAS_REQUIRE(nseg >= 1 && nseg <= 3, "cell segments");
    LAS_REQUIRE(Hs % 4 == 0, "speller hidden size must be a multiple of 4");
    CellParams p;
    for (int i = 0; i < nseg; ++i) {
        p.seg[i] = segs[i];
        LAS_REQUIRE(al16(segs[i].x, segs[i].ldx) && al16(segs[i].w, segs[i].ldw), "cell operands must be 16-byte aligned");
        LAS_REQUIRE(segs[i].K % 16 == 0, "cell segment widths must be multiples of 16 (pad the label segment)");
    }
    for (int i = nseg; i < 3; ++i) p.seg[i] = segs[0];
    p.nseg = nseg; p.b_ih = b_ih; p.b_hh = b_hh; p.c_prev = c_prev; p.h_out = h_out; p.c_out = c_out; p.gates_out = gates_out;
    p.B = B; p.Hs = Hs; p.linear_out = nullptr; p.ldo = 0; p.linear_bias = nullptr;
    // one 16-utterance M-tile per workgroup while that keeps the grid within ~two waves of the 256 CUs
    const int mt_env = (int)opt_get(OPT_CELL_MT);
    const int mt = mt_env ? mt_env : ((long)(Hs / 4) * cdiv(B, 16) <= 1024 ? 1 : 2);
    dim3 grid(Hs / 4, cdiv(B, 16 * mt)), block(CELL_THREADS);
    if (mt == 1) hipLaunchKernelGGL((lstm_cell_fwd_kernel<1>), grid, block, 0, stream, p);
    else hipLaunchKernelGGL((lstm_cell_fwd_kernel<2>), grid, block, 0, stream, p);
    LAS_LAUNCH_CHECK();
    return LAS_OK;
}

int smallm_linear_nt(const CellSeg* segs, int nseg, const float* bias, float* out, long ldo, int B, int N, hipStream_t stream) {
    LAS_REQUIRE(nseg >= 1 && nseg <= 3 && N % 16 == 0, "linear segments / width");
    CellParams p;
    for (int i = 0; i < nseg; ++i) {
        p.seg[i] = segs[i];
        LAS_REQUIRE(al16(segs[i].x, segs[i].ldx) && al16(segs[i].w, segs[i].ldw) && segs[i].K % 16 == 0, "linear operands");
    }
    for (int i = nseg; i < 3; ++i) p.seg[i] = segs[0];
    p.nseg = nseg; p.b_ih = nullptr; p.b_hh = nullptr; p.c_prev = nullptr; p.h_out = nullptr; p.c_out = nullptr; p.gates_out = nullptr;
    p.B = B; p.Hs = 0; p.linear_out = out; p.ldo = ldo; p.linear_bias = bias;
    dim3 grid(N / 16, cdiv(B, 16)), block(CELL_THREADS);
    hipLaunchKernelGGL((lstm_cell_fwd_kernel<1>), grid, block, 0, stream, p);
    LAS_LAUNCH_CHECK();
    return LAS_OK;
}

// one (utterance, unit) of the cell backward pointwise step (single step)
__device__ __forceinline__ void cell_bwd_point(const CellPw& pw, long b, int j, int Hs, float dh) {
    const long idx = b * Hs + j;
    if (pw.dh_carry) dh += pw.dh_carry[idx];
    const float* gp = pw.gates + b * 4 * Hs + j;
    const float ig = gp[0], fg = gp[Hs], gg = gp[2 * Hs], og = gp[3 * Hs];
    const float tc = tanhf_acc(pw.c[idx]);
    const float cp = pw.c_prev ? pw.c_prev[idx] : 0.f;
    const float dct = (pw.dc_in ? pw.dc_in[idx] : 0.f) + dh * og * (1.f - tc * tc);
    float* dp = pw.dG + b * 4 * Hs + j;
    dp[0] = dct * gg * ig * (1.f - ig);
    dp[Hs] = dct * cp * fg * (1.f - fg);
    dp[2 * Hs] = dct * ig * (1.f - gg * gg);
    dp[3 * Hs] = dh * tc * og * (1.f - og);
    pw.dc_out[idx] = dct * fg;
}

// ------------------------------------------------------------------------------------------------
// cell backward: pointwise part
// ------------------------------------------------------------------------------------------------
__global__ void lstm_cell_bwd_pointwise_kernel(const float* __restrict__ dh_a, int nparts, long part_stride,
                                               const float* __restrict__ dh_b,
                                               const float* dc_in, const float* __restrict__ gates,
                                               const float* __restrict__ c, const float* __restrict__ c_prev,
                                               float* __restrict__ dG, float* dc_prev, int B, int Hs) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)B * Hs) return;
    const int b = idx / Hs, j = idx % Hs;
    float dh = 0.f;
    for (int q = 0; q < nparts; ++q) dh += dh_a[idx + q * part_stride];
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

int lstm_cell_bwd_pointwise(const float* dh_a, int nparts, long part_stride, const float* dh_b, const float* dc_in,
                            const float* gates, const float* c, const float* c_prev, float* dG, float* dc_prev, int B, int Hs,
                            hipStream_t stream) {
    const long n = (long)B * Hs;
    hipLaunchKernelGGL(lstm_cell_bwd_pointwise_kernel, dim3(cdiv(n, 256)), dim3(256), 0, stream, dh_a, nparts, part_stride, dh_b, dc_in, gates, c,
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
    CellPw pw; int Hs;
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
    const int n = min(n0 + r, N - 1);                // clamped column: out-of-range columns are never stored

    f32x4 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nkb = (p.K + 15) / 16;
    for (int kb0 = wave * CELL_UNR; kb0 < nkb; kb0 += CELL_NW * CELL_UNR) {
        float bw[CELL_UNR][4];
        float ax[CELL_UNR][MT][4];
#pragma unroll
        for (int u = 0; u < CELL_UNR; ++u) {
            const int k = (kb0 + u) * 16 + kq * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) bw[u][e] = (k + e < p.K) ? W[(long)(k + e) * ldw + n] : 0.f;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int b = b0 + mt * 16 + r;
                load_vec<4>(p.a + (long)min(b, p.B - 1) * p.lda + k, b < p.B ? p.K - k : 0, ax[u][mt]);
            }
        }
#pragma unroll
        for (int u = 0; u < CELL_UNR; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ax[u][mt][e], bw[u][e], acc[mt], 0, 0, 0);
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
            if (p.out[set]) p.out[set][(long)b * p.ldo[set] + n0 + c] = s;
            if (set == 0 && p.pw.gates) cell_bwd_point(p.pw, b, n0 + c, p.Hs, s);   // next lower cell's pointwise step
        }
    }
}

int smallm_gemm_nn2(const float* a, long lda, int B, int K, const float* w0, long ldw0, float* out0, long ldo0, int N0,
                    const float* w1, long ldw1, float* out1, long ldo1, int N1, const CellPw& pw, int Hs, hipStream_t stream) {
    LAS_REQUIRE(((uintptr_t)a % 16 == 0) && (lda % 4 == 0), "smallm A alignment");
    LAS_REQUIRE(!pw.gates || N0 == Hs, "fused pointwise needs N0 == Hs");
    LAS_REQUIRE(out0 || pw.gates, "smallm output 0");
    SmallMParams p;
    p.pw = pw; p.Hs = Hs;
    p.a = a; p.lda = lda; p.B = B; p.K = K;
    p.w[0] = w0; p.ldw[0] = ldw0; p.out[0] = out0; p.ldo[0] = ldo0; p.N[0] = N0;
    p.w[1] = w1; p.ldw[1] = ldw1; p.out[1] = out1; p.ldo[1] = ldo1; p.N[1] = w1 ? N1 : 0;
    p.tiles0 = cdiv(N0, 16);
    const int tiles = p.tiles0 + (w1 ? cdiv(N1, 16) : 0);
    const int mt = (long)tiles * cdiv(B, 16) <= 512 ? 1 : (B <= 32 ? 2 : 4);
    dim3 grid(tiles, cdiv(B, 16 * mt)), block(CELL_THREADS);
    if (mt == 1) hipLaunchKernelGGL((smallm_gemm_nn_kernel<1>), grid, block, 0, stream, p);
    else if (mt == 2) hipLaunchKernelGGL((smallm_gemm_nn_kernel<2>), grid, block, 0, stream, p);
    else hipLaunchKernelGGL((smallm_gemm_nn_kernel<4>), grid, block, 0, stream, p);
    LAS_LAUNCH_CHECK();
    return LAS_OK;
}

// ------------------------------------------------------------------------------------------------
// attention + character distribution.  One workgroup of 1024 threads (16 wave64) per utterance.
// Every phase is shaped so that ALL of its global loads are independent and issued together (one L2 round trip
// per phase): the decode step is a chain of ~6 dependent phases and each costs a memory latency, not bandwidth.
// ------------------------------------------------------------------------------------------------
constexpr int ATT_THREADS = 1024, ATT_MAX_TP = 4096, ATT_MAX_V = 128;

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
template <int W>
__device__ __forceinline__ float group_sum(float v) {       // sum over aligned groups of W lanes
#pragma unroll
    for (int m = W / 2; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}

// rows x K (row-major, ld) times a vector: out[row] for rows [0,R), G lanes per row, vector read through `vec(k)`.
// All loads of a pass are independent.  K % 4 == 0.

__global__ __launch_bounds__(ATT_THREADS) void attn_step_fwd_kernel(AttnFwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int Mq = a.use_mlp ? a.M : a.Hs;
    const int C4 = a.D / 4;                               // float4 column groups of a feature row
    const int NTQ = min(ATT_THREADS / C4, 64);            // time slices of the context reduction
    // layout: q[Mq] | e[Tp] | ctx[D] | logit[V(+pad)] | part[NTQ*D]
    float* qs = smem;
    float* es = qs + Mq;
    float* ctxs = es + ((a.Tp + 3) & ~3);
    float* lg = ctxs + a.D;
    float* part = lg + ((a.V + 3) & ~3);
    const int b = blockIdx.x, hd = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
    const float* __restrict__ hb = a.h_top + (long)b * a.Hs;
    const long ldq = a.ldq ? a.ldq : a.M, ldctx = a.ldctx ? a.ldctx : a.D;
    const float* __restrict__ w_phi = a.w_phi + (long)hd * a.M * a.Hs;      // this head's rows of phi
    const float* __restrict__ b_phi = a.b_phi + (long)hd * a.M;
    float mx = 0.f, inv = 0.f;

    if (!(a.phases & 1)) {
        // character-distribution-only pass (multi-head, free running): the context was produced by dim_reduce
        for (int d = tid; d < a.D; d += ATT_THREADS) ctxs[d] = a.ctx_in[(long)b * a.D + d];
        __syncthreads();
    } else {
    // 1. query q = act(W_phi h + b_phi): 16 lanes per output row
    if (a.use_mlp) {
        const int ks = tid & 15;
        for (int m = tid >> 4; m < a.M; m += ATT_THREADS / 16) {
            const float* wr = w_phi + (long)m * a.Hs;
            float acc = 0.f;
#pragma unroll 8
            for (int k = ks * 4; k < a.Hs; k += 64) acc = dot4(ld4(wr + k), ld4(hb + k), acc);
            acc = group_sum<16>(acc);
            if (ks == 0) {
                acc += b_phi[m];
                acc = act_apply(acc, a.relu);
                qs[m] = acc;
                if (a.q_out) a.q_out[(long)b * ldq + (long)hd * a.M + m] = acc;
            }
        }
    } else {
        for (int k = tid; k < a.Hs; k += ATT_THREADS) qs[k] = hb[k];
    }
    __syncthreads();
    // 2. energies e[t] = q . keys[t]: 16 lanes per frame
    {
        const float* kb = a.keys + (long)b * a.Tp * Mq;
        const int ms = tid & 15;
        for (int t = tid >> 4; t < a.Tp; t += ATT_THREADS / 16) {
            const float* kr = kb + (long)t * Mq;
            float acc = 0.f;
#pragma unroll 4
            for (int m = ms * 4; m < Mq; m += 64) acc = dot4(ld4(kr + m), ld4(qs + m), acc);
            acc = group_sum<16>(acc);
            if (ms == 0) es[t] = acc;
        }
    }
    __syncthreads();
    // 3. softmax statistics over ALL frames (no mask, reference las_model.py:292), redundantly per wave (no barrier)
    mx = -INFINITY;
    for (int t = lane; t < a.Tp; t += 64) mx = fmaxf(mx, es[t]);
    mx = wave_max(mx);
    float sm = 0.f;
    for (int t = lane; t < a.Tp; t += 64) sm += expf(es[t] - mx);
    inv = 1.0f / wave_sum(sm);
    for (int t = tid; t < a.Tp; t += ATT_THREADS) a.att_out[(long)hd * a.att_hs + (long)b * a.Tp + t] = expf(es[t] - mx) * inv;
    // 4. context = sum_t a_t feat_t : (column group, time slice) per thread, then LDS reduce over slices
    {
        const float* fb = a.feat + (long)b * a.Tp * a.D;
        const int c4 = tid % C4, tq = tid / C4;
        if (tq < NTQ) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
            for (int t = tq; t < a.Tp; t += NTQ) {
                const float w = expf(es[t] - mx) * inv;
                const f32x4 f = ld4(fb + (long)t * a.D + c4 * 4);
                acc[0] = fmaf(w, f[0], acc[0]); acc[1] = fmaf(w, f[1], acc[1]);
                acc[2] = fmaf(w, f[2], acc[2]); acc[3] = fmaf(w, f[3], acc[3]);
            }
            *reinterpret_cast<f32x4*>(part + (long)tq * a.D + c4 * 4) = acc;
        }
    }
    __syncthreads();
    for (int d = tid; d < a.D; d += ATT_THREADS) {
        float acc = 0.f;
        for (int q = 0; q < NTQ; ++q) acc += part[(long)q * a.D + d];
        ctxs[d] = acc;
        a.ctx_out[(long)b * ldctx + (long)hd * a.D + d] = acc;
    }
    __syncthreads();
    }   // phases & 1
    if (a.logp_out == nullptr || !(a.phases & 2)) return;   // deferred: one GEMM + log-softmax over all steps after the loop
    // 5. character distribution logits = W_c [h | ctx] + b_c : 32 lanes per output row
    {
        const int ks = tid & 31;
        const int K2 = a.Hs + a.D;
        for (int v = tid >> 5; v < a.V; v += ATT_THREADS / 32) {
            const float* wr = a.w_c + (long)v * K2;
            float acc = 0.f;
#pragma unroll 8
            for (int k = ks * 4; k < K2; k += 128) {
                const f32x4 x = (k < a.Hs) ? ld4(hb + k) : ld4(ctxs + (k - a.Hs));
                acc = dot4(ld4(wr + k), x, acc);
            }
            acc = group_sum<32>(acc);
            if (ks == 0) lg[v] = acc + a.b_c[v];
        }
    }
    __syncthreads();
    // 6. log-softmax + argmax (first maximal index, as torch.topk/argmax)
    if (tid < 64) {
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
            int pick = best;
            if (a.y_mode == 2) {
                // decode_mode 2 (las_model.py:229-234): Categorical(raw_pred) is handed LOG-probabilities as `probs`; torch
                // renormalises them (p_v = logp_v / sum_v logp_v, i.e. proportional to |log p|) and draws one sample as
                // argmax_v p_v / q_v with q ~ Exp(1) (torch.multinomial's single-sample path).  The caller supplies q.
                float sl = 0.f;
                for (int v = lane; v < a.V; v += 64) sl += lg[v] - lse;
                sl = wave_sum(sl);
                float kbest = -INFINITY;
                pick = 0x7fffffff;
                for (int v = lane; v < a.V; v += 64) {
                    const float key = ((lg[v] - lse) / sl) / a.sample_noise[(long)b * a.V + v];
                    if (key > kbest) { kbest = key; pick = v; }
                }
#pragma unroll
                for (int mm = 32; mm >= 1; mm >>= 1) {
                    const float ok = __shfl_xor(kbest, mm);
                    const int op = __shfl_xor(pick, mm);
                    if (ok > kbest || (ok == kbest && op < pick)) { kbest = ok; pick = op; }
                }
            }
            // next-step input: log-probs (decode_mode 0, las_model.py:220-221), one-hot argmax (mode 1, :223-227) or one-hot sample
            for (int v = lane; v < a.V; v += 64)
                a.y_next[(long)b * a.ldy + v] = (a.y_mode == 0) ? (lg[v] - lse) : (v == pick ? 1.0f : 0.f);
        }
    }
}

static int attn_dims_ok(int Hs, int D, int Mq) {
    return Hs % 4 == 0 && D % 4 == 0 && Mq % 4 == 0 && D / 4 <= ATT_THREADS && Mq / 4 <= ATT_THREADS;
}

int attn_step_fwd(const AttnFwdArgs& a, hipStream_t stream) {
    LAS_REQUIRE(a.Tp <= ATT_MAX_TP, "attention length above kernel limit (4096 encoder frames)");
    LAS_REQUIRE(a.V <= ATT_MAX_V, "vocab above kernel limit");
    const int Mq = a.use_mlp ? a.M : a.Hs;
    LAS_REQUIRE(a.use_mlp || a.D == a.Hs, "attention without MLP needs decoder dim == feature dim");
    LAS_REQUIRE(attn_dims_ok(a.Hs, a.D, Mq), "attention dims must be multiples of 4 (and <= 4096)");
    const int NTQ = min(ATT_THREADS / (a.D / 4), 64);
    const size_t smem = sizeof(float) * (size_t)(Mq + a.Tp + 4 + a.D + a.V + 4 + (size_t)NTQ * a.D);
    LAS_REQUIRE(a.heads >= 1 && ((a.phases & 1) || a.ctx_in), "attention phases");
    hipLaunchKernelGGL(attn_step_fwd_kernel, dim3(a.B, (a.phases & 1) ? a.heads : 1), dim3(ATT_THREADS), smem, stream, a);
    LAS_LAUNCH_CHECK();
    return LAS_OK;
}

// ------------------------------------------------------------------------------------------------
// backward of one decode step's attention + character distribution (one workgroup per utterance)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(ATT_THREADS) void attn_step_bwd_kernel(AttnBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int Mq = a.use_mlp ? a.M : a.Hs;
    const int Q4 = Mq / 4;
    const int NTQ = min(ATT_THREADS / Q4, 64);
    // layout: dz[V(+pad)] | dctx[D] | de[Tp(+pad)] | dq[Mq] | part[NTQ*Mq]
    float* dz = smem;
    float* dctx = dz + ((a.V + 3) & ~3);
    float* de = dctx + a.D;
    float* dq = de + ((a.Tp + 3) & ~3);
    float* part = dq + Mq;
    const int b = blockIdx.x, hd = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
    const long ldq = a.ldq ? a.ldq : a.M;
    const float* __restrict__ w_phi = a.w_phi + (long)hd * a.M * a.Hs;     // this head's rows of phi

    const int K2 = a.Hs + a.D;
    float dh_keep[2] = {0.f, 0.f};                        // this thread's dh_top columns (k = tid, tid + 1024)
    if (!(a.phases & 1)) {
        // per-head attention part only: this head's slice of the dim_reduce input gradient is the context gradient
        for (int d = tid; d < a.D; d += ATT_THREADS) dctx[d] = a.dctx_in[(long)b * a.ld_dctx_in + (long)hd * a.D + d];
    } else if (a.dcat_pre) {
        // [dh_top | dctx] = dz W_c was computed for every step by one GEMM before the loop
        for (int k = tid, it = 0; k < K2; k += ATT_THREADS, ++it) {
            float acc = a.dcat_pre[(long)b * K2 + k];
            if (k >= a.Hs) {
                if (a.dctx_carry) acc += a.dctx_carry[(long)b * a.ldc + (k - a.Hs)];
                a.dctx_out[(long)b * a.D + (k - a.Hs)] = acc;
                dctx[k - a.Hs] = acc;
            } else if (it < 2) {
                dh_keep[it] = acc;
            }
        }
    } else {
    // 1. log-softmax backward: dz = g - exp(logp) * sum(g), g = dlogp (+ fed-back gradient in decode_mode 0)
    {
        float ps = 0.f;
        for (int v = lane; v < a.V; v += 64) {
            float g = a.dlogp[(long)b * a.V + v];
            if (a.dy_carry) g += a.dy_carry[(long)b * a.ldy + v];
            ps += g;
        }
        const float gsum = wave_sum(ps);                  // every wave computes it redundantly (V is tiny)
        if (tid < 64) {
            for (int v = lane; v < a.V; v += 64) {
                float g = a.dlogp[(long)b * a.V + v];
                if (a.dy_carry) g += a.dy_carry[(long)b * a.ldy + v];
                const float z = g - expf(a.logp[(long)b * a.V + v]) * gsum;
                dz[v] = z;
                a.dz_out[(long)b * a.V + v] = z;
            }
        }
    }
    __syncthreads();
    // 2. [dh_top | dctx] = dz W_c ; dctx += carry.  One column per thread, V independent coalesced loads.
    for (int k = tid, it = 0; k < K2; k += ATT_THREADS, ++it) {
        float acc = 0.f;
#pragma unroll 6
        for (int v = 0; v < a.V; ++v) acc = fmaf(dz[v], a.w_c[(long)v * K2 + k], acc);
        if (k >= a.Hs) {
            if (a.dctx_carry) acc += a.dctx_carry[(long)b * a.ldc + (k - a.Hs)];
            a.dctx_out[(long)b * a.D + (k - a.Hs)] = acc;
            dctx[k - a.Hs] = acc;
        } else if (it < 2) {
            dh_keep[it] = acc;
        }
    }
    }
    if (!(a.phases & 2)) {
        // character-distribution part only (multi-head): hand the decoder-state gradient part to the summing kernel
        for (int k = tid, it = 0; k < a.Hs && it < 2; k += ATT_THREADS, ++it) a.dh_top_out[(long)b * a.Hs + k] = dh_keep[it];
        return;
    }
    __syncthreads();
    // 3. da[t] = dctx . feat_t   (16 lanes per frame)
    {
        const float* fb = a.feat + (long)b * a.Tp * a.D;
        const int ds = tid & 15;
        for (int t = tid >> 4; t < a.Tp; t += ATT_THREADS / 16) {
            const float* fr = fb + (long)t * a.D;
            float acc = 0.f;
#pragma unroll 8
            for (int d = ds * 4; d < a.D; d += 64) acc = dot4(ld4(fr + d), ld4(dctx + d), acc);
            acc = group_sum<16>(acc);
            if (ds == 0) de[t] = acc;
        }
    }
    __syncthreads();
    // 4. softmax backward: de = a * (da - sum_t a_t da_t)   (statistic redundantly per wave)
    {
        const float* ab = a.att + (long)hd * a.att_hs + (long)b * a.Tp;
        float ls = 0.f;
        for (int t = lane; t < a.Tp; t += 64) ls = fmaf(ab[t], de[t], ls);
        const float sdot = wave_sum(ls);
        __syncthreads();                                  // everyone has read da before it is overwritten
        for (int t = tid; t < a.Tp; t += ATT_THREADS) {
            const float v = ab[t] * (de[t] - sdot);
            de[t] = v;
            a.de_out[(long)hd * a.att_hs + (long)b * a.Tp + t] = v;
        }
    }
    __syncthreads();
    // 5. dq[m] = sum_t de[t] keys[t][m] : (column group, time slice) per thread, LDS reduce over slices
    {
        const float* kb = a.keys + (long)b * a.Tp * Mq;
        const int m4 = tid % Q4, tq = tid / Q4;
        if (tq < NTQ) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
            for (int t = tq; t < a.Tp; t += NTQ) {
                const float w = de[t];
                const f32x4 f = ld4(kb + (long)t * Mq + m4 * 4);
                acc[0] = fmaf(w, f[0], acc[0]); acc[1] = fmaf(w, f[1], acc[1]);
                acc[2] = fmaf(w, f[2], acc[2]); acc[3] = fmaf(w, f[3], acc[3]);
            }
            *reinterpret_cast<f32x4*>(part + (long)tq * Mq + m4 * 4) = acc;
        }
    }
    __syncthreads();
    for (int m = tid; m < Mq; m += ATT_THREADS) {
        float v = 0.f;
        for (int q = 0; q < NTQ; ++q) v += part[(long)q * Mq + m];
        if (a.use_mlp) {
            if (a.relu) v *= act_grad(a.q[(long)b * ldq + (long)hd * a.M + m], a.relu);
            a.dqpre_out[(long)b * ldq + (long)hd * a.M + m] = v;
        }
        dq[m] = v;
    }
    __syncthreads();
    // 6. dh_top += dqpre W_phi  (or += dq when there is no MLP)
    for (int k = tid, it = 0; k < a.Hs; k += ATT_THREADS, ++it) {
        float acc = 0.f;
        if (it < 2) acc = dh_keep[it];
        else {  // columns beyond 2048: recompute the W_c part (never at the reference's sizes)
            for (int v = 0; v < a.V; ++v) acc = fmaf(dz[v], a.w_c[(long)v * K2 + k], acc);
        }
        if (a.use_mlp) {
#pragma unroll 8
            for (int m = 0; m < a.M; ++m) acc = fmaf(dq[m], w_phi[(long)m * a.Hs + k], acc);
        } else {
            acc += dq[k];
        }
        if (a.dh_top_out) a.dh_top_out[(long)hd * a.dh_hs + (long)b * a.Hs + k] = acc;
        if (a.pw.gates) cell_bwd_point(a.pw, b, k, a.Hs, acc);      // top LSTM layer's pointwise step, fused
    }
}

int attn_step_bwd(const AttnBwdArgs& a, hipStream_t stream) {
    LAS_REQUIRE(a.Tp <= ATT_MAX_TP && a.V <= ATT_MAX_V, "attention backward limits");
    const int Mq = a.use_mlp ? a.M : a.Hs;
    LAS_REQUIRE(attn_dims_ok(a.Hs, a.D, Mq), "attention dims must be multiples of 4 (and <= 4096)");
    const int NTQ = min(ATT_THREADS / (Mq / 4), 64);
    const size_t smem = sizeof(float) * (size_t)(a.V + 4 + a.D + a.Tp + 4 + Mq + (size_t)NTQ * Mq);
    LAS_REQUIRE(a.heads >= 1 && ((a.phases & 1) || a.dctx_in), "attention backward phases");
    LAS_REQUIRE(a.Hs <= 2 * ATT_THREADS || (a.phases == 3), "split phases need Hs <= 2048");
    hipLaunchKernelGGL(attn_step_bwd_kernel, dim3(a.B, (a.phases & 1) ? 1 : a.heads), dim3(ATT_THREADS), smem, stream, a);
    LAS_LAUNCH_CHECK();
    return LAS_OK;
}

}  // namespace las
