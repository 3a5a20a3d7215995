// Persistent teacher-forced Speller decode loop for gfx950: ONE launch runs all U decode steps.
//
// Replaces the per-step launch chain (lstm_cell_fwd x2 + attn_step_fwd, speller.hip) for the training forward of the
// reference's Speller.forward (model/las_model.py:186-238, teacher-forced branch :207-209) when the shapes allow it.
// The stepwise kernels stay the general path (decode mode 2, multi-head, very long T'); batches above 32 utterances are decoded in
// slices of 32 by the caller (las_speller_decode_batch, include/las_hip.h).
//
// Why: a decode step is a chain of dependent phases.  As three kernels each phase pays ~4 us of launch/drain floor plus ~4 us of
// L2->CU operand traffic (every workgroup re-reads the weights it used one step ago).  Three role sets live in this file:
//   * classic (CellRole / AttnRole; free-running decode modes 0 / 1, vocabularies above 32, T' beyond the PRE table): Hs/4 "cell"
//     workgroups own 4 hidden units of BOTH LSTM layers for the whole batch (weight rows in VGPRs, batch = M of v_mfma_f32_16x16x4_f32,
//     the 16 waves split K); split*B "attention" workgroups keep their slice of the listener features in VGPRs, the keys in LDS;
//   * PRE, round 3 (CellPreRole / AttnPreRole; teacher forcing — the training case): the context enters the bottom cell as
//     sum_t a_t (W_ctx feat_t) from a register-resident slice of P = feat . W_ctx^T, and the ATTENTION workgroups apply the bottom cell
//     themselves (their lanes hold the four gates of 128 units after the reduction); the cell workgroups run the top cell and prepare
//     R0 = W_hh0 h0 + W_y y + b a whole attention phase ahead.  TWO cross-CU hops per decode step; resident products on the bf16
//     matrix pipe (exact three-way operand split, persist_common.h).
//   * phases hand data over through per-step slabs that the host pre-fills with a sentinel bit pattern (0xFFFFFFFF, never produced by
//     the kernels): producers write whole cache lines with agent-scope (write-through) stores; one wave of a consumer workgroup
//     watches one dword per producer workgroup, then every wave reads its tile with ordinary (L2-shared) loads, multiplies, and
//     checks every consumed word against the sentinel — the data is its own flag, no counters, no fences, no epochs.
// All spins are bounded and report through the device error word (results are then invalid, never a hang).
#include "las_common.h"
#include "las_kernels.h"
#include "persist_common.h"
#include "options.h"
#include <algorithm>

namespace las {

namespace {
constexpr int PS_NI = 7;           // listener frames held per attention lane
}  // namespace

struct PersistArgs {
    const float* w0p; long ldw0; int Vp;      // [W_y | 0 | W_ctx] shadow of W_ih0 (ld = Vp + Hs)
    const float* w_hh0; const float* w_ih1; const float* w_hh1;
    const float* b_ih0; const float* b_hh0; const float* b_ih1; const float* b_hh1;
    const float* w_phi; const float* b_phi;
    const float* feat; const float* keys; float* y_all;
    float* ctx_all; float* h_all; float* c_all; float* gates_all; float* q_all; float* att;
    float* hx;                     // hand-off copy of h: [layer][step][unit tile Hs/4][row 32][4], sentinel-prefilled
    // free-running decode (mode 1: feed the one-hot arg-max, reference decode_mode 1, las_model.py:223-227;
    // mode 2: feed the log-probabilities, decode_mode 0, :220-221; mode 0: teacher forcing)
    int mode, V;
    const float* w_c; const float* b_c;   // (V, 2Hs), (V)
    float* logp; int* argmax_out;          // (U,B,V), (U,B) or null
    float* lgx;                            // [U][B][split][32] partial logits of the attention workgroups, sentinel-prefilled
    const float* pctx; float* gx;          // PRE variant: feat . W_ctx^T (B*Tp, 4Hs) and its per-step weighted sums [U][B][4Hs] (stash for the backward)
    float* r0x;                            // PRE variant: layer-0 gates minus the context half, [U][Hs/4][32][16] (cell -> attention, sentinel-prefilled)
    const float* yw;                       // PRE variant: y_s W_y^T + b_ih0 + b_hh0 for every step, (U*B, 4Hs), columns in unit*4 + gate order
    // PRE variant, free-running (mode 1): Q^T = W_c[:, Hs:] feat^T per utterance (B, 32, Tp); W_y^T in the permuted gate-column order
    // (Vp, 4Hs); partial logits W_c[:, units] h1 of every cell workgroup, [U][Hs/4][16 utterances][32], sentinel-prefilled
    const float* qct; const float* wyT; float* plx;
    // PRE variant, multi-head (heads = NH > 1, teacher forcing; las_model.py:298-314): pctx is (B*Tp, NH*4Hs) — head h's block is
    // feat . (W_ctx W_dr[:, h])^T, head 0's carrying W_ctx b_dr as a bias (the attention weights sum to 1) — gx is [U][B][NH][4Hs] and doubles as
    // the heads' exchange slab (sentinel-prefilled, agent-scope stores); p0 = feat[:, 0] . W_ctx^T (B, 4Hs) is the step-0 context product
    int NH; const float* p0;
    float* ex;                             // PRE variant, Hs = 256 with 16 workgroups per utterance: the frame slices' energies, [U][B][16][64], sentinel-prefilled
    int B, Tp, U, relu;
    int split;                     // attention workgroups per utterance (each owns D/split context columns)
    unsigned* err;
    unsigned long long* trace;     // profiling aid (tools/ubench_persist_trace.py): shader-clock stamps of workgroup 0 of each role
};

#define PS_STAMP(role, s, k) do { if (a.trace && first_wg && threadIdx.x == 0) a.trace[((size_t)(role) * a.U + (s)) * 8 + (k)] = wall_clock64(); } while (0)

// ------------------------------------------------------------------------------------------------ cell workgroups
// PRE ("pre-multiplied context", teacher forcing only): layer 0's context half W_ctx . ctx_{s-1} arrives already multiplied —
// the attention workgroups publish sum_t a_t (W_ctx feat_t) from a register-resident slice of feat . W_ctx^T — so layer 0 is
// off the MFMA chain: its recurrent / label halves are reduced ahead of time and the cell lanes only add 16 bytes per
// (utterance, unit) that they poll themselves.
template <int HS, bool GREEDY>
struct CellRole {
    static constexpr int NF = HS / 256;                 // 16-wide k-blocks of an Hs-wide operand per wave
    static constexpr int RLD = 20;                      // row stride of a partial tile: 16 columns + pad, 16-byte aligned
    static constexpr int RED = PS_NW * 2 * 16 * RLD;    // floats of the per-wave partial tile buffer
    static constexpr int GREEDY_FLOATS = 32 * 32 + 32 * 32 + 32 + 16 * 32;    // logits, fed-back input, arg-max, W_y rows
    static constexpr int LDS_FLOATS = 2 * RED + 2 * 4 * 128 + 4 + GREEDY_FLOATS;   // reduction buffer per layer + biases + canary flags

    // x tile of this wave: rows = utterances (two 16-row M-tiles), columns = its k-blocks.
    // KIND 0: tile of h (each 4-column group comes from one cell workgroup); KIND 1: tile of the context (each row part
    // comes from one attention workgroup).
    //  1. canary (wg_canary_wait): the first wave(s) watch one agent-scope dword per producer workgroup;
    //  2. the tile itself with PLAIN loads: the 16 cell workgroups of an XCD share them through its L2 instead of each
    //     pulling 64 KB over the fabric (agent-scope loads bypass the L2: 8.4 MB per phase, measured ~3.5 us);
    //  3. every consumed word is still checked against the sentinel; a slot that raced ahead of its producer (its stale
    //     line now sits in this XCD's L2) is re-read with agent-scope loads until it is complete.
    struct TileAddr {            // element offsets of this lane inside a (B, HS) slab; identical for every tile
        // unsigned 32-bit element offsets from a wave-uniform base: the loads use the SGPR-base + VGPR-offset form
        // instead of one hoisted 64-bit address pair per slot
        unsigned x[NF][2];       // (bytes) its float4 of k-block f, M-tile mt (row-major slab: the context)
        unsigned hx[NF][2];      // the same float4 in the tiled hand-off copy of h
        bool ok[2];              // row < B
        unsigned canary[2];      // [KIND]: the producer dword this lane watches when its wave is a canary wave
        bool cact[2];            // ... and whether that producer exists
        int npw1;                // canary waves of a context tile: ceil(split*B / 64)
    };
    static __device__ __forceinline__ TileAddr tile_addr(int B, int split, int wave, int lane) {
        TileAddr t;
        const int r = lane & 15, kq = lane >> 4;
        const int kwave = wave;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            t.ok[mt] = mt * 16 + r < B;
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                t.x[f][mt] = 4u * ((t.ok[mt] ? mt * 16 + r : 0) * HS + (kwave * NF + f) * 16 + kq * 4);
                t.hx[f][mt] = 4u * ((((kwave * NF + f) * 4 + kq) * 32 + (t.ok[mt] ? mt * 16 + r : 0)) * 4);
            }
        }
        {
            // KIND 0: producer p = cell workgroup p (its 4 units of every row); watch the last row's last unit
            const int p0 = wave * 64 + lane;
            t.cact[0] = p0 < HS / 4;
            t.canary[0] = 4u * (((t.cact[0] ? p0 : 0) * 32 + (B - 1)) * 4 + 3);
            // KIND 1: producer p = attention workgroup (utterance p / split, column part p % split)
            t.npw1 = (split * B + 63) / 64;
            t.cact[1] = p0 < split * B;
            const int pb = t.cact[1] ? p0 / split : 0, pp = p0 % split;
            t.canary[1] = 4u * (pb * HS + (pp + 1) * (HS / split) - 1);
        }
        return t;
    }
    // acc += tile * W.  The product is started speculatively as the tile's loads land (load and MFMA time overlap); the
    // sentinel check comes afterwards and a tile that was not complete is repaired and multiplied again.
    template <int KIND, class WT>
    static __device__ __forceinline__ int poll_mul(const float* base, const TileAddr& t, f32x4 (&x)[NF][2],
                                                   const WT& W, f32x4 (&acc)[2],
                                                   unsigned* err, volatile unsigned* flags, unsigned& ep) {
        unsigned spins = 0;
        int slow = 0;
        {
            constexpr int K = KIND == 0 ? 0 : 1;
            const int npw = KIND == 0 ? (HS / 4 + 63) / 64 : t.npw1;      // canary waves: one lane per producer workgroup
            const unsigned* cp = reinterpret_cast<const unsigned*>(at_bytes(base, opaque(t.canary[K])));
            wg_canary_wait(flags, ++ep, npw, __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), threadIdx.x & 63, cp, t.cact[K], err, 0xDEAD0011u);
        }
        asm volatile("" ::: "memory");
        bool need[NF][2];
        bool bad = false;
        // ordinary (L2-cacheable) loads; rows past B keep the sentinel for ever and only feed output rows nobody reads
#pragma unroll
        for (int f = 0; f < NF; ++f)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
                x[f][mt] = *reinterpret_cast<const f32x4*>(at_bytes(base, opaque(KIND == 0 ? t.hx[f][mt] : t.x[f][mt])));
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        f32x4 p[2] = {zero, zero};
        mfma_tile(x, W, p);
#pragma unroll
        for (int f = 0; f < NF; ++f)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                need[f][mt] = __any(t.ok[mt] && has_sentinel(x[f][mt]));
                bad |= need[f][mt];
            }
        asm volatile("" ::: "memory");
        const bool redo = bad;
        // slow path: only the (wave-uniform) slots in which some lane still saw the sentinel, with L2-bypassing loads
        while (bad) {
            if (spin_expired(spins, err, 0xDEAD0012u)) break;
            bad = false;
            ++slow;
#pragma unroll
            for (int f = 0; f < NF; ++f)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    if (need[f][mt]) {
                        x[f][mt] = ld4_agent(at_bytes(base, opaque(KIND == 0 ? t.hx[f][mt] : t.x[f][mt])));
                        need[f][mt] = __any(t.ok[mt] && has_sentinel(x[f][mt]));
                        bad |= need[f][mt];
                    }
                }
        }
        if (redo) { p[0] = p[1] = zero; mfma_tile(x, W, p); }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[mt][i] += p[mt][i];
        return slow;
    }

    static __device__ __forceinline__ void mfma_tile(const f32x4 (&x)[NF][2], const float (&w)[NF][4], f32x4 (&acc)[2]) {
#pragma unroll
        for (int f = 0; f < NF; ++f)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
                    acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[f][mt][e], w[f][e], acc[mt], 0, 0, 0);
    }
    // the same product on the bf16 matrix pipe (persist_common.h: exact three-way operand split, six partial products): the
    // lane's 4 NF k-slots of every k-block it owns form ONE 16x16x(16 NF) operand; the weights are split once per launch
    static constexpr int NK = 4 * NF;
    using WSplit = PsPlanes<NK>;
    static __device__ __forceinline__ WSplit split_w(const float (&w)[NF][4]) {
        float v[NK];
#pragma unroll
        for (int f = 0; f < NF; ++f)
#pragma unroll
            for (int e = 0; e < 4; ++e) v[f * 4 + e] = w[f][e];
        return ps_split<NK>(v);
    }
    static __device__ __forceinline__ void mfma_tile(const f32x4 (&x)[NF][2], const WSplit& w, f32x4 (&acc)[2]) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            float v[NK];
#pragma unroll
            for (int f = 0; f < NF; ++f)
#pragma unroll
                for (int e = 0; e < 4; ++e) v[f * 4 + e] = x[f][mt][e];
            acc[mt] = ps_mfma6<NK>(ps_split<NK>(v), w, acc[mt]);
        }
    }

    static __device__ void run(const PersistArgs& a, float* smem) {
        const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave: scalar register
        const int r = lane & 15, kq = lane >> 4;
        const int j0 = blockIdx.x * 4;
        const bool first_wg = blockIdx.x == 0;
        const int B = a.B, U = a.U;
        const long wrow = (long)(r >> 2) * HS + j0 + (r & 3);       // tile column n = gate*4 + unit
        const size_t sH = (size_t)B * HS;
        constexpr size_t HXS = (size_t)32 * HS;          // floats of one tiled hand-off slab

        // ---- resident weights (MFMA B operands): W[wrow][16*blk + 4*kq + e]
        float Wc0[NF][4], Wh0[NF][4], Wi1[NF][4], Wh1[NF][4], Wy[1][4];
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            const int k = (wave * NF + f) * 16 + kq * 4;
            const f32x4 c0 = ld4p(a.w0p + wrow * a.ldw0 + a.Vp + k);
            const f32x4 h0 = ld4p(a.w_hh0 + wrow * HS + k);
            const f32x4 i1 = ld4p(a.w_ih1 + wrow * HS + k);
            const f32x4 h1 = ld4p(a.w_hh1 + wrow * HS + k);
#pragma unroll
            for (int e = 0; e < 4; ++e) { Wc0[f][e] = c0[e]; Wh0[f][e] = h0[e]; Wi1[f][e] = i1[e]; Wh1[f][e] = h1[e]; }
        }
        const bool ywave = wave * 16 < a.Vp;            // label columns: one k-block per wave (Vp <= 256)
        {
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            const f32x4 y = ywave ? ld4p(a.w0p + wrow * a.ldw0 + wave * 16 + kq * 4) : z;
#pragma unroll
            for (int e = 0; e < 4; ++e) Wy[0][e] = y[e];
        }
        // ---- cell non-linearity lanes: (utterance pb, unit pu)
        const int pb = tid >> 2, pu = tid & 3;
        const bool pw = tid < 128 && pb < B;
        const int pmt = (pb >> 4) & 1, pm = pb & 15;
        float c0 = 0.f, c1 = 0.f;
        float* bias = smem + 2 * RED;                 // [layer][gate][cell lane]: registers are for the weights
        if (tid < 128) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int row = g * HS + j0 + pu;
                bias[g * 128 + tid] = a.b_ih0[row] + a.b_hh0[row];
                bias[(4 + g) * 128 + tid] = a.b_ih1[row] + a.b_hh1[row];
            }
        }
        // ---- free-running decode state (LDS): summed logits of the previous step, the input fed back, its arg-max, W_y rows
        const int mode = GREEDY ? a.mode : 0;        // compile-time 0 in the teacher-forced instantiation
        float* lsum = smem + 2 * RED + 2 * 4 * 128 + 4;
        float* yv = lsum + 32 * 32;
        int* amax = reinterpret_cast<int*>(yv + 32 * 32);
        float* wy = yv + 32 * 32 + 32;
        if (GREEDY) {
            if (tid < 512) {
                const int n = tid >> 5, v = tid & 31;
                wy[n * 32 + v] = v < a.Vp ? a.w0p[((long)(n >> 2) * HS + j0 + (n & 3)) * a.ldw0 + v] : 0.f;
            }
            yv[tid] = (tid & 31) == 0 ? 1.f : 0.f;       // step 0 is fed <sos> = one-hot(0) (las_model.py:193-195)
            if (tid < 32) amax[tid] = 0;
        }
        // the attention workgroups' partial logits of step sp -> lsum (threads 768..1023: 8 float4 per utterance)
        auto collect_logits = [&](int sp) {
            if (tid >= 768) {
                const int t = tid - 768, b = t >> 3, q = t & 7;
                if (b < B) {
                    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                    for (int p = 0; p < a.split; ++p) {
                        const float* src = a.lgx + (((size_t)sp * B + b) * a.split + p) * 32 + q * 4;
                        unsigned spins = 0;
                        f32x4 v;
                        for (;;) {
                            v = ld4_agent(src);
                            if (!has_sentinel(v)) break;
                            if (spin_expired(spins, a.err, 0xDEAD0014u)) break;
                        }
                        acc[0] += v[0]; acc[1] += v[1]; acc[2] += v[2]; acc[3] += v[3];
                    }
                    *reinterpret_cast<f32x4*>(lsum + b * 32 + q * 4) = acc;
                }
            }
        };
        // logits of step sp (in lsum) -> arg-max / log-probabilities and the next input (LDS): one 32-lane group per
        // utterance.  The outputs of utterance b (log-probs, arg-max, the fed-back input for the backward pass) are
        // written by cell workgroup b, so that no single workgroup carries all of that off-chip traffic.
        auto choose_next = [&](int sp) {
            const int b = tid >> 5, v = tid & 31;
            const float val = v < a.V ? lsum[b * 32 + v] : -INFINITY;
            float m = val;
            m = fmaxf(m, dpp_f(m, 0)); m = fmaxf(m, dpp_f(m, 1)); m = fmaxf(m, dpp_f(m, 2)); m = fmaxf(m, dpp_f(m, 3));
            m = fmaxf(m, __shfl_xor(m, 16));
            int best = val == m ? v : 64;                                                 // first maximal index
            best = min(best, __shfl_xor(best, 1)); best = min(best, __shfl_xor(best, 2)); best = min(best, __shfl_xor(best, 4));
            best = min(best, __shfl_xor(best, 8)); best = min(best, __shfl_xor(best, 16));
            const bool writer = (int)blockIdx.x == b;
            float lp = 0.f;
            if (mode == 2 || writer) {
                float se = v < a.V ? expf(val - m) : 0.f;
                se = gsum<16>(se);
                se += __shfl_xor(se, 16);
                lp = val - (m + logf(se));
            }
            const float y = v < a.V ? (mode == 1 ? (v == best ? 1.f : 0.f) : lp) : 0.f;
            if (b < B) {
                yv[b * 32 + v] = y;
                if (v == 0) amax[b] = best;
                if (writer) {
                    // (addresses from 32-bit offsets the optimiser cannot hoist: hoisted 64-bit pointers spilled here, and every spill reload
                    // waits on vmcnt(0), i.e. on the acknowledgement of the store in front of it — three store round trips per step on the chain
                    // of the writing workgroup)
                    if (v < a.V) *at_bytes(a.logp, opaque(4u * (unsigned)((sp * B + b) * a.V + v))) = lp;
                    if (v < a.Vp) *at_bytes(a.y_all, opaque(4u * (unsigned)(((sp + 1) * B + b) * a.Vp + v))) = y;
                    if (v == 0 && a.argmax_out) *at_bytes(a.argmax_out, opaque(4u * (unsigned)(sp * B + b))) = best;
                }
            }
        };
        const int rcol = (r & 3) * 4 + (r >> 2);        // tile column (gate*4 + unit) stored as unit*4 + gate: a cell lane
                                                        // reads the four gates of its unit as one 16-byte LDS word
        auto load_y = [&](int s, f32x4 (&y)[1][2]) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const bool ok = ywave && mt * 16 + r < B;
                const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                const f32x4 v = ld4p(at_bytes(a.y_all + (size_t)s * B * a.Vp,
                                              opaque(4u * ((ok ? mt * 16 + r : 0) * a.Vp + (ok ? wave * 16 + kq * 4 : 0)))));
                y[0][mt] = ok ? v : z;
            }
        };
        // gates of one layer: reduce the 16 waves' partial tiles (-> the cell lanes' registers) ...
        auto reduce_gates = [&](const f32x4 (&acc)[2], int layer, int s) -> f32x4 {
            // one buffer per layer: the single barrier below then also separates this buffer's readers from its next writers
            float (*red)[2][16][RLD] = reinterpret_cast<float (*)[2][16][RLD]>(smem + layer * RED);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int i = 0; i < 4; ++i) red[wave][mt][kq * 4 + i][rcol] = acc[mt][i];
            lds_barrier();
            if (layer == 1) PS_STAMP(1, s, 6); else PS_STAMP(1, s, 7);       // (cell wg 0) all 16 waves' products are in
            f32x4 g4 = {0.f, 0.f, 0.f, 0.f};
            if (pw) {
#pragma unroll
                for (int w = 0; w < PS_NW; ++w) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(&red[w][pmt][pm][pu * 4]);
                    g4[0] += v[0]; g4[1] += v[1]; g4[2] += v[2]; g4[3] += v[3];
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) g4[g] += bias[(layer * 4 + g) * 128 + tid];
            }
            return g4;
        };
        // ... apply the cell, publish h, stash c / gates
        auto cell = [&](f32x4 g4, float& c, int layer, int s) {
            if (pw) {
                if (GREEDY && layer == 0) {
                    // free-running: the label half of the gates comes from what the previous step produced
                    if (mode == 1) {
                        const int am = amax[pb];
#pragma unroll
                        for (int g = 0; g < 4; ++g) g4[g] += wy[(g * 4 + pu) * 32 + am];
                    } else {
                        for (int v = 0; v < a.V; ++v) {
                            const float yvv = yv[pb * 32 + v];
#pragma unroll
                            for (int g = 0; g < 4; ++g) g4[g] = fmaf(wy[(g * 4 + pu) * 32 + v], yvv, g4[g]);
                        }
                    }
                }
                const float ig = sigmoidf_acc(g4[0]), fg = sigmoidf_acc(g4[1]), gg = tanhf_acc(g4[2]), og = sigmoidf_acc(g4[3]);
                c = fg * c + ig * gg;
                const float h = og * tanhf_acc(c);
                const size_t slab = ((size_t)layer * U + s) * sH;           // wave-uniform
                // lane offsets re-derived from the thread id in every step: as loop invariants they would be kept live (and spilled)
                const unsigned tq = opaque((unsigned)tid), pbq = tq >> 2, puq = tq & 3;
                const unsigned o = 4u * (pbq * HS + j0 + puq);
                st1_agent(at_bytes(a.hx + ((size_t)layer * U + s) * HXS, 4u * (((unsigned)blockIdx.x * 32 + pbq) * 4 + puq)), h);
                *at_bytes(a.h_all + slab, o) = h;
                *at_bytes(a.c_all + slab, o) = c;
                float* go = at_bytes(a.gates_all + 4 * slab, 4u * (pbq * 4 * HS + j0 + puq));
                go[0] = ig; go[HS] = fg; go[2 * HS] = gg; go[3 * HS] = og;
            }
        };
        auto finish = [&](const f32x4 (&acc)[2], float& c, int layer, int s) { cell(reduce_gates(acc, layer, s), c, layer, s); };

        // ---- prologue: label half of step 0's layer-0 gates
        f32x4 accR0[2], accR1[2];
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        accR0[0] = accR0[1] = accR1[0] = accR1[1] = zero;
        if (!GREEDY) {
            f32x4 y[1][2];
            load_y(0, y);
            if (ywave) CellRole<256, GREEDY>::mfma_tile(y, Wy, accR0);
        }
        f32x4 x[NF][2];
        const TileAddr ta = tile_addr(B, a.split, wave, lane);
        volatile unsigned* cflags = reinterpret_cast<volatile unsigned*>(smem + 2 * RED + 2 * 4 * 128);
        unsigned cep = 0;
        if (tid < 4) cflags[tid] = 0u;
        lds_barrier();
        int nslow[3] = {0, 0, 0};       // slow-path rounds per tile kind (reported through the trace buffer)
        for (int s = 0; s < U; ++s) {
            // next step's labels: issued now, consumed after layer 1 (plain load, its latency is off the chain)
            f32x4 ynext[1][2];
            if (!GREEDY && s + 1 < U) load_y(s + 1, ynext);
            // layer 0: gates = W_ctx ctx_{s-1} + [W_hh0 h0_{s-1} + W_y y_s  (accumulated ahead)]
            PS_STAMP(0, s, 0);
            nslow[1] += poll_mul<1>(a.ctx_all + (size_t)s * B * HS, ta, x, Wc0, accR0, a.err, cflags, cep);
            if (GREEDY && s > 0) {      // the logits of step s-1 land just after its context: fetched behind the product
                collect_logits(s - 1);
                lds_barrier();
                choose_next(s - 1);
            }
            PS_STAMP(0, s, 1);
            finish(accR0, c0, 0, s);
            PS_STAMP(0, s, 2);
            // layer 1: gates = W_ih1 h0_s + W_hh1 h1_{s-1}
            nslow[0] += poll_mul<0>(a.hx + (size_t)s * HXS, ta, x, Wi1, accR1, a.err, cflags, cep);
            PS_STAMP(0, s, 4);
            finish(accR1, c1, 1, s);
            PS_STAMP(0, s, 5);
            if (s + 1 == U) break;
            // off the critical chain (the attention workgroups are working now): recurrent / label halves of layer 0
            accR0[0] = accR0[1] = zero;
            mfma_tile(x, Wh0, accR0);                              // the h0_s tile is still in registers
            if (!GREEDY && ywave) CellRole<256, GREEDY>::mfma_tile(ynext, Wy, accR0);
            PS_STAMP(0, s, 6);
            // ... and of layer 1, as soon as every cell workgroup's h1_s has arrived (still inside the attention window)
            accR1[0] = accR1[1] = zero;
            nslow[2] += poll_mul<0>(a.hx + ((size_t)U + s) * HXS, ta, x, Wh1, accR1, a.err, cflags, cep);
            PS_STAMP(0, s, 7);
        }
        if (GREEDY && (int)blockIdx.x < B) {        // the last step's character distribution (writer workgroups only)
            lds_barrier();
            collect_logits(U - 1);
            lds_barrier();
            choose_next(U - 1);
        }
        if (a.trace && first_wg && tid == 0)
            for (int k = 0; k < 3; ++k) a.trace[((size_t)a.U + k) * 8 + 7] = (unsigned long long)nslow[k];
    }
};

// ------------------------------------------------------------------------------------------------ cell workgroups, PRE variant
// Teacher forcing with the pre-multiplied context (AttnPreRole below): the attention workgroups own the BOTTOM cell, this role runs
// the TOP cell and prepares what the bottom cell needs besides the context term,
//     R0_s = W_hh0 h0_{s-1} + W_y y_s + b_ih0 + b_hh0            (the "recurrent / label / bias part" of the bottom-layer gates),
// published per workgroup as a tile [16 utterances][8 units x 4 gates] (whole 128-byte lines) a whole attention phase ahead of its
// use.  Chain per decode step:  attention + bottom cell -> h0_s slab -> W_ih1 product, reduce, top cell -> h1_s : TWO cross-CU hops.
//   * partition: a workgroup owns EIGHT hidden units (two 16-column gate tiles) for SIXTEEN utterances (one 16-row M-tile), i.e. it
//     pulls a (16, Hs) tile of h0_s per step — 32 KB.  The first version of this role (4 units x 32 utterances, 64 KB per step) spent
//     0.6 us longer in the product: a CU takes in freshly written lines at ~32 B/clk, so bytes-in per workgroup, not matrix-pipe
//     work (identical: two 16x16 output tiles either way), is what the last of the 16 waves waits for;
//   * every product runs on the bf16 matrix pipe (persist_common.h: exact three-way split, six partial products, fp32 accumulation).
//     W_ih1 (on the chain) lives in registers as bf16 planes; W_hh0 (R0: needed by the attention workgroups before their weighted sum
//     ends) as bf16 planes in LDS; W_hh1 (needed last) as fp32 in registers, split again in every step — three resident split
//     matrices do not fit the 128 registers a lane has at 1024 threads;
//   * the label half W_y y_s + biases comes from ONE GEMM before the launch (a.yw) and is staged in LDS a step ahead;
//   * one LDS exchange buffer serves both layers in turn: top-layer gates -> barrier -> waves 0-1 reduce / apply / publish the top
//     cell; then R0's product (same h0_s tile, still in registers) -> barrier -> waves 2-3 reduce and publish R0.
// GREEDY (free-running decode, mode 1): the workgroup also publishes its units' share of the character-distribution logits,
//     PL_s[utterance][v] = sum_{u in its 8 units} W_c[v][u] h1_s[utterance][u],
// one 2 KB tile per step next to h1_s; the attention workgroups add the 64 tiles of their utterance to the context share they compute
// themselves and pick the arg-max (AttnPreRole) — the label half of R0 then holds the biases only (the caller's yw is built from y = 0).
template <int HS, bool GREEDY = false>
struct CellPreRole {
    static constexpr int NF = HS / 256;                 // 16-wide k-blocks of an Hs-wide operand per wave
    static constexpr int NK = 4 * NF;
    static constexpr int RLD = 36;                      // row stride of a partial tile: 32 columns + pad, 16-byte aligned
    static constexpr int RED = PS_NW * 16 * RLD;
    static constexpr int WPL = 2 * 3 * (NK / 2) * PS_THREADS;      // dwords of one matrix' bf16 planes (two column tiles)
    static constexpr int LDS_BASE = RED + 4 * 128 + 4 + 2 * 128 * 4 + WPL + 32;
    static constexpr int LDS_FLOATS = LDS_BASE + (GREEDY ? 128 + 32 * 8 : 0);      // + h1 of the step, W_c columns of the 8 units
    static constexpr int NWG = 2 * (HS / 8);
    using WSplit = PsPlanes<NK>;

    struct TileAddr {
        unsigned x[NF], hx[NF];      // (bytes) this lane's float4 of k-block f in the row-major h0 slab / the tiled h1 slab
        bool ok;                     // its row is an utterance
        unsigned canary[2];
        bool cact[2];
        int npw1;
    };
    // acc[nt] += tile . W[nt]   (see CellRole::poll_mul: canary, plain loads, speculative product, sentinel check, repair)
    template <int KIND, class WF>
    static __device__ __forceinline__ int poll_mul(const float* base, const TileAddr& t, f32x4 (&x)[NF], const WF& wf, f32x4 (&acc)[2],
                                                   unsigned* err, volatile unsigned* flags, unsigned& ep) {
        unsigned spins = 0;
        int slow = 0;
        {
            constexpr int K = KIND == 0 ? 0 : 1;
            const int npw = KIND == 0 ? 1 : t.npw1;
            const unsigned* cp = reinterpret_cast<const unsigned*>(at_bytes(base, opaque(t.canary[K])));
            wg_canary_wait(flags, ++ep, npw, __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), threadIdx.x & 63, cp, t.cact[K], err, 0xDEAD0011u);
        }
        asm volatile("" ::: "memory");
#pragma unroll
        for (int f = 0; f < NF; ++f) x[f] = *reinterpret_cast<const f32x4*>(at_bytes(base, opaque(KIND == 0 ? t.hx[f] : t.x[f])));
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        f32x4 p[2] = {zero, zero};
        mul(x, wf, p);
        bool need[NF];
        bool bad = false;
#pragma unroll
        for (int f = 0; f < NF; ++f) { need[f] = __any(t.ok && has_sentinel(x[f])); bad |= need[f]; }
        asm volatile("" ::: "memory");
        const bool redo = bad;
        while (bad) {
            if (spin_expired(spins, err, 0xDEAD0012u)) break;
            bad = false;
            ++slow;
#pragma unroll
            for (int f = 0; f < NF; ++f)
                if (need[f]) {
                    x[f] = ld4_agent(at_bytes(base, opaque(KIND == 0 ? t.hx[f] : t.x[f])));
                    need[f] = __any(t.ok && has_sentinel(x[f]));
                    bad |= need[f];
                }
        }
        if (redo) { p[0] = p[1] = zero; mul(x, wf, p); }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[nt][i] += p[nt][i];
        return slow;
    }
    template <class WF>
    static __device__ __forceinline__ void mul(const f32x4 (&x)[NF], const WF& wf, f32x4 (&acc)[2]) {
        float v[NK];
#pragma unroll
        for (int f = 0; f < NF; ++f)
#pragma unroll
            for (int e = 0; e < 4; ++e) v[f * 4 + e] = x[f][e];
        const WSplit xs = ps_split<NK>(v);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) acc[nt] = ps_mfma6<NK>(xs, wf(nt), acc[nt]);
    }

    static __device__ void run(const PersistArgs& a, float* smem) {
        const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int r = lane & 15, kq = lane >> 4;
        const int j8 = blockIdx.x >> 1, mh = blockIdx.x & 1;           // its 8 units start at 8 j8; its utterances at 16 mh
        const bool first_wg = blockIdx.x == 0;
        const int B = a.B, U = a.U;
        const int nrows = min(16, B - 16 * mh);
        if (nrows <= 0) return;                                         // B <= 16: the second half of the workgroups has no utterances
        const size_t sH = (size_t)B * HS;
        constexpr size_t HXS = (size_t)32 * HS;
        float* bias = smem + RED;                     // [gate][cell lane]: b_ih1 + b_hh1
        volatile unsigned* cflags = reinterpret_cast<volatile unsigned*>(smem + RED + 4 * 128);
        float* ywl = smem + RED + 4 * 128 + 4;        // [step parity][R0 lane][4 gates]
        unsigned* wl = reinterpret_cast<unsigned*>(ywl + 2 * 128 * 4);     // W_hh0: [column tile][plane][thread][NK / 2]
        float* hl = smem + LDS_BASE;                  // GREEDY: h1_s of the 128 cell lanes, [utterance][unit]
        float* wcl = hl + 128;                        // GREEDY: W_c[v][8 j8 + u], [32][8]
        if (GREEDY && tid < 256) {
            const int v = tid >> 3, u = tid & 7;
            wcl[tid] = v < a.V ? a.w_c[(size_t)v * 2 * HS + j8 * 8 + u] : 0.f;
        }

        // ---- weights: column n of tile nt = gate n / 4 of unit 8 j8 + 4 nt + n % 4;  W[row][16*blk + 4*kq + e]
        WSplit Si1[2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const long wrow = (long)(r >> 2) * HS + j8 * 8 + nt * 4 + (r & 3);
            float wi1[NK], wh0[NK];
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                const int k = (wave * NF + f) * 16 + kq * 4;
                const f32x4 h0 = ld4p(a.w_hh0 + wrow * HS + k), i1 = ld4p(a.w_ih1 + wrow * HS + k);
#pragma unroll
                for (int e = 0; e < 4; ++e) { wh0[f * 4 + e] = h0[e]; wi1[f * 4 + e] = i1[e]; }
            }
            Si1[nt] = ps_split<NK>(wi1);
            const WSplit S0 = ps_split<NK>(wh0);
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                for (int i = 0; i < NK / 2; ++i) wl[((nt * 3 + pl) * PS_THREADS + tid) * (NK / 2) + i] = S0.p[pl][i];
        }
        auto w_i1 = [&](int nt) -> const WSplit& { return Si1[nt]; };
        auto w_h0 = [&](int nt) {            // re-read per step, not kept in registers
            WSplit w;
            const unsigned* src = at_bytes(wl + nt * 3 * PS_THREADS * (NK / 2), opaque(4u * (unsigned)tid * (NK / 2)));
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                for (int i = 0; i < NK / 2; ++i) w.p[pl][i] = src[pl * PS_THREADS * (NK / 2) + i];
            return w;
        };
        // W_hh1: 64 KB of fp32 per workgroup that neither the registers nor the LDS have room for — re-read from the L2 in every step
        // (its product waits for h1_s anyway; the offsets are opaque so that the loads are not hoisted back into registers)
        auto w_h1 = [&](int nt) {
            const unsigned ln = opaque((unsigned)tid) & 63u, rr = ln & 15u, kk = ln >> 4;
            const float* src = a.w_hh1 + ((size_t)(j8 * 8 + nt * 4) * HS + (size_t)wave * NF * 16);
            float v[NK];
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                const f32x4 h1 = ld4p(at_bytes(src, 4u * (((rr >> 2) * HS + (rr & 3)) * HS + f * 16 + kk * 4)));
#pragma unroll
                for (int e = 0; e < 4; ++e) v[f * 4 + e] = h1[e];
            }
            return ps_split<NK>(v);
        };
        if (tid < 128) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int row = g * HS + j8 * 8 + (tid & 7);
                bias[g * 128 + tid] = a.b_ih1[row] + a.b_hh1[row];
            }
        }
        const int rcol = (r & 3) * 4 + (r >> 2);        // tile column (gate*4 + unit) stored as unit*4 + gate
        auto write_red = [&](const f32x4 (&acc)[2]) {
            float (*red)[16][RLD] = reinterpret_cast<float (*)[16][RLD]>(smem);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int i = 0; i < 4; ++i) red[wave][kq * 4 + i][nt * 16 + rcol] = acc[nt][i];
        };
        // reduced gates of cell lane lt = (utterance lt / 8, unit lt % 8)
        auto read_red = [&](unsigned lt) -> f32x4 {
            float (*red)[16][RLD] = reinterpret_cast<float (*)[16][RLD]>(smem);
            const unsigned lb = lt >> 3, lu = lt & 7;
            f32x4 g4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int w0 = 0; w0 < PS_NW; w0 += 4) {
#pragma unroll
                for (int w = w0; w < w0 + 4; ++w) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(&red[w][lb][lu * 4]);
                    g4[0] += v[0]; g4[1] += v[1]; g4[2] += v[2]; g4[3] += v[3];
                }
            }
            return g4;
        };
        auto label_half = [&](int s) {
            const unsigned t = opaque((unsigned)tid) - 128u;
            if (t < 128u && (int)(t >> 3) < nrows)
                *reinterpret_cast<f32x4*>(ywl + ((s & 1) * 128 + t) * 4) =
                    // (GREEDY: the caller's yw holds two step blocks — <sos> for step 0, the biases alone for every later step)
                    ld4p(at_bytes(a.yw + (size_t)(GREEDY ? min(s, 1) : s) * B * (4 * HS), 4u * ((16u * mh + (t >> 3)) * (4 * HS) + ((unsigned)j8 * 8 + (t & 7)) * 4)));
        };
        auto publish_r0 = [&](int s, bool with_red) {
            const unsigned t = opaque((unsigned)tid) - 128u;
            if (t < 128u && (int)(t >> 3) < nrows) {
                f32x4 g4 = *reinterpret_cast<const f32x4*>(ywl + ((s & 1) * 128 + t) * 4);
                if (with_red) {
                    const f32x4 v = read_red(t);
                    g4[0] += v[0]; g4[1] += v[1]; g4[2] += v[2]; g4[3] += v[3];
                }
                st4_agent(at_bytes(a.r0x + (size_t)s * ((size_t)NWG * 128 * 4), 4u * (((unsigned)blockIdx.x * 128 + t) * 4)), g4);
            }
        };
        float c1 = 0.f;
        auto top_cell = [&](f32x4 g4, int s) {
            const unsigned tq = opaque((unsigned)tid), pbq = 16u * mh + (tq >> 3), unit = (unsigned)j8 * 8 + (tq & 7);
            if ((int)(tq >> 3) < nrows) {
#pragma unroll
                for (int g = 0; g < 4; ++g) g4[g] += bias[g * 128 + tq];
                const float ig = sigmoidf_acc(g4[0]), fg = sigmoidf_acc(g4[1]), gg = tanhf_acc(g4[2]), og = sigmoidf_acc(g4[3]);
                c1 = fg * c1 + ig * gg;
                const float h = og * tanhf_acc(c1);
                if (GREEDY) hl[tq] = h;
                const size_t slab = ((size_t)U + s) * sH;
                const unsigned o = 4u * (pbq * HS + unit);
                st1_agent(at_bytes(a.hx + ((size_t)U + s) * HXS, 4u * (((unit >> 2) * 32 + pbq) * 4 + (unit & 3))), h);
                *at_bytes(a.h_all + slab, o) = h;
                *at_bytes(a.c_all + slab, o) = c1;
                float* go = at_bytes(a.gates_all + 4 * slab, 4u * (pbq * 4 * HS + unit));
                go[0] = ig; go[HS] = fg; go[2 * HS] = gg; go[3 * HS] = og;
            }
        };

        TileAddr ta;
        {
            ta.ok = r < nrows;
            const int row = 16 * mh + (ta.ok ? r : 0);
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                ta.x[f] = 4u * (row * HS + (wave * NF + f) * 16 + kq * 4);
                ta.hx[f] = 4u * ((((wave * NF + f) * 4 + kq) * 32 + row) * 4);
            }
            // KIND 0: producer p = the cell workgroup of units 8p.. for these utterances; watch its last unit of the last row
            const int p0 = wave * 64 + lane;
            ta.cact[0] = p0 < HS / 8;
            ta.canary[0] = 4u * ((((ta.cact[0] ? p0 : 0) * 2 + 1) * 32 + 16 * mh + nrows - 1) * 4 + 3);
            // KIND 1: producer p = attention workgroup (utterance 16 mh + p / split, column part p % split)
            ta.npw1 = (a.split * nrows + 63) / 64;
            ta.cact[1] = p0 < a.split * nrows;
            const int pb = 16 * mh + (ta.cact[1] ? p0 / a.split : 0), pp = p0 % a.split;
            ta.canary[1] = 4u * (pb * HS + (pp + 1) * (HS / a.split) - 1);
        }
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        f32x4 accR0[2], accR1[2];
        accR1[0] = accR1[1] = zero;
        f32x4 x[NF];
        unsigned cep = 0;
        if (tid < 4) cflags[tid] = 0u;
        label_half(0);
        if (U > 1) label_half(1);
        lds_barrier();
        publish_r0(0, false);
        int nslow[3] = {0, 0, 0};
        for (int s = 0; s < U; ++s) {
            const bool more = s + 1 < U;
            PS_STAMP(0, s, 0);
            // top layer: gates = W_ih1 h0_s + W_hh1 h1_{s-1}; h0_s arrives from the attention workgroups as a row-major (B, Hs) slab
            nslow[0] += poll_mul<1>(a.hx + (size_t)s * HXS, ta, x, w_i1, accR1, a.err, cflags, cep);
            PS_STAMP(0, s, 4);
            write_red(accR1);
            lds_barrier();
            PS_STAMP(1, s, 6);
            if (tid < 128) top_cell(read_red(opaque((unsigned)tid)), s);
            PS_STAMP(0, s, 5);
            if (GREEDY) {      // partial logits of h1_s: thread (utterance, v) of the upper eight waves, one 16-byte store per four v
                lds_barrier();
                const unsigned t = opaque((unsigned)tid) - 512u;
                if (t < 512u) {
                    const unsigned utt = t >> 5, v = t & 31u;
                    const f32x4 w0 = *reinterpret_cast<const f32x4*>(wcl + v * 8), w1 = *reinterpret_cast<const f32x4*>(wcl + v * 8 + 4);
                    const f32x4 h0 = *reinterpret_cast<const f32x4*>(hl + utt * 8), h1 = *reinterpret_cast<const f32x4*>(hl + utt * 8 + 4);
                    const float acc = dot4p(w1, h1, dot4p(w0, h0, 0.f));
                    const int ai = __builtin_bit_cast(int, acc);
                    f32x4 q4;
                    q4[0] = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(ai, 0x00, 0xF, 0xF, true));      // quad_perm broadcasts
                    q4[1] = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(ai, 0x55, 0xF, 0xF, true));
                    q4[2] = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(ai, 0xAA, 0xF, 0xF, true));
                    q4[3] = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(ai, 0xFF, 0xF, 0xF, true));
                    if ((v & 3u) == 0u && (int)utt < nrows)
                        st4_agent(at_bytes(a.plx + (size_t)s * ((size_t)NWG * 512), 4u * (((unsigned)blockIdx.x * 16 + utt) * 32 + v)), q4);
                }
            }
            if (!more) break;
            // ---- off the chain (the attention workgroups are working now): next step's R0 from the h0_s tile still in registers
            accR0[0] = accR0[1] = zero;
            mul(x, w_h0, accR0);
            lds_barrier();                       // the top-cell lanes have left the reduction buffer
            write_red(accR0);
            lds_barrier();
            publish_r0(s + 1, true);
            if (s + 2 < U) label_half(s + 2);
            PS_STAMP(0, s, 6);
            // the top layer's recurrent half
            accR1[0] = accR1[1] = zero;
            nslow[2] += poll_mul<0>(a.hx + ((size_t)U + s) * HXS, ta, x, w_h1, accR1, a.err, cflags, cep);
            PS_STAMP(0, s, 7);
        }
        if (a.trace && first_wg && tid == 0)
            for (int k = 0; k < 3; ++k) a.trace[((size_t)a.U + k) * 8 + 7] = (unsigned long long)nslow[k];
    }
};

// ------------------------------------------------------------------------------------------------ attention workgroups
template <int HS, int SPLIT, bool GREEDY>
struct AttnRole {
    static constexpr int D = HS, DW = D / SPLIT;                         // this workgroup's slice of the context columns
    static constexpr int C4 = DW / 4, TQ = PS_THREADS / C4;              // context lanes: (column group, time slice)
    static constexpr int TQW = C4 >= 64 ? 1 : 64 / C4;                   // time slices that share a wave
    static constexpr int NPART = TQ / TQW;                               // partial contexts that meet in LDS (<= 16 waves)
    static constexpr int NJ = HS / 64;                                   // float4 of W_phi per lane (16 lanes per row)
    static constexpr int MAX_TP = PS_NI * TQ;
    static constexpr int EP = (MAX_TP + 63) & ~63;                       // energies padded to whole waves (pad = -inf)
    static constexpr int WCL = HS + DW;                                  // W_c columns a workgroup multiplies: [h | its context slice]
    static __host__ __device__ constexpr int lds_floats(int Tp, int V, bool greedy) {
        return HS + PS_M + EP + MAX_TP + NPART * DW + Tp * PS_KLD + (greedy ? V * WCL + DW + 32 : 0);
    }

    static __device__ void run(const PersistArgs& a, float* smem, const int widx) {
        const int b = widx / SPLIT, part_id = widx % SPLIT;
        const int col0 = part_id * DW;
        const bool first_wg = widx == 0;
        const int tid = threadIdx.x, lane = tid & 63;
        const int B = a.B, U = a.U, Tp = a.Tp;
        float* hs = smem;
        float* qs = hs + HS;
        float* es = qs + PS_M;
        float* as = es + EP;
        float* part = as + MAX_TP;
        float* ks = part + NPART * DW;
        float* wcl = ks + Tp * PS_KLD;                 // free-running only: W_c rows [h columns | own context columns]
        float* ctxl = wcl + a.V * WCL;
        float* lgl = ctxl + DW;

        // ---- resident operands
        const int c4 = tid % C4, tq = tid / C4;
        f32x4 f[PS_NI];
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < PS_NI; ++i) {
            const int t = tq + TQ * i;
            const f32x4 v = ld4p(a.feat + ((size_t)b * Tp + (t < Tp ? t : 0)) * D + col0 + c4 * 4);
            f[i] = t < Tp ? v : zero;
        }
        if (tid >= Tp && tid < MAX_TP) as[tid] = 0.f;         // frames past T' carry zero weight
        if (tid >= Tp && tid < EP) es[tid] = -INFINITY;
        for (int idx = tid; idx < Tp * (PS_M / 4); idx += PS_THREADS) {
            const int t = idx / (PS_M / 4), m4 = idx % (PS_M / 4);
            *reinterpret_cast<f32x4*>(ks + t * PS_KLD + m4 * 4) = ld4p(a.keys + ((size_t)b * Tp + t) * PS_M + m4 * 4);
        }
        const int prow = tid >> 4, pk = tid & 15;         // phi: 64 rows x 16 lanes
        f32x4 wphi[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) wphi[j] = ld4p(a.w_phi + (size_t)prow * HS + 4 * (pk + 16 * j));
        const float bphi = a.b_phi[prow];
        if (GREEDY) {
            for (int idx = tid; idx < a.V * WCL; idx += PS_THREADS) {
                const int v = idx / WCL, k = idx % WCL;
                wcl[idx] = a.w_c[(size_t)v * 2 * HS + (k < HS ? k : HS + col0 + (k - HS))];
            }
        }
        lds_barrier();

        for (int s = 0; s < U; ++s) {
            PS_STAMP(1, s, 0);
            // ---- decoder state of this utterance (published by the cell workgroups)
            if (tid < HS / 4) {
                const float* p = a.hx + ((size_t)U + s) * ((size_t)32 * HS) + ((size_t)tid * 32 + b) * 4;
                unsigned spins = 0;
                f32x4 v;
                for (;;) {
                    v = ld4_agent(p);
                    if (!__any(has_sentinel(v))) break;
                    if (spin_expired(spins, a.err, 0xDEAD0013u)) break;
                }
                *reinterpret_cast<f32x4*>(hs + tid * 4) = v;
            }
            PS_STAMP(1, s, 1);
            lds_barrier();
            // ---- query q = act(W_phi h + b_phi)
            {
                float acc = 0.f;
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc = dot4p(wphi[j], *reinterpret_cast<const f32x4*>(hs + 4 * (pk + 16 * j)), acc);
                acc = gsum<16>(acc);
                if (pk == 0) {
                    acc += bphi;
                    acc = act_apply(acc, a.relu);      // (a.relu: the activation code of las_speller_desc — none / relu / tanh / sigmoid, wave-uniform)
                    qs[prow] = acc;
                    if (part_id == 0) a.q_all[((size_t)s * B + b) * PS_M + prow] = acc;
                }
            }
            lds_barrier();
            PS_STAMP(1, s, 2);
            // ---- energies e[t] = q . keys[t]: 8 lanes per frame
            {
                const int sub = tid & 7;
                const f32x4 q0 = *reinterpret_cast<const f32x4*>(qs + sub * 8), q1 = *reinterpret_cast<const f32x4*>(qs + sub * 8 + 4);
#pragma unroll
                for (int t = tid >> 3; t < EP; t += PS_THREADS / 8) {
                    if (t >= Tp) break;
                    const float* kr = ks + t * PS_KLD + sub * 8;
                    float acc = dot4p(*reinterpret_cast<const f32x4*>(kr), q0, 0.f);
                    acc = dot4p(*reinterpret_cast<const f32x4*>(kr + 4), q1, acc);
                    acc = gsum<8>(acc);
                    if (sub == 0) es[t] = acc;
                }
            }
            lds_barrier();
            PS_STAMP(1, s, 3);
            // ---- softmax over ALL frames (no mask, reference las_model.py:292), statistics redundantly per wave
            float ev[EP / 64];
            float mx = -INFINITY;
#pragma unroll
            for (int k = 0; k < EP / 64; ++k) { ev[k] = es[lane + 64 * k]; mx = fmaxf(mx, ev[k]); }
            mx = wmax(mx);
            float sm = 0.f;
#pragma unroll
            for (int k = 0; k < EP / 64; ++k) sm += __builtin_amdgcn_exp2f((ev[k] - mx) * 1.4426950408889634f);
            const float inv = 1.0f / wsum(sm);
            if (tid < Tp) {
                const float w = __builtin_amdgcn_exp2f((es[tid] - mx) * 1.4426950408889634f) * inv;
                as[tid] = w;
                if (part_id == 0) a.att[((size_t)s * B + b) * Tp + tid] = w;
            }
            lds_barrier();
            PS_STAMP(1, s, 4);
            // ---- context = sum_t a_t feat_t from the register-resident features
            {
                f32x4 acc = zero;
#pragma unroll
                for (int i = 0; i < PS_NI; ++i) {
                    const float w = as[tq + TQ * i];
                    acc[0] = fmaf(w, f[i][0], acc[0]); acc[1] = fmaf(w, f[i][1], acc[1]);
                    acc[2] = fmaf(w, f[i][2], acc[2]); acc[3] = fmaf(w, f[i][3], acc[3]);
                }
                // time slices that share a wave meet by cross-lane adds, the rest through LDS
#pragma unroll
                for (int off = C4; off < 64; off <<= 1) {
                    acc[0] += __shfl_xor(acc[0], off); acc[1] += __shfl_xor(acc[1], off);
                    acc[2] += __shfl_xor(acc[2], off); acc[3] += __shfl_xor(acc[3], off);
                }
                if (TQW == 1 || lane < C4) *reinterpret_cast<f32x4*>(part + (tq / TQW) * DW + c4 * 4) = acc;
            }
            lds_barrier();
            if (tid < C4) {
                f32x4 acc = zero;
#pragma unroll 4
                for (int q = 0; q < NPART; ++q) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(part + q * DW + tid * 4);
                    acc[0] += v[0]; acc[1] += v[1]; acc[2] += v[2]; acc[3] += v[3];
                }
                st4_agent(a.ctx_all + ((size_t)(s + 1) * B + b) * D + col0 + tid * 4, acc);
                if (GREEDY) *reinterpret_cast<f32x4*>(ctxl + tid * 4) = acc;
            }
            PS_STAMP(1, s, 5);
            if (GREEDY) {
                // ---- free-running: this workgroup's part of the character-distribution logits W_c [h | ctx] + b_c
                // (the context columns are split over the workgroups of an utterance, part 0 also takes the h columns and
                //  the bias); the cell workgroups sum the parts, pick the arg-max and feed it back
                lds_barrier();
                {
                    const int v = tid >> 5, l = tid & 31;
                    float acc = 0.f;
                    if (v < a.V) {
                        const float* wr = wcl + v * WCL;
#pragma unroll 4
                        for (int k = l; k < DW; k += 32) acc = fmaf(wr[HS + k], ctxl[k], acc);
                        if (part_id == 0) {
#pragma unroll 4
                            for (int k = l; k < HS; k += 32) acc = fmaf(wr[k], hs[k], acc);
                        }
                    }
                    acc = gsum<16>(acc);
                    acc += __shfl_xor(acc, 16);
                    if (l == 0 && v < 32) lgl[v] = v < a.V ? acc + (part_id == 0 ? a.b_c[v] : 0.f) : 0.f;
                }
                lds_barrier();
                if (tid < 32) st1_agent(a.lgx + (((size_t)s * B + b) * SPLIT + part_id) * 32 + tid, lgl[tid]);
            }
        }
    }
};


// ------------------------------------------------------------------------------------------------ attention workgroups, PRE variant
// Four workgroups per utterance, each owning 4Hs/4 of the layer-0 gate columns: its slice of P = feat . W_ctx^T (T' x Hs
// floats, register-resident: lane = (column group, time slice)) is contracted with the step's attention weights and published
// as whole 128-byte lines.  Query, energies and softmax are computed redundantly by the four (as the two of AttnRole do);
// the context itself is not needed on the chain any more and is left to one batched GEMM after the launch.
// GREEDY (free-running decode, mode 1; round 5): the workgroup also forms the character distribution of its utterance — the decoder-state
// share arrives as 64 partial tiles from the cell workgroups (CellPreRole), the context share is sum_t a_t Q_t with Q = feat W_c[:, Hs:]^T
// resident in LDS (the context itself never exists on the chain) — every wave picks the arg-max itself, and the bottom cell's lanes fetch the
// symbol's W_y column (2 KB per workgroup from the L2: the one dependent load this mode adds to the chain).
template <int HS, int WS, bool GREEDY = false, bool MH = false>
struct AttnPreRole {
    static constexpr int SPLIT = WS;                     // workgroups per utterance: 4, 8 or 16 (longer T' at smaller batches)
    static constexpr int GC = 4 * HS / SPLIT;            // gate columns of this workgroup
    static constexpr int CG = GC / 4;                    // column groups (one float4 each)
    static constexpr int TS = PS_THREADS / CG;           // time slices: 8, 16 or 32 adjacent lanes
    static constexpr int NIP = 14;                       // frames per lane (56 VGPRs of P)
    static constexpr int MAX_TP = NIP * TS;              // 112, 224 or 448 frames
    static constexpr int EP = (MAX_TP + 63) & ~63;       // energies padded to whole waves (pad = -inf)
    static constexpr int NJ = HS / 64;
    static_assert(TS == 8 || TS == 16 || TS == 32 || TS == 64, "time-slice layout");
    // Hs = 256 with 16 workgroups per utterance (T' up to 896, B <= 12: BASELINE configs[4] for the small model, T' = 750): a column group spans a
    // whole wave, and the KEYS no longer fit one workgroup's LDS (192 KB at T' = 750) — each workgroup keeps the keys of ITS ceil(T'/16) frames,
    // computes their energies and the sixteen exchange them (64 floats each, one more hand-off per step) before the softmax
    // ... and the free-running form with 16 workgroups per utterance (either Hs): Q^T (up to 115 KB) takes the LDS the whole keys would need
    static constexpr bool FSPLIT = TS == 64 || (GREEDY && WS == 16);
    static_assert(!FSPLIT || !MH, "the frame-split form is single-head");
    // float4 of a W_phi row slice kept in registers; the rest lives in LDS.  With 16 workgroups per utterance (T' up to 448: the keys
    // alone take up to 122 KB of LDS) all of it stays in registers and R0 is fetched with a blocking load instead
    static constexpr int NJR = WS == 16 ? NJ : NJ / 2;
    static constexpr bool R0_BLOCKING = WS == 16;
    static constexpr int NJ8 = HS / 8;                   // cell-workgroup column groups = partial-logit tiles per utterance
    static __host__ __device__ constexpr int lds_base(int Tp) {
        return HS + PS_M + EP + MAX_TP + 2 * GC + (FSPLIT ? (Tp + SPLIT - 1) / SPLIT : Tp) * PS_KLD + PS_M * 4 * (NJ - NJR) * 16 + (FSPLIT ? 64 : 0);
    }
    static __host__ __device__ constexpr int lds_floats(int Tp) { return lds_base(Tp) + (GREEDY ? 32 * MAX_TP + 32 * NJ8 + 64 : 0); }

    static __host__ __device__ constexpr int fnt_max(int Tp) { return FSPLIT ? (Tp + SPLIT - 1) / SPLIT : Tp; }      // key rows held in LDS
    static __device__ void run(const PersistArgs& a, float* smem, const int widx) {
        // MH: one set of SPLIT workgroups per (utterance, head) — query rows, softmax and P block of that head; the heads' weighted sums meet
        // through gx before the bottom cell, which every head's workgroup applies (identical arithmetic) and head 0 publishes
        const int NH = MH ? a.NH : 1;
        // (MH placement: the NH workgroups that exchange their sums — same utterance, same column part — get block indices 8 apart whenever
        // B * SPLIT is a multiple of 8, i.e. ONE XCD under round-robin dispatch; verified at run time below, never assumed)
        const bool xmap = MH && ((a.B * SPLIT) & 7) == 0;
        // (FSPLIT placement: the SPLIT workgroups of an utterance exchange their energies — with B a multiple of 8 they get block indices 8 apart)
        const bool fmap = FSPLIT && (a.B & 7) == 0;
        const int qx = xmap ? (widx / (8 * NH)) * 8 + (widx & 7) : 0;
        const int pu = xmap ? 0 : widx / SPLIT, part_id = xmap ? qx % SPLIT : fmap ? (widx >> 3) % SPLIT : widx % SPLIT;
        const int b = xmap ? qx / SPLIT : fmap ? (widx / (8 * SPLIT)) * 8 + (widx & 7) : (MH ? pu / NH : pu), hd = xmap ? (widx >> 3) % NH : (MH ? pu % NH : 0);
        const int col0 = part_id * GC;
        const bool first_wg = widx == 0;
        const int tid = threadIdx.x, lane = tid & 63;
        const int B = a.B, U = a.U, Tp = a.Tp;
        float* hs = smem;
        float* qs = hs + HS;
        float* es = qs + PS_M;
        float* as = es + EP;
        float* gxl = as + MAX_TP;
        float* r0l = gxl + GC;                   // R0 of the step in flight: 4 gates x this workgroup's CG units
        float* ks = r0l + GC;

        // ---- resident operands
        const int ts = tid % TS, cg = tid / TS;
        f32x4 pr[NIP];
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < NIP; ++i) {
            const int t = ts + TS * i;
            const f32x4 v = ld4p(a.pctx + (((size_t)b * Tp + (t < Tp ? t : 0)) * NH + hd) * (4 * HS) + col0 + cg * 4);
            pr[i] = t < Tp ? v : zero;
        }
        if (tid >= Tp && tid < MAX_TP) as[tid] = 0.f;         // frames past T' carry zero weight (MAX_TP <= 448 < threads)
        if (tid >= Tp && tid < EP) es[tid] = -INFINITY;
        // (FSPLIT: the keys of this workgroup's frames [ft0, ft0 + fnt) only)
        const int fth = (Tp + SPLIT - 1) / SPLIT, ft0 = FSPLIT ? part_id * fth : 0, fnt = FSPLIT ? max(0, min(fth, Tp - ft0)) : Tp;
        for (int idx = tid; idx < fnt * (PS_M / 4); idx += PS_THREADS) {
            const int t = idx / (PS_M / 4), m4 = idx % (PS_M / 4);
            *reinterpret_cast<f32x4*>(ks + t * PS_KLD + m4 * 4) = ld4p(a.keys + ((size_t)b * Tp + ft0 + t) * PS_M + m4 * 4);
        }
        const int prow = tid >> 4, pk = tid & 15;         // phi: 64 rows x 16 lanes
        // W_phi row prow, columns 4 (pk + 16 j): j < NJR in registers, the rest in LDS ([row][HS / 2], a wave's 16-lane row groups read
        // 256 contiguous bytes each: conflict-free b128 reads) — the P rows and the query weights together do not fit 128 registers
        f32x4 wphi[NJR];
        float* wpl = ks + fnt_max(Tp) * PS_KLD;
        float* esp = wpl + PS_M * 4 * (NJ - NJR) * 16;      // FSPLIT: this workgroup's energies before the exchange (64 floats)
#pragma unroll
        for (int j = 0; j < NJR; ++j) wphi[j] = ld4p(a.w_phi + (size_t)(hd * PS_M + prow) * HS + 4 * (pk + 16 * j));
#pragma unroll
        for (int j = NJR; j < NJ; ++j)
            *reinterpret_cast<f32x4*>(wpl + prow * (64 * (NJ - NJR)) + 4 * (pk + 16 * (j - NJR))) = ld4p(a.w_phi + (size_t)(hd * PS_M + prow) * HS + 4 * (pk + 16 * j));
        const float bphi = a.b_phi[hd * PS_M + prow];
        // GREEDY: Q^T of this utterance [v][t] (zero past V / T'), the staging area of the partial logits [v][tile], the step's logits
        float* qt = smem + lds_base(Tp);
        float* plv = qt + 32 * MAX_TP;
        float* lgl = plv + 32 * NJ8;
        float bcv = 0.f;
        float* lgp = lgl + 32;      // GREEDY, MH: this head's share of the logits before the heads' shares meet
        if (GREEDY) {
            // (MH: Q^T per (utterance, head) with dim_reduce folded in; the decoder-state share and the bias enter through head 0 only)
            for (int idx = tid; idx < 32 * MAX_TP; idx += PS_THREADS) {
                const int v = idx / MAX_TP, t = idx % MAX_TP;
                qt[idx] = (v < a.V && t < Tp) ? a.qct[(((size_t)b * NH + hd) * 32 + v) * Tp + t] : 0.f;
            }
            bcv = ((tid >> 5) < a.V && (!MH || hd == 0)) ? a.b_c[tid >> 5] : 0.f;
            if (MH && hd != 0) for (int idx = tid; idx < 32 * NJ8; idx += PS_THREADS) plv[idx] = 0.f;
        }
        lds_barrier();
        bool l2x = false;
        if (FSPLIT && fmap) {
            // the utterance's workgroups publish their XCC ids behind the energy slabs (sentinel-prefilled with them) and compare
            unsigned xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcc));
            unsigned* ids = reinterpret_cast<unsigned*>(a.ex + (size_t)a.U * a.B * (SPLIT * 64)) + (size_t)b * SPLIT;
            volatile int* flag = reinterpret_cast<volatile int*>(gxl);
            if (tid == 0) { *flag = 1; __hip_atomic_store(ids + part_id, 0xC0DE0000u | xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
            lds_barrier();
            if (tid < SPLIT) {
                unsigned spins = 0, v;
                for (;;) {
                    v = __hip_atomic_load(ids + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (v != PS_SENT) break;
                    if (spin_expired(spins, a.err, 0xDEAD001Bu)) break;
                }
                if (v != (0xC0DE0000u | xcc)) *flag = 0;
            }
            lds_barrier();
            l2x = *flag != 0;
            lds_barrier();
        }
        if (MH) {
            // every workgroup publishes its XCC id behind the gx slabs (sentinel-prefilled with them) and reads its partners'
            unsigned xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcc));
            unsigned* ids = reinterpret_cast<unsigned*>(a.gx + (size_t)a.U * a.B * NH * (4 * HS)) + ((size_t)b * SPLIT + part_id) * 4;
            volatile int* flag = reinterpret_cast<volatile int*>(gxl);
            if (tid == 0) { *flag = 1; __hip_atomic_store(ids + hd, 0xC0DE0000u | xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
            lds_barrier();
            if (tid < NH) {
                unsigned spins = 0, v;
                for (;;) {
                    v = __hip_atomic_load(ids + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (v != PS_SENT) break;
                    if (spin_expired(spins, a.err, 0xDEAD0018u)) break;
                }
                if (v != (0xC0DE0000u | xcc)) *flag = 0;
            }
            lds_barrier();
            l2x = *flag != 0;
            lds_barrier();
        }

        // ---- the bottom LSTM cell of this workgroup's CG hidden units, one per lane of the first CG/64 waves, state in a register:
        //      gates_s = sum_t a_{s-1,t} P_t (reduced by the ts == 0 lanes, handed over through LDS) + R0_s (the cell workgroups'
        //      recurrent / label / bias part, fetched while the attention of step s-1 is computed)
        const bool clane = ts == 0;
        const size_t sH = (size_t)B * HS;
        constexpr size_t HXS = (size_t)32 * HS, R0S = (size_t)(HS / 4) * 32 * 16;
        float c0 = 0.f;
        // R0_s is published ~0.7 us after h1_{s-1}: waves 2.. issue its loads once the query is done (four agent-scope dword loads
        // the compiler schedules itself: their latency hides under the energies / softmax / weighted sum) and hand it to the bottom
        // cell's lanes through LDS; a slot that was not complete yet is re-polled when it is consumed
        const bool rlane = tid >= HS / 4 && tid < HS / 4 + CG;
        auto r0_src = [&](int s) {
            // cell workgroup (u / 8, b / 16) publishes a tile [16 utterances][8 units x 4 gates]; offsets re-derived per use (opaque)
            const unsigned u = part_id * CG + (rlane ? opaque((unsigned)tid) - HS / 4 : 0u);
            return at_bytes(a.r0x + (size_t)s * R0S + ((size_t)(b >> 4) * 16 + (b & 15)) * 32, ((u & ~7u) << 9) + ((u & 7u) << 4));
        };
        auto r0_issue = [&](int s, unsigned (&rv)[4]) {
            const unsigned* src = reinterpret_cast<const unsigned*>(r0_src(s));
#pragma unroll
            for (int k = 0; k < 4; ++k) rv[k] = rlane ? __hip_atomic_load(src + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
        };
        auto r0_land = [&](int s, const unsigned (&rv)[4]) {
            if (tid >= HS / 4 && tid < HS / 4 + CG) {       // whole waves
                f32x4 v;
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = __uint_as_float(rv[k]);
                if (__any(has_sentinel(v))) {
                    const float* src = r0_src(s);
                    unsigned spins = 0;
                    for (;;) {
                        v = ld4_agent(src);
                        if (!__any(has_sentinel(v))) break;
                        if (spin_expired(spins, a.err, 0xDEAD0015u)) break;
                    }
                }
                *reinterpret_cast<f32x4*>(r0l + (tid - HS / 4) * 4) = v;
            }
        };
        // after the barrier that follows the ts == 0 lanes' LDS hand-over of the reduced sums (gxl): apply the cell, publish h0_s
        auto bottom_cell = [&](int s, const f32x4 wy = f32x4{0.f, 0.f, 0.f, 0.f}) {      // wy: GREEDY, the fed-back symbol's W_y entries of this unit's gates
            if (tid >= CG) return;                               // whole waves (CG is a multiple of 64)
            const int unit = part_id * CG + tid;                 // hidden unit; its four gates are columns col0 + 4 tid .. + 3
            f32x4 g4 = *reinterpret_cast<const f32x4*>(gxl + tid * 4);
            const f32x4 r = *reinterpret_cast<const f32x4*>(r0l + tid * 4);
            if (GREEDY) { g4[0] += wy[0]; g4[1] += wy[1]; g4[2] += wy[2]; g4[3] += wy[3]; }
            const float ig = sigmoidf_acc(g4[0] + r[0]), fg = sigmoidf_acc(g4[1] + r[1]), gg = tanhf_acc(g4[2] + r[2]), og = sigmoidf_acc(g4[3] + r[3]);
            c0 = fg * c0 + ig * gg;
            const float h = og * tanhf_acc(c0);
            if (MH && hd != 0) return;                               // (the other heads only carry the cell state along)
            const unsigned o = opaque(4u * ((unsigned)b * HS + unit));
            st1_agent(at_bytes(a.hx + (size_t)s * HXS, o), h);       // hand-off to the cell workgroups: a wave writes 256 contiguous bytes
            *at_bytes(a.h_all + (size_t)s * sH, o) = h;              // stash for the backward pass
            *at_bytes(a.c_all + (size_t)s * sH, o) = c0;
            float* go = at_bytes(a.gates_all + 4 * (size_t)s * sH, opaque(4u * ((unsigned)b * 4 * HS + unit)));
            go[0] = ig; go[HS] = fg; go[2 * HS] = gg; go[3 * HS] = og;
        };
        // GREEDY: the partial logits of h1_s (64 tiles of 32 floats for this utterance): waves 8.. fetch one 16-byte piece each while the
        // energies are computed (like R0) and drop it transposed into LDS; a piece that was not complete yet is re-polled when it lands
        const bool plane = GREEDY && (!MH || hd == 0) && tid >= 512 && tid < 512 + NJ8 * 8;
        auto pl_src = [&](int s) {
            const unsigned k = plane ? opaque((unsigned)tid) - 512u : 0u;
            return at_bytes(a.plx + (size_t)s * ((size_t)(HS / 4) * 512), 4u * ((((k >> 3) * 2 + ((unsigned)b >> 4)) * 16 + ((unsigned)b & 15u)) * 32 + (k & 7u) * 4));
        };
        // (first read: ONE ordinary 16-byte load — a slot of this step's slab is either still the sentinel, possibly a stale copy of it in
        // this XCD's L2, or final: whatever looks incomplete is re-read with agent-scope loads when it lands)
        auto pl_issue = [&](int s, f32x4& pv) { pv = plane ? *reinterpret_cast<const f32x4*>(pl_src(s)) : f32x4{0.f, 0.f, 0.f, 0.f}; };
        auto pl_land = [&](int s, const f32x4& pv) {
            if ((!MH || hd == 0) && tid >= 512 && tid < 512 + NJ8 * 8) {       // whole waves
                f32x4 v = pv;
                if (__any(has_sentinel(v))) {
                    const float* src = pl_src(s);
                    unsigned spins = 0;
                    for (;;) {
                        v = ld4_agent(src);
                        if (!__any(has_sentinel(v))) break;
                        if (spin_expired(spins, a.err, 0xDEAD0016u)) break;
                    }
                }
                const unsigned k = (unsigned)tid - 512u, j = k >> 3, q = k & 7u;
#pragma unroll
                for (int e = 0; e < 4; ++e) plv[(q * 4 + e) * NJ8 + j] = v[e];
            }
        };
        {   // step 0: the context is the first listener frame (las_model.py:198): W_ctx feat_0 = P row 0 = pr[0] of the ts == 0 lanes
            unsigned rv0[4];
            r0_issue(0, rv0);
            r0_land(0, rv0);
            if (clane) *reinterpret_cast<f32x4*>(gxl + cg * 4) = MH ? ld4p(a.p0 + (size_t)b * (4 * HS) + col0 + cg * 4) : pr[0];
            lds_barrier();
            bottom_cell(0);
        }

        for (int s = 0; s < U; ++s) {
            PS_STAMP(1, s, 0);
            // ---- decoder state of this utterance (published by the cell workgroups)
            if (tid < HS / 4) {
                const float* p = at_bytes(a.hx + ((size_t)U + s) * ((size_t)32 * HS), opaque(4u * (((unsigned)tid * 32 + b) * 4)));
                unsigned spins = 0;
                f32x4 v;
                for (;;) {
                    v = ld4_agent(p);
                    if (!__any(has_sentinel(v))) break;
                    if (spin_expired(spins, a.err, 0xDEAD0013u)) break;
                }
                *reinterpret_cast<f32x4*>(hs + tid * 4) = v;
            }
            PS_STAMP(1, s, 1);
            lds_barrier();
            // ---- query q = act(W_phi h + b_phi)
            {
                float acc = 0.f;
#pragma unroll
                for (int j = 0; j < NJR; ++j) acc = dot4p(wphi[j], *reinterpret_cast<const f32x4*>(hs + 4 * (pk + 16 * j)), acc);
                if (NJR < NJ) {
                    const float* wr = at_bytes(wpl, opaque(4u * (unsigned)(prow * (64 * (NJ - NJR)) + 4 * pk)));
#pragma unroll
                    for (int j = NJR; j < NJ; ++j)
                        acc = dot4p(*reinterpret_cast<const f32x4*>(wr + 64 * (j - NJR)), *reinterpret_cast<const f32x4*>(hs + 4 * (pk + 16 * j)), acc);
                }
                acc = gsum<16>(acc);
                if (pk == 0) {
                    acc += bphi;
                    acc = act_apply(acc, a.relu);      // (a.relu: the activation code of las_speller_desc — none / relu / tanh / sigmoid, wave-uniform)
                    qs[prow] = acc;
                    if (part_id == 0) *at_bytes(a.q_all + (((size_t)s * B + b) * NH + hd) * PS_M, opaque(4u * (unsigned)prow)) = acc;
                }
            }
            lds_barrier();
            PS_STAMP(1, s, 2);
            unsigned rv[4];
            if (!R0_BLOCKING && s + 1 < U) r0_issue(s + 1, rv);
            // ---- energies e[t] = q . keys[t]: 8 lanes per frame
            {
                const int sub = tid & 7;
                const f32x4 q0 = *reinterpret_cast<const f32x4*>(qs + sub * 8), q1 = *reinterpret_cast<const f32x4*>(qs + sub * 8 + 4);
                if (!FSPLIT) {
#pragma unroll
                    for (int t = tid >> 3; t < EP; t += PS_THREADS / 8) {
                        if (t >= Tp) break;
                        const float* kr = ks + t * PS_KLD + sub * 8;
                        float acc = dot4p(*reinterpret_cast<const f32x4*>(kr), q0, 0.f);
                        acc = dot4p(*reinterpret_cast<const f32x4*>(kr + 4), q1, acc);
                        acc = gsum<8>(acc);
                        if (sub == 0) es[t] = acc;
                    }
                } else if (tid < 64 * 8) {      // own frames: at most 64, 8 lanes each (slots past the slice publish 0: a slot is complete or the sentinel)
                    const int t = tid >> 3;
                    float acc = 0.f;
                    if (t < fnt) {
                        const float* kr = ks + t * PS_KLD + sub * 8;
                        acc = dot4p(*reinterpret_cast<const f32x4*>(kr), q0, 0.f);
                        acc = dot4p(*reinterpret_cast<const f32x4*>(kr + 4), q1, acc);
                    }
                    acc = gsum<8>(acc);
                    if (sub == 0) esp[t] = acc;
                }
            }
            lds_barrier();
            if (FSPLIT) {
                // ---- the sixteen slices of the energies meet: publish 256 bytes (two whole lines), collect every part (the own one included)
                float* ex = a.ex + ((size_t)s * B + b) * (SPLIT * 64);
                if (tid < 16) {
                    const f32x4 ev = *reinterpret_cast<const f32x4*>(esp + tid * 4);
                    float* dste = at_bytes(ex + part_id * 64, opaque(16u * (unsigned)tid));
                    if (l2x) {      // same XCD: plain store into the shared L2
                        f32x4 cv;
#pragma unroll
                        for (int k = 0; k < 4; ++k) cv[k] = __uint_as_float(pub_bits(ev[k]));
                        *reinterpret_cast<f32x4*>(dste) = cv;
                    } else {
                        st4_agent(dste, ev);
                    }
                }
                if (tid < SPLIT * 16) {
                    const float* src = at_bytes(ex, opaque(16u * (unsigned)tid));
                    unsigned spins = 0;
                    f32x4 v;
                    for (;;) {
                        v = ld4_agent(src);
                        if (!__any(has_sentinel(v))) break;
                        if (spin_expired(spins, a.err, 0xDEAD001Au)) break;
                    }
                    const int pp = tid >> 4, j0 = (tid & 15) * 4;
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (j0 + k < fth && pp * fth + j0 + k < Tp) es[pp * fth + j0 + k] = v[k];
                }
                lds_barrier();
            }
            PS_STAMP(1, s, 3);
            f32x4 pv = zero;
            if (GREEDY) pl_issue(s, pv);         // (published ~0.4 us after h1_s: issued now, the first read usually finds them complete)
            // ---- softmax over ALL frames (no mask, reference las_model.py:292), statistics redundantly per wave
            float mx = -INFINITY;
#pragma unroll
            for (int k = 0; k < EP / 64; ++k) mx = fmaxf(mx, es[lane + 64 * k]);
            mx = wmax(mx);
            float sm = 0.f;
#pragma unroll
            for (int k = 0; k < EP / 64; ++k) sm += __builtin_amdgcn_exp2f((es[lane + 64 * k] - mx) * 1.4426950408889634f);
            const float inv = 1.0f / wsum(sm);
            if (tid < Tp) {
                const float w = __builtin_amdgcn_exp2f((es[tid] - mx) * 1.4426950408889634f) * inv;
                as[tid] = w;
                if (part_id == 0) *at_bytes(a.att + (((size_t)s * NH + hd) * B + b) * Tp, opaque(4u * (unsigned)tid)) = w;
            }
            if (GREEDY) pl_land(s, pv);          // (visible to every wave behind the barrier)
            lds_barrier();
            PS_STAMP(1, s, 4);
            f32x4 wy = zero;
            int ysym = 0;
            float lval = 0.f, lmax = 0.f;
            if (GREEDY) {
                // ---- character distribution of step s: logit[v] = b_c[v] + sum_tiles PL[v] + sum_t a_t Q[v][t]; 32 lanes per symbol.  It comes
                // BEFORE the weighted sum of P: the fetch of the symbol's W_y column, the one dependent load of this mode, then runs under it
                {
                    const int v = tid >> 5, l = tid & 31;
                    float acc = 0.f;
#pragma unroll
                    for (int i = 0; i < (MAX_TP + 31) / 32; ++i) {
                        const int t = l + 32 * i;
                        if (MAX_TP % 32 == 0 || t < MAX_TP) acc = fmaf(as[t], qt[v * MAX_TP + t], acc);
                    }
#pragma unroll
                    for (int i = 0; i < NJ8 / 32; ++i) acc += plv[v * NJ8 + l + 32 * i];
                    acc = gsum<16>(acc);                                  // row sums: the two 16-lane rows of a symbol meet as scalars
                    const float s0 = lane_f(acc, 0) + lane_f(acc, 16), s1 = lane_f(acc, 32) + lane_f(acc, 48);
                    if (l == 0) {
                        if (MH) lgp[v] = (lane < 32 ? s0 : s1) + bcv;
                        else lgl[v] = v < a.V ? (lane < 32 ? s0 : s1) + bcv : -INFINITY;
                    }
                }
                lds_barrier();
                if (MH) {
                    // the heads' shares of the logits meet (one 128-byte line per workgroup and step, as the weighted sums below do)
                    if (tid < 8) {
                        float* lx = a.lgx + ((((size_t)s * B + b) * SPLIT + part_id) * NH) * 32;
                        const f32x4 own = *reinterpret_cast<const f32x4*>(lgp + tid * 4);
                        float* dstl = at_bytes(lx + hd * 32, opaque(16u * (unsigned)tid));
                        if (l2x) {
                            f32x4 cv;
#pragma unroll
                            for (int k = 0; k < 4; ++k) cv[k] = __uint_as_float(pub_bits(own[k]));
                            *reinterpret_cast<f32x4*>(dstl) = cv;
                        } else {
                            st4_agent(dstl, own);
                        }
                        f32x4 tot = zero;
                        for (int h2 = 0; h2 < NH; ++h2) {
                            f32x4 v4 = own;
                            if (h2 != hd) {
                                const float* src = at_bytes(lx + h2 * 32, opaque(16u * (unsigned)tid));
                                unsigned spins = 0;
                                for (;;) {
                                    v4 = ld4_agent(src);
                                    if (!has_sentinel(v4)) break;
                                    if (spin_expired(spins, a.err, 0xDEAD0019u)) break;
                                }
                            }
                            tot[0] += v4[0]; tot[1] += v4[1]; tot[2] += v4[2]; tot[3] += v4[3];
                        }
#pragma unroll
                        for (int k = 0; k < 4; ++k) lgl[tid * 4 + k] = (tid * 4 + k) < a.V ? tot[k] : -INFINITY;
                    }
                    lds_barrier();
                }
                // every wave: arg-max (first maximal index, as torch.argmax / torch.max do) from the 32 logits: DPP row reductions, the two
                // rows of 16 meet as scalars (no LDS round trips on the chain)
                lval = lgl[lane & 31];
                float m = lval;
                m = fmaxf(m, dpp_f(m, 0)); m = fmaxf(m, dpp_f(m, 1)); m = fmaxf(m, dpp_f(m, 2)); m = fmaxf(m, dpp_f(m, 3));
                m = fmaxf(lane_f(m, 0), lane_f(m, 16));
                int best = lval == m ? (lane & 31) : 64;
                {
                    float bf = __builtin_bit_cast(float, best);               // (non-negative ints order like their float bit patterns)
                    bf = fminf(bf, dpp_f(bf, 0)); bf = fminf(bf, dpp_f(bf, 1)); bf = fminf(bf, dpp_f(bf, 2)); bf = fminf(bf, dpp_f(bf, 3));
                    best = min(__builtin_amdgcn_readlane(__builtin_bit_cast(int, bf), 0), __builtin_amdgcn_readlane(__builtin_bit_cast(int, bf), 16));
                }
                ysym = best;
                lmax = m;
                if (s + 1 < U && tid < CG)        // the symbol's W_y entries of this lane's four gates (permuted column order)
                    wy = ld4p(at_bytes(a.wyT + (size_t)ysym * (4 * HS), opaque(4u * (unsigned)(col0 + tid * 4))));
            }
            // ---- sum_t a_t P_t over this workgroup's gate columns; the TS time slices of a column group are adjacent lanes
            {
                f32x4 acc = zero;
#pragma unroll
                for (int i = 0; i < NIP; ++i) {
                    const float w = as[ts + TS * i];
                    acc[0] = fmaf(w, pr[i][0], acc[0]); acc[1] = fmaf(w, pr[i][1], acc[1]);
                    acc[2] = fmaf(w, pr[i][2], acc[2]); acc[3] = fmaf(w, pr[i][3], acc[3]);
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    acc[k] = gsum<(TS < 16 ? TS : 16)>(acc[k]);
                    if (TS == 32) {          // two 16-lane rows per column group: add the row sums (wave-uniform scalars)
                        const float lo = lane_f(acc[k], 0) + lane_f(acc[k], 16), hi = lane_f(acc[k], 32) + lane_f(acc[k], 48);
                        acc[k] = lane < 32 ? lo : hi;
                    }
                    if (TS == 64) acc[k] = (lane_f(acc[k], 0) + lane_f(acc[k], 16)) + (lane_f(acc[k], 32) + lane_f(acc[k], 48));      // a whole wave per column group
                }
                // stash for the backward pass (its softmax-backward statistic reuses this sum): off the chain, plain stores
                float* dst = a.gx + (((size_t)s * B + b) * NH + hd) * (4 * HS) + col0;
                if (!MH) {
                    if (clane) {
                        *reinterpret_cast<f32x4*>(at_bytes(dst, opaque(16u * (unsigned)cg))) = acc;
                        *reinterpret_cast<f32x4*>(gxl + cg * 4) = acc;
                    }
                } else if (clane) {
                    // the heads' sums meet here: publish this head's (agent scope; the slab is the backward's stash as well), collect the others'
                    // and add them in head order, so that every head's workgroup holds bit-identical gate pre-activations
                    if (l2x) {      // same XCD: an ordinary store reaches the shared L2, where the partners' L1-bypassing polls find it
                        f32x4 cv;
#pragma unroll
                        for (int k = 0; k < 4; ++k) cv[k] = __uint_as_float(pub_bits(acc[k]));
                        *reinterpret_cast<f32x4*>(at_bytes(dst, opaque(16u * (unsigned)cg))) = cv;
                    } else {
                        st4_agent(at_bytes(dst, opaque(16u * (unsigned)cg)), acc);
                    }
                    f32x4 tot = zero;
                    for (int h2 = 0; h2 < NH; ++h2) {
                        f32x4 v = acc;
                        if (h2 != hd) {
                            const float* src = at_bytes(a.gx + (((size_t)s * B + b) * NH + h2) * (4 * HS) + col0, opaque(16u * (unsigned)cg));
                            unsigned spins = 0;
                            for (;;) {
                                v = ld4_agent(src);
                                if (!has_sentinel(v)) break;
                                if (spin_expired(spins, a.err, 0xDEAD0017u)) break;
                            }
                        }
                        tot[0] += v[0]; tot[1] += v[1]; tot[2] += v[2]; tot[3] += v[3];
                    }
                    *reinterpret_cast<f32x4*>(gxl + cg * 4) = tot;
                }
            }
            // ---- bottom cell of step s+1 and its hand-off to the top layer
            if (s + 1 < U) {
                if (R0_BLOCKING) r0_issue(s + 1, rv);
                r0_land(s + 1, rv);
                lds_barrier();
                PS_STAMP(2, s + 1, 0);
                bottom_cell(s + 1, wy);
            }
            if (GREEDY && part_id == 0 && (!MH || hd == 0) && tid < 32) {
                // outputs of step s (off the chain): log-probabilities, the arg-max, the one-hot row fed to step s+1
                float se = tid < a.V ? expf(lval - lmax) : 0.f;
                se = gsum<16>(se);
                se += __shfl_xor(se, 16);
                const float lp = lval - (lmax + logf(se));
                if (tid < a.V) *at_bytes(a.logp, opaque(4u * (unsigned)((s * B + b) * a.V + tid))) = lp;
                if (tid == 0 && a.argmax_out) *at_bytes(a.argmax_out, opaque(4u * (unsigned)(s * B + b))) = ysym;
                if (a.y_all && tid < a.Vp) *at_bytes(a.y_all, opaque(4u * (unsigned)(((s + 1) * B + b) * a.Vp + tid))) = tid == ysym ? 1.f : 0.f;
            }
            PS_STAMP(1, s, 5);
        }
    }
};

template <int HS, int WS, bool GREEDY, bool MH = false>
__global__ __launch_bounds__(PS_THREADS) void speller_persist_fwd_pre_kernel(PersistArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NC = HS / 4;
    if ((int)blockIdx.x >= NC) AttnPreRole<HS, WS, GREEDY, MH>::run(a, smem, blockIdx.x - NC);
    else CellPreRole<HS, GREEDY>::run(a, smem);
}

template <int HS, int SPLIT, bool GREEDY>
__global__ __launch_bounds__(PS_THREADS) void speller_persist_fwd_kernel(PersistArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NC = HS / 4;
#if defined(PS_ONLY_CELL)
    CellRole<HS, GREEDY>::run(a, smem);
#elif defined(PS_ONLY_ATTN)
    AttnRole<HS, SPLIT, GREEDY>::run(a, smem, blockIdx.x - NC);
#else
    if ((int)blockIdx.x < NC) CellRole<HS, GREEDY>::run(a, smem);
    else AttnRole<HS, SPLIT, GREEDY>::run(a, smem, blockIdx.x - NC);
#endif
}

// ------------------------------------------------------------------------------------------------ host side
static unsigned long long* g_persist_trace = nullptr;      // device buffer of 2*U*8 stamps, or null (normal operation)
void speller_persist_set_trace(unsigned long long* dev_buf) { g_persist_trace = dev_buf; }

// attention workgroups per utterance: the smallest split whose lanes can hold T' frames (7 float4 per lane) and that
// leaves every workgroup resident at once (one per CU); 0 = the persistent kernel does not apply
static int persist_split(int B, int Tp, int Hs, int V, bool greedy) {
    int cus = 0, dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
        return 0;
    for (int split = 2; split <= 8; split *= 2) {
        const int dw = Hs / split, c4 = dw / 4, tq = PS_THREADS / c4, max_tp = PS_NI * tq;
        const int npart = tq / (c4 >= 64 ? 1 : 64 / c4);
        const long lds = Hs + PS_M + ((max_tp + 63) & ~63) + max_tp + (long)npart * dw + (long)Tp * PS_KLD +
                         (greedy ? (long)V * (Hs + dw) + dw + 32 : 0);                                          // AttnRole::lds_floats
        if (Tp <= max_tp && Hs / 4 + split * B <= cus && lds * 4 <= 160 * 1024) return split;
    }
    return 0;
}

bool speller_persist_eligible(int B, int Tp, int Hs, int D, int M, int V, int L, int heads, int use_mlp, int free_running) {
    if (L != 2 || heads != 1 || !use_mlp || M != PS_M || D != Hs) return false;
    if (Hs != 256 && Hs != 512) return false;
    if (B < 1 || B > 32 || ((V + 15) & ~15) > 256) return false;
    if (free_running && V > 32) return false;                 // the fed-back symbol rows are 32 floats wide
    return persist_split(B, Tp, Hs, V, free_running != 0) != 0;
}

// attention workgroups per utterance of the PRE variant: the smallest of 4 / 8 / 16 whose lanes hold T' frames of P
// (14 per lane, 1024*ws/Hs... time slices of 8 / 16 / 32 lanes) and that leaves every workgroup resident; 0 = not applicable.
// cus < 0: shape check only (sizes the reserve).
int speller_persist_pre_ws(int B, int Tp, int Hs, int cus) {
    for (int ws = 4; ws <= 16; ws *= 2) {
        const int ts = PS_THREADS / (Hs / ws);           // time slices = threads / column groups
        if (ts > 64) break;                              // (64: Hs = 256 with 16 workgroups per utterance, the frame-split form)
        if (Tp <= 14 * ts && (cus < 0 || Hs / 4 + ws * B <= cus)) return ws;
    }
    return 0;
}
bool speller_persist_pre_shape(int B, int Tp, int Hs, int D, int M, int V, int L, int heads, int use_mlp) {
    if (L != 2 || heads != 1 || !use_mlp || M != PS_M || D != Hs) return false;
    if (Hs != 256 && Hs != 512) return false;
    if (B < 1 || B > 32 || ((V + 15) & ~15) > 32) return false;      // the label half is a 32-wide dot product from LDS
    return speller_persist_pre_ws(B, Tp, Hs, -1) != 0;
}
template <int HS, int WS, bool GREEDY = false>
static size_t persist_fwd_pre_smem(int Tp) {
    return sizeof(float) * (size_t)std::max(CellPreRole<HS, GREEDY>::LDS_FLOATS, AttnPreRole<HS, WS, GREEDY>::lds_floats(Tp));
}
template <int HS, int WS, bool GREEDY = false, bool MH = false>
static bool persist_fwd_pre_fits(int Tp, int grid) {
    const size_t smem = persist_fwd_pre_smem<HS, WS, GREEDY>(Tp);
    if (smem > 160 * 1024) return false;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&speller_persist_fwd_pre_kernel<HS, WS, GREEDY, MH>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)smem) != hipSuccess)
        return false;
    return persistent_launch_fits(speller_persist_fwd_pre_kernel<HS, WS, GREEDY, MH>, PS_THREADS, smem, grid);
}
static bool persist_fwd_pre_mh_greedy_fits_rt(int Hs, int ws, int Tp, int grid) {      // free-running, several heads: 4 or 8 workgroups per (utterance, head)
    if (ws == 16) return false;
    if (Hs == 512) return ws == 4 ? persist_fwd_pre_fits<512, 4, true, true>(Tp, grid) : persist_fwd_pre_fits<512, 8, true, true>(Tp, grid);
    return ws == 4 ? persist_fwd_pre_fits<256, 4, true, true>(Tp, grid) : persist_fwd_pre_fits<256, 8, true, true>(Tp, grid);
}
static bool persist_fwd_pre_mh_fits_rt(int Hs, int ws, int Tp, int grid) {
    if (Hs == 256 && ws == 16) return false;      // (the frame-split form is single-head)
    if (Hs == 512) return ws == 4 ? persist_fwd_pre_fits<512, 4, false, true>(Tp, grid) : ws == 8 ? persist_fwd_pre_fits<512, 8, false, true>(Tp, grid)
                                                                                                  : persist_fwd_pre_fits<512, 16, false, true>(Tp, grid);
    return ws == 4 ? persist_fwd_pre_fits<256, 4, false, true>(Tp, grid) : persist_fwd_pre_fits<256, 8, false, true>(Tp, grid);
}
static bool persist_fwd_pre_fits_rt(int Hs, int ws, int Tp, int grid, bool greedy = false) {
    if (greedy) {      // free-running instantiations: 4 or 8 attention workgroups per utterance with the whole keys in LDS beside Q^T; 16 with the keys split by frames
        if (Hs == 512) return ws == 4 ? persist_fwd_pre_fits<512, 4, true>(Tp, grid) : ws == 8 ? persist_fwd_pre_fits<512, 8, true>(Tp, grid) : persist_fwd_pre_fits<512, 16, true>(Tp, grid);
        return ws == 4 ? persist_fwd_pre_fits<256, 4, true>(Tp, grid) : ws == 8 ? persist_fwd_pre_fits<256, 8, true>(Tp, grid) : persist_fwd_pre_fits<256, 16, true>(Tp, grid);
    }
    if (Hs == 512) return ws == 4 ? persist_fwd_pre_fits<512, 4>(Tp, grid) : ws == 8 ? persist_fwd_pre_fits<512, 8>(Tp, grid) : persist_fwd_pre_fits<512, 16>(Tp, grid);
    return ws == 4 ? persist_fwd_pre_fits<256, 4>(Tp, grid) : ws == 8 ? persist_fwd_pre_fits<256, 8>(Tp, grid) : persist_fwd_pre_fits<256, 16>(Tp, grid);
}
// Shape, switch, CU count AND the occupancy calculator: las_speller_bwd repeats this call to learn what las_speller_fwd did
// (its PRE variant needs the P matrix and the gx slabs the forward's PRE variant left in the reserve).
bool speller_persist_pre_eligible(int B, int Tp, int Hs, int D, int M, int V, int L, int heads, int use_mlp) {
    const bool on = opt_get(OPT_SPELLER_PRE) != 0;
    if (!on || !speller_persist_pre_shape(B, Tp, Hs, D, M, V, L, heads, use_mlp)) return false;
    int cus = 0, dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
        return false;
    const int ws = speller_persist_pre_ws(B, Tp, Hs, cus);
    if (ws == 0) return false;
    return persist_fwd_pre_fits_rt(Hs, ws, Tp, Hs / 4 + ws * B);
}
// Multi-head attention (heads 2 or 4, teacher forcing; reference las_model.py:298-314) on the same kernel: one set of attention workgroups per
// (utterance, head), so B * heads takes the place of B in the workgroup budget (heads = 2: 16 utterances per launch at T' <= 112)
bool speller_persist_pre_mh_shape(int B, int Tp, int Hs, int D, int M, int V, int L, int heads, int use_mlp) {
    if (L != 2 || (heads != 2 && heads != 4) || !use_mlp || M != PS_M || D != Hs) return false;      // (the backward's unit slices are whole 16-unit tiles)
    if (Hs != 256 && Hs != 512) return false;
    if (B < 1 || B * heads > 32 || ((V + 15) & ~15) > 32) return false;
    return speller_persist_pre_ws(B * heads, Tp, Hs, -1) != 0;
}
bool speller_persist_pre_mh_eligible(int B, int Tp, int Hs, int D, int M, int V, int L, int heads, int use_mlp) {
    if (opt_get(OPT_SPELLER_PRE) == 0 || opt_get(OPT_SPELLER_PRE_MH) == 0) return false;
    if (!speller_persist_pre_mh_shape(B, Tp, Hs, D, M, V, L, heads, use_mlp)) return false;
    int cus = 0, dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
        return false;
    const int ws = speller_persist_pre_ws(B * heads, Tp, Hs, cus);
    if (ws == 0) return false;
    return persist_fwd_pre_mh_fits_rt(Hs, ws, Tp, Hs / 4 + ws * B * heads);
}
// ... free-running (decode_mode 1) with several heads: the heads' shares of the character distribution meet like their weighted sums
bool speller_persist_pre_mh_greedy_eligible(int B, int Tp, int Hs, int D, int M, int V, int L, int heads, int use_mlp) {
    if (opt_get(OPT_SPELLER_PRE) == 0 || opt_get(OPT_SPELLER_PRE_MH) == 0 || opt_get(OPT_SPELLER_PRE_GREEDY) == 0 || V > 32) return false;
    if (!speller_persist_pre_mh_shape(B, Tp, Hs, D, M, V, L, heads, use_mlp)) return false;
    int cus = 0, dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
        return false;
    const int ws = speller_persist_pre_ws(B * heads, Tp, Hs, cus);
    if (ws == 0) return false;
    return persist_fwd_pre_mh_greedy_fits_rt(Hs, ws, Tp, Hs / 4 + ws * B * heads);
}
// ... and its free-running (decode_mode 1: fed-back arg-max) form: the same structure with the character distribution inside the attention
// workgroups (AttnPreRole<.., GREEDY>); forward only — a stashing forward (free-running training step) keeps the classic kernels, whose
// backward needs the context the PRE forward never forms on the chain
bool speller_persist_pre_greedy_eligible(int B, int Tp, int Hs, int D, int M, int V, int L, int heads, int use_mlp) {
    if (opt_get(OPT_SPELLER_PRE) == 0 || opt_get(OPT_SPELLER_PRE_GREEDY) == 0) return false;
    if (!speller_persist_pre_shape(B, Tp, Hs, D, M, V, L, heads, use_mlp) || V > 32) return false;
    int cus = 0, dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
        return false;
    const int ws = speller_persist_pre_ws(B, Tp, Hs, cus);
    if (ws == 0) return false;
    return persist_fwd_pre_fits_rt(Hs, ws, Tp, Hs / 4 + ws * B, true);
}

template <int HS, int WS, bool GREEDY = false, bool MH = false>
static int launch_persist_fwd_pre(const PersistArgs& a, int grid, hipStream_t stream) {
    const size_t smem = persist_fwd_pre_smem<HS, WS, GREEDY>(a.Tp);
    if (!persist_fwd_pre_fits<HS, WS, GREEDY, MH>(a.Tp, grid))
        return fail(LAS_ERR_UNSUPPORTED, "persistent decode kernel: %s%ld workgroups cannot all be resident", "", (long)grid);
    {
        KernelTimer timer(TIMED_DECODE_FWD, stream);
        hipLaunchKernelGGL((speller_persist_fwd_pre_kernel<HS, WS, GREEDY, MH>), dim3(grid), dim3(PS_THREADS), smem, stream, a);
    }
    LAS_LAUNCH_CHECK();
    path_note(PATH_DECODE_FWD, GREEDY ? (MH ? "persist_pre_mh_greedy" : "persist_pre_greedy") : (MH ? "persist_pre_mh" : "persist_pre"));
    return LAS_OK;
}

template <int HS, int SPLIT, bool GREEDY>
static int launch_persist_fwd2(const PersistArgs& a, int grid, hipStream_t stream) {
    const size_t smem = sizeof(float) * (size_t)std::max(CellRole<HS, GREEDY>::LDS_FLOATS, AttnRole<HS, SPLIT, GREEDY>::lds_floats(a.Tp, a.V, GREEDY));
    LAS_REQUIRE(smem <= 160 * 1024, "persistent speller LDS budget");
    LAS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&speller_persist_fwd_kernel<HS, SPLIT, GREEDY>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    if (!persistent_launch_fits(speller_persist_fwd_kernel<HS, SPLIT, GREEDY>, PS_THREADS, smem, grid))
        return fail(LAS_ERR_UNSUPPORTED, "persistent decode kernel: %s%ld workgroups cannot all be resident", "", (long)grid);
    {
        KernelTimer timer(TIMED_DECODE_FWD, stream);
        hipLaunchKernelGGL((speller_persist_fwd_kernel<HS, SPLIT, GREEDY>), dim3(grid), dim3(PS_THREADS), smem, stream, a);
    }
    LAS_LAUNCH_CHECK();
    path_note(PATH_DECODE_FWD, "persist");
    return LAS_OK;
}
template <int HS, int SPLIT>
static int launch_persist_fwd(const PersistArgs& a, int grid, hipStream_t stream) {
    return a.mode != 0 ? launch_persist_fwd2<HS, SPLIT, true>(a, grid, stream) : launch_persist_fwd2<HS, SPLIT, false>(a, grid, stream);
}

// sentinel-fill what the phases of the PRE variant hand over: h0 (row-major slabs, first half of hx), h1 (tiled, second half), R0 tiles
int speller_persist_fwd_fill(const PersistFwd& p, hipStream_t stream) {
    const size_t r0_floats = (size_t)p.U * 32 * 4 * p.Hs;
    if (p.r0x == p.hx + (size_t)2 * p.U * 32 * p.Hs) {      // adjacent (the layout las_capi.hip uses): one fill
        LAS_HIP_CHECK(hipMemsetAsync(p.hx, 0xFF, sizeof(float) * ((size_t)2 * p.U * 32 * p.Hs + r0_floats), stream));
    } else {
        LAS_HIP_CHECK(hipMemsetAsync(p.hx, 0xFF, sizeof(float) * 2 * p.U * 32 * p.Hs, stream));
        LAS_HIP_CHECK(hipMemsetAsync(p.r0x, 0xFF, sizeof(float) * r0_floats, stream));
    }
    return LAS_OK;
}

int speller_persist_fwd(const PersistFwd& p, hipStream_t stream) {
    LAS_REQUIRE(p.NH >= 1, "attention heads");
    LAS_REQUIRE(p.pctx != nullptr || speller_persist_eligible(p.B, p.Tp, p.Hs, p.Hs, PS_M, p.V, 2, 1, 1, p.mode != 0), "persistent speller shape");
    LAS_REQUIRE(p.mode >= 0 && p.mode <= 2, "persistent speller mode");
    LAS_REQUIRE(p.mode == 0 || (p.w_c && p.b_c && p.logp && (p.lgx || p.pctx)), "free-running decode needs the character distribution");
    PersistArgs a;
    a.mode = p.mode; a.V = p.V; a.w_c = p.w_c; a.b_c = p.b_c; a.logp = p.logp; a.argmax_out = p.argmax; a.lgx = p.lgx;
    a.w0p = p.w0p; a.ldw0 = p.Vp + p.Hs; a.Vp = p.Vp;
    a.w_hh0 = p.w_hh0; a.w_ih1 = p.w_ih1; a.w_hh1 = p.w_hh1;
    a.b_ih0 = p.b_ih0; a.b_hh0 = p.b_hh0; a.b_ih1 = p.b_ih1; a.b_hh1 = p.b_hh1;
    a.w_phi = p.w_phi; a.b_phi = p.b_phi;
    a.feat = p.feat; a.keys = p.keys; a.y_all = p.y_all;
    a.ctx_all = p.ctx_all; a.h_all = p.h_all; a.c_all = p.c_all; a.gates_all = p.gates_all; a.q_all = p.q_all; a.att = p.att;
    a.hx = p.hx;
    a.B = p.B; a.Tp = p.Tp; a.U = p.U; a.relu = p.relu; a.err = p.err;
    a.split = persist_split(p.B, p.Tp, p.Hs, p.V, p.mode != 0);
    a.trace = g_persist_trace;
    a.pctx = p.pctx; a.gx = p.gx; a.r0x = p.r0x; a.yw = p.yw;
    a.qct = p.qct; a.wyT = p.wyT; a.plx = p.plx;
    a.NH = p.NH; a.p0 = p.p0; a.ex = p.ex;
    LAS_REQUIRE(p.err != nullptr, "the persistent speller needs the device error word");
    if (p.pctx && p.NH > 1) {      // multi-head form of the pre-multiplied context variant (the caller checked speller_persist_pre_mh_eligible)
        int cus = 0, dev = 0;
        LAS_HIP_CHECK(hipGetDevice(&dev));
        LAS_HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        const int ws = speller_persist_pre_ws(p.B * p.NH, p.Tp, p.Hs, cus);
        LAS_REQUIRE((p.mode == 0 || p.mode == 1) && p.gx && p.r0x && p.yw && p.p0 && ws != 0 &&
                    speller_persist_pre_mh_shape(p.B, p.Tp, p.Hs, p.Hs, PS_M, p.V, 2, p.NH, 1), "persistent speller (pre, multi-head) shape");
        a.split = ws;
        if (!p.prefilled) LAS_TRY(speller_persist_fwd_fill(p, stream));
        // the heads' exchange slab + 4 words per (utterance, column part) of XCC ids behind it
        LAS_HIP_CHECK(hipMemsetAsync(p.gx, 0xFF, sizeof(float) * ((size_t)p.U * p.B * p.NH * 4 * p.Hs + (size_t)p.B * 16 * 4), stream));
        const int grid = p.Hs / 4 + ws * p.B * p.NH;
        if (p.mode == 1) {      // free-running: partial logits of h1 (cell workgroups -> head 0) and the heads' logit shares, both sentinel-prefilled
            LAS_REQUIRE(p.qct && p.wyT && p.plx && p.lgx && p.logp && p.w_c && p.b_c && ws != 16, "persistent speller (pre, multi-head, free-running) buffers");
            LAS_HIP_CHECK(hipMemsetAsync(p.plx, 0xFF, sizeof(float) * (size_t)p.U * (p.Hs / 4) * 512, stream));
            LAS_HIP_CHECK(hipMemsetAsync(p.lgx, 0xFF, sizeof(float) * (size_t)p.U * p.B * ws * p.NH * 32, stream));
            if (p.Hs == 512) return ws == 4 ? launch_persist_fwd_pre<512, 4, true, true>(a, grid, stream) : launch_persist_fwd_pre<512, 8, true, true>(a, grid, stream);
            return ws == 4 ? launch_persist_fwd_pre<256, 4, true, true>(a, grid, stream) : launch_persist_fwd_pre<256, 8, true, true>(a, grid, stream);
        }
        if (p.Hs == 512)
            return ws == 4 ? launch_persist_fwd_pre<512, 4, false, true>(a, grid, stream)
                           : ws == 8 ? launch_persist_fwd_pre<512, 8, false, true>(a, grid, stream) : launch_persist_fwd_pre<512, 16, false, true>(a, grid, stream);
        return ws == 4 ? launch_persist_fwd_pre<256, 4, false, true>(a, grid, stream) : launch_persist_fwd_pre<256, 8, false, true>(a, grid, stream);
    }
    if (p.pctx) {      // pre-multiplied context variant (the caller checked speller_persist_pre_eligible)
        int cus = 0, dev = 0;
        LAS_HIP_CHECK(hipGetDevice(&dev));
        LAS_HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        const int ws = speller_persist_pre_ws(p.B, p.Tp, p.Hs, cus);
        LAS_REQUIRE((p.mode == 0 || p.mode == 1) && p.gx && p.r0x && p.yw && ws != 0 && speller_persist_pre_shape(p.B, p.Tp, p.Hs, p.Hs, PS_M, p.V, 2, 1, 1), "persistent speller (pre) shape");
        a.split = ws;
        if (!p.prefilled) LAS_TRY(speller_persist_fwd_fill(p, stream));
        const int grid = p.Hs / 4 + ws * p.B;
        if (p.mode == 1) {      // free-running: the character distribution inside the attention workgroups
            LAS_REQUIRE(p.qct && p.wyT && p.plx && p.logp && p.w_c && p.b_c && (ws != 16 || p.ex), "persistent speller (pre, free-running) buffers");
            LAS_HIP_CHECK(hipMemsetAsync(p.plx, 0xFF, sizeof(float) * (size_t)p.U * (p.Hs / 4) * 512, stream));
            if (ws == 16)      // keys split by frames: the slices' energies meet through this slab (+ the XCC ids of the placement check)
                LAS_HIP_CHECK(hipMemsetAsync(p.ex, 0xFF, sizeof(float) * ((size_t)p.U * p.B * 16 * 64 + (size_t)p.B * 16), stream));
            if (p.Hs == 512)
                return ws == 4 ? launch_persist_fwd_pre<512, 4, true>(a, grid, stream)
                               : ws == 8 ? launch_persist_fwd_pre<512, 8, true>(a, grid, stream) : launch_persist_fwd_pre<512, 16, true>(a, grid, stream);
            return ws == 4 ? launch_persist_fwd_pre<256, 4, true>(a, grid, stream)
                           : ws == 8 ? launch_persist_fwd_pre<256, 8, true>(a, grid, stream) : launch_persist_fwd_pre<256, 16, true>(a, grid, stream);
        }
        if (p.Hs == 512)
            return ws == 4 ? launch_persist_fwd_pre<512, 4>(a, grid, stream)
                           : ws == 8 ? launch_persist_fwd_pre<512, 8>(a, grid, stream) : launch_persist_fwd_pre<512, 16>(a, grid, stream);
        if (ws == 16) {      // frame-split keys: the slices' energies are exchanged through a sentinel-prefilled slab
            LAS_REQUIRE(p.ex != nullptr, "persistent speller (pre, frame-split keys) exchange slab");
            LAS_HIP_CHECK(hipMemsetAsync(p.ex, 0xFF, sizeof(float) * ((size_t)p.U * p.B * 16 * 64 + (size_t)p.B * 16), stream));      // (+ the XCC ids of the placement check)
            return launch_persist_fwd_pre<256, 16>(a, grid, stream);
        }
        return ws == 4 ? launch_persist_fwd_pre<256, 4>(a, grid, stream) : launch_persist_fwd_pre<256, 8>(a, grid, stream);
    }
    // sentinel-fill what the phases hand over: every h of both layers and the contexts of steps 1..U
    const size_t sH = (size_t)p.B * p.Hs;
    LAS_HIP_CHECK(hipMemsetAsync(p.hx, 0xFF, sizeof(float) * 2 * p.U * 32 * p.Hs, stream));
    LAS_HIP_CHECK(hipMemsetAsync(p.ctx_all + sH, 0xFF, sizeof(float) * p.U * sH, stream));
    if (p.mode != 0) LAS_HIP_CHECK(hipMemsetAsync(p.lgx, 0xFF, sizeof(float) * (size_t)p.U * p.B * a.split * 32, stream));
    const int grid = p.Hs / 4 + a.split * p.B;
    if (p.Hs == 512) {
        if (a.split == 2) return launch_persist_fwd<512, 2>(a, grid, stream);
        if (a.split == 4) return launch_persist_fwd<512, 4>(a, grid, stream);
        return launch_persist_fwd<512, 8>(a, grid, stream);
    }
    if (a.split == 2) return launch_persist_fwd<256, 2>(a, grid, stream);
    if (a.split == 4) return launch_persist_fwd<256, 4>(a, grid, stream);
    return launch_persist_fwd<256, 8>(a, grid, stream);
}

}  // namespace las
