// Persistent-RNN time recurrence of one bidirectional LSTM layer (forward and BPTT) for gfx950.
//
// Replaces the recurrent half of nn.LSTM(bidirectional=True) that the reference's pBLSTMLayer calls
// (reference model/las_model.py:72-79,90; cell equations are torch's: gates i,f,g,o, zero initial state,
// reverse direction runs t = T-1..0).  The input half (x_t W_ih^T + biases) is the MFMA GEMM in gemm_f32.hip.
//
// Design (CDNA4):
//   * One *group* of G workgroups (1024 threads = 16 wave64 each, one per CU) owns ONE (utterance, direction)
//     for all T steps.  W_hh (4H x H fp32: 256 KB at H=128, 1 MB at H=256) never leaves the register file:
//     each thread keeps 64 weights (4 gates x 16 k) in VGPRs, so a CU holds 256 KB and G = H^2/16384 CUs hold
//     the whole matrix (G=1 at H=128, G=4 at H=256, G=16 at H=512).  Per step only h_{t-1} (H floats) moves.
//   * Inside a CU h_{t-1} lives in LDS (double buffered, one barrier per step); each unit's 4 gate rows are
//     split over LPU = H/16 lanes, reduced with cross-lane shuffles, and the owning lane applies the cell.
//   * Between the G CUs of a group the new h slice travels as 8-byte {epoch tag, value} granules written with
//     ONE agent-scope (sc1, write-through) store each and polled with relaxed agent-scope loads: the data is
//     the flag, no fence, placement-independent (MI355X per-XCD L2s are not coherent).  Two parity slots
//     make the protocol race-free; every spin is bounded and reports through a device error word.
//   * Group members are placed on the same XCD when the dispatcher's observed round-robin holds (speed only).
// A generic fallback (any H, weights streamed from L2) keeps every shape correct.
#include "las_common.h"
#include "las_kernels.h"
#include "options.h"
#include <stdlib.h>
#include <algorithm>
#include <mutex>

namespace las {

constexpr int REC_THREADS = 1024;
#ifdef LAS_REC_TRACE
// Profiling build only (make CXXFLAGS_EXTRA=-DLAS_REC_TRACE): per-phase wall-clock stamps of workgroup 0 of the backward
// kernel (tools/ubench_rec_trace.py).  Each stamp costs ~0.15 us: read ratios, not absolute times.
__device__ unsigned long long* g_rec_trace = nullptr;
#define REC_STAMP(who, step, k) do { if (g_rec_trace && blockIdx.x == 0 && threadIdx.x == (who)) g_rec_trace[(((who) ? 1 : 0) * 4096 + (step)) * 8 + (k)] = wall_clock64(); } while (0)
#else
#define REC_STAMP(who, step, k) do { } while (0)
#endif
constexpr unsigned SPIN_LIMIT = 1u << 18;      // ~40 ms of bounded spinning (see persist_common.h)
using u64 = unsigned long long;

__device__ __forceinline__ float poll_granule(u64* g, unsigned epoch, unsigned* err) {
    unsigned spins = 0;
    for (;;) {
        const u64 x = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((unsigned)(x >> 32) == epoch) return __uint_as_float((unsigned)x);
        ++spins;
        if ((spins & 255u) == 0) {
            if (spins > SPIN_LIMIT && spins > __hip_atomic_load(err + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {      // err[1]: the caller's extended budget
                atomicExch(err, 0xDEAD0001u); return 0.f;
            }
            if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return 0.f;
        }
        __builtin_amdgcn_s_sleep(1);
    }
}

__device__ __forceinline__ void publish_granule(u64* g, unsigned epoch, float v) {
    __hip_atomic_store(g, ((u64)epoch << 32) | (u64)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Same-XCD publish: a plain 8-byte store stays in the XCD's (shared, coherent) L2, where the group's other CUs read
// it with an L1-bypassing load at L2 latency; the agent-scope (sc1) form above writes through to memory and drops
// the line, so every reader pays a fabric round trip.  ONLY valid when every member of the group has verified at
// run time that the whole group sits on one XCD (same_xcd_group below); otherwise the agent-scope form is used.
__device__ __forceinline__ void publish_granule_l2(u64* g, unsigned epoch, float v) {
    *reinterpret_cast<volatile u64*>(g) = ((u64)epoch << 32) | (u64)__float_as_uint(v);
}

constexpr int XID_SLOTS = 32;        // id granules per group (G <= 32)

// Run-time placement check, placement-independent itself: every member publishes its XCC id with the agent-scope
// protocol and reads all the others'.  Returns true iff all G members of this group run on the same XCD.
// `tag` marks this launch's ids: 0xC0DE0001 on a buffer the host zeroed, the launch's epoch base on the library's persistent scratch
// (rec_scratch below), whose slots still hold the ids of earlier launches under THEIR (smaller) bases.
template <int G>
__device__ __forceinline__ bool same_xcd_group(u64* idbuf, int member, unsigned* err, int* lds_flag, unsigned tag = 0xC0DE0001u) {
    if (G == 1) return true;
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcc));
    const int tid = threadIdx.x;
    if (tid == 0) {
        *lds_flag = 1;
        __hip_atomic_store(idbuf + member, ((u64)tag << 32) | (u64)xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (tid < G) {
        unsigned spins = 0;
        u64 x;
        for (;;) {
            x = __hip_atomic_load(idbuf + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((unsigned)(x >> 32) == tag) break;
            if (++spins > SPIN_LIMIT && spins > __hip_atomic_load(err + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { atomicExch(err, 0xDEAD0002u); break; }
            __builtin_amdgcn_s_sleep(2);
        }
        if ((unsigned)x != xcc || (unsigned)(x >> 32) != tag) *lds_flag = 0;
    }
    __syncthreads();
    return *lds_flag != 0;
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains the vector-memory counter, i.e. it
// waits for this step's global stores (h, stash) to be acknowledged — ~0.3 us per recurrent step for nothing.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// group / member decode shared by forward and backward
template <int G>
__device__ __forceinline__ void decode_block(int ngroups, int& group, int& member) {
    const int bid = blockIdx.x;
    if (G > 1 && (ngroups & 7) == 0) {      // XCD-local groups under round-robin dispatch (block b -> XCD b%8)
        const int xcd = bid & 7, q = bid >> 3;
        member = q % G;
        group = (q / G) * 8 + xcd;
    } else {
        member = bid % G;
        group = bid / G;
    }
}

// sum over aligned groups of W (<= 16) lanes with DPP row operations: 1 VALU instruction per level, no LDS traffic
template <int W>
__device__ __forceinline__ float row_sum(float v) {
    if (W >= 2) v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
    if (W >= 4) v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
    if (W >= 8) v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));  // row_half_mirror
    if (W >= 16) v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true)); // row_mirror
    return v;
}

// ------------------------------------------------------------------------------------------------
// Forward, register-resident W_hh.
// Per step: (1) every wave does its share of the mat-vec (4 LDS reads, 64 FMAs, DPP reduction) and drops the gate
// sums of its units into LDS; (2) ONE wave per 64 units applies the cell with all lanes active (coalesced
// pre-activation loads and h / stash stores) and publishes h, while the other waves already poll the other CUs'
// granules.  The per-step instruction stream — the real bound of this latency-critical loop — stays ~100
// instructions per wave instead of every wave running the cell code with 4 of 64 lanes active.
// ------------------------------------------------------------------------------------------------
template <int H, int UW, bool STASH>
__global__ __launch_bounds__(UW * (H / 16)) void rec_fwd_fast(float* __restrict__ gates, const float* __restrict__ w_hh_f,
                                                              const float* __restrict__ w_hh_r, float* __restrict__ out,
                                                              float* __restrict__ cbuf, float* __restrict__ hprev, int B,
                                                              int T, u64* xbuf, unsigned* err, int force_agent, unsigned ebase) {
    // ebase: epoch base of this launch's hand-off granules (0: the host zeroed xbuf; else xbuf is the library's persistent scratch, whose
    // stale granules all carry smaller epochs — no fill between launches, see rec_scratch)
    constexpr int LPU = H / 16;             // lanes cooperating on one hidden unit (16 k-values each)
    constexpr int NT = UW * LPU;            // threads: UW hidden units owned by this workgroup
    constexpr int G = H / UW;               // workgroups per (utterance, direction)
    constexpr int PS = (UW + 63) / 64 * 64;   // pollers start on a wave boundary: a wave that both publishes and polls
                                              // could run its poll branch first and deadlock the group
    static_assert(G == 1 || NT > PS, "no polling threads");
    __shared__ __attribute__((aligned(16))) float hs[2][H];
    __shared__ __attribute__((aligned(16))) float gsum[UW][4];
    __shared__ int xcd_flag;

    int group, member;
    decode_block<G>(2 * B, group, member);
    const bool l2x = !force_agent && same_xcd_group<G>(xbuf + (long)2 * B * 2 * H + (long)group * XID_SLOTS, member, err, &xcd_flag, ebase ? ebase : 0xC0DE0001u);
    const int dir = group & 1, b = group >> 1;
    const float* __restrict__ w_hh = dir ? w_hh_r : w_hh_f;
    const int tid = threadIdx.x;
    const int kc = tid % LPU, ul = tid / LPU;

    f32x4 w[4][4];
    {
        const int j = member * UW + ul;
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                w[g][i] = *reinterpret_cast<const f32x4*>(w_hh + (long)(g * H + j) * H + i * 4 * LPU + kc * 4);
    }
    for (int i = tid; i < 2 * H; i += NT) (&hs[0][0])[i] = 0.f;
    __syncthreads();

    // cell role: thread tid < UW owns hidden unit jc
    const bool cell = tid < UW;
#ifdef LAS_REC_PRIO
    if (cell) __builtin_amdgcn_s_setprio(LAS_REC_PRIO);      // the step's dependent chain runs through this wave
#endif
    const int jc = member * UW + (cell ? tid : 0);
    const long seq = (long)(dir * B + b) * T;
    float* gb = gates + seq * 4 * H + jc;
    float c = 0.f;
    float pre[4] = {0.f, 0.f, 0.f, 0.f};
    if (cell) {
        const int t0 = dir ? T - 1 : 0;
#pragma unroll
        for (int g = 0; g < 4; ++g) pre[g] = gb[(long)t0 * 4 * H + g * H];
    }
    // polling role: threads of the waves after the cell waves fetch the H - UW foreign units
    const bool poller = G > 1 && tid >= PS;
    u64* xg = xbuf + (long)group * 2 * H;

    int cur = 0;
    for (int step = 0; step < T; ++step) {
        const int t = dir ? T - 1 - step : step;
        const unsigned epoch = ebase + (unsigned)step + 1u;
        REC_STAMP(0, step, 0); REC_STAMP(512, step, 0);
        float nxt[4] = {0.f, 0.f, 0.f, 0.f};
        if (cell && step + 1 < T) {
            const int tn = dir ? t - 1 : t + 1;
#pragma unroll
            for (int g = 0; g < 4; ++g) nxt[g] = gb[(long)tn * 4 * H + g * H];
        }
        f32x4 hv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) hv[i] = *reinterpret_cast<const f32x4*>(&hs[cur][i * 4 * LPU + kc * 4]);
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int g = 0; g < 4; ++g) acc[g] = fmaf(w[g][i][e], hv[i][e], acc[g]);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            acc[g] = row_sum<(LPU < 16 ? LPU : 16)>(acc[g]);
            if (LPU > 16) acc[g] += __shfl_xor(acc[g], 16);
        }
        if (kc == 0) *reinterpret_cast<f32x4*>(&gsum[ul][0]) = f32x4{acc[0], acc[1], acc[2], acc[3]};
        REC_STAMP(0, step, 1); REC_STAMP(512, step, 1);
        lds_barrier();
        REC_STAMP(0, step, 2); REC_STAMP(512, step, 2);

        if (cell) {
            const f32x4 s4 = *reinterpret_cast<const f32x4*>(&gsum[tid][0]);
            const float ig = sigmoidf_acc(s4[0] + pre[0]);
            const float fg = sigmoidf_acc(s4[1] + pre[1]);
            const float gg = tanhf_acc(s4[2] + pre[2]);
            const float og = sigmoidf_acc(s4[3] + pre[3]);
            c = fg * c + ig * gg;
            const float h = og * tanhf_acc(c);
            if (G > 1) {
                u64* gp64 = xg + (step & 1) * H + jc;
                if (l2x) publish_granule_l2(gp64, epoch, h); else publish_granule(gp64, epoch, h);
            }
            REC_STAMP(0, step, 5);
            const float hp = hs[cur][jc];
            hs[cur ^ 1][jc] = h;
            out[((long)b * T + t) * 2 * H + dir * H + jc] = h;
            if (STASH) {
                hprev[(seq + t) * H + jc] = hp;
                cbuf[(seq + t) * H + jc] = c;
                float* gp = gb + (long)t * 4 * H;
                gp[0] = ig; gp[H] = fg; gp[2 * H] = gg; gp[3 * H] = og;
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) pre[g] = nxt[g];
        } else if (poller) {
            for (int fidx = tid - PS; fidx < H - UW; fidx += NT - PS) {
                const int fu = fidx < member * UW ? fidx : fidx + UW;       // foreign hidden unit index
                hs[cur ^ 1][fu] = poll_granule(xg + (step & 1) * H + fu, epoch, err);
            }
        }
        REC_STAMP(0, step, 3); REC_STAMP(512, step, 3);
        lds_barrier();
        REC_STAMP(0, step, 4); REC_STAMP(512, step, 4);
        cur ^= 1;
    }
}


// ------------------------------------------------------------------------------------------------
// Forward, register-resident W_hh, NB utterances per group ("multi").
// One group (G workgroups, one per CU) keeps its W_hh slice in VGPRs and steps NB utterances of the same direction in
// lock-step: per round every wave does its share of NB mat-vecs against the SAME resident weights, then NB*UW cell
// threads (one wave per utterance at UW = 64) apply the cells and publish, while the remaining waves gather the other
// members' h for all NB utterances.  The two barriers and the inter-CU hand-off latency of a step are paid once per NB
// utterances, and the launch never needs more than 2*ceil(B/NB)*G <= #CU resident workgroups, whatever the batch.
// ------------------------------------------------------------------------------------------------
// PIPE (G > 1): the block's utterances are stepped as two halves half a step apart — while every thread multiplies one half's
// h by the resident W_hh slice, the other half's cell waves apply the cell and publish, and its foreign h components travel
// between the CUs: the inter-CU hand-off and the cell wave's dependent chain hide behind the other half's mat-vecs.
template <int H, int UW, int NB, bool STASH, bool PIPE = false>
__global__ __launch_bounds__(UW * (H / 16)) void rec_fwd_multi(float* __restrict__ gates, const float* __restrict__ w_hh_f,
                                                               const float* __restrict__ w_hh_r, float* __restrict__ out,
                                                               float* __restrict__ cbuf, float* __restrict__ hprev, int B,
                                                               int T, u64* xbuf, unsigned* err, int force_agent, int b0,
                                                               int Bc, unsigned ebase) {
    // b0 / Bc: this launch steps utterances [b0, b0 + Bc) of the B in the buffers (host-side chunking of very large batches)
    constexpr int LPU = H / 16;
    constexpr int NT = UW * LPU;
    constexpr int G = H / UW;
    constexpr int NC = NB * UW;                 // cell threads: (utterance u, unit) = (tid / UW, tid % UW)
    constexpr int PS = (NC + 63) / 64 * 64;     // pollers start on a wave boundary (see rec_fwd_fast)
    static_assert(NC <= NT, "more cells than threads");
    static_assert(G == 1 || NT > PS, "no polling threads");
    __shared__ __attribute__((aligned(16))) float hs[2][NB][H];
    __shared__ __attribute__((aligned(16))) float gsum[NB][UW][4];
    __shared__ int xcd_flag;

    const int nblk = (Bc + NB - 1) / NB;        // utterance blocks per direction
    int group, member;
    decode_block<G>(2 * nblk, group, member);
    const bool l2x = !force_agent && same_xcd_group<G>(xbuf + (long)2 * nblk * 2 * NB * H + (long)group * XID_SLOTS, member, err, &xcd_flag, ebase ? ebase : 0xC0DE0001u);
    const int dir = group & 1, blk = group >> 1;
    const float* __restrict__ w_hh = dir ? w_hh_r : w_hh_f;
    const int tid = threadIdx.x;
    const int kc = tid % LPU, ul = tid / LPU;

    f32x4 w[4][4];
    {
        const int j = member * UW + ul;
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                w[g][i] = *reinterpret_cast<const f32x4*>(w_hh + (long)(g * H + j) * H + i * 4 * LPU + kc * 4);
    }
    for (int i = tid; i < 2 * NB * H; i += NT) (&hs[0][0][0])[i] = 0.f;
    __syncthreads();

    const bool cellt = tid < NC;
    const int cu = cellt ? tid / UW : 0;
    const int jc = member * UW + (cellt ? tid % UW : 0);
    const int b = b0 + blk * NB + cu;
    const bool cell = cellt && blk * NB + cu < Bc;   // utterances past the chunk end are idle slots of the last block
    const long seq = (long)(dir * B + (cell ? b : 0)) * T;
    float* gb = gates + seq * 4 * H + jc;
    float c = 0.f;
    float pre[4] = {0.f, 0.f, 0.f, 0.f};
    if (cell) {
        const int t0 = dir ? T - 1 : 0;
#pragma unroll
        for (int g = 0; g < 4; ++g) pre[g] = gb[(long)t0 * 4 * H + g * H];
    }
    const bool poller = G > 1 && tid >= PS;
    u64* xg = xbuf + (long)group * 2 * NB * H;   // [step parity][utterance][unit]
    const int nvalid = min(NB, Bc - blk * NB);   // live utterances of this block

    int cur = 0;
    if constexpr (PIPE) {
        constexpr int NBH = NB / 2;
        static_assert(!PIPE || (NB >= 2 && NB % 2 == 0 && G > 1), "pipelined stepping needs two halves and a hand-off to hide");
        auto matvec = [&](int u0, int buf) {          // gsum[u] = W_hh slice . h_u for the NBH utterances from u0
#pragma unroll
            for (int uu = 0; uu < NBH; ++uu) {
                const int u = u0 + uu;
                f32x4 hv[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) hv[i] = *reinterpret_cast<const f32x4*>(&hs[buf][u][i * 4 * LPU + kc * 4]);
                float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int g = 0; g < 4; ++g) acc[g] = fmaf(w[g][i][e], hv[i][e], acc[g]);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    acc[g] = row_sum<(LPU < 16 ? LPU : 16)>(acc[g]);
                    if (LPU > 16) acc[g] += __shfl_xor(acc[g], 16);
                }
                if (kc == 0) *reinterpret_cast<f32x4*>(&gsum[u][ul][0]) = f32x4{acc[0], acc[1], acc[2], acc[3]};
            }
        };
        auto cellstep = [&](int step, int t, unsigned epoch) {
            const f32x4 s4 = *reinterpret_cast<const f32x4*>(&gsum[cu][tid % UW][0]);
            const float ig = sigmoidf_acc(s4[0] + pre[0]);
            const float fg = sigmoidf_acc(s4[1] + pre[1]);
            const float gg = tanhf_acc(s4[2] + pre[2]);
            const float og = sigmoidf_acc(s4[3] + pre[3]);
            c = fg * c + ig * gg;
            const float h = og * tanhf_acc(c);
            u64* gp64 = xg + ((long)(step & 1) * NB + cu) * H + jc;
            if (l2x) publish_granule_l2(gp64, epoch, h); else publish_granule(gp64, epoch, h);
            const float hp = hs[cur][cu][jc];
            hs[cur ^ 1][cu][jc] = h;
            out[((long)b * T + t) * 2 * H + dir * H + jc] = h;
            if (STASH) {
                hprev[(seq + t) * H + jc] = hp;
                cbuf[(seq + t) * H + jc] = c;
                float* gp = gb + (long)t * 4 * H;
                gp[0] = ig; gp[H] = fg; gp[2 * H] = gg; gp[3 * H] = og;
            }
            if (step + 1 < T) {      // next step's pre-activations: consumed a whole step from now
                const int tn = dir ? t - 1 : t + 1;
#pragma unroll
                for (int g = 0; g < 4; ++g) pre[g] = gb[(long)tn * 4 * H + g * H];
            }
        };
        auto gather = [&](int u0, int step, unsigned epoch) {     // foreign h of the live utterances among [u0, u0 + NBH)
            const int nv = max(0, min(NBH, nvalid - u0));
            for (int fidx = tid - PS; fidx < nv * (H - UW); fidx += NT - PS) {
                const int u = u0 + fidx / (H - UW), f = fidx % (H - UW);
                const int fu = f < member * UW ? f : f + UW;
                hs[cur ^ 1][u][fu] = poll_granule(xg + ((long)(step & 1) * NB + u) * H + fu, epoch, err);
            }
        };
        matvec(0, 0);                  // h_{-1} = 0
        lds_barrier();
        for (int step = 0; step < T; ++step) {
            const int t = dir ? T - 1 - step : step;
            const unsigned epoch = ebase + (unsigned)step + 1u;
            // phase 1: cells of the first half | mat-vecs of the second half | foreign h of the first half
            if (cell && cu < NBH) cellstep(step, t, epoch);
            matvec(NBH, cur);
            if (poller) gather(0, step, epoch);
            lds_barrier();
            // phase 2: cells of the second half | next step's mat-vecs of the first half | foreign h of the second half
            if (cell && cu >= NBH) cellstep(step, t, epoch);
            if (step + 1 < T) matvec(0, cur ^ 1);
            if (poller) gather(NBH, step, epoch);
            lds_barrier();
            cur ^= 1;
        }
        return;
    }
    for (int step = 0; step < T; ++step) {
        const int t = dir ? T - 1 - step : step;
        const unsigned epoch = ebase + (unsigned)step + 1u;
        REC_STAMP(0, step, 0); REC_STAMP(512, step, 0);
        float nxt[4] = {0.f, 0.f, 0.f, 0.f};
        if (cell && step + 1 < T) {
            const int tn = dir ? t - 1 : t + 1;
#pragma unroll
            for (int g = 0; g < 4; ++g) nxt[g] = gb[(long)tn * 4 * H + g * H];
        }
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            f32x4 hv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) hv[i] = *reinterpret_cast<const f32x4*>(&hs[cur][u][i * 4 * LPU + kc * 4]);
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int g = 0; g < 4; ++g) acc[g] = fmaf(w[g][i][e], hv[i][e], acc[g]);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                acc[g] = row_sum<(LPU < 16 ? LPU : 16)>(acc[g]);
                if (LPU > 16) acc[g] += __shfl_xor(acc[g], 16);
            }
            if (kc == 0) *reinterpret_cast<f32x4*>(&gsum[u][ul][0]) = f32x4{acc[0], acc[1], acc[2], acc[3]};
        }
        REC_STAMP(0, step, 1); REC_STAMP(512, step, 1);
        lds_barrier();
        REC_STAMP(0, step, 2); REC_STAMP(512, step, 2);

        if (cell) {
            const f32x4 s4 = *reinterpret_cast<const f32x4*>(&gsum[cu][tid % UW][0]);
            const float ig = sigmoidf_acc(s4[0] + pre[0]);
            const float fg = sigmoidf_acc(s4[1] + pre[1]);
            const float gg = tanhf_acc(s4[2] + pre[2]);
            const float og = sigmoidf_acc(s4[3] + pre[3]);
            c = fg * c + ig * gg;
            const float h = og * tanhf_acc(c);
            if (G > 1) {
                u64* gp64 = xg + ((long)(step & 1) * NB + cu) * H + jc;
                if (l2x) publish_granule_l2(gp64, epoch, h); else publish_granule(gp64, epoch, h);
            }
            const float hp = hs[cur][cu][jc];
            hs[cur ^ 1][cu][jc] = h;
            out[((long)b * T + t) * 2 * H + dir * H + jc] = h;
            if (STASH) {
                hprev[(seq + t) * H + jc] = hp;
                cbuf[(seq + t) * H + jc] = c;
                float* gp = gb + (long)t * 4 * H;
                gp[0] = ig; gp[H] = fg; gp[2 * H] = gg; gp[3 * H] = og;
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) pre[g] = nxt[g];
        } else if (poller) {
            for (int fidx = tid - PS; fidx < nvalid * (H - UW); fidx += NT - PS) {
                const int u = fidx / (H - UW), f = fidx % (H - UW);
                const int fu = f < member * UW ? f : f + UW;                // foreign hidden unit index
                hs[cur ^ 1][u][fu] = poll_granule(xg + ((long)(step & 1) * NB + u) * H + fu, epoch, err);
            }
        }
        REC_STAMP(0, step, 3); REC_STAMP(512, step, 3);
        lds_barrier();
        REC_STAMP(0, step, 4); REC_STAMP(512, step, 4);
        cur ^= 1;
    }
}

// ------------------------------------------------------------------------------------------------
// Forward, generic fallback: one workgroup per (utterance, direction), W_hh streamed from L2
// ------------------------------------------------------------------------------------------------
template <bool STASH>
__global__ __launch_bounds__(256) void rec_fwd_generic(float* __restrict__ gates, const float* __restrict__ w_hh_f,
                                                       const float* __restrict__ w_hh_r, float* __restrict__ out,
                                                       float* __restrict__ cbuf, float* __restrict__ hprev, int B, int T,
                                                       int H) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* hs0 = smem; float* hs1 = smem + H; float* cs = smem + 2 * H;
    const int dir = blockIdx.x & 1, b = blockIdx.x >> 1;
    const float* __restrict__ w_hh = dir ? w_hh_r : w_hh_f;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
    for (int i = tid; i < H; i += blockDim.x) { hs0[i] = 0.f; cs[i] = 0.f; }
    __syncthreads();
    float* gb = gates + ((long)(dir * B + b) * T) * 4 * H;
    for (int step = 0; step < T; ++step) {
        const int t = dir ? T - 1 - step : step;
        float* hc = (step & 1) ? hs1 : hs0;
        float* hn = (step & 1) ? hs0 : hs1;
        float* gp = gb + (long)t * 4 * H;
        for (int j = wave; j < H; j += nw) {
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
            for (int k = lane; k < H; k += 64) {
                const float hv = hc[k];
#pragma unroll
                for (int g = 0; g < 4; ++g) acc[g] = fmaf(w_hh[(long)(g * H + j) * H + k], hv, acc[g]);
            }
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1)
#pragma unroll
                for (int g = 0; g < 4; ++g) acc[g] += __shfl_xor(acc[g], m);
            if (lane == 0) {
                const float ig = sigmoidf_acc(acc[0] + gp[j]);
                const float fg = sigmoidf_acc(acc[1] + gp[H + j]);
                const float gg = tanhf_acc(acc[2] + gp[2 * H + j]);
                const float og = sigmoidf_acc(acc[3] + gp[3 * H + j]);
                const float c = fg * cs[j] + ig * gg;
                const float h = og * tanhf_acc(c);
                cs[j] = c;
                hn[j] = h;
                out[((long)b * T + t) * 2 * H + dir * H + j] = h;
                if (STASH) {
                    const long o = ((long)(dir * B + b) * T + t) * H + j;
                    hprev[o] = hc[j];
                    cbuf[o] = c;
                    gp[j] = ig; gp[H + j] = fg; gp[2 * H + j] = gg; gp[3 * H + j] = og;
                }
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// Backward (BPTT), register-resident W_hh^T.  Every workgroup of a group recomputes the full dG_t (cheap,
// H threads) so that only dh (H floats) is exchanged per step, exactly like the forward pass.
// ------------------------------------------------------------------------------------------------
template <int H>
__global__ __launch_bounds__(REC_THREADS) void rec_bwd_fast(const float* __restrict__ dout, const float* __restrict__ gates,
                                                            const float* __restrict__ cbuf, const float* __restrict__ w_hh_t,
                                                            float* __restrict__ dgates, int B, int T, u64* xbuf,
                                                            unsigned* err, float* __restrict__ db_f, float* __restrict__ db_r, unsigned ebase,
                                                            int nx, unsigned* probe) {
    // nx in {2, 4}: XCD-confined launch — the grid is over-subscribed 8 / nx times and only the workgroups that round-robin dispatch puts on
    // XCDs [0, nx) take part (the placement check below still decides which hand-off protocol is safe); 0: the whole chip
    // db_f / db_r (optional): 2 x (4H) bias gradients per direction [b_ih | b_hh], pre-zeroed by the caller: the threads
    // that copy this workgroup's dG rows to memory also sum them over time and add the totals once at the end
    // (replaces a column-sum kernel over the (B*T, 4H) gate gradients per direction).
    constexpr int LPU = H / 16;
    constexpr int UW = REC_THREADS / LPU;
    constexpr int G = H / UW;
    // Thread roles besides the mat-vec that every thread does.  The step's dependent chain runs through the H "cell"
    // threads (dh_t -> dG_t) — everything else is kept off them: other waves poll the foreign dh components, compute the
    // dh-independent factors of the NEXT step's cell backward (tanh(c), gate derivatives) from the prefetched stash and
    // leave them in LDS, and write this workgroup's share of dG to memory.
    constexpr bool SPLIT = 4 * H <= REC_THREADS;          // enough waves for separate roles (H <= 256)
    constexpr int PB = SPLIT ? H : 0;                     // pollers: threads [H, 2H), or the cell threads themselves
    constexpr int QB = SPLIT ? 2 * H : 0;                 // factor threads: [2H, 3H), or the cell threads themselves
    constexpr int SB = SPLIT ? 3 * H : 0;                 // stash writers: [3H, 4H), or the cell threads themselves
    __shared__ __attribute__((aligned(16))) float dhs[2][H];
    __shared__ __attribute__((aligned(16))) float dgs[4 * H];
    __shared__ __attribute__((aligned(16))) float fac[2][H][8];      // per unit: beta, a_i, a_f, a_g | a_o, f, dout, -
    __shared__ int xcd_flag;

    int group, member;
    if (nx > 0) {
        // Which XCD a block lands on is (blockIdx + c) % 8 with a queue-history-dependent rotation c (measured: tools/xcd_probe.py), so the
        // workgroups that take part are chosen by their ACTUAL XCC id — the ones on XCDs [0, nx), which the partitioned GEMM group avoids by the
        // same test — and numbered within their XCD by blockIdx / 8 (one residue class of blockIdx per XCD).  Should the dispatcher ever place
        // blocks differently, a role is missing and the bounded hand-off spins report it through the error word (never a wrong result).
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcc));
        const int xcd = (int)xcc, q = (int)(blockIdx.x >> 3);
        if (probe && threadIdx.x == 0 && blockIdx.x < 1024) probe[blockIdx.x] = xcc + 1;
        if (xcd >= nx) return;                       // (whole workgroup, before any barrier)
        member = q % G;
        group = (q / G) * nx + xcd;
    } else {
        decode_block<G>(2 * B, group, member);
    }
    const bool l2x = same_xcd_group<G>(xbuf + (long)2 * B * 2 * H + (long)group * XID_SLOTS, member, err, &xcd_flag, ebase ? ebase : 0xC0DE0001u);
    const int dir = group & 1, b = group >> 1;
    const int tid = threadIdx.x;
    const int rc = tid % LPU, kl = tid / LPU;
    const int k = member * UW + kl;                      // the dh_{t-1} component this thread group produces
    const float* __restrict__ wt = w_hh_t + ((long)dir * H + k) * 4 * H;

    f32x4 w[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) w[i] = *reinterpret_cast<const f32x4*>(wt + i * 4 * LPU + rc * 4);

    const long seq = (long)(dir * B + b) * T;
    const float* gb = gates + seq * 4 * H;
    const float* cb = cbuf + seq * H;
    float* dgb = dgates + seq * 4 * H;
    const float* dob = dout + ((long)b * T) * 2 * H + dir * H;
    u64* xg = xbuf + (long)group * 2 * H;

    for (int i = tid; i < H; i += REC_THREADS) dhs[0][i] = 0.f;

    const bool cellt = tid < H;                          // cell role: unit j = tid
#ifdef LAS_REC_PRIO
    if (cellt) __builtin_amdgcn_s_setprio(LAS_REC_PRIO);
#endif
    const bool fact = tid >= QB && tid < QB + H;         // factor role: unit tid - QB
    const int jf = fact ? tid - QB : 0;
    float dc = 0.f;
    // raw stash values of a coming step (factor threads), fetched one step ahead
    float p_i = 0, p_f = 0, p_g = 0, p_o = 0, p_c = 0, p_cp = 0, p_do = 0;
    auto load_step = [&](int st) {                       // st: step in processing order
        if (st >= T) return;
        const int t = dir ? st : T - 1 - st;
        const float* gp = gb + (long)t * 4 * H + jf;
        p_i = gp[0]; p_f = gp[H]; p_g = gp[2 * H]; p_o = gp[3 * H];
        p_c = cb[(long)t * H + jf];
        const int tp = dir ? t + 1 : t - 1;              // previously processed time in the FORWARD pass
        p_cp = (tp >= 0 && tp < T) ? cb[(long)tp * H + jf] : 0.f;
        p_do = dob[(long)t * 2 * H + jf];
    };
    auto prepare = [&](int par) {                        // factors of the step whose raw values are in p_*
        const float tc = tanhf_acc(p_c);
        const f32x4 lo = {p_o * (1.f - tc * tc), p_g * p_i * (1.f - p_i), p_cp * p_f * (1.f - p_f), p_i * (1.f - p_g * p_g)};
        const f32x4 hi = {tc * p_o * (1.f - p_o), p_f, p_do, 0.f};
        *reinterpret_cast<f32x4*>(&fac[par][jf][0]) = lo;
        *reinterpret_cast<f32x4*>(&fac[par][jf][4]) = hi;
    };
    if (fact) { load_step(0); prepare(0); load_step(1); }
    __syncthreads();

    constexpr int NBS = (4 * UW + H - 1) / H;             // dG entries per stash-writer thread
    float bsum[NBS];
#pragma unroll
    for (int i = 0; i < NBS; ++i) bsum[i] = 0.f;

    int cur = 0;
    for (int step = 0; step < T; ++step) {
        const int t = dir ? step : T - 1 - step;         // reverse of the forward processing order
        const int par = step & 1;
        REC_STAMP(0, step, 0); REC_STAMP(512, step, 0);
        if (cellt) {
            const int j = tid;
            const f32x4 lo = *reinterpret_cast<const f32x4*>(&fac[par][j][0]);
            const f32x4 hi = *reinterpret_cast<const f32x4*>(&fac[par][j][4]);
            const float dh = hi[2] + dhs[cur][j];
            const float dct = dc + dh * lo[0];
            dc = dct * hi[1];
            dgs[j] = dct * lo[1]; dgs[H + j] = dct * lo[2];
            dgs[2 * H + j] = dct * lo[3]; dgs[3 * H + j] = dh * hi[0];
        }
        REC_STAMP(0, step, 1);
        lds_barrier();
        REC_STAMP(0, step, 2); REC_STAMP(512, step, 2);
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const f32x4 d = *reinterpret_cast<const f32x4*>(&dgs[i * 4 * LPU + rc * 4]);
            a0 = fmaf(w[i][0], d[0], a0); a1 = fmaf(w[i][1], d[1], a1);
            a2 = fmaf(w[i][2], d[2], a2); a3 = fmaf(w[i][3], d[3], a3);
        }
        float acc = (a0 + a1) + (a2 + a3);
        acc = row_sum<(LPU < 16 ? LPU : 16)>(acc);
        if (LPU > 16) acc += __shfl_xor(acc, 16);
        REC_STAMP(0, step, 3); REC_STAMP(512, step, 3);
        if (rc == 0) {
            if (G > 1) {
                u64* gp64 = xg + (step & 1) * H + k;
                if (l2x) publish_granule_l2(gp64, ebase + (unsigned)step + 1u, acc); else publish_granule(gp64, ebase + (unsigned)step + 1u, acc);
            }
            dhs[cur ^ 1][k] = acc;
        }
        if (fact && step + 1 < T) {                      // next step's factors, then fetch the stash of the step after it
            prepare(par ^ 1);
            load_step(step + 2);
        }
        if (tid >= SB && tid < SB + H) {                 // this workgroup's dG rows of step t -> memory (off the chain)
#pragma unroll
            for (int i = 0; i < NBS; ++i) {
                const int e = tid - SB + i * H;
                if (e < 4 * UW) {
                    const int g = e / UW, j = member * UW + e % UW;
                    const float v = dgs[g * H + j];
                    dgb[(long)t * 4 * H + g * H + j] = v;
                    bsum[i] += v;
                }
            }
        }
        if (G > 1) {
            const int u = tid - PB;
            if (tid >= PB && u < H && u / UW != member)
                dhs[cur ^ 1][u] = poll_granule(xg + (step & 1) * H + u, ebase + (unsigned)step + 1u, err);
        }
        REC_STAMP(0, step, 4); REC_STAMP(512, step, 4);
        lds_barrier();
        REC_STAMP(0, step, 5); REC_STAMP(512, step, 5);
        cur ^= 1;
    }
    float* db = dir ? db_r : db_f;
    if (db != nullptr && tid >= SB && tid < SB + H) {
#pragma unroll
        for (int i = 0; i < NBS; ++i) {
            const int e = tid - SB + i * H;
            if (e < 4 * UW) {
                const int g = e / UW, j = member * UW + e % UW;
                atomicAdd(db + g * H + j, bsum[i]);              // b_ih
                atomicAdd(db + 4 * H + g * H + j, bsum[i]);      // b_hh receives the same gradient
            }
        }
    }
}


// ------------------------------------------------------------------------------------------------
// Backward (BPTT), register-resident W_hh^T, NB utterances per group.  Per round: every thread applies the cell
// backward of its (utterance, unit) items (factors of the step were prepared in registers during the previous round),
// barrier, NB mat-vecs dG_t W_hh against the resident weights, then — off the chain — the factors of the next step and
// the gather of the other members' dh.  Same residency guarantee as rec_fwd_multi.
// ------------------------------------------------------------------------------------------------
template <int H, int NB>
__global__ __launch_bounds__(REC_THREADS) void rec_bwd_multi(const float* __restrict__ dout, const float* __restrict__ gates,
                                                             const float* __restrict__ cbuf, const float* __restrict__ w_hh_t,
                                                             float* __restrict__ dgates, int B, int T, u64* xbuf,
                                                             unsigned* err, int b0, int Bc, float* __restrict__ db_f,
                                                             float* __restrict__ db_r, unsigned ebase) {
    constexpr int LPU = H / 16;
    constexpr int UW = REC_THREADS / LPU;
    constexpr int G = H / UW;
    constexpr int NI = NB * H;                                   // (utterance, unit) items per round
    constexpr int CI = (NI + REC_THREADS - 1) / REC_THREADS;     // items per thread
    extern __shared__ __attribute__((aligned(16))) float smem_rb[];
    float (*dhs)[NB][H] = reinterpret_cast<float (*)[NB][H]>(smem_rb);                       // [2][NB][H]
    float (*dgs)[4 * H] = reinterpret_cast<float (*)[4 * H]>(smem_rb + 2 * NB * H);          // [NB][4H]
    __shared__ int xcd_flag;

    const int nblk = (Bc + NB - 1) / NB;
    int group, member;
    decode_block<G>(2 * nblk, group, member);
    const bool l2x = same_xcd_group<G>(xbuf + (long)2 * nblk * 2 * NB * H + (long)group * XID_SLOTS, member, err, &xcd_flag, ebase ? ebase : 0xC0DE0001u);
    const int dir = group & 1, blk = group >> 1;
    const int tid = threadIdx.x;
    const int rc = tid % LPU, kl = tid / LPU;
    const int k = member * UW + kl;
    const float* __restrict__ wt = w_hh_t + ((long)dir * H + k) * 4 * H;

    f32x4 w[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) w[i] = *reinterpret_cast<const f32x4*>(wt + i * 4 * LPU + rc * 4);

    for (int i = tid; i < 2 * NB * H; i += REC_THREADS) (&dhs[0][0][0])[i] = 0.f;
    u64* xg = xbuf + (long)group * 2 * NB * H;
    const int nvalid = min(NB, Bc - blk * NB);

    // per-item state: item index it = tid + q*REC_THREADS -> (u, j) = (it / H, it % H)
    float dc[CI], fa[CI][7], p[CI][7], bs[CI][4];
    bool live[CI];
    long seqs[CI];
#pragma unroll
    for (int q = 0; q < CI; ++q) {
        const int it = tid + q * REC_THREADS;
        const int u = it / H;
        live[q] = it < NI && blk * NB + u < Bc;
        seqs[q] = (long)(dir * B + (live[q] ? b0 + blk * NB + u : 0)) * T;
        dc[q] = 0.f;
#pragma unroll
        for (int e = 0; e < 7; ++e) { fa[q][e] = 0.f; p[q][e] = 0.f; }
#pragma unroll
        for (int e = 0; e < 4; ++e) bs[q][e] = 0.f;
    }
    auto load_step = [&](int q, int st) {                 // raw stash of processing step st -> p[q]
        if (!live[q] || st >= T) return;
        const int j = (tid + q * REC_THREADS) % H;
        const int b = b0 + blk * NB + (tid + q * REC_THREADS) / H;
        const int t = dir ? st : T - 1 - st;
        const float* gp = gates + (seqs[q] + t) * 4 * H + j;
        p[q][0] = gp[0]; p[q][1] = gp[H]; p[q][2] = gp[2 * H]; p[q][3] = gp[3 * H];
        p[q][4] = cbuf[(seqs[q] + t) * H + j];
        const int tp = dir ? t + 1 : t - 1;
        p[q][5] = (tp >= 0 && tp < T) ? cbuf[(seqs[q] + tp) * H + j] : 0.f;
        p[q][6] = dout[((long)b * T + t) * 2 * H + dir * H + j];
    };
    auto prepare = [&](int q) {                           // p[q] -> factors fa[q]: beta, a_i, a_f, a_g, a_o, f, dout
        const float pi = p[q][0], pf = p[q][1], pg = p[q][2], po = p[q][3];
        const float tc = tanhf_acc(p[q][4]);
        fa[q][0] = po * (1.f - tc * tc); fa[q][1] = pg * pi * (1.f - pi); fa[q][2] = p[q][5] * pf * (1.f - pf);
        fa[q][3] = pi * (1.f - pg * pg); fa[q][4] = tc * po * (1.f - po); fa[q][5] = pf; fa[q][6] = p[q][6];
    };
#pragma unroll
    for (int q = 0; q < CI; ++q) { load_step(q, 0); prepare(q); load_step(q, 1); }
    __syncthreads();

    int cur = 0;
    for (int step = 0; step < T; ++step) {
        const int t = dir ? step : T - 1 - step;
        const unsigned epoch = ebase + (unsigned)step + 1u;
#pragma unroll
        for (int q = 0; q < CI; ++q) {
            const int it = tid + q * REC_THREADS;
            if (it < NI) {
                const int u = it / H, j = it % H;
                const float dh = fa[q][6] + dhs[cur][u][j];
                const float dct = dc[q] + dh * fa[q][0];
                dc[q] = dct * fa[q][5];
                const float gi = dct * fa[q][1], gf = dct * fa[q][2], gg = dct * fa[q][3], go = dh * fa[q][4];
                dgs[u][j] = gi; dgs[u][H + j] = gf; dgs[u][2 * H + j] = gg; dgs[u][3 * H + j] = go;
                if (live[q] && j / UW == member) {            // this workgroup's share of dG_t -> memory (+ bias-gradient sums)
                    float* dp = dgates + (seqs[q] + t) * 4 * H + j;
                    dp[0] = gi; dp[H] = gf; dp[2 * H] = gg; dp[3 * H] = go;
                    bs[q][0] += gi; bs[q][1] += gf; bs[q][2] += gg; bs[q][3] += go;
                }
            }
        }
        lds_barrier();
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const f32x4 d = *reinterpret_cast<const f32x4*>(&dgs[u][i * 4 * LPU + rc * 4]);
                a0 = fmaf(w[i][0], d[0], a0); a1 = fmaf(w[i][1], d[1], a1);
                a2 = fmaf(w[i][2], d[2], a2); a3 = fmaf(w[i][3], d[3], a3);
            }
            float acc = (a0 + a1) + (a2 + a3);
            acc = row_sum<(LPU < 16 ? LPU : 16)>(acc);
            if (LPU > 16) acc += __shfl_xor(acc, 16);
            if (rc == 0) {
                if (G > 1 && u < nvalid) {
                    u64* gp64 = xg + ((long)(step & 1) * NB + u) * H + k;
                    if (l2x) publish_granule_l2(gp64, epoch, acc); else publish_granule(gp64, epoch, acc);
                }
                dhs[cur ^ 1][u][k] = acc;
            }
        }
        if (step + 1 < T) {
#pragma unroll
            for (int q = 0; q < CI; ++q) { prepare(q); load_step(q, step + 2); }
        }
        if (G > 1) {
            for (int fidx = tid; fidx < nvalid * (H - UW); fidx += REC_THREADS) {
                const int u = fidx / (H - UW), f = fidx % (H - UW);
                const int fu = f < member * UW ? f : f + UW;
                dhs[cur ^ 1][u][fu] = poll_granule(xg + ((long)(step & 1) * NB + u) * H + fu, epoch, err);
            }
        }
        lds_barrier();
        cur ^= 1;
    }
    float* db = dir ? db_r : db_f;
    if (db != nullptr) {
#pragma unroll
        for (int q = 0; q < CI; ++q) {
            const int it = tid + q * REC_THREADS;
            const int j = it % H;
            if (it < NI && live[q] && j / UW == member) {
#pragma unroll
                for (int g = 0; g < 4; ++g) { atomicAdd(db + g * H + j, bs[q][g]); atomicAdd(db + 4 * H + g * H + j, bs[q][g]); }
            }
        }
    }
}

__global__ __launch_bounds__(256) void rec_bwd_generic(const float* __restrict__ dout, const float* __restrict__ gates,
                                                       const float* __restrict__ cbuf, const float* __restrict__ w_hh_t,
                                                       float* __restrict__ dgates, int B, int T, int H) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* dh0 = smem; float* dh1 = smem + H; float* dcs = smem + 2 * H; float* dgs = smem + 3 * H;
    const int dir = blockIdx.x & 1, b = blockIdx.x >> 1;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
    for (int i = tid; i < H; i += blockDim.x) { dh0[i] = 0.f; dcs[i] = 0.f; }
    __syncthreads();
    const long seq = (long)(dir * B + b) * T;
    const float* wt = w_hh_t + (long)dir * H * 4 * H;
    for (int step = 0; step < T; ++step) {
        const int t = dir ? step : T - 1 - step;
        float* dhc = (step & 1) ? dh1 : dh0;
        float* dhn = (step & 1) ? dh0 : dh1;
        for (int j = tid; j < H; j += blockDim.x) {
            const float* gp = gates + (seq + t) * 4 * H + j;
            const float ig = gp[0], fg = gp[H], gg = gp[2 * H], og = gp[3 * H];
            const float c_t = cbuf[(seq + t) * H + j];
            const int tp = dir ? t + 1 : t - 1;
            const float c_prev = (tp >= 0 && tp < T) ? cbuf[(seq + tp) * H + j] : 0.f;
            const float dh = dout[((long)b * T + t) * 2 * H + dir * H + j] + dhc[j];
            const float tc = tanhf_acc(c_t);
            const float dct = dcs[j] + dh * og * (1.f - tc * tc);
            const float dGi = dct * gg * ig * (1.f - ig);
            const float dGf = dct * c_prev * fg * (1.f - fg);
            const float dGg = dct * ig * (1.f - gg * gg);
            const float dGo = dh * tc * og * (1.f - og);
            dcs[j] = dct * fg;
            dgs[j] = dGi; dgs[H + j] = dGf; dgs[2 * H + j] = dGg; dgs[3 * H + j] = dGo;
            float* dp = dgates + (seq + t) * 4 * H + j;
            dp[0] = dGi; dp[H] = dGf; dp[2 * H] = dGg; dp[3 * H] = dGo;
        }
        __syncthreads();
        for (int k = wave; k < H; k += nw) {
            float acc = 0.f;
            for (int r = lane; r < 4 * H; r += 64) acc = fmaf(wt[(long)k * 4 * H + r], dgs[r], acc);
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m);
            if (lane == 0) dhn[k] = acc;
        }
        __syncthreads();
    }
}

// blockIdx.z selects (src, dst) pair 0 or 1: both directions' W_hh in one launch
__global__ void transpose2d_kernel(const float* __restrict__ src0, float* __restrict__ dst0, const float* __restrict__ src1,
                                   float* __restrict__ dst1, int rows, int cols) {
    __shared__ float tile[32][33];
    const float* __restrict__ src = blockIdx.z ? src1 : src0;
    float* __restrict__ dst = blockIdx.z ? dst1 : dst0;
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    for (int i = threadIdx.y; i < 32; i += blockDim.y) {
        const int r = r0 + i, c = c0 + threadIdx.x;
        tile[i][threadIdx.x] = (r < rows && c < cols) ? src[(long)r * cols + c] : 0.f;
    }
    __syncthreads();
    for (int i = threadIdx.y; i < 32; i += blockDim.y) {
        const int c = c0 + i, r = r0 + threadIdx.x;
        if (r < rows && c < cols) dst[(long)c * rows + r] = tile[threadIdx.x][i];
    }
}

int transpose2d(const float* src, float* dst, int rows, int cols, hipStream_t stream, const float* src1, float* dst1) {
    dim3 grid(cdiv(cols, 32), cdiv(rows, 32), src1 ? 2 : 1), block(32, 8);
    hipLaunchKernelGGL(transpose2d_kernel, grid, block, 0, stream, src, dst, src1, dst1, rows, cols);
    LAS_LAUNCH_CHECK();
    return LAS_OK;
}

// granules [group][parity][utterance][unit] + XCC-id slots per group; groups*NB <= 2*(B + 15) for every NB <= 16
// (+ the matrix-pipe forward kernel's hand-off ring at the shapes it serves: the two paths never use the buffer at the same time)
size_t rec_xbuf_bytes(int B, int H) {
    const size_t base = ((size_t)2 * (B + 15) * 2 * H + (size_t)2 * (B + 15) * XID_SLOTS) * sizeof(u64);
    return std::max(base, rec_mfma_xbuf_extra_bytes(B, H));
}

// ---- hand-off scratch of the register-resident recurrences: one per (device, stream), created on first use ------------------------------
// The granules between the CUs of a group carry {epoch, value}; a poller waits for EXACTLY its step's epoch.  On a buffer the caller owns
// (torch.empty: arbitrary contents) every launch needed a fill first — six 5-us launches per training step.  The library's own scratch is
// zeroed once, when it is created, and every launch draws a fresh, monotonically increasing epoch range [base, base + T]: everything an
// earlier launch left behind carries a smaller epoch and can never match.  Kernels of one stream run in order, so one scratch per stream is
// race-free.  No scratch during stream capture (a replayed graph would re-use its epoch range), on allocation failure, or beyond
// RS_MAX streams: those launches fill the caller's buffer as before (ebase = 0).  Wrap-around (2^32 epochs: ~10^6 training steps) re-zeroes.
struct RecScratch { int dev; hipStream_t stream; u64* buf; size_t bytes; unsigned next; };
constexpr int RS_MAX = 16;
static std::mutex g_rs_mu;
static RecScratch g_rs[RS_MAX];
static int g_rs_n = 0;

static u64* rec_scratch(size_t bytes, unsigned epochs, hipStream_t stream, unsigned* ebase) {
    *ebase = 0;
    if (opt_get(OPT_REC_EPOCH_SCRATCH) == 0) return nullptr;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) return nullptr;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lk(g_rs_mu);
    RecScratch* r = nullptr;
    for (int i = 0; i < g_rs_n; ++i)
        if (g_rs[i].dev == dev && g_rs[i].stream == stream) { r = &g_rs[i]; break; }
    if (r == nullptr) {
        if (g_rs_n >= RS_MAX) return nullptr;
        r = &g_rs[g_rs_n];
        *r = RecScratch{dev, stream, nullptr, 0, 1u};
        ++g_rs_n;
    }
    if (r->bytes < bytes) {      // (re)allocate: rare (first use, a larger batch); hipFree waits for the stream's earlier launches
        u64* nb = nullptr;
        const size_t want = std::max(bytes, (size_t)1 << 20);
        if (hipMalloc(&nb, want) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        if (hipMemsetAsync(nb, 0, want, stream) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(nb); return nullptr; }
        if (r->buf) (void)hipFree(r->buf);
        r->buf = nb; r->bytes = want; r->next = 1u;
    }
    const unsigned seed = (unsigned)opt_get(OPT_REC_EPOCH_SEED);      // test hook: while set, every launch starts at least there (towards the wrap-around)
    if (seed > r->next) r->next = seed;
    if (r->next > 0xFFFF0000u - epochs - 2u) {
        if (hipMemsetAsync(r->buf, 0, r->bytes, stream) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        r->next = 1u;
    }
    *ebase = r->next;
    r->next += epochs + 2u;
    return r->buf;
}

static bool fast_h(int H) { return H == 128 || H == 256 || H == 512; }

static int device_cus() {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    return cus;
}

// Residency plan of the persistent recurrences: every workgroup of a launch must be resident at once (the members of a
// group spin on each other), one workgroup per CU (1024 threads at >= 100 VGPRs).  Utterances per group (NB) are raised
// until 2*ceil(Bc/NB)*G workgroups fit the device's CU count; batches beyond NB_max per group are stepped in several
// launches of `chunk` utterances.  nb == 0: the device cannot hold even one group -> generic kernels.
struct RecPlan { int nb, chunk; };
static RecPlan rec_plan(int B, int G, int nb_min, int nb_max, int cus) {
    const int blocks = cus / (2 * G);              // utterance blocks per direction that fit
    if (blocks < 1) return {0, 0};
    for (int nb = std::max(1, nb_min); nb <= nb_max; nb *= 2)
        if ((B + nb - 1) / nb <= blocks) return {nb, B};
    return {nb_max, blocks * nb_max};
}

int pblstm_rec_fwd(float* gates, const float* w_hh_f, const float* w_hh_r, float* out, float* cbuf, float* hprev, int B,
                   int T, int H, int stash, u64* xbuf, unsigned* err, int force_generic, hipStream_t stream) {
    LAS_REQUIRE(B > 0 && T > 0 && H > 0, "rec dims");
    LAS_REQUIRE(!stash || (cbuf && hprev), "stash buffers");
    if (!force_generic && err && xbuf && rec_fwd_mfma_eligible(B, H)) {      // large batches: 16 utterances per group on the matrix pipe
        const int rc = rec_fwd_mfma(gates, w_hh_f, w_hh_r, out, cbuf, hprev, B, T, H, stash, xbuf, err, stream);
        if (rc != LAS_ERR_UNSUPPORTED) return rc;                    // residency check failed: the kernels below run instead
    }
    const int ngroups = 2 * B;
    // LAS_REC_AGENT_HANDOFF=1 forces the placement-independent agent-scope hand-off even when a group shares an XCD (A/B tests)
    const int dbg = (int)opt_get(OPT_REC_AGENT_HANDOFF);
    const int uw_env = (int)opt_get(OPT_REC_UW);
    const int nb_env = (int)opt_get(OPT_REC_NB);          // force the utterances per group (A/B tests)
    const bool pipe_on = opt_get(OPT_REC_PIPE) != 0;  // 0: lock-step multi-utterance kernels instead of the pipelined halves
    // UW: hidden units per workgroup (measured best on MI355X: one CU holds 256 KB of W_hh)
    const int uw = uw_env > 0 ? uw_env : (H == 128 ? 128 : (H == 256 ? 64 : 32));
    RecPlan plan = {0, 0};
    if (fast_h(H) && !force_generic && H % uw == 0) {
        const int nb_max = H == 128 ? 8 : (H == 256 ? 8 : 4);
        plan = rec_plan(B, H / uw, std::min(nb_env, nb_max), nb_max, device_cus());
        if (plan.nb > 1 && uw != (H == 128 ? 128 : (H == 256 ? 64 : 32))) plan.nb = 0;      // multi kernels exist for the default UW only
    }
    bool fits = true;       // occupancy calculator admits one workgroup of the chosen kernel per CU
    if (plan.nb > 0) {
        LAS_REQUIRE(xbuf && err, "hand-off buffers");
        const int G = H / uw;
        for (int b0 = 0; b0 < B; b0 += plan.chunk) {
            const int Bc = std::min(plan.chunk, B - b0);
            const int nblk = (Bc + plan.nb - 1) / plan.nb;
            unsigned ebase = 0;
            u64* xb = G > 1 ? rec_scratch(rec_xbuf_bytes(B, H), (unsigned)T, stream, &ebase) : nullptr;
            if (xb == nullptr) {
                xb = xbuf; ebase = 0;
                if (G > 1) LAS_HIP_CHECK(hipMemsetAsync(xbuf, 0, rec_xbuf_bytes(B, H), stream));      // (G == 1: nothing is handed between CUs)
            }
            dim3 grid(2 * nblk * G), block(uw * (H / 16));
            bool launched = false;
#define TRY_FWD(HH, UWV)                                                                                                  \
    if (!launched && plan.nb == 1 && H == HH && uw == UWV) {                                                             \
        launched = true;                                                                                                  \
        if (b0 != 0 || Bc != B) return fail(LAS_ERR_UNSUPPORTED, "chunked launch needs the multi kernel%s", "");          \
        fits = persistent_launch_fits(rec_fwd_fast<HH, UWV, true>, block.x, 0, grid.x);                                   \
        if (!fits) break;                                                                                                 \
        if (stash) hipLaunchKernelGGL((rec_fwd_fast<HH, UWV, true>), grid, block, 0, stream, gates, w_hh_f, w_hh_r, out,  \
                                      cbuf, hprev, B, T, xb, err, dbg, ebase);                                            \
        else hipLaunchKernelGGL((rec_fwd_fast<HH, UWV, false>), grid, block, 0, stream, gates, w_hh_f, w_hh_r, out, cbuf, \
                                hprev, B, T, xb, err, dbg, ebase);                                                        \
    }
#define TRY_FWD_M(HH, UWV, NBV)                                                                                           \
    if (!launched && plan.nb == NBV && H == HH && uw == UWV) {                                                           \
        launched = true;                                                                                                  \
        constexpr bool PP = (HH / UWV) > 1;      /* pipelined halves where there is an inter-CU hand-off to hide */       \
        if (PP && pipe_on) {                                                                                              \
            fits = persistent_launch_fits(rec_fwd_multi<HH, UWV, NBV, true, PP>, block.x, 0, grid.x);                     \
            if (!fits) break;                                                                                             \
            if (stash) hipLaunchKernelGGL((rec_fwd_multi<HH, UWV, NBV, true, PP>), grid, block, 0, stream, gates, w_hh_f, \
                                          w_hh_r, out, cbuf, hprev, B, T, xb, err, dbg, b0, Bc, ebase);                   \
            else hipLaunchKernelGGL((rec_fwd_multi<HH, UWV, NBV, false, PP>), grid, block, 0, stream, gates, w_hh_f,      \
                                    w_hh_r, out, cbuf, hprev, B, T, xb, err, dbg, b0, Bc, ebase);                         \
        } else {                                                                                                          \
            fits = persistent_launch_fits(rec_fwd_multi<HH, UWV, NBV, true>, block.x, 0, grid.x);                         \
            if (!fits) break;                                                                                             \
            if (stash) hipLaunchKernelGGL((rec_fwd_multi<HH, UWV, NBV, true>), grid, block, 0, stream, gates, w_hh_f,     \
                                          w_hh_r, out, cbuf, hprev, B, T, xb, err, dbg, b0, Bc, ebase);                   \
            else hipLaunchKernelGGL((rec_fwd_multi<HH, UWV, NBV, false>), grid, block, 0, stream, gates, w_hh_f, w_hh_r,  \
                                    out, cbuf, hprev, B, T, xb, err, dbg, b0, Bc, ebase);                                 \
        }                                                                                                                 \
    }
            TRY_FWD(128, 128) TRY_FWD(128, 64) TRY_FWD(128, 32) TRY_FWD(128, 16)
            TRY_FWD(256, 64) TRY_FWD(256, 32) TRY_FWD(256, 16)
            TRY_FWD(512, 32) TRY_FWD(512, 16)
            TRY_FWD_M(128, 128, 2) TRY_FWD_M(128, 128, 4) TRY_FWD_M(128, 128, 8)
            TRY_FWD_M(256, 64, 2) TRY_FWD_M(256, 64, 4) TRY_FWD_M(256, 64, 8)
            TRY_FWD_M(512, 32, 2) TRY_FWD_M(512, 32, 4)
#undef TRY_FWD
#undef TRY_FWD_M
            if (!launched) return fail(LAS_ERR_UNSUPPORTED, "no recurrence kernel for %s H=%ld uw=%ld", "", (long)H, (long)uw);
            LAS_LAUNCH_CHECK();
            path_note(PATH_REC_FWD, plan.nb == 1 ? "rec_fwd_fast" : "rec_fwd_multi");
        }
    }
    if (plan.nb == 0 || !fits) {
        const size_t smem = sizeof(float) * 3 * H;
        if (stash) hipLaunchKernelGGL((rec_fwd_generic<true>), dim3(ngroups), dim3(256), smem, stream, gates, w_hh_f, w_hh_r,
                                      out, cbuf, hprev, B, T, H);
        else hipLaunchKernelGGL((rec_fwd_generic<false>), dim3(ngroups), dim3(256), smem, stream, gates, w_hh_f, w_hh_r, out,
                                cbuf, hprev, B, T, H);
        path_note(PATH_REC_FWD, "rec_fwd_generic");
    }
    LAS_LAUNCH_CHECK();
    return LAS_OK;
}

template <int H, int NB>
static int launch_bwd_multi(const float* dout, const float* gates, const float* cbuf, const float* w_hh_t, float* dgates, int B, int T,
                            u64* xbuf, unsigned* err, int b0, int Bc, int grid, float* db_f, float* db_r, hipStream_t stream, unsigned ebase) {
    const size_t smem = sizeof(float) * ((size_t)2 * NB * H + (size_t)NB * 4 * H);
    LAS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&rec_bwd_multi<H, NB>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    if (!persistent_launch_fits(rec_bwd_multi<H, NB>, REC_THREADS, smem, grid)) return fail(LAS_ERR_UNSUPPORTED, "backward recurrence: %s%ld workgroups cannot all be resident", "", (long)grid);
    hipLaunchKernelGGL((rec_bwd_multi<H, NB>), dim3(grid), dim3(REC_THREADS), smem, stream, dout, gates, cbuf, w_hh_t, dgates, B, T, xbuf,
                       err, b0, Bc, db_f, db_r, ebase);
    LAS_LAUNCH_CHECK();
    return LAS_OK;
}

// XCDs the one-utterance-per-group backward recurrence of this batch can be confined to (2 or 4 of the 8; 0: not at all): a device of
// 8 XCDs x 32 CUs, G > 1 workgroups per group, whole groups per XCD, at most one workgroup per CU
int rec_confine_xcds(int B, int H) {
    if (!fast_h(H) || device_cus() != 256) return 0;
    const int G = H * H / 16384;
    if (G < 2) return 0;
    // (2 XCDs only: measured, tools/ab_step_option.py DEFER_DW — at 4 of 8 XCDs (B = 16 at H = 256) the deferred groups slow the critical-path
    // launches around the short recurrences more than they save: 5.23 against 5.19 ms per step)
    for (int nx = 2; nx <= 2; nx *= 2)
        if ((2 * B) % nx == 0 && 2 * B * G <= 32 * nx) return nx;
    return 0;
}

int pblstm_rec_bwd(const float* dout, const float* gates, const float* cbuf, const float* w_hh_t, float* dgates, int B,
                   int T, int H, u64* xbuf, unsigned* err, int force_generic, hipStream_t stream, float* db_f, float* db_r,
                   int* db_done, int confine_nx) {
    LAS_REQUIRE(B > 0 && T > 0 && H > 0, "rec dims");
    if (db_done) *db_done = 0;
    if (!force_generic && err && xbuf && rec_bwd_mfma_eligible(B, H)) {      // large batches: 16 utterances per group on the matrix pipe
        const int rc = rec_bwd_mfma(dout, gates, cbuf, w_hh_t, dgates, B, T, H, xbuf, err, db_f, db_r, stream);
        if (rc != LAS_ERR_UNSUPPORTED) {
            if (rc == LAS_OK && db_done && db_f && db_r) *db_done = 1;
            return rc;
        }
    }
    const int ngroups = 2 * B;
    const int nb_env = (int)opt_get(OPT_REC_NB);
    RecPlan plan = {0, 0};
    const int G = H * H / 16384 > 0 ? H * H / 16384 : 1;
    if (fast_h(H) && !force_generic) {
        const int nb_max = H == 128 ? 8 : (H == 256 ? 8 : 4);
        plan = rec_plan(B, G, std::min(nb_env, nb_max), nb_max, device_cus());
    }
    bool fits = true;
    if (plan.nb > 0) {
        LAS_REQUIRE(xbuf && err, "hand-off buffers");
        for (int b0 = 0; b0 < B && fits; b0 += plan.chunk) {
            const int Bc = std::min(plan.chunk, B - b0);
            const int nblk = (Bc + plan.nb - 1) / plan.nb;
            int grid = 2 * nblk * G;
            const int nx = (plan.nb == 1 && confine_nx > 0 && confine_nx == rec_confine_xcds(B, H)) ? confine_nx : 0;
            if (nx > 0) grid = grid * 8 / nx;
            unsigned ebase = 0;
            u64* xb = G > 1 ? rec_scratch(rec_xbuf_bytes(B, H), (unsigned)T, stream, &ebase) : nullptr;
            if (xb == nullptr) {
                xb = xbuf; ebase = 0;
                if (G > 1) LAS_HIP_CHECK(hipMemsetAsync(xbuf, 0, rec_xbuf_bytes(B, H), stream));
            }
            if (plan.nb == 1) {
                if (b0 != 0 || Bc != B) return fail(LAS_ERR_UNSUPPORTED, "chunked launch needs the multi kernel%s", "");
                fits = H == 128 ? persistent_launch_fits(rec_bwd_fast<128>, REC_THREADS, 0, grid)
                     : H == 256 ? persistent_launch_fits(rec_bwd_fast<256>, REC_THREADS, 0, grid)
                                : persistent_launch_fits(rec_bwd_fast<512>, REC_THREADS, 0, grid);
                if (!fits) break;
                if (H == 128) hipLaunchKernelGGL((rec_bwd_fast<128>), dim3(grid), dim3(REC_THREADS), 0, stream, dout, gates, cbuf, w_hh_t, dgates, B, T, xb, err, db_f, db_r, ebase, nx, nx > 0 ? xcd_probe_ptr() : nullptr);
                else if (H == 256) hipLaunchKernelGGL((rec_bwd_fast<256>), dim3(grid), dim3(REC_THREADS), 0, stream, dout, gates, cbuf, w_hh_t, dgates, B, T, xb, err, db_f, db_r, ebase, nx, nx > 0 ? xcd_probe_ptr() : nullptr);
                else hipLaunchKernelGGL((rec_bwd_fast<512>), dim3(grid), dim3(REC_THREADS), 0, stream, dout, gates, cbuf, w_hh_t, dgates, B, T, xb, err, db_f, db_r, ebase, nx, nx > 0 ? xcd_probe_ptr() : nullptr);
                LAS_LAUNCH_CHECK();
                path_note(PATH_REC_BWD, "rec_bwd_fast");
                if (db_done && db_f && db_r) *db_done = 1;
                continue;
            }
#define TRY_BWD_M(HH, NBV) if (H == HH && plan.nb == NBV) {                                                                          \
        const int rc = launch_bwd_multi<HH, NBV>(dout, gates, cbuf, w_hh_t, dgates, B, T, xb, err, b0, Bc, grid, db_f, db_r, stream, ebase); \
        if (rc == LAS_ERR_UNSUPPORTED) { fits = false; break; }                                                                       \
        LAS_TRY(rc); path_note(PATH_REC_BWD, "rec_bwd_multi"); if (db_done && db_f && db_r) *db_done = 1; continue; }
            TRY_BWD_M(128, 2) TRY_BWD_M(128, 4) TRY_BWD_M(128, 8)
            TRY_BWD_M(256, 2) TRY_BWD_M(256, 4) TRY_BWD_M(256, 8)
            TRY_BWD_M(512, 2) TRY_BWD_M(512, 4)
#undef TRY_BWD_M
            return fail(LAS_ERR_UNSUPPORTED, "no backward recurrence kernel for %s H=%ld nb=%ld", "", (long)H, (long)plan.nb);
        }
    }
    if (plan.nb == 0 || !fits) {
        if (db_done) *db_done = 0;          // the residency check fails before the first launch: no partial sums were added
        const size_t smem = sizeof(float) * 7 * H;
        hipLaunchKernelGGL(rec_bwd_generic, dim3(ngroups), dim3(256), smem, stream, dout, gates, cbuf, w_hh_t, dgates, B, T, H);
        path_note(PATH_REC_BWD, "rec_bwd_generic");
    }
    LAS_LAUNCH_CHECK();
    return LAS_OK;
}

#ifdef LAS_REC_TRACE
void rec_set_trace(unsigned long long* dev_buf) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_rec_trace), &dev_buf, sizeof(dev_buf)); }
#endif

}  // namespace las
