// One-launch decode loop, forward AND backward, for the reference's SHIPPED model size (config/librispeech-config.yaml:16-34:
// Listener 512x3 -> 1024 features, Speller 1024x2, attention MLP 64, batch 16, decode_mode 1) on gfx950.  DESIGN.md 4.3a.
//
// Replaces, for that size, the 3 + 3 launches per decode step of speller.hip (reference model/las_model.py:178-184, 205-236, 275-297 and
// their autograd), each of which re-streams 34 MB of LSTM weights from the L2 / Infinity Cache.  Here the four fp32 matrices of the two
// cells (67 MB) stay in the REGISTERS of 256 workgroups x 512 threads (128 VGPRs per lane) for all U steps.
//
// Forward (speller_big_fwd_kernel<GREEDY, LONG>):
//   * workgroup w owns 4 hidden units (16 gate rows) of BOTH layers; its 8 waves split K, a lane holds 8 k-blocks of each of
//     [W_ctx | W_hh0 | W_ih1 | W_hh1] as v_mfma_f32_16x16x4_f32 B operands, the 16 utterances are the M dimension;
//   * the attention of utterance b is split over 16 workgroups BY FEATURE COLUMNS (64 columns x all T' frames and all keys of b in LDS): every
//     one computes all energies and the whole softmax itself, so its 64 context columns are final (no exchange of partial contexts); the 16
//     workgroups of an utterance sit on one XCD.  LONG (T' > 256): the keys no longer fit, the 16 workgroups split the energies by frames
//     and exchange the energy row (one more hand-off);
//   * the query needs no hop of its own: with the top cell's h the owner publishes its K-slice of phi (64 x 4 weights), an attention
//     workgroup adds the 256 slices of its utterance;
//   * GREEDY (decode_mode 1): the column blocks add their part of the logits W_c [h1 | ctx], workgroup (b, 0) takes log-softmax / arg-max
//     and publishes the symbol, the cells add W_y[:, symbol].
//   Per step the chain is three hand-offs (ctx -> bottom cell -> h0 -> top cell -> h1 + query parts -> attention -> ctx): the data itself in
//   per-step slabs pre-filled with the sentinel 0xFFFFFFFF (persist_common.h) plus one flag dword per producer (8 copies, one per consumer
//   XCD); operand loads go through the L2 once the flags are in, every piece is still checked for the sentinel.  The recurrent halves
//   W_hh . h of the next step are multiplied inside the hand-off windows.  The stash (h_all, c_all, gates_all, q_all, ctx_all, att, y_all) is
//   the per-step kernels'.
// Backward (speller_big_bwd_kernel): weights held column-wise (workgroup = 16 columns x 4096 rows of ONE matrix), the cell backward fused
//   behind the products, attention backward sliced by time; five hand-offs per step.  See the comment in front of it.
#include "las_common.h"
#include "las_kernels.h"
#include "options.h"
#include "persist_common.h"

namespace las {
namespace {

constexpr int BG_THREADS = 512, BG_NW = 8, BG_HS = 1024, BG_M = 64, BG_WGS = 256, BG_NB = 16;
constexpr int BG_KLD = BG_M + 4;         // LDS row stride of the keys
constexpr int BG_FLW = 3 * 8 * 256;      // flag words per step of the forward kernel: 3 hand-offs x 8 copies x 256 producers
constexpr int BG_MAXTP = 512;            // encoder frames (one thread each).  Up to 256 the features AND all keys of the utterance stay in LDS (528 B per frame);
                                         // above, the 16 workgroups of an utterance split the energies by frames and exchange them (LONG instantiations)
constexpr int BG_SHORT_TP = 256;
constexpr float BG_LOG2E = 1.4426950408889634f;
constexpr float BG_NEG = -3.0e38f;

struct BigArgs {
    const float* w0p; long ldw0; int Vp;                 // [W_y | 0 | W_ctx] shadow of W_ih0
    const float* w_hh0; const float* w_ih1; const float* w_hh1;
    const float* b_ih0; const float* b_hh0; const float* b_ih1; const float* b_hh1;
    const float* w_phi; const float* b_phi;
    const float* feat; const float* keys; const float* yw;
    float* ctx_all; float* h_all; float* c_all; float* gates_all; float* q_all; float* att;
    float* hx; float* qp; unsigned* flags;
    // free-running greedy decode (mode 1: the arg-max symbol is fed back, reference las_model.py:223-227): character distribution inside the loop
    float* eg; int FRL;                                  // T' > 256: the utterances' energy rows (U x B x 512), frames per workgroup
    int mode; int V; const float* w_c; const float* b_c; float* logp; int* argmax; float* y_all; float* lgp; float* ysym;
    int B, Tp, U, relu;
    int tune;                                            // poll pacing (option SPELLER_BIG_TUNE)
    unsigned* err;
    u64* trace;
};

__device__ __forceinline__ void st2_agent(float* p, float x, float y) {
    ps_f32x2 b;
    b[0] = __uint_as_float(pub_bits(x)); b[1] = __uint_as_float(pub_bits(y));
    asm volatile("global_store_dwordx2 %0, %1, off sc1\n\ts_nop 2" : : "v"(p), "v"(b) : "memory");
}
__device__ __forceinline__ unsigned ld1_agent(const float* p) {
    return __hip_atomic_load(reinterpret_cast<const unsigned*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// eight agent-scope 16-byte loads of one lane, 64 bytes apart (8 k-blocks of one row), in flight together
__device__ __forceinline__ void ld4x8_row(const float* p, f32x4 (&v)[8]) {
    asm volatile(
        "global_load_dwordx4 %0, %8, off sc1\n\t"
        "global_load_dwordx4 %1, %8, off offset:64 sc1\n\t"
        "global_load_dwordx4 %2, %8, off offset:128 sc1\n\t"
        "global_load_dwordx4 %3, %8, off offset:192 sc1\n\t"
        "global_load_dwordx4 %4, %8, off offset:256 sc1\n\t"
        "global_load_dwordx4 %5, %8, off offset:320 sc1\n\t"
        "global_load_dwordx4 %6, %8, off offset:384 sc1\n\t"
        "global_load_dwordx4 %7, %8, off offset:448 sc1\n\t"
        "s_waitcnt vmcnt(0)"
        : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
        : "v"(p)
        : "memory");
}
// the same through the L2 (no sc1): after the canaries of ALL producers are in, the 32 workgroups of an XCD share one fetch of the slab
// instead of 32 trips across the fabric (16 MB per hand-off chip-wide).  A line that was fetched too early still shows the sentinel
// and sends its readers to the agent-scope form above.
__device__ __forceinline__ void ld4x8_row_l2(const float* p, f32x4 (&v)[8]) {
    asm volatile(
        "global_load_dwordx4 %0, %8, off\n\t"
        "global_load_dwordx4 %1, %8, off offset:64\n\t"
        "global_load_dwordx4 %2, %8, off offset:128\n\t"
        "global_load_dwordx4 %3, %8, off offset:192\n\t"
        "global_load_dwordx4 %4, %8, off offset:256\n\t"
        "global_load_dwordx4 %5, %8, off offset:320\n\t"
        "global_load_dwordx4 %6, %8, off offset:384\n\t"
        "global_load_dwordx4 %7, %8, off offset:448\n\t"
        "s_waitcnt vmcnt(0)"
        : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
        : "v"(p)
        : "memory");
}
// eight k-blocks of a producer-major h slab [256 producers][16 utterances][4 units] (or of a context slab [64 k-blocks][16][16]): 1 KB
// apart (the 13-bit offset field ends at 4095)
template <bool L2>
__device__ __forceinline__ void ld4x8_kb(const float* p, f32x4 (&v)[8]) {
    const float* p2 = p + 1024;
    if (L2) {
        asm volatile(
            "global_load_dwordx4 %0, %8, off\n\t"
            "global_load_dwordx4 %1, %8, off offset:1024\n\t"
            "global_load_dwordx4 %2, %8, off offset:2048\n\t"
            "global_load_dwordx4 %3, %8, off offset:3072\n\t"
            "global_load_dwordx4 %4, %9, off\n\t"
            "global_load_dwordx4 %5, %9, off offset:1024\n\t"
            "global_load_dwordx4 %6, %9, off offset:2048\n\t"
            "global_load_dwordx4 %7, %9, off offset:3072\n\t"
            "s_waitcnt vmcnt(0)"
            : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
            : "v"(p), "v"(p2)
            : "memory");
        return;
    }
    asm volatile(
        "global_load_dwordx4 %0, %8, off sc1\n\t"
        "global_load_dwordx4 %1, %8, off offset:1024 sc1\n\t"
        "global_load_dwordx4 %2, %8, off offset:2048 sc1\n\t"
        "global_load_dwordx4 %3, %8, off offset:3072 sc1\n\t"
        "global_load_dwordx4 %4, %9, off sc1\n\t"
        "global_load_dwordx4 %5, %9, off offset:1024 sc1\n\t"
        "global_load_dwordx4 %6, %9, off offset:2048 sc1\n\t"
        "global_load_dwordx4 %7, %9, off offset:3072 sc1\n\t"
        "s_waitcnt vmcnt(0)"
        : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
        : "v"(p), "v"(p2)
        : "memory");
}
// the same loads WITHOUT the wait (software pipelining: the matrix pipe works on one chunk while the next is in flight); kb_wait() ends the
// flight.  Between the two calls nothing may touch the registers (they are outputs of the first block only so that the allocator keeps them).
__device__ __forceinline__ void kb_issue_l2(const float* p, f32x4 (&v)[8]) {
    const float* p2 = p + 1024;
    asm volatile(
        "global_load_dwordx4 %0, %8, off\n\t"
        "global_load_dwordx4 %1, %8, off offset:1024\n\t"
        "global_load_dwordx4 %2, %8, off offset:2048\n\t"
        "global_load_dwordx4 %3, %8, off offset:3072\n\t"
        "global_load_dwordx4 %4, %9, off\n\t"
        "global_load_dwordx4 %5, %9, off offset:1024\n\t"
        "global_load_dwordx4 %6, %9, off offset:2048\n\t"
        "global_load_dwordx4 %7, %9, off offset:3072"
        : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
        : "v"(p), "v"(p2)
        : "memory");
}
__device__ __forceinline__ void kb_wait(f32x4 (&v)[8]) {
    asm volatile("s_waitcnt vmcnt(0)"
                 : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7])
                 :
                 : "memory");
}
// ... at an arbitrary byte stride (the 256 query slices of an utterance)
template <bool L2>
__device__ __forceinline__ void ld4x8_strided(const float* p, long stride_floats, f32x4 (&v)[8]) {
    const float* p1 = p + stride_floats; const float* p2 = p1 + stride_floats; const float* p3 = p2 + stride_floats;
    const float* p4 = p3 + stride_floats; const float* p5 = p4 + stride_floats; const float* p6 = p5 + stride_floats;
    const float* p7 = p6 + stride_floats;
    if (L2) {
        asm volatile(
            "global_load_dwordx4 %0, %8, off\n\t"
            "global_load_dwordx4 %1, %9, off\n\t"
            "global_load_dwordx4 %2, %10, off\n\t"
            "global_load_dwordx4 %3, %11, off\n\t"
            "global_load_dwordx4 %4, %12, off\n\t"
            "global_load_dwordx4 %5, %13, off\n\t"
            "global_load_dwordx4 %6, %14, off\n\t"
            "global_load_dwordx4 %7, %15, off\n\t"
            "s_waitcnt vmcnt(0)"
            : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
            : "v"(p), "v"(p1), "v"(p2), "v"(p3), "v"(p4), "v"(p5), "v"(p6), "v"(p7)
            : "memory");
        return;
    }
    asm volatile(
        "global_load_dwordx4 %0, %8, off sc1\n\t"
        "global_load_dwordx4 %1, %9, off sc1\n\t"
        "global_load_dwordx4 %2, %10, off sc1\n\t"
        "global_load_dwordx4 %3, %11, off sc1\n\t"
        "global_load_dwordx4 %4, %12, off sc1\n\t"
        "global_load_dwordx4 %5, %13, off sc1\n\t"
        "global_load_dwordx4 %6, %14, off sc1\n\t"
        "global_load_dwordx4 %7, %15, off sc1\n\t"
        "s_waitcnt vmcnt(0)"
        : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
        : "v"(p), "v"(p1), "v"(p2), "v"(p3), "v"(p4), "v"(p5), "v"(p6), "v"(p7)
        : "memory");
}
__device__ __forceinline__ bool any_sentinel8(const f32x4 (&v)[8]) {
    bool bad = false;
#pragma unroll
    for (int i = 0; i < 8; ++i) bad |= has_sentinel(v[i]);
    return bad;
}
// (two accumulation chains: a wave's dependent v_mfma_f32_16x16x4_f32 issue ~100 clocks apart, so one chain per wave and two waves per
// SIMD leave the matrix pipe at ~60 %)
__device__ __forceinline__ f32x4 seg_mfma(const f32x4 (&ax)[8], const f32x4 (&w)[8], f32x4 acc) {
    f32x4 t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 8; i += 2)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ax[i][e], w[i][e], acc, 0, 0, 0);
            t = __builtin_amdgcn_mfma_f32_16x16x4f32(ax[i + 1][e], w[i + 1][e], t, 0, 0, 0);
        }
    return acc + t;
}
__device__ __forceinline__ float quad_bcast(float v, int u) {      // value of lane (quad base + u), u a compile-time constant at the call sites
    switch (u) {
        case 0: return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x00, 0xF, 0xF, true));
        case 1: return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x55, 0xF, 0xF, true));
        case 2: return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xAA, 0xF, 0xF, true));
        default: return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xFF, 0xF, 0xF, true));
    }
}
__device__ __forceinline__ float gmax16(float v) {
    v = fmaxf(v, dpp_f(v, 0)); v = fmaxf(v, dpp_f(v, 1)); v = fmaxf(v, dpp_f(v, 2)); v = fmaxf(v, dpp_f(v, 3));
    return v;
}

// per-workgroup stamps of step 10 (skew of the hand-offs over the chip): slots 1024 + 4 wg + k
#define BG_WSTAMP(k) do { if (a.trace && s == 10 && tid == 0) a.trace[1024 + wg * 8 + (k)] = wall_clock64(); } while (0)
#define BG_STAMP(slot) do { if (a.trace && wg == 0 && tid == 0 && s < 64) a.trace[s * 16 + (slot)] = wall_clock64(); } while (0)

// one flag dword per producer and hand-off, four per lane: wave 0 watches the 256 (or 16 B) flags of a slab with ONE 1 KB request per poll.
// A flag is stored right behind its data by the same wave; it is a HINT (the stores may land in either order): every consumer still
// checks the data it loads for the sentinel.
__device__ __forceinline__ void sleep_units(int n) {      // n x 64 clocks (s_sleep takes an immediate)
    for (; n >= 8; n -= 8) __builtin_amdgcn_s_sleep(8);
    for (; n > 0; --n) __builtin_amdgcn_s_sleep(1);
}
__device__ __forceinline__ bool flags_wait(const unsigned* fl, bool active, unsigned* err, unsigned code, int first, int gap, int ncomp = 4) {
    unsigned spins = 0;
    sleep_units(first);
    for (;;) {
        const f32x4 v = ld4_agent(reinterpret_cast<const float*>(fl));
        const bool pending = (ncomp > 0 && __float_as_uint(v[0]) == PS_SENT) || (ncomp > 1 && __float_as_uint(v[1]) == PS_SENT) ||
                             (ncomp > 2 && __float_as_uint(v[2]) == PS_SENT) || (ncomp > 3 && __float_as_uint(v[3]) == PS_SENT);
        if (!__any(active && pending)) return false;
        if (spin_expired(spins, err, code)) return true;
        sleep_units(gap);
    }
}

template <bool GREEDY, bool LONG>
__global__ __launch_bounds__(BG_THREADS, 1) void speller_big_fwd_kernel(const BigArgs a) {
    extern __shared__ float lds[];
    float* red = lds;                              // [8 waves][16 utterances][17]: K reduction of a gate tile
    float* qred = red + BG_NW * 16 * 17;           // [8 waves][64]: query reduction
    float* cred = qred + BG_NW * BG_M;             // [8 waves][64]: context reduction
    float* wphiS = cred + BG_NW * 64;              // [64][4]: this workgroup's K-slice of phi
    float* wst = wphiS + BG_M * 4;                 // [8 waves][2]: (max, sum) of a wave's energies
    float* aS = wst + 16;                          // [256]: attention weights of the utterance
    float* keysS = aS + BG_MAXTP;                  // [Tp][68]; LONG (T' > 256): only the FRL = ceil(T'/16) frames whose energies this workgroup computes
    float* featS = keysS + (LONG ? a.FRL : a.Tp) * BG_KLD;      // [Tp][64]: this workgroup's 64 feature columns of every frame
    float* wcS = featS + a.Tp * 64;                // greedy: [32 symbols][64 h1 columns | 64 context columns] of W_c for this column block
    float* wyS = wcS + 32 * 128;                   // greedy: [16 gate rows][32 symbols] of W_y
    int* symS = reinterpret_cast<int*>(wyS + 16 * 32);      // greedy: the fed-back symbol of every utterance
    __shared__ int dead_s;

    // (per-thread indices are re-derived from an opaque copy of threadIdx.x at the start of each phase: see the backward kernel)
    const int wg = blockIdx.x;
    int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, kq = lane >> 4;
    const int B = a.B, U = a.U, Tp = a.Tp;
    const int j0 = wg * 4;
    const size_t sH = (size_t)B * BG_HS;

    // ---- resident weights: 8 k-blocks of each of the four matrices, rows = gate*Hs + unit of this workgroup's 4 units
    f32x4 wc[8], wh0[8], wi1[8], wh1[8];
    {
        const long row = (long)(r >> 2) * BG_HS + j0 + (r & 3);
        const int k0 = wave * 128 + kq * 4;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            wc[i] = ld4p(a.w0p + row * a.ldw0 + a.Vp + k0 + i * 16);
            wh0[i] = ld4p(a.w_hh0 + row * BG_HS + k0 + i * 16);
            wi1[i] = ld4p(a.w_ih1 + row * BG_HS + k0 + i * 16);
            wh1[i] = ld4p(a.w_hh1 + row * BG_HS + k0 + i * 16);
        }
    }
    // ---- cell lanes (wave 0): utterance cb, unit cu
    int cb = lane >> 2, cu = lane & 3;
    bool cell_on = wave == 0 && cb < B;
    float bias0[4] = {0.f, 0.f, 0.f, 0.f}, bias1[4] = {0.f, 0.f, 0.f, 0.f};
    if (wave == 0) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            bias0[g] = a.b_ih0[g * BG_HS + j0 + cu] + a.b_hh0[g * BG_HS + j0 + cu];
            bias1[g] = a.b_ih1[g * BG_HS + j0 + cu] + a.b_hh1[g * BG_HS + j0 + cu];
        }
    }
    float c0 = 0.f, c1 = 0.f;

    // ---- attention role: 64 feature columns of utterance ab, every frame.  The 16 column blocks of an utterance sit on ONE XCD
    //      (workgroup w runs on XCD w % 8): its 64 KB of query parts cross the fabric once.
    const int ab = 2 * (wg & 7) + ((wg >> 3) & 1), aj = wg >> 4;
    const bool att_on = ab < B;
    if (att_on) {
        for (int idx = tid; idx < Tp * 16; idx += BG_THREADS) {
            const int t = idx >> 4, c4 = idx & 15;
            *reinterpret_cast<f32x4*>(featS + t * 64 + c4 * 4) = ld4p(a.feat + ((size_t)ab * Tp + t) * BG_HS + aj * 64 + c4 * 4);
            if (!LONG) *reinterpret_cast<f32x4*>(keysS + t * BG_KLD + c4 * 4) = ld4p(a.keys + ((size_t)ab * Tp + t) * BG_M + c4 * 4);
        }
        if (LONG) {      // the keys of frames [aj FRL, aj FRL + FRL): the 16 workgroups of an utterance share the energies (one more hand-off per step)
            for (int idx = tid; idx < a.FRL * 16; idx += BG_THREADS) {
                const int t = aj * a.FRL + (idx >> 4), c4 = idx & 15;
                if (t < Tp) *reinterpret_cast<f32x4*>(keysS + (idx >> 4) * BG_KLD + c4 * 4) = ld4p(a.keys + ((size_t)ab * Tp + t) * BG_M + c4 * 4);
            }
        }
    }
    if (tid < 256) wphiS[tid] = a.w_phi[(size_t)(tid >> 2) * BG_HS + j0 + (tid & 3)];
    const float bphi = a.b_phi[lane];
    constexpr bool greedy = GREEDY;      // (a template parameter: the teacher-forced instantiation keeps its registers — 0 spills)
    float bc = 0.f;
    if (greedy) {
        if (att_on) {
            for (int idx = tid; idx < 32 * 128; idx += BG_THREADS) {
                const int v = idx >> 7, k = idx & 127;
                wcS[idx] = v < a.V ? a.w_c[(size_t)v * (2 * BG_HS) + (k < 64 ? aj * 64 + k : BG_HS + aj * 64 + (k - 64))] : 0.f;
            }
            if (lane < a.V) bc = a.b_c[lane];
        }
        {
            const int n = tid >> 5, v = tid & 31;      // 512 threads = 16 gate rows x 32 symbols
            wyS[tid] = a.w0p[((size_t)(n >> 2) * BG_HS + j0 + (n & 3)) * a.ldw0 + v];
        }
        if (tid < 16) symS[tid] = 0;                     // <sos> = symbol 0 (las_model.py:193-195)
    }
    if (tid == 0) dead_s = 0;
    __syncthreads();
    bool dead = false;

    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    int rowb = min(r, B - 1);                            // rows beyond the batch repeat the last utterance (never stored)
    unsigned aoff = (unsigned)(rowb * BG_HS + wave * 128 + kq * 4) * 4u;      // this lane's first k-block in a row-major (B,1024) slab
    unsigned hoff = (unsigned)(((wave * 32 + kq) * 16 + rowb) * 4) * 4u;      // ... in a producer-major [256][16][4] slab of h

    // checked loads of this lane's 8 k-blocks of a hand-off slab: through the L2 first (the flags said every producer is through), agent
    // scope if a line was fetched too early
    auto load_ctx = [&](const float* slab, f32x4 (&ax)[8], unsigned code) {
        const float* p = at_bytes(slab, opaque(aoff));
        ld4x8_row_l2(p, ax);
        if (!__any(any_sentinel8(ax))) return;
        if (a.trace && lane == 0) atomicAdd(a.trace + 14, 1ull);
        unsigned spins = 0;
        for (;;) {
            ld4x8_row(p, ax);
            if (!__any(any_sentinel8(ax))) break;
            if (dead || spin_expired(spins, a.err, code)) { dead = true; break; }
        }
    };
    unsigned coff = (unsigned)((wave * 8 * 16 + rowb) * 16 + kq * 4) * 4u;      // ... in a context slab [64 k-blocks][16 utterances][16 columns]
    auto derive = [&]() {
        tid = (int)opaque(threadIdx.x); lane = tid & 63; wave = tid >> 6; r = lane & 15; kq = lane >> 4;
        cb = lane >> 2; cu = lane & 3; cell_on = wave == 0 && cb < B; rowb = min(r, B - 1);
        aoff = (unsigned)(rowb * BG_HS + wave * 128 + kq * 4) * 4u;
        hoff = (unsigned)(((wave * 32 + kq) * 16 + rowb) * 4) * 4u;
        coff = (unsigned)((wave * 8 * 16 + rowb) * 16 + kq * 4) * 4u;
    };
    auto load_cx = [&](const float* slab, f32x4 (&ax)[8], unsigned code) {
        const float* p = at_bytes(slab, opaque(coff));
        ld4x8_kb<true>(p, ax);
        if (!__any(any_sentinel8(ax))) return;
        if (a.trace && lane == 0) atomicAdd(a.trace + 14, 1ull);
        unsigned spins = 0;
        for (;;) {
            ld4x8_kb<false>(p, ax);
            if (!__any(any_sentinel8(ax))) break;
            if (dead || spin_expired(spins, a.err, code)) { dead = true; break; }
        }
    };
    auto load_h = [&](const float* slab, f32x4 (&ax)[8], unsigned code) {
        const float* p = at_bytes(slab, opaque(hoff));
        ld4x8_kb<true>(p, ax);
        if (!__any(any_sentinel8(ax))) return;
        if (a.trace && lane == 0) atomicAdd(a.trace + 14, 1ull);
        unsigned spins = 0;
        for (;;) {
            ld4x8_kb<false>(p, ax);
            if (!__any(any_sentinel8(ax))) break;
            if (dead || spin_expired(spins, a.err, code)) { dead = true; break; }
        }
    };

    for (int s = 0; s < U; ++s) {
        derive();
        BG_STAMP(0);
        if (a.trace && wg == 0 && tid == 0 && s < 64) a.trace[s * 16 + 13] = __builtin_readcyclecounter();      // shader clock (the stamps are 100 MHz)
        const unsigned* fl = a.flags + (size_t)s * BG_FLW;             // flags of this step: [h0 | h1 + query parts | ctx] x 8 copies x 256 producers
        const unsigned* flc = fl + (wg & 7) * BG_WGS;                    // the copy this workgroup polls (one per XCD: 256 pollers on one 1 KB array serialise on its channel)
        // label half of the bottom-layer gates (one GEMM before the launch), off the chain
        float ywv[4] = {0.f, 0.f, 0.f, 0.f};
        if (cell_on && !greedy) {
#pragma unroll
            for (int g = 0; g < 4; ++g) ywv[g] = a.yw[((size_t)s * B + cb) * (4 * BG_HS) + g * BG_HS + j0 + cu];
        }
        f32x4 ax[8];
        // ================= [1] bottom cell: gates0 = yw + W_ctx ctx_{s-1} + W_hh0 h0_{s-1}
        {
            if (wave == 0 && s > 0 && !dead) {
                if (flags_wait(flc - BG_FLW + 2 * 8 * BG_WGS + lane * 4, lane * 4 < B * 16, a.err, 0xB1600001u, (a.tune >> 16) & 255, (a.tune >> 8) & 255)) dead_s = 1;
            }
            BG_WSTAMP(0);
            __syncthreads();
            dead |= dead_s != 0;
            if (s == 0) load_ctx(a.ctx_all, ax, 0xB1600002u);        // ctx_{-1} = feat[:,0,:], row-major, written before the launch
            else load_cx(a.hx + ((size_t)2 * U + s - 1) * BG_WGS * 64, ax, 0xB1600002u);
            acc0 = seg_mfma(ax, wc, acc0);
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) red[(wave * 16 + kq * 4 + rr) * 17 + r] = acc0[rr];
            acc0 = f32x4{0.f, 0.f, 0.f, 0.f};
            __syncthreads();
            BG_STAMP(1);
            if (wave == 0) {
                if (greedy) {      // the symbol every utterance emitted at step s-1: published ~2 us behind its context, i.e. while the product above ran
                    if (s > 0) {
                        unsigned v = 0, spins = 0;
                        for (;;) {
                            v = ld1_agent(a.ysym + (size_t)(s - 1) * 16 + min(lane, B - 1));
                            if (!__any(v == PS_SENT)) break;
                            if (dead || spin_expired(spins, a.err, 0xB1600009u)) { dead = true; dead_s = 1; v = 0; break; }
                        }
                        if (lane < 16) symS[lane] = (int)v & 31;
                    }
                    const int sym = symS[cb];
#pragma unroll
                    for (int g = 0; g < 4; ++g) ywv[g] = wyS[(g * 4 + cu) * 32 + sym];
                }
                float g4[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float sum = 0.f;
#pragma unroll
                    for (int w = 0; w < BG_NW; ++w) sum += red[(w * 16 + cb) * 17 + g * 4 + cu];
                    g4[g] = sum + (ywv[g] + bias0[g]);
                }
                const float ig = sigmoidf_acc(g4[0]), fg = sigmoidf_acc(g4[1]), gg = tanhf_acc(g4[2]), og = sigmoidf_acc(g4[3]);
                c0 = fg * c0 + ig * gg;
                const float h = og * tanhf_acc(c0);
                if (cell_on) st1_agent(a.hx + ((size_t)s * BG_WGS + wg) * 64 + lane, h);          // hand-off copy: 256 contiguous bytes
                if (lane < 8) st1_agent(reinterpret_cast<float*>(const_cast<unsigned*>(fl)) + lane * BG_WGS + wg, 0.f);
                BG_WSTAMP(1);
                if (cell_on) {
                    const size_t o = (size_t)s * sH + (size_t)cb * BG_HS + j0 + cu;       // layer 0
                    a.h_all[o] = h;
                    a.c_all[o] = c0;
                    float* go = a.gates_all + 4 * ((size_t)s * sH) + (size_t)cb * 4 * BG_HS + j0 + cu;
                    go[0] = ig; go[BG_HS] = fg; go[2 * BG_HS] = gg; go[3 * BG_HS] = og;
                }
            }
        }
        BG_STAMP(2);
        // ================= [2] top cell: gates1 = b + W_ih1 h0_s + W_hh1 h1_{s-1}
        derive();
        {
            // the recurrent half W_hh1 h1_{s-1} is multiplied while this step's h0 travels (its operand arrived in [3] of the last step; polling the
            // flags right after publishing only queues reads in front of the flag stores they are waiting for)
            if (s > 0) {
                load_h(a.hx + ((size_t)U + s - 1) * BG_WGS * 64, ax, 0xB1600008u);
                acc1 = seg_mfma(ax, wh1, acc1);
            }
            BG_WSTAMP(6);
            if (wave == 0 && !dead) {
                if (flags_wait(flc + lane * 4, true, a.err, 0xB1600003u, a.tune & 255, (a.tune >> 8) & 255)) dead_s = 1;
            }
            BG_STAMP(9);
            BG_WSTAMP(2);
            __syncthreads();
            BG_STAMP(10);
            dead |= dead_s != 0;
            load_h(a.hx + (size_t)s * BG_WGS * 64, ax, 0xB1600004u);
            BG_STAMP(11);
            acc1 = seg_mfma(ax, wi1, acc1);
            BG_STAMP(12);
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) red[(wave * 16 + kq * 4 + rr) * 17 + r] = acc1[rr];
            acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
            __syncthreads();
            BG_STAMP(3);
            if (wave == 0) {
                float g4[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float sum = 0.f;
#pragma unroll
                    for (int w = 0; w < BG_NW; ++w) sum += red[(w * 16 + cb) * 17 + g * 4 + cu];
                    g4[g] = sum + bias1[g];
                }
                const float ig = sigmoidf_acc(g4[0]), fg = sigmoidf_acc(g4[1]), gg = tanhf_acc(g4[2]), og = sigmoidf_acc(g4[3]);
                c1 = fg * c1 + ig * gg;
                const float h = og * tanhf_acc(c1);
                if (cell_on) st1_agent(a.hx + (((size_t)U + s) * BG_WGS + wg) * 64 + lane, h);
                // this workgroup's K-slice of the query of every utterance: qp[s][b][wg][m] = sum_u W_phi[m][j0+u] h1[b][j0+u]
                const float h_0 = quad_bcast(h, 0), h_1 = quad_bcast(h, 1), h_2 = quad_bcast(h, 2), h_3 = quad_bcast(h, 3);
                float* qo = a.qp + (((size_t)s * B + min(cb, B - 1)) * BG_WGS + wg) * BG_M + cu * 16;
#pragma unroll
                for (int i4 = 0; i4 < 4; ++i4) {
                    f32x4 o4;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const f32x4 w4 = *reinterpret_cast<const f32x4*>(wphiS + (cu * 16 + i4 * 4 + e) * 4);
                        o4[e] = w4[0] * h_0 + w4[1] * h_1 + w4[2] * h_2 + w4[3] * h_3;
                    }
                    if (cell_on) st4_agent(qo + i4 * 4, o4);
                }
                if (lane < 8) st1_agent(reinterpret_cast<float*>(const_cast<unsigned*>(fl)) + (8 + lane) * BG_WGS + wg, 0.f);
                BG_WSTAMP(3);
                if (cell_on) {
                    const size_t o = ((size_t)U + s) * sH + (size_t)cb * BG_HS + j0 + cu;  // layer 1
                    a.h_all[o] = h;
                    a.c_all[o] = c1;
                    float* go = a.gates_all + 4 * (((size_t)U + s) * sH) + (size_t)cb * 4 * BG_HS + j0 + cu;
                    go[0] = ig; go[BG_HS] = fg; go[2 * BG_HS] = gg; go[3 * BG_HS] = og;
                }
            }
            // recurrent half of the NEXT bottom cell while h1 / the query parts travel (the A operand is still in registers)
            acc0 = seg_mfma(ax, wh0, acc0);
        }
        BG_STAMP(4);
        derive();
        // ================= [3] attention of utterance ab, feature columns [64 aj, 64 aj + 64): query, energies of every frame, softmax, context
        {
            const float* qps = a.qp + ((size_t)s * B + min(ab, B - 1)) * BG_WGS * BG_M;
            if (wave == 0 && !dead) {
                if (flags_wait(flc + 8 * BG_WGS + lane * 4, true, a.err, 0xB1600005u, (a.tune >> 24) & 255, (a.tune >> 8) & 255)) dead_s = 1;
            }
            BG_WSTAMP(4);
            __syncthreads();
            dead |= dead_s != 0;
            if (att_on) {
                const int m4 = tid & 15, pg = tid >> 4;
                const float* p = qps + (size_t)pg * BG_M + m4 * 4;
                ld4x8_strided<true>(p, 32 * BG_M, ax);
                if (__any(any_sentinel8(ax))) {
                    if (a.trace && lane == 0) atomicAdd(a.trace + 15, 1ull);
                    unsigned spins = 0;
                    for (;;) {
                        ld4x8_strided<false>(p, 32 * BG_M, ax);
                        if (!__any(any_sentinel8(ax))) break;
                        if (dead || spin_expired(spins, a.err, 0xB1600006u)) { dead = true; break; }
                    }
                }
                f32x4 sum = ax[0];
#pragma unroll
                for (int i = 1; i < 8; ++i) sum += ax[i];
                // the wave's four producer groups (its four rows of 16 lanes) added in registers, one LDS row per wave
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const unsigned x = __float_as_uint(sum[e]);
                    auto r16 = __builtin_amdgcn_permlane16_swap(x, x, false, false);
                    const unsigned y = __float_as_uint(__uint_as_float(r16[0]) + __uint_as_float(r16[1]));
                    auto r32 = __builtin_amdgcn_permlane32_swap(y, y, false, false);
                    sum[e] = __uint_as_float(r32[0]) + __uint_as_float(r32[1]);
                }
                if (lane < 16) *reinterpret_cast<f32x4*>(qred + wave * BG_M + m4 * 4) = sum;
            }
            __syncthreads();
            BG_STAMP(5);
            float pw = 0.f, mw = BG_NEG;
            const bool fr_on = att_on && tid < Tp;                       // thread t owns frame t
            if (att_on && wave * 64 < Tp) {
                float q = bphi;
#pragma unroll
                for (int w = 0; w < BG_NW; ++w) q += qred[w * BG_M + lane];
                q = act_apply(q, a.relu);
                if (aj == 0 && wave == 0) a.q_all[((size_t)s * B + ab) * BG_M + lane] = q;
                float e = 0.f;
                if (!LONG || wave == 0) {
                    const float* kr = keysS + (LONG ? min(lane, a.FRL - 1) : min(tid, Tp - 1)) * BG_KLD;
#pragma unroll
                    for (int m4 = 0; m4 < BG_M / 4; ++m4) {
                        const f32x4 kv = *reinterpret_cast<const f32x4*>(kr + m4 * 4);
                        e = fmaf(lane_f(q, m4 * 4), kv[0], e); e = fmaf(lane_f(q, m4 * 4 + 1), kv[1], e);
                        e = fmaf(lane_f(q, m4 * 4 + 2), kv[2], e); e = fmaf(lane_f(q, m4 * 4 + 3), kv[3], e);
                    }
                }
                if (LONG) {      // this workgroup's frames -> the utterance's energy row; every workgroup of the utterance then reads the whole row
                    float* eg = a.eg + ((size_t)s * B + ab) * 512;
                    if (wave == 0 && lane < a.FRL && aj * a.FRL + lane < Tp) st1_agent(eg + aj * a.FRL + lane, e);
                    unsigned x = 0, spins = 0;
                    for (;;) {
                        x = ld1_agent(eg + min(tid, Tp - 1));
                        if (!__any(x == PS_SENT)) break;
                        if (dead || spin_expired(spins, a.err, 0xB160000Cu)) { dead = true; break; }
                    }
                    e = __uint_as_float(x);
                }
                e = fr_on ? e : BG_NEG;
                mw = wmax(e);
                pw = fr_on ? __builtin_amdgcn_exp2f((e - mw) * BG_LOG2E) : 0.f;
                const float lw = wsum(pw);
                if (lane == 0) { wst[wave * 2] = mw; wst[wave * 2 + 1] = lw; }
            } else if (lane == 0) {
                wst[wave * 2] = BG_NEG; wst[wave * 2 + 1] = 0.f;
            }
            __syncthreads();
            if (att_on) {
                float mx = BG_NEG;
#pragma unroll
                for (int w = 0; w < BG_NW; ++w) mx = fmaxf(mx, wst[w * 2]);
                float tot = 0.f;
#pragma unroll
                for (int w = 0; w < BG_NW; ++w) tot += wst[w * 2 + 1] * __builtin_amdgcn_exp2f((wst[w * 2] - mx) * BG_LOG2E);
                const float at = pw * __builtin_amdgcn_exp2f((mw - mx) * BG_LOG2E) / tot;
                if (fr_on) {
                    aS[tid] = at;
                    if ((tid & 15) == aj) a.att[((size_t)s * B + ab) * Tp + tid] = at;       // the 16 column blocks share the row's stores
                }
            }
            __syncthreads();
            if (att_on) {
                float c0a = 0.f, c1a = 0.f, c2a = 0.f, c3a = 0.f;
                const float* fc = featS + lane;
                int t = wave;
                for (; t + 3 * BG_NW < Tp; t += 4 * BG_NW) {
                    const float a0 = aS[t], a1 = aS[t + BG_NW], a2 = aS[t + 2 * BG_NW], a3 = aS[t + 3 * BG_NW];
                    const float x0 = fc[t * 64], x1 = fc[(t + BG_NW) * 64], x2 = fc[(t + 2 * BG_NW) * 64], x3 = fc[(t + 3 * BG_NW) * 64];
                    c0a = fmaf(a0, x0, c0a); c1a = fmaf(a1, x1, c1a); c2a = fmaf(a2, x2, c2a); c3a = fmaf(a3, x3, c3a);
                }
                for (; t < Tp; t += BG_NW) c0a = fmaf(aS[t], fc[t * 64], c0a);
                cred[wave * 64 + lane] = (c0a + c1a) + (c2a + c3a);
            }
            __syncthreads();
            BG_STAMP(6);
            if (wave == 0 && att_on) {
                float c = 0.f;
#pragma unroll
                for (int w = 0; w < BG_NW; ++w) c += cred[w * 64 + lane];
                // hand-off copy [k-block][utterance][16]: half lines from the producer, 1 KB per load instruction of the cells; row-major stash for the
                // backward pass and the logits
                st1_agent(a.hx + ((size_t)2 * U + s) * BG_WGS * 64 + (((aj * 4 + (lane >> 4)) * 16 + ab) * 16 + (lane & 15)), c);
                a.ctx_all[((size_t)(s + 1) * B + ab) * BG_HS + aj * 64 + lane] = c;
                if (lane < 8) st1_agent(reinterpret_cast<float*>(const_cast<unsigned*>(fl)) + (16 + lane) * BG_WGS + ab * 16 + aj, 0.f);
                BG_WSTAMP(5);
                if (greedy) {
                    // this column block's part of the logits W_c [h1 | ctx] (reference las_model.py:181-182): lane = column
                    const float* hp = a.hx + ((size_t)U + s) * BG_WGS * 64 + ((size_t)((aj * 64 + lane) >> 2) * 16 + ab) * 4 + (lane & 3);
                    unsigned hb = __builtin_nontemporal_load(reinterpret_cast<const unsigned*>(hp));
                    {
                        unsigned spins = 0;
                        while (__any(hb == PS_SENT)) {
                            hb = ld1_agent(hp);
                            if (dead || spin_expired(spins, a.err, 0xB160000Au)) { dead = true; dead_s = 1; break; }
                        }
                    }
                    const float h1v = __uint_as_float(hb);
                    float mine = 0.f;
#pragma unroll
                    for (int v = 0; v < 32; ++v) {
                        const float t = wsum(fmaf(wcS[v * 128 + lane], h1v, wcS[v * 128 + 64 + lane] * c));
                        mine = lane == v ? t : mine;
                    }
                    if (lane < 32) st1_agent(a.lgp + (((size_t)s * B + ab) * 16 + aj) * 32 + lane, mine);
                    if (aj == 0) {      // workgroup (ab, 0): add the 16 parts, log-softmax, first maximal index, feed it back
                        float lg = 0.f;
                        unsigned spins = 0;
                        for (;;) {
                            bool bad = false;
                            lg = 0.f;
#pragma unroll
                            for (int i = 0; i < 16; ++i) {
                                const unsigned x = ld1_agent(a.lgp + (((size_t)s * B + ab) * 16 + i) * 32 + (lane & 31));
                                bad |= x == PS_SENT;
                                lg += __uint_as_float(x);
                            }
                            if (!__any(bad)) break;
                            if (dead || spin_expired(spins, a.err, 0xB160000Bu)) { dead = true; dead_s = 1; break; }
                        }
                        const bool vv = lane < a.V;
                        lg = vv ? lg + bc : -INFINITY;
                        const float m = wmax(lg);
                        const float se = wsum(vv ? expf(lg - m) : 0.f);
                        const float lse = m + logf(se);
                        int best = (vv && lg == m) ? lane : 0x7fffffff;
#pragma unroll
                        for (int mm = 32; mm >= 1; mm >>= 1) best = min(best, __shfl_xor(best, mm));
                        if (vv) a.logp[((size_t)s * B + ab) * a.V + lane] = lg - lse;
                        if (lane == 0) {
                            if (a.argmax) a.argmax[(size_t)s * B + ab] = best;
                            st1_agent(a.ysym + (size_t)s * 16 + ab, __int_as_float(best));
                        }
                        if (lane < 32) a.y_all[((size_t)(s + 1) * B + ab) * 32 + lane] = lane == best ? 1.f : 0.f;
                    }
                }
            }
        }
        BG_STAMP(7);
        BG_STAMP(8);
    }
}

size_t big_fwd_smem(int Tp, int greedy) {
    const size_t krows = Tp > BG_SHORT_TP ? (size_t)((Tp + 15) / 16) : (size_t)Tp;
    return (greedy ? sizeof(float) * (32 * 128 + 16 * 32 + 16) : 0) + sizeof(float) * ((size_t)BG_NW * 16 * 17 + BG_NW * BG_M + BG_NW * 64 + BG_M * 4 + 16 + BG_MAXTP + krows * BG_KLD + (size_t)Tp * 64);
}

// ==================================================================================================================================
// Backward of the same loop in one launch (reference: autograd through model/las_model.py:178-184, 205-236, 275-297 for all U steps).
//
// Per step the four products dX = dG . W contract over the 4096 gate rows, so the weights are held COLUMN-wise: workgroup (matrix m, block j)
// keeps 16 columns x 4096 rows of ONE of [W_ih1 | W_hh1 | W_ctx | W_hh0] in registers (128 VGPRs per lane as v_mfma_f32_16x16x4_f32 B
// operands, 8 waves split K) and multiplies the 16 utterances' gate gradients (a [256 k-blocks][16][16] slab, 256 KB per step and layer)
// by them.  The workgroup that produces a block of dh also applies that block's cell backward (gates, c from the forward stash; dc is a
// register carried across steps) and publishes the resulting dG block as four whole 1 KB lines of the next slab:
//   W_hh1 block j: keeps its product (the recurrent carry of dh1) in registers, adds dcat_h + dqpre . W_phi (16 more MFMAs) at the next step
//                  and applies the top cell's backward             -> dG1
//   W_ih1 block j: dh0 = dG1 . W_ih1 + carry (from W_hh0 block j)  -> bottom cell's backward -> dG0
//   W_ctx block j: the context gradient carried into the previous step; W_hh0 block j: the recurrent carry of dh0.
// Attention backward of utterance b is sliced over 16 workgroups BY TIME (features of T'/16 frames in LDS): da_t = dctx . feat_t, then the
// three slice sums S = sum a da, P1 = sum a da keys, P2 = sum a keys; workgroup (b, 0) adds the 16 triples, dq = P1 - S P2, dqpre = dq act'(q).
// Chain per step: dctx -> slices -> combine -> top cell -> W_ih1 product + bottom cell -> W_ctx product: five hand-offs (flags + sentinel
// slabs as in the forward kernel).
constexpr int BB_APLD = 132;     // one slice triple: S, 3 pad, P1[64], P2[64]
constexpr int BB_FLK = 4;        // flag words per step / 256: [dG1: 8 copies x 64 producers | dG0: 8 x 64] — a consumer polls the copy of its XCD (w % 8):
                                 // 128 pollers on ONE 256-byte flag array serialise on its memory channel, the flag stores queue behind them
constexpr int BB_MAXFR = 32;      // frames per time slice (T' <= 512; the LDS budget — 4 KB of features per frame — admits 30)

struct BigBwdArgs {
    const float* w_ih1; const float* w_hh1; const float* w_hh0; const float* w0p; long ldw0; int Vp;
    const float* w_phi;
    const float* feat; const float* keys; const float* att; const float* q_all; const float* gates_all; const float* c_all;
    const float* dcat_all;
    float* dG_all; float* dctx_all; float* de_all; float* dqpre_all;
    float* dg1x; float* dg0x; float* dcx; float* dh0c; float* apart; unsigned* flags;
    int B, Tp, U, FR, relu, tune;
    unsigned* err;
    u64* trace;
};

__device__ __forceinline__ float ld1_checked(const float* p, unsigned* err, unsigned code, bool& dead) {
    unsigned v = __builtin_nontemporal_load(reinterpret_cast<const unsigned*>(p));
    unsigned spins = 0;
    while (v == PS_SENT) {
        v = ld1_agent(p);
        if (v != PS_SENT) break;
        if (dead || spin_expired(spins, err, code)) { dead = true; break; }
    }
    return __uint_as_float(v);
}

// per-workgroup wall-clock stamps of the 11th step from the end (hop latencies over the chip): slots 4096 + 8 wg + k
#define BB_WSTAMP(k) do { if (a.trace && s == U - 11 && tid == 0) a.trace[4096 + wg * 8 + (k)] = wall_clock64(); } while (0)
#define BB_STAMP(slot) do { if (a.trace && (wg & 63) == 0 && tid == 0 && (U - 1 - s) < 64) a.trace[((wg >> 6) * 64 + (U - 1 - s)) * 16 + (slot)] = wall_clock64(); } while (0)

__global__ __launch_bounds__(BG_THREADS, 1) void speller_big_bwd_kernel(const BigBwdArgs a) {
    extern __shared__ float lds[];
    float* red = lds;                              // [8 waves + phi part][16 utterances][17]
    float* dctxS = red + 9 * 16 * 17;              // [1024]: context gradient of this workgroup's utterance
    float* part = dctxS + BG_HS;                   // [8 waves][132]: slice triple parts
    float* adS = part + BG_NW * BB_APLD;           // [16 frames][2]: (a_t, da_t) of the slice
    float* wphT = adS + 2 * BB_MAXFR;              // [4][64 lanes][4]: phi operands of the W_hh1 role
    float* keysS = wphT + 1024;                    // [FR][68]
    float* featS = keysS + a.FR * BG_KLD;          // [FR][1024]
    __shared__ int dead_s;

    // Every per-thread index is re-derived (from an opaque copy of threadIdx.x) at the start of each phase: derived offsets that the optimiser
    // hoists out of the step loop stay live across the products (128 weight + 32 operand registers) and spill — and a spill reload waits on
    // vmcnt(0), i.e. on the acknowledgement of every store still in flight (1.8 us on the chain when it happened in the slice role).
    const int wg = blockIdx.x;
    int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, kq = lane >> 4;
    const int B = a.B, U = a.U, Tp = a.Tp, FR = a.FR;
    const size_t sH = (size_t)B * BG_HS;
    const int mt = wg >> 6, jb = wg & 63;          // matrix role: 0 W_ih1, 1 W_hh1, 2 W_ctx, 3 W_hh0; column block jb (16 columns)

    // ---- resident weights: k-blocks [32 wave, 32 wave + 32) of the 4096 gate rows, columns 16 jb + (lane & 15)
    f32x4 wreg[32];
    {
        const float* W = mt == 0 ? a.w_ih1 : mt == 1 ? a.w_hh1 : mt == 2 ? a.w0p + a.Vp : a.w_hh0;
        const long ld = mt == 2 ? a.ldw0 : BG_HS;
        const float* wp = W + (long)(wave * 512 + kq * 4) * ld + jb * 16 + r;
#pragma unroll
        for (int i = 0; i < 32; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) wreg[i][e] = wp[(long)(i * 16 + e) * ld];
    }
    if (mt == 1 && wave == 0) {                     // W_hh1 role: phi rows [16 i + 4 kq, +4) x this block's columns, B operands of wave 0
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) wphT[(i * 64 + lane) * 4 + e] = a.w_phi[(long)(i * 16 + kq * 4 + e) * BG_HS + jb * 16 + r];
    }

    // ---- cell-backward threads: utterance pb, column pn of the block
    int pb = tid >> 4, pn = tid & 15;
    bool pw_on = tid < 256 && pb < B;
    float dc_st = 0.f, carry1 = 0.f;

    // ---- attention slice of this workgroup: frames [t0, t0 + nfr) of utterance ab
    const int ab = 2 * (wg & 7) + ((wg >> 3) & 1), aj = wg >> 4;
    const bool att_on = ab < B;
    const int t0 = aj * FR;
    const int nfr = att_on ? max(0, min(FR, Tp - t0)) : 0;
    for (int idx = tid; idx < nfr * 256; idx += BG_THREADS) {
        const int f = idx >> 8, c4 = idx & 255;
        *reinterpret_cast<f32x4*>(featS + f * BG_HS + c4 * 4) = ld4p(a.feat + ((size_t)ab * Tp + t0 + f) * BG_HS + c4 * 4);
    }
    for (int idx = tid; idx < nfr * 16; idx += BG_THREADS) {
        const int f = idx >> 4, m4 = idx & 15;
        *reinterpret_cast<f32x4*>(keysS + f * BG_KLD + m4 * 4) = ld4p(a.keys + ((size_t)ab * Tp + t0 + f) * BG_M + m4 * 4);
    }
    if (tid == 0) dead_s = 0;
    __syncthreads();
    bool dead = false;
    int rowb = min(r, B - 1);
    auto derive = [&]() {
        tid = (int)opaque(threadIdx.x); lane = tid & 63; wave = tid >> 6; r = lane & 15; kq = lane >> 4;
        pb = tid >> 4; pn = tid & 15; pw_on = tid < 256 && pb < B; rowb = min(r, B - 1);
    };
    const int gap = (a.tune >> 8) & 255;

    // product of one 256 KB gate-gradient slab with the resident columns: four chunks of 8 k-blocks per wave
    auto slab_product = [&](const float* slab, unsigned code) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};      // two chains, see seg_mfma
        const float* base = slab + ((size_t)(wave * 32) * 16 + rowb) * 16 + kq * 4;
        f32x4 ax[8];
        // a chunk whose lines were fetched into the L2 too early still shows the sentinel: re-read it at agent scope
        auto settle = [&](const float* p) {
            // one dword per 16-byte piece: a piece is 4 lanes of ONE store instruction inside one 64-byte segment
            bool bad = false;
#pragma unroll
            for (int i = 0; i < 8; ++i) bad |= __float_as_uint(ax[i][3]) == PS_SENT;
            if (!__any(bad)) return;
            unsigned spins = 0;
            for (;;) {
                ld4x8_kb<false>(p, ax);
                if (!__any(any_sentinel8(ax))) break;
                if (dead || spin_expired(spins, a.err, code)) { dead = true; break; }
            }
        };
        // (loading chunk c+1 into a second register set under the MFMAs of chunk c measured the same 6.3 us per product and cost 32 VGPRs,
        // i.e. spills — whose reloads wait on vmcnt(0) and with it on the acknowledgement of every store still in flight: 1.8 us on the chain)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float* p = at_bytes(base, opaque((unsigned)(c * 8192)));
            ld4x8_kb<true>(p, ax);
            settle(p);
#pragma unroll
            for (int i = 0; i < 8; i += 2)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ax[i][e], wreg[c * 8 + i][e], acc, 0, 0, 0);
                    acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(ax[i + 1][e], wreg[c * 8 + i + 1][e], acc2, 0, 0, 0);
                }
        }
        acc += acc2;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) red[(wave * 16 + kq * 4 + rr) * 17 + r] = acc[rr];
    };
    // (D) / (E) waits last ~15 us: a workgroup that polls all the while only queues reads in front of the flag stores it is waiting for.  It sleeps
    // 14/16 (option SPELLER_BIG_TUNE byte 2: sixteenths) of the wait it saw at the previous step before the first poll.
    unsigned est_d = 0, est_e = 0;
    auto long_wait = [&](const unsigned* flp, unsigned code, unsigned& est) {
        const u64 t0 = wall_clock64();
        const unsigned frac = ((a.tune >> 16) & 255) ? ((a.tune >> 16) & 255) : 14u;      // sixteenths of the last wait to sleep through
        sleep_units((int)min((est * frac * 23u) >> 10, 4000u));          // est in 10 ns ticks, a sleep unit is 64 clocks ~ 28 ns: 10 / 28 / 16 ~ 23 / 1024
        const bool gave_up = flags_wait(flp, lane < 16, a.err, code, 0, gap);
        est = (unsigned)(wall_clock64() - t0);
        return gave_up;
    };
    // cell backward of (utterance pb, unit 16 jb + pn) of layer l at step s; publishes the four gate gradients
    // its operands from the forward stash (and the addend of dh) are fetched at the top of the step, long before dh arrives
    float st_g[4] = {0.f, 0.f, 0.f, 0.f}, st_c = 0.f, st_cp = 0.f, st_add = 0.f;
    auto cell_fetch = [&](int l, int s) {
        const int unit = jb * 16 + pn;
        const size_t o = ((size_t)l * U + s) * sH + (size_t)pb * BG_HS + unit;
        const float* gp = a.gates_all + 4 * (((size_t)l * U + s) * sH) + (size_t)pb * 4 * BG_HS + unit;
        st_g[0] = gp[0]; st_g[1] = gp[BG_HS]; st_g[2] = gp[2 * BG_HS]; st_g[3] = gp[3 * BG_HS];
        st_c = a.c_all[o];
        st_cp = s > 0 ? a.c_all[o - sH] : 0.f;
    };
    auto cell_bwd = [&](int l, int s, float dh, float* slab) {
        const int unit = jb * 16 + pn;
        const float ig = st_g[0], fg = st_g[1], gg = st_g[2], og = st_g[3];
        const float tc = tanhf_acc(st_c);
        const float cp = st_cp;
        const float dct = dc_st + dh * og * (1.f - tc * tc);
        float dg[4];
        dg[0] = dct * gg * ig * (1.f - ig);
        dg[1] = dct * cp * fg * (1.f - fg);
        dg[2] = dct * ig * (1.f - gg * gg);
        dg[3] = dh * tc * og * (1.f - og);
        dc_st = dct * fg;
        float* go = a.dG_all + 4 * (((size_t)l * U + s) * sH) + (size_t)pb * 4 * BG_HS + unit;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            st1_agent(slab + ((size_t)(g * 64 + jb) * 16 + pb) * 16 + pn, dg[g]);      // k-block (gate, block): [16 utterances][16 units]
            go[g * BG_HS] = dg[g];
        }
    };

    float nd0 = 0.f, nd1 = 0.f;      // this thread's two columns of dcat's context half for the coming step
    float naf[4] = {0.f, 0.f, 0.f, 0.f}, nqv = 0.f;      // attention weights of this wave's frames (wave, wave + 8, + 16, + 24)
    if (att_on) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (wave + i * BG_NW < nfr) naf[i] = a.att[((size_t)(U - 1) * B + ab) * Tp + t0 + wave + i * BG_NW];
        if (aj == 0 && wave == 0) nqv = a.q_all[((size_t)(U - 1) * B + ab) * BG_M + lane];
        const float* dn = a.dcat_all + ((size_t)(U - 1) * B + ab) * (2 * BG_HS) + BG_HS + tid * 2;
        nd0 = dn[0]; nd1 = dn[1];
    }
    for (int s = U - 1; s >= 0; --s) {
        const bool last = s == U - 1;
        unsigned* fl = a.flags + (size_t)s * BB_FLK * 256;
        derive();
        BB_STAMP(0);
        if (pw_on && mt <= 1) {      // the cell-backward threads' stash operands of this step, and what is added to the product: dcat_h (top) / nothing yet (bottom)
            cell_fetch(mt == 1 ? 1 : 0, s);
            if (mt == 1) st_add = a.dcat_all[((size_t)s * B + pb) * (2 * BG_HS) + jb * 16 + pn];
        }
        if (att_on) {
            // ---- attention weights' gradient of the PREVIOUS iteration's step, off the chain: de_t = a_t (da_t - S), S from the 16 slice triples
            float pa = 0.f, pd = 0.f;
            if (!last && wave == 1 && lane < nfr) { pa = adS[lane * 2]; pd = adS[lane * 2 + 1]; }      // finished behind this step's triple (below)
            // ================= (A) slice triple of this step
            // (the attention weights of this wave's two frames and the combine role's query were fetched one step ahead: a cold global load takes
            // 2 - 3 us here)
            const float af[4] = {naf[0], naf[1], naf[2], naf[3]};
            const float qv = nqv;
            {
                const int c = tid * 2;
                float v0 = nd0, v1 = nd1;
                if (!last) {      // the context gradient carried out of step s+1 (W_ctx blocks): polled at agent scope, 4 KB per workgroup
                    const float* cx = a.dcx + ((size_t)(s + 1) * BG_NB + ab) * BG_HS + c;
                    unsigned x0, x1, spins = 0;
                    for (;;) {
                        x0 = ld1_agent(cx); x1 = ld1_agent(cx + 1);
                        if (!__any(x0 == PS_SENT || x1 == PS_SENT)) break;
                        if (dead || spin_expired(spins, a.err, 0xB1610003u)) { dead = true; break; }
                        sleep_units(gap);
                    }
                    v0 += __uint_as_float(x0); v1 += __uint_as_float(x1);
                }
                BB_WSTAMP(0);
                dctxS[c] = v0; dctxS[c + 1] = v1;
                if (aj == 0) { float* o = a.dctx_all + ((size_t)s * B + ab) * BG_HS + c; o[0] = v0; o[1] = v1; }
            }
            __syncthreads();
            BB_STAMP(1);
            float p1 = 0.f, p2 = 0.f, ssum = 0.f;
            for (int f = wave; f < nfr; f += BG_NW) {
                const float* fr = featS + f * BG_HS + lane * 4;
                float acc = 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    acc = dot4p(*reinterpret_cast<const f32x4*>(fr + 256 * i), *reinterpret_cast<const f32x4*>(dctxS + lane * 4 + 256 * i), acc);
                const float da = wsum(acc);
                const int fi = (f - wave) >> 3;
                const float afv = fi == 0 ? af[0] : fi == 1 ? af[1] : fi == 2 ? af[2] : af[3];
                const float kv = keysS[f * BG_KLD + lane];
                p1 = fmaf(afv * da, kv, p1); p2 = fmaf(afv, kv, p2); ssum = fmaf(afv, da, ssum);
                if (lane == 0) { adS[f * 2] = afv; adS[f * 2 + 1] = da; }
            }
            BB_STAMP(9);
            part[wave * BB_APLD + 4 + lane] = p1; part[wave * BB_APLD + 68 + lane] = p2;
            if (lane == 0) part[wave * BB_APLD] = ssum;
            __syncthreads();
            BB_STAMP(10);
            if (wave == 0) {
                float q1 = 0.f, q2 = 0.f, qs = 0.f;
#pragma unroll
                for (int w = 0; w < BG_NW; ++w) { q1 += part[w * BB_APLD + 4 + lane]; q2 += part[w * BB_APLD + 68 + lane]; qs += part[w * BB_APLD]; }
                float* ap = a.apart + (((size_t)s * B + ab) * 16 + aj) * BB_APLD;
                st1_agent(ap + 4 + lane, q1); st1_agent(ap + 68 + lane, q2);
                if (lane == 0) st1_agent(ap, qs);
                BB_WSTAMP(1);
            } else if (wave == 1 && !last) {
                const float* apb = a.apart + ((size_t)(s + 1) * B + ab) * 16 * BB_APLD;
                const float si = ld1_checked(apb + (size_t)(lane & 15) * BB_APLD, a.err, 0xB1610001u, dead);
                const float st = lane_f(gsum<16>(si), 0);
                if (lane < nfr) a.de_all[((size_t)(s + 1) * B + ab) * Tp + t0 + lane] = pa * (pd - st);
            }
            if (s > 0) {      // operands of the NEXT step, issued behind this step's last chain-critical load of the slice role: loads return in order,
                              // so a cold one (2 - 3 us) in front of a poll would sit on the chain
                const float* dn = a.dcat_all + ((size_t)(s - 1) * B + ab) * (2 * BG_HS) + BG_HS + tid * 2;
                nd0 = dn[0]; nd1 = dn[1];
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (wave + i * BG_NW < nfr) naf[i] = a.att[((size_t)(s - 1) * B + ab) * Tp + t0 + wave + i * BG_NW];
                if (aj == 0 && wave == 0) nqv = a.q_all[((size_t)(s - 1) * B + ab) * BG_M + lane];
            }
            if (wave == 0) {
                BB_STAMP(2);
                // ================= (B) workgroup (ab, 0): add the 16 triples, dq = P1 - S P2, dqpre = dq act'(q)
                if (aj == 0) {
                    const float* apb = a.apart + ((size_t)s * B + ab) * 16 * BB_APLD;
                    unsigned vs = 0;
                    float t1 = 0.f, t2 = 0.f;
                    unsigned spins = 0;
                    for (;;) {
                        bool bad = false;
                        t1 = 0.f; t2 = 0.f;
#pragma unroll
                        for (int i = 0; i < 16; ++i) {
                            const unsigned x1 = ld1_agent(apb + (size_t)i * BB_APLD + 4 + lane), x2 = ld1_agent(apb + (size_t)i * BB_APLD + 68 + lane);
                            bad |= x1 == PS_SENT || x2 == PS_SENT;
                            t1 += __uint_as_float(x1); t2 += __uint_as_float(x2);
                        }
                        vs = ld1_agent(apb + (size_t)(lane & 15) * BB_APLD);
                        bad |= vs == PS_SENT;
                        if (!__any(bad)) break;
                        if (dead || spin_expired(spins, a.err, 0xB1610005u)) { dead = true; dead_s = 1; break; }
                        sleep_units(gap);
                    }
                    const float st = lane_f(gsum<16>(__uint_as_float(vs)), 0);
                    float dq = t1 - st * t2;
                    if (a.relu) dq *= act_grad(qv, a.relu);
                    st1_agent(a.dqpre_all + ((size_t)s * B + ab) * BG_M + lane, dq);
                    BB_WSTAMP(2);
                    BB_STAMP(3);
                }
            }
        }
        // ================= (C) top cell's backward: W_hh1 blocks
        derive();
        if (mt == 1) {
            if (wave == 0) {
                f32x4 aq[4];
                const float* qp = a.dqpre_all + ((size_t)s * B + rowb) * BG_M + kq * 4;
                unsigned spins = 0;
                for (;;) {
                    asm volatile(
                        "global_load_dwordx4 %0, %4, off sc1\n\t"
                        "global_load_dwordx4 %1, %4, off offset:64 sc1\n\t"
                        "global_load_dwordx4 %2, %4, off offset:128 sc1\n\t"
                        "global_load_dwordx4 %3, %4, off offset:192 sc1\n\t"
                        "s_waitcnt vmcnt(0)"
                        : "=&v"(aq[0]), "=&v"(aq[1]), "=&v"(aq[2]), "=&v"(aq[3]) : "v"(qp) : "memory");
                    if (!__any(has_sentinel(aq[0]) || has_sentinel(aq[1]) || has_sentinel(aq[2]) || has_sentinel(aq[3]))) break;
                    if (dead || spin_expired(spins, a.err, 0xB1610007u)) { dead = true; dead_s = 1; break; }
                    sleep_units(gap);
                }
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const f32x4 wv = *reinterpret_cast<const f32x4*>(wphT + (i * 64 + lane) * 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[i][e], wv[e], acc, 0, 0, 0);
                }
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) red[(8 * 16 + kq * 4 + rr) * 17 + r] = acc[rr];
            }
            __syncthreads();
            dead |= dead_s != 0;
            BB_STAMP(4);
            if (pw_on) {
                const float dh = red[(8 * 16 + pb) * 17 + pn] + st_add + carry1;
                cell_bwd(1, s, dh, a.dg1x + (size_t)s * 65536);
            }
            __syncthreads();
            if (tid < 8) st1_agent(reinterpret_cast<float*>(fl) + tid * 64 + jb, 0.f);
            BB_WSTAMP(3);
            BB_STAMP(5);
        }
        derive();
        // ================= (D) products with dG1: W_ih1 blocks (+ bottom cell's backward), W_hh1 blocks (recurrent carry of dh1)
        if (mt <= 1) {
            if (wave == 0 && !dead) {
                if (long_wait(fl + (wg & 7) * 64 + lane * 4, 0xB1610008u, est_d)) dead_s = 1;
            }
            BB_WSTAMP(4);
            __syncthreads();
            dead |= dead_s != 0;
            BB_STAMP(6);
            unsigned h0c_raw = 0;      // W_ih1 role: the recurrent carry of dh0 from the W_hh0 block of the same columns (published during the last step)
            if (mt == 0 && pw_on && !last) h0c_raw = __builtin_nontemporal_load(reinterpret_cast<const unsigned*>(a.dh0c + (((size_t)(s + 1) * 64 + jb) * 16 + pb) * 16 + pn));
            slab_product(a.dg1x + (size_t)s * 65536, 0xB1610009u);
            __syncthreads();
            derive();
            BB_STAMP(7);
            float cs = 0.f;
            if (tid < 256) {
#pragma unroll
                for (int w = 0; w < BG_NW; ++w) cs += red[(w * 16 + pb) * 17 + pn];
            }
            if (mt == 1) {
                carry1 = cs;
            } else {
                if (pw_on) {
                    float dh = cs;
                    if (!last) {
                        if (h0c_raw == PS_SENT) h0c_raw = __float_as_uint(ld1_checked(a.dh0c + (((size_t)(s + 1) * 64 + jb) * 16 + pb) * 16 + pn, a.err, 0xB161000Au, dead));
                        dh += __uint_as_float(h0c_raw);
                    }
                    cell_bwd(0, s, dh, a.dg0x + (size_t)s * 65536);
                }
                __syncthreads();
                if (tid < 8) st1_agent(reinterpret_cast<float*>(fl) + 512 + tid * 64 + jb, 0.f);
                BB_WSTAMP(5);
            }
            BB_STAMP(8);
        }
        derive();
        // ================= (E) products with dG0: W_ctx blocks (context gradient carried into step s-1), W_hh0 blocks (recurrent carry of dh0)
        if (mt >= 2) {
            if (wave == 0 && !dead) {
                if (long_wait(fl + 512 + (wg & 7) * 64 + lane * 4, 0xB161000Bu, est_e)) dead_s = 1;
            }
            BB_WSTAMP(6);
            __syncthreads();
            dead |= dead_s != 0;
            BB_STAMP(6);
            slab_product(a.dg0x + (size_t)s * 65536, 0xB161000Cu);
            __syncthreads();
            derive();
            BB_STAMP(7);
            if (pw_on) {
                float cs = 0.f;
#pragma unroll
                for (int w = 0; w < BG_NW; ++w) cs += red[(w * 16 + pb) * 17 + pn];
                if (mt == 2) st1_agent(a.dcx + ((size_t)s * BG_NB + pb) * BG_HS + jb * 16 + pn, cs);
                else st1_agent(a.dh0c + (((size_t)s * 64 + jb) * 16 + pb) * 16 + pn, cs);
            }
            BB_WSTAMP(7);
            BB_STAMP(8);
        }
    }
    // attention weights' gradient of step 0
    if (att_on && wave == 1) {
        const float* apb = a.apart + ((size_t)ab) * 16 * BB_APLD;
        const float si = ld1_checked(apb + (size_t)(lane & 15) * BB_APLD, a.err, 0xB161000Du, dead);
        const float st = lane_f(gsum<16>(si), 0);
        if (lane < nfr) a.de_all[((size_t)ab) * Tp + t0 + lane] = adS[lane * 2] * (adS[lane * 2 + 1] - st);
    }
}

size_t big_bwd_smem(int FR) {
    return sizeof(float) * ((size_t)9 * 16 * 17 + BG_HS + BG_NW * BB_APLD + 2 * BB_MAXFR + 1024 + (size_t)FR * BG_KLD + (size_t)FR * BG_HS);
}

u64* g_big_bwd_trace = nullptr;
u64* g_big_trace = nullptr;

}  // namespace

void speller_big_set_trace(unsigned long long* dev_buf) { g_big_trace = dev_buf; }
void speller_big_bwd_set_trace(unsigned long long* dev_buf) { g_big_bwd_trace = dev_buf; }

bool speller_big_shape(int B, int Tp, int Hs, int D, int M, int V, int L, int heads, int use_mlp) {
    if (L != 2 || heads != 1 || !use_mlp || M != BG_M || Hs != BG_HS || D != BG_HS) return false;
    if (B < 1 || B > BG_NB || Tp < 1 || Tp > BG_MAXTP) return false;
    return true;
}

size_t speller_big_hx_floats(int U) { return (size_t)3 * U * BG_WGS * 64; }
size_t speller_big_qp_floats(int B, int U) { return (size_t)U * B * BG_WGS * BG_M; }
size_t speller_big_flag_words(int U) { return (size_t)U * BG_FLW; }

static bool big_fits(int Tp, int greedy) {
    const size_t smem = big_fwd_smem(Tp, greedy);
    if (smem > 160 * 1024) return false;
    const bool lng = Tp > BG_SHORT_TP;
#define BG_FITS(G, L)                                                                                                                        \
    if ((bool)greedy == G && lng == L) {                                                                                                     \
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&speller_big_fwd_kernel<G, L>), hipFuncAttributeMaxDynamicSharedMemorySize,  \
                                (int)smem) != hipSuccess)                                                                                    \
            return false;                                                                                                                    \
        return persistent_launch_fits(speller_big_fwd_kernel<G, L>, BG_THREADS, smem, BG_WGS);                                               \
    }
    BG_FITS(false, false) BG_FITS(false, true) BG_FITS(true, false) BG_FITS(true, true)
#undef BG_FITS
    return false;
}

bool speller_big_eligible(int B, int Tp, int Hs, int D, int M, int V, int L, int heads, int use_mlp, int greedy) {
    if (opt_get(OPT_SPELLER_BIG) == 0 || !speller_big_shape(B, Tp, Hs, D, M, V, L, heads, use_mlp)) return false;
    if (greedy && (V > 32 || V <= 16)) return false;      // the fed-back symbol rows are 32 floats wide (Vp = 32)
    return big_fits(Tp, greedy);
}
size_t speller_big_greedy_floats(int B, int U) { return (size_t)U * B * 16 * 32 + (size_t)U * 16; }

int speller_big_fwd(const BigFwd& p, hipStream_t stream) {
    LAS_REQUIRE(speller_big_shape(p.B, p.Tp, BG_HS, BG_HS, BG_M, p.V, 2, 1, 1), "one-launch decode (Hs = 1024) shape");
    LAS_REQUIRE(p.err != nullptr, "the persistent speller needs the device error word");
    LAS_REQUIRE((uintptr_t)p.flags % 16 == 0 && (uintptr_t)p.hx % 16 == 0 && (uintptr_t)p.qp % 16 == 0, "hand-off slab alignment");
    BigArgs a;
    a.w0p = p.w0p; a.ldw0 = p.Vp + BG_HS; a.Vp = p.Vp;
    a.w_hh0 = p.w_hh0; a.w_ih1 = p.w_ih1; a.w_hh1 = p.w_hh1;
    a.b_ih0 = p.b_ih0; a.b_hh0 = p.b_hh0; a.b_ih1 = p.b_ih1; a.b_hh1 = p.b_hh1;
    a.w_phi = p.w_phi; a.b_phi = p.b_phi;
    a.feat = p.feat; a.keys = p.keys; a.yw = p.yw;
    a.ctx_all = p.ctx_all; a.h_all = p.h_all; a.c_all = p.c_all; a.gates_all = p.gates_all; a.q_all = p.q_all; a.att = p.att;
    a.hx = p.hx; a.qp = p.qp; a.flags = p.flags;
    const bool lng = p.Tp > BG_SHORT_TP;
    a.eg = p.eg; a.FRL = (p.Tp + 15) / 16;
    LAS_REQUIRE(!lng || p.eg, "energy rows for T' > 256");
    a.mode = p.mode; a.V = p.V; a.w_c = p.w_c; a.b_c = p.b_c; a.logp = p.logp; a.argmax = p.argmax; a.y_all = p.y_all;
    a.lgp = p.lgp; a.ysym = p.lgp ? p.lgp + (size_t)p.U * p.B * 16 * 32 : nullptr;
    LAS_REQUIRE(p.mode == 0 || (p.mode == 1 && p.w_c && p.b_c && p.logp && p.y_all && p.lgp && p.Vp == 32), "greedy decode operands");
    a.B = p.B; a.Tp = p.Tp; a.U = p.U; a.relu = p.relu; a.err = p.err;
    a.trace = g_big_trace;
    a.tune = (int)opt_get(OPT_SPELLER_BIG_TUNE);
    if (!big_fits(p.Tp, p.mode))
        return fail(LAS_ERR_UNSUPPORTED, "one-launch decode (Hs = 1024): %s%ld workgroups cannot all be resident", "", (long)BG_WGS);
    // sentinel-fill what the phases hand over: the operand-order copies of h and of the contexts, the query slices and the flags
    LAS_HIP_CHECK(hipMemsetAsync(p.hx, 0xFF, sizeof(float) * speller_big_hx_floats(p.U), stream));
    if ((float*)p.flags == p.qp + speller_big_qp_floats(p.B, p.U)) {      // adjacent (the layout las_capi.hip uses): one fill
        LAS_HIP_CHECK(hipMemsetAsync(p.qp, 0xFF, sizeof(float) * (speller_big_qp_floats(p.B, p.U) + speller_big_flag_words(p.U)), stream));
    } else {
        LAS_HIP_CHECK(hipMemsetAsync(p.qp, 0xFF, sizeof(float) * speller_big_qp_floats(p.B, p.U), stream));
        LAS_HIP_CHECK(hipMemsetAsync(p.flags, 0xFF, sizeof(unsigned) * speller_big_flag_words(p.U), stream));
    }
    if (lng) LAS_HIP_CHECK(hipMemsetAsync(p.eg, 0xFF, sizeof(float) * (size_t)p.U * p.B * 512, stream));
    if (p.mode == 1) LAS_HIP_CHECK(hipMemsetAsync(p.lgp, 0xFF, sizeof(float) * speller_big_greedy_floats(p.B, p.U), stream));
    {
        KernelTimer timer(TIMED_DECODE_FWD, stream);
        const size_t smem = big_fwd_smem(p.Tp, p.mode);
        if (p.mode == 1 && lng) hipLaunchKernelGGL((speller_big_fwd_kernel<true, true>), dim3(BG_WGS), dim3(BG_THREADS), smem, stream, a);
        else if (p.mode == 1) hipLaunchKernelGGL((speller_big_fwd_kernel<true, false>), dim3(BG_WGS), dim3(BG_THREADS), smem, stream, a);
        else if (lng) hipLaunchKernelGGL((speller_big_fwd_kernel<false, true>), dim3(BG_WGS), dim3(BG_THREADS), smem, stream, a);
        else hipLaunchKernelGGL((speller_big_fwd_kernel<false, false>), dim3(BG_WGS), dim3(BG_THREADS), smem, stream, a);
    }
    LAS_LAUNCH_CHECK();
    return LAS_OK;
}

// ---- backward
static int big_bwd_frames(int Tp) { return (Tp + 15) / 16; }
size_t speller_big_bwd_workspace_floats(int B, int U) {
    return (size_t)U * (2 * 65536 + BG_NB * BG_HS + 64 * 256 + (size_t)B * 16 * BB_APLD + BB_FLK * 256);
}
static bool big_bwd_fits(int FR) {
    const size_t smem = big_bwd_smem(FR);
    if (smem > 160 * 1024) return false;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&speller_big_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
        return false;
    return persistent_launch_fits(speller_big_bwd_kernel, BG_THREADS, smem, BG_WGS);
}
bool speller_big_bwd_eligible(int B, int Tp, int Hs, int D, int M, int V, int L, int heads, int use_mlp) {
    if (opt_get(OPT_SPELLER_BIG_BWD) == 0 || !speller_big_shape(B, Tp, Hs, D, M, V, L, heads, use_mlp)) return false;
    return big_bwd_fits(big_bwd_frames(Tp));
}

int speller_big_bwd(const BigBwd& p, hipStream_t stream) {
    LAS_REQUIRE(speller_big_shape(p.B, p.Tp, BG_HS, BG_HS, BG_M, p.V, 2, 1, 1), "one-launch decode backward (Hs = 1024) shape");
    LAS_REQUIRE(p.err != nullptr, "the persistent speller needs the device error word");
    LAS_REQUIRE((uintptr_t)p.xbuf % 16 == 0, "hand-off slab alignment");
    BigBwdArgs a;
    a.w_ih1 = p.w_ih1; a.w_hh1 = p.w_hh1; a.w_hh0 = p.w_hh0; a.w0p = p.w0p; a.ldw0 = p.Vp + BG_HS; a.Vp = p.Vp; a.w_phi = p.w_phi;
    a.feat = p.feat; a.keys = p.keys; a.att = p.att; a.q_all = p.q_all; a.gates_all = p.gates_all; a.c_all = p.c_all; a.dcat_all = p.dcat_all;
    a.dG_all = p.dG_all; a.dctx_all = p.dctx_all; a.de_all = p.de_all; a.dqpre_all = p.dqpre_all;
    float* o = p.xbuf;
    a.dg1x = o; o += (size_t)p.U * 65536;
    a.dg0x = o; o += (size_t)p.U * 65536;
    a.dcx = o; o += (size_t)p.U * BG_NB * BG_HS;
    a.dh0c = o; o += (size_t)p.U * 64 * 256;
    a.apart = o; o += (size_t)p.U * p.B * 16 * BB_APLD;
    a.flags = reinterpret_cast<unsigned*>(o);
    a.B = p.B; a.Tp = p.Tp; a.U = p.U; a.FR = big_bwd_frames(p.Tp); a.relu = p.relu; a.err = p.err;
    a.tune = (int)opt_get(OPT_SPELLER_BIG_TUNE);
    a.trace = g_big_bwd_trace;
    if (!big_bwd_fits(a.FR))
        return fail(LAS_ERR_UNSUPPORTED, "one-launch decode backward (Hs = 1024): %s%ld workgroups cannot all be resident", "", (long)BG_WGS);
    LAS_HIP_CHECK(hipMemsetAsync(p.xbuf, 0xFF, sizeof(float) * speller_big_bwd_workspace_floats(p.B, p.U), stream));
    LAS_HIP_CHECK(hipMemsetAsync(p.dqpre_all, 0xFF, sizeof(float) * (size_t)p.U * p.B * BG_M, stream));
    {
        KernelTimer timer(TIMED_DECODE_BWD, stream);
        hipLaunchKernelGGL(speller_big_bwd_kernel, dim3(BG_WGS), dim3(BG_THREADS), big_bwd_smem(a.FR), stream, a);
    }
    LAS_LAUNCH_CHECK();
    return LAS_OK;
}
// the context gradient carried out of step 0 (gradient of the initial context feat[:,0,:]): row-major (16, 1024) block inside xbuf
const float* speller_big_bwd_dx0_ctx(const float* xbuf, int U) { return xbuf + (size_t)U * 2 * 65536; }

}  // namespace las
