// One-launch teacher-forced decode forward for the reference's SHIPPED model size (config/librispeech-config.yaml:16-34:
// Listener 512x3 -> 1024 features, Speller 1024x2, attention MLP 64, batch 16) on gfx950.
//
// Replaces, for that size, the 3 launches per decode step of speller.hip (reference model/las_model.py:178-184, 205-236, 275-297),
// each of which re-streams 34 MB of LSTM weights from the L2 / Infinity Cache.  Here the four fp32 matrices of the two cells
// (67 MB) stay in the REGISTERS of 256 workgroups for all U steps:
//   * workgroup w owns 4 hidden units (16 gate rows) of BOTH layers; its 8 waves split K, a lane holds 8 k-blocks of each of
//     [W_ctx | W_hh0 | W_ih1 | W_hh1] as v_mfma_f32_16x16x4_f32 B operands (128 VGPRs), the 16 utterances are the M dimension;
//   * the attention of utterance b is sliced over the 16 workgroups 16b..16b+15 (T'/16 frames each, features and keys in LDS):
//     partial softmax (local max / sum / unnormalised partial context), combined per 64-column block by the same workgroups;
//   * the query never needs its own hop: with the top cell's h the owner publishes its K-slice of phi (64 x 4 weights), and an
//     attention workgroup adds the 256 slices of its utterance.
// Per step the chain is four hand-offs (ctx -> bottom cell -> top cell -> query parts -> partial contexts -> ctx); every
// hand-off is the data itself, written agent-scope into slabs pre-filled with the sentinel 0xFFFFFFFF (persist_common.h).
// The recurrent halves W_hh . h of the next step are multiplied while the hand-offs are in flight.
// Stash layout (h_all, c_all, gates_all, q_all, ctx_all, att) is the per-step kernels', so las_speller_bwd is unchanged.
#include "las_common.h"
#include "las_kernels.h"
#include "options.h"
#include "persist_common.h"

namespace las {
namespace {

constexpr int BG_THREADS = 512, BG_NW = 8, BG_HS = 1024, BG_M = 64, BG_WGS = 256, BG_NB = 16;
constexpr int BG_KLD = BG_M + 4;         // LDS row stride of the keys
constexpr int BG_MAXTP = 256;            // encoder frames: features and keys of the workgroup's column block stay in LDS (528 B per frame)
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
__device__ __forceinline__ f32x4 seg_mfma(const f32x4 (&ax)[8], const f32x4 (&w)[8], f32x4 acc) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ax[i][e], w[i][e], acc, 0, 0, 0);
    return acc;
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
__device__ __forceinline__ bool flags_wait(const unsigned* fl, bool active, unsigned* err, unsigned code, int first, int gap) {
    unsigned spins = 0;
    sleep_units(first);
    for (;;) {
        const f32x4 v = ld4_agent(reinterpret_cast<const float*>(fl));
        if (!__any(active && has_sentinel(v))) return false;
        if (spin_expired(spins, err, code)) return true;
        sleep_units(gap);
    }
}

__global__ __launch_bounds__(BG_THREADS, 1) void speller_big_fwd_kernel(const BigArgs a) {
    extern __shared__ float lds[];
    float* red = lds;                              // [8 waves][16 utterances][17]: K reduction of a gate tile
    float* qred = red + BG_NW * 16 * 17;           // [8 waves][64]: query reduction
    float* cred = qred + BG_NW * BG_M;             // [8 waves][64]: context reduction
    float* wphiS = cred + BG_NW * 64;              // [64][4]: this workgroup's K-slice of phi
    float* wst = wphiS + BG_M * 4;                 // [8 waves][2]: (max, sum) of a wave's energies
    float* aS = wst + 16;                          // [256]: attention weights of the utterance
    float* keysS = aS + BG_MAXTP;                  // [Tp][68]
    float* featS = keysS + a.Tp * BG_KLD;          // [Tp][64]: this workgroup's 64 feature columns of every frame
    __shared__ int dead_s;

    const int wg = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, kq = lane >> 4;
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
    const int cb = lane >> 2, cu = lane & 3;
    const bool cell_on = wave == 0 && cb < B;
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
            *reinterpret_cast<f32x4*>(keysS + t * BG_KLD + c4 * 4) = ld4p(a.keys + ((size_t)ab * Tp + t) * BG_M + c4 * 4);
        }
    }
    if (tid < 256) wphiS[tid] = a.w_phi[(size_t)(tid >> 2) * BG_HS + j0 + (tid & 3)];
    const float bphi = a.b_phi[lane];
    if (tid == 0) dead_s = 0;
    __syncthreads();
    bool dead = false;

    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    const int rowb = min(r, B - 1);                      // rows beyond the batch repeat the last utterance (never stored)
    const unsigned aoff = (unsigned)(rowb * BG_HS + wave * 128 + kq * 4) * 4u;      // this lane's first k-block in a row-major (B,1024) slab
    const unsigned hoff = (unsigned)(((wave * 32 + kq) * 16 + rowb) * 4) * 4u;      // ... in a producer-major [256][16][4] slab of h

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
    const unsigned coff = (unsigned)((wave * 8 * 16 + rowb) * 16 + kq * 4) * 4u;      // ... in a context slab [64 k-blocks][16 utterances][16 columns]
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
        BG_STAMP(0);
        if (a.trace && wg == 0 && tid == 0 && s < 64) a.trace[s * 16 + 13] = __builtin_readcyclecounter();      // shader clock (the stamps are 100 MHz)
        const unsigned* fl = a.flags + (size_t)s * 3 * BG_WGS;          // flags of this step: [h0 | h1 + query parts | ctx]
        // label half of the bottom-layer gates (one GEMM before the launch), off the chain
        float ywv[4] = {0.f, 0.f, 0.f, 0.f};
        if (cell_on) {
#pragma unroll
            for (int g = 0; g < 4; ++g) ywv[g] = a.yw[((size_t)s * B + cb) * (4 * BG_HS) + g * BG_HS + j0 + cu];
        }
        f32x4 ax[8];
        // ================= [1] bottom cell: gates0 = yw + W_ctx ctx_{s-1} + W_hh0 h0_{s-1}
        {
            if (wave == 0 && s > 0 && !dead) {
                if (flags_wait(fl - 3 * BG_WGS + 2 * BG_WGS + lane * 4, lane * 4 < B * 16, a.err, 0xB1600001u, (a.tune >> 16) & 255, (a.tune >> 8) & 255)) dead_s = 1;
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
                if (lane == 0) st1_agent(reinterpret_cast<float*>(const_cast<unsigned*>(fl)) + wg, 0.f);
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
        {
            // the recurrent half W_hh1 h1_{s-1} is multiplied while this step's h0 travels (its operand arrived in [3] of the last step; polling the
            // flags right after publishing only queues reads in front of the flag stores they are waiting for)
            if (s > 0) {
                load_h(a.hx + ((size_t)U + s - 1) * BG_WGS * 64, ax, 0xB1600008u);
                acc1 = seg_mfma(ax, wh1, acc1);
            }
            BG_WSTAMP(6);
            if (wave == 0 && !dead) {
                if (flags_wait(fl + lane * 4, true, a.err, 0xB1600003u, a.tune & 255, (a.tune >> 8) & 255)) dead_s = 1;
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
                if (lane == 0) st1_agent(reinterpret_cast<float*>(const_cast<unsigned*>(fl)) + BG_WGS + wg, 0.f);
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
        // ================= [3] attention of utterance ab, feature columns [64 aj, 64 aj + 64): query, energies of every frame, softmax, context
        {
            const float* qps = a.qp + ((size_t)s * B + min(ab, B - 1)) * BG_WGS * BG_M;
            if (wave == 0 && !dead) {
                if (flags_wait(fl + BG_WGS + lane * 4, true, a.err, 0xB1600005u, (a.tune >> 24) & 255, (a.tune >> 8) & 255)) dead_s = 1;
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
                const float* kr = keysS + min(tid, Tp - 1) * BG_KLD;
                float e = 0.f;
#pragma unroll
                for (int m4 = 0; m4 < BG_M / 4; ++m4) {
                    const f32x4 kv = *reinterpret_cast<const f32x4*>(kr + m4 * 4);
                    e = fmaf(lane_f(q, m4 * 4), kv[0], e); e = fmaf(lane_f(q, m4 * 4 + 1), kv[1], e);
                    e = fmaf(lane_f(q, m4 * 4 + 2), kv[2], e); e = fmaf(lane_f(q, m4 * 4 + 3), kv[3], e);
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
                if (lane == 0) st1_agent(reinterpret_cast<float*>(const_cast<unsigned*>(fl)) + 2 * BG_WGS + ab * 16 + aj, 0.f);
                BG_WSTAMP(5);
            }
        }
        BG_STAMP(7);
        BG_STAMP(8);
    }
}

size_t big_fwd_smem(int Tp) {
    return sizeof(float) * ((size_t)BG_NW * 16 * 17 + BG_NW * BG_M + BG_NW * 64 + BG_M * 4 + 16 + BG_MAXTP + (size_t)Tp * BG_KLD + (size_t)Tp * 64);
}

u64* g_big_trace = nullptr;

}  // namespace

void speller_big_set_trace(unsigned long long* dev_buf) { g_big_trace = dev_buf; }

bool speller_big_shape(int B, int Tp, int Hs, int D, int M, int V, int L, int heads, int use_mlp) {
    if (L != 2 || heads != 1 || !use_mlp || M != BG_M || Hs != BG_HS || D != BG_HS) return false;
    if (B < 1 || B > BG_NB || Tp < 1 || Tp > BG_MAXTP) return false;
    return true;
}

size_t speller_big_hx_floats(int U) { return (size_t)3 * U * BG_WGS * 64; }
size_t speller_big_qp_floats(int B, int U) { return (size_t)U * B * BG_WGS * BG_M; }
size_t speller_big_flag_words(int U) { return (size_t)U * 3 * BG_WGS; }

static bool big_fits(int Tp) {
    const size_t smem = big_fwd_smem(Tp);
    if (smem > 160 * 1024) return false;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&speller_big_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
        return false;
    return persistent_launch_fits(speller_big_fwd_kernel, BG_THREADS, smem, BG_WGS);
}

bool speller_big_eligible(int B, int Tp, int Hs, int D, int M, int V, int L, int heads, int use_mlp) {
    if (opt_get(OPT_SPELLER_BIG) == 0 || !speller_big_shape(B, Tp, Hs, D, M, V, L, heads, use_mlp)) return false;
    return big_fits(Tp);
}

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
    a.B = p.B; a.Tp = p.Tp; a.U = p.U; a.relu = p.relu; a.err = p.err;
    a.trace = g_big_trace;
    a.tune = (int)opt_get(OPT_SPELLER_BIG_TUNE);
    if (!big_fits(p.Tp))
        return fail(LAS_ERR_UNSUPPORTED, "one-launch decode (Hs = 1024): %s%ld workgroups cannot all be resident", "", (long)BG_WGS);
    // sentinel-fill what the phases hand over: the operand-order copies of h and of the contexts, the query slices and the flags
    LAS_HIP_CHECK(hipMemsetAsync(p.hx, 0xFF, sizeof(float) * speller_big_hx_floats(p.U), stream));
    if ((float*)p.flags == p.qp + speller_big_qp_floats(p.B, p.U)) {      // adjacent (the layout las_capi.hip uses): one fill
        LAS_HIP_CHECK(hipMemsetAsync(p.qp, 0xFF, sizeof(float) * (speller_big_qp_floats(p.B, p.U) + speller_big_flag_words(p.U)), stream));
    } else {
        LAS_HIP_CHECK(hipMemsetAsync(p.qp, 0xFF, sizeof(float) * speller_big_qp_floats(p.B, p.U), stream));
        LAS_HIP_CHECK(hipMemsetAsync(p.flags, 0xFF, sizeof(unsigned) * speller_big_flag_words(p.U), stream));
    }
    {
        KernelTimer timer(TIMED_DECODE_FWD, stream);
        hipLaunchKernelGGL(speller_big_fwd_kernel, dim3(BG_WGS), dim3(BG_THREADS), big_fwd_smem(p.Tp), stream, a);
    }
    LAS_LAUNCH_CHECK();
    return LAS_OK;
}

}  // namespace las
