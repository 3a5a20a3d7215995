// Multi-utterance forward recurrence of a BiLSTM layer on the matrix pipe, second form: a WAVE-SPECIALISED pipeline.
//
// Same job, group geometry and arithmetic as pblstm_rec_mfma.hip (nn.LSTM(bidirectional=True) behind the reference's pBLSTMLayer,
// model/las_model.py:72-79,90; 16 sequences of one direction per group of G = H / 32 workgroups, exact three-way bf16 operand split,
// fp32 accumulation) — what changes is WHO does what inside a workgroup.  In the first form all 16 waves walk through the phases of a
// step together (tile in + split | MFMA | K reduction | cell), separated by workgroup barriers: every phase uses one resource (memory /
// matrix pipe / LDS / VALU) while the others idle, and two alternating batches only hide the hand-off latency, not the phases (2.6 us of
// local time per batch-step against 0.73 us of matrix-pipe time).  Here a workgroup has 12 waves with fixed roles that meet only
// through LDS flags (monotonic counters; no workgroup barrier in the step loop), and up to three batches of 16 sequences in flight:
//   * waves 0-3  M (one per SIMD): wave w multiplies N-tiles 2w, 2w+1 (32 gate rows = 8 hidden units x 4 gates) over the WHOLE K = H: its
//                   W_hh rows as three bf16 planes in 192 registers, 96 v_mfma_f32_16x16x32_bf16 per batch-step, no K split and hence no
//                   reduction; it then transposes the accumulators inside quads (a lane holds the four gates of its two cells — the
//                   ADJACENT units 2p, 2p+1), applies the cells (c in registers), publishes h and stages the stash.
//   * waves 4-5  L : poll the h tile of the ring (twelve 16-byte agent-scope loads per lane in flight) and copy it to this batch's planes.
//   * wave 6     F : feeds the cells: the input half of the gates (x W_ih^T + b, HBM) into LDS two steps ahead — loads only.
//   * wave 7     S : stores what the cells made (`out`, and gates / c / h_prev for the backward pass) from the LDS staging — stores only.
// What the first version of this file (three roles, fp32 ring) taught, kept here because each cost a measured microsecond:
//   * the ring carries h as its three bf16 PLANES, split once by the producing lane (11 VALU per pair) instead of by each of the group's
//     eight consumers (176 VALU per L lane and batch-step); a dword = one k-pair, written by one lane in one store, so it is either the
//     sentinel or complete, and the consumer's sentinel check is a running unsigned maximum;
//   * the accumulators START from the input half of the gates, read in the MFMA's own C layout: nothing is added after the product;
//   * a wave's loads and stores retire through ONE in-order counter with at most 63 operations outstanding: a wave that stores the
//     stash with single-word stores is throttled to ~30 stores per microsecond (1.5 us per batch-step) and every wait of a wave that
//     mixes loads with conditional stores becomes a vmcnt(0) — hence separate F (wide loads) and S (wide stores) waves, gate-major
//     staging rows, and pollers (L) that issue nothing else;
//   * pointer selects compile to exec-masked branches; 32-bit offset selects do not (lanes without a sequence publish to a dump word).
// Also measured and NOT kept (git history): (a) software-pipelining the M wave (product of the next batch-step in one basic block with the
// cell work of the pending one): at 247-256 registers the scheduler keeps the 96 MFMAs together and the cell's VALU work behind them,
// sched_group_barrier or not, and any forced interleave spills; (b) 12 waves with the K-halves on separate M waves (96 weight registers,
// two M waves per SIMD so that one wave's cells overlap the other's MFMAs): the accumulator exchange through LDS and three waves per
// SIMD cost more than the overlap gained (2.9 against 2.2 us per batch-step).
// All spins are bounded and report through the device error word.
#include "las_common.h"
#include "las_kernels.h"
#include "options.h"
#include "persist_common.h"
#include "rec_mfma_common.h"
#include <algorithm>
#include <type_traits>

namespace las {

namespace {

constexpr int RM2_THREADS = 512, RM2_NBMAX = 3, RM2_MW = 4;

template <int H>
struct RecMfma2 {
    static constexpr int G = H / RM_UW;                 // workgroups (CUs) per group
    static constexpr int KS = H / 32;                   // 32-deep k-steps (whole K per M wave)
    static constexpr int PLD = H / 2 + 4;               // LDS row stride (dwords = bf16 pairs) of one plane of the h tile
    static constexpr int PLANE = RM_NB * PLD;           // dwords of one plane
    // gate-major per-sequence rows [gate][32 units]: the F / S waves move 16 bytes (four units of one gate) per lane, the M waves single
    // words; GS / GLD chosen by exhaustive search over the three access patterns (conflict-free b128, 2-way b32)
    static constexpr int GS = RM_UW + 4;                // stride between the gates of a row
    static constexpr int GLD = 4 * GS + 4;              // row stride (floats) of a [16 sequences][4 gates][32 units] array
    static constexpr int G4 = RM_NB * GLD;              // floats of such an array
    static constexpr int SLD = 2 * GS + 4;              // row stride of the [16][(c, h)][32 units] array
    static constexpr int S2 = RM_NB * SLD;
    // per batch: three planes of the h tile | input half of the gates of the coming step (prel) | stash staging: gates (4) and (c, h) per cell
    static constexpr int PER_BATCH = 3 * PLANE + G4 + G4 + S2;
    static constexpr int NFLAGS = 128;
    static constexpr int LDS_FLOATS = RM2_NBMAX * PER_BATCH + NFLAGS;
    static_assert(H == 256, "register budget of the M waves (2 N-tiles x KS x 12 plane registers) and twelve tile float4 per L lane");
    static_assert(LDS_FLOATS * 4 <= 160 * 1024, "LDS of one workgroup");
};

// flag words (LDS, monotonic counters).  Per batch b (stride 24): step counters; XW / XR: batch-step sequence numbers
__device__ __forceinline__ int FL_PR(int b, int w) { return b * 24 + w; }           // L wave w stored the planes for step s: s            (2)
__device__ __forceinline__ int FL_PF(int b) { return b * 24 + 2; }                  // the F wave wrote prel for step s: s
__device__ __forceinline__ int FL_SF(int b) { return b * 24 + 3; }                  // the S wave took the staging of step s: s + 1
__device__ __forceinline__ int FL_MF(int b, int w) { return b * 24 + 8 + w; }       // M wave w finished reading the planes of step s: s    (4)
__device__ __forceinline__ int FL_CD(int b, int w) { return b * 24 + 16 + w; }      // M wave w applied its cells of step s, staging written: s + 1 (8)
constexpr int FL_INIT = 96, FL_XCD = 100;

typedef __attribute__((address_space(3))) unsigned rm2_lds_u32;

template <int N>
__device__ __forceinline__ bool rm2_wait(volatile unsigned* flags_generic, int first, unsigned target, unsigned* err, unsigned code) {
    volatile rm2_lds_u32* f = (volatile rm2_lds_u32*)flags_generic;
    unsigned spins = 0, tries = 0;
    for (;;) {
        bool ok = true;
#pragma unroll
        for (int k = 0; k < N; ++k) ok = ok && (int)(f[first + k] - target) >= 0;
        if (ok) break;
        __builtin_amdgcn_s_sleep(1);                    // (a tight loop of twelve pollers takes issue slots and LDS cycles from the M waves)
        if (++tries < 64u) continue;
        if (spin_expired(spins, err, code)) return false;
    }
    asm volatile("" ::: "memory");
    return true;
}
__device__ __forceinline__ void rm2_post(volatile unsigned* flags_generic, int idx, unsigned value, int lane) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // this wave's LDS traffic before the flag (LDS operations of a wave complete in order)
    if (lane == 0) ((volatile rm2_lds_u32*)flags_generic)[idx] = value;
}

// twelve 16-byte agent-scope loads in flight, one wait
__device__ __forceinline__ void rm2_ld12(const float* const (&p)[12], f32x4 (&v)[12]) {
    asm volatile("global_load_dwordx4 %0, %12, off sc1\n\t"
                 "global_load_dwordx4 %1, %13, off sc1\n\t"
                 "global_load_dwordx4 %2, %14, off sc1\n\t"
                 "global_load_dwordx4 %3, %15, off sc1\n\t"
                 "global_load_dwordx4 %4, %16, off sc1\n\t"
                 "global_load_dwordx4 %5, %17, off sc1\n\t"
                 "global_load_dwordx4 %6, %18, off sc1\n\t"
                 "global_load_dwordx4 %7, %19, off sc1\n\t"
                 "global_load_dwordx4 %8, %20, off sc1\n\t"
                 "global_load_dwordx4 %9, %21, off sc1\n\t"
                 "global_load_dwordx4 %10, %22, off sc1\n\t"
                 "global_load_dwordx4 %11, %23, off sc1\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7]),
                   "=&v"(v[8]), "=&v"(v[9]), "=&v"(v[10]), "=&v"(v[11])
                 : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]), "v"(p[5]), "v"(p[6]), "v"(p[7]), "v"(p[8]), "v"(p[9]), "v"(p[10]), "v"(p[11])
                 : "memory");
}

// 4 x 4 transpose inside a quad in two butterfly stages (8 DPP moves + 8 selects): lane g holds v[i] = X[i][g] on entry and
// X[g][k] in v[k] on exit.  Stage 1 transposes the 2 x 2 blocks (partner lane g ^ 1), stage 2 swaps the off-diagonal blocks (g ^ 2).
__device__ __forceinline__ void quad_transpose(f32x4& v, int g) {
    auto x1 = [](float x) { return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, true)); };   // quad_perm [1,0,3,2]
    auto x2 = [](float x) { return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0x4E, 0xF, 0xF, true)); };   // quad_perm [2,3,0,1]
    const bool odd = g & 1, hi = g & 2;
    f32x4 a;
    {
        const float p0 = x1(v[1]), p1 = x1(v[0]), p2 = x1(v[3]), p3 = x1(v[2]);      // the partner's element i ^ 1
        a[0] = odd ? p0 : v[0]; a[1] = odd ? v[1] : p1; a[2] = odd ? p2 : v[2]; a[3] = odd ? v[3] : p3;
    }
    {
        const float p0 = x2(a[2]), p1 = x2(a[3]), p2 = x2(a[0]), p3 = x2(a[1]);      // the partner's element i ^ 2
        v[0] = hi ? p0 : a[0]; v[1] = hi ? p1 : a[1]; v[2] = hi ? a[2] : p2; v[3] = hi ? a[3] : p3;
    }
}

// phase stamps (option REC_TRACE; tools/ubench_rec_mfma.py TRACE=1): workgroup 0, batch 0, 100 MHz wall clock
#define RM2_STAMP(k) do { if (a.trace && blockIdx.x == 0 && bi == 0 && lane == 0 && step < 256) a.idbuf[4096 + step * 8 + (k)] = wall_clock64(); } while (0)

template <int H, bool STASH>
__global__ __launch_bounds__(RM2_THREADS) void rec_fwd_mfma2_kernel(RecMfmaArgs a) {
    using C = RecMfma2<H>;
    constexpr int G = C::G, KS = C::KS, PLD = C::PLD, PLANE = C::PLANE, GLD = C::GLD, SLD = C::SLD, GS = C::GS, NBM = RM2_NBMAX;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    auto planes_of = [&](int b) { return reinterpret_cast<unsigned*>(smem + b * C::PER_BATCH); };     // [plane][16 sequences][PLD] bf16 pairs
    auto prel_of = [&](int b) { return smem + b * C::PER_BATCH + 3 * PLANE; };                        // [16][GLD]: x W_ih^T + b of the coming step
    auto stg4_of = [&](int b) { return smem + b * C::PER_BATCH + 3 * PLANE + C::G4; };               // [16][GLD]: i, f, g, o after activation
    auto stg2_of = [&](int b) { return smem + b * C::PER_BATCH + 3 * PLANE + 2 * C::G4; };           // [16][SLD]: c_t, h_t
    volatile unsigned* flags = reinterpret_cast<volatile unsigned*>(smem + NBM * C::PER_BATCH);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int T = a.T, B = a.B;
    const int nbat = a.nbat;
    const int per_group = nbat * RM_NB;
    const int ngb = (a.Bc + per_group - 1) / per_group;                     // groups per direction
    const int ngroups = 2 * ngb;
    int group, member;
    {
        // XCD-local groups under round-robin dispatch (block b -> XCD b % 8, verified at run time below): the grid is padded to whole
        // rounds of eight groups and the workgroups of the missing groups leave at once
        const int bid = blockIdx.x, q = bid >> 3;
        member = q % G; group = (q / G) * 8 + (bid & 7);
        if (group >= ngroups) return;
    }
    const int dir = group >= ngb ? 1 : 0;
    const float* __restrict__ w_hh = dir ? a.w_hh_r : a.w_hh_f;
    const int u0 = member * RM_UW;
    int bbase[NBM], nvalid[NBM];
#pragma unroll
    for (int bi = 0; bi < NBM; ++bi) {
        bbase[bi] = a.b0 + ((group - dir * ngb) * nbat + bi) * RM_NB;
        nvalid[bi] = bi < nbat ? max(0, min(RM_NB, a.b0 + a.Bc - bbase[bi])) : 0;
    }
    if (tid < C::NFLAGS) flags[tid] = 0u;
    __syncthreads();
    const bool l2x = !a.force_agent && rm_same_xcd<G>(a.idbuf + (size_t)group * 32, member, a.err, flags + FL_XCD);
    // The ring: [batch][slot][plane][16 sequences][H / 2 dwords], a dword = the k-adjacent pair (unit 2p, unit 2p + 1) of one bf16 plane
    constexpr int RROW = H / 2, RPLANE = RM_NB * RROW, RSLOT = 3 * RPLANE;                       // dwords
    constexpr int RGROUP = NBM * 4 * RSLOT + 1024;      // a group's slots + dump words (where lanes without a sequence "publish")
    unsigned* ring = reinterpret_cast<unsigned*>(a.ring) + (size_t)group * RGROUP;

    if (wave < RM2_MW) {
        // ================================================================================================ M: the recurrent product and the cells
        // N-tile j of wave w, tile column c <-> unit 8 w + 2 (c / 4) + j, gate c % 4: after the quad transpose a lane holds the cells of the
        // ADJACENT units 2 p, 2 p + 1 (p = 4 w + uq) of sequence 4 kq + g4 — one k-pair dword per plane to publish
        const int w = wave;
        PsPlanes<8> Wp[2][KS];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int r16 = lane & 15, kq = lane >> 4;
            const long wrow = (long)(r16 & 3) * H + u0 + 8 * w + 2 * (r16 >> 2) + j;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const float* src = w_hh + wrow * H + ks * 32 + kq * 8;
                const f32x4 w0 = ld4p(src), w1 = ld4p(src + 4);
                const float v[8] = {w0[0], w0[1], w0[2], w0[3], w1[0], w1[1], w1[2], w1[3]};
                Wp[j][ks] = ps_split<8>(v);
            }
        }
        float c[NBM][2];
#pragma unroll
        for (int bi = 0; bi < NBM; ++bi) c[bi][0] = c[bi][1] = 0.f;
        if (!rm2_wait<1>(flags, FL_INIT, 1u, a.err, 0xDEAD0047u)) return;   // the F wave has put step 0's input half into LDS
        const unsigned dump_off = (unsigned)(NBM * 4 * RSLOT) + tid;         // (32-bit offsets from the group's ring: selects, not branches)
        auto m_role = [&](auto L2XC) {
            constexpr bool L2X = decltype(L2XC)::value;                      // (compile time: a run-time branch would split the cell block)
            for (int step = 0; step < T; ++step) {
#pragma unroll
                for (int bi = 0; bi < NBM; ++bi) {
                    if (nvalid[bi] == 0) continue;                          // (uniform)
                    // lane geometry re-derived per batch-step behind opaque(): as loop invariants the LDS / ring offsets of three batches are
                    // hoisted out of the step loop and spill (scratch reloads inside the cell phase, on the chain)
                    const int ln = (int)opaque((unsigned)lane);
                    const int r16 = ln & 15, kq = ln >> 4, g4 = ln & 3, uq = (ln >> 2) & 3;
                    const int cs = 4 * kq + g4, ul0 = 8 * w + 2 * uq;       // this lane's sequence and units ul0, ul0 + 1
                    if (wave == 0) RM2_STAMP(6);
                    // the accumulators START from the input half of this step's gates (x W_ih^T + b; the F wave put it into LDS a step ago) in
                    // the MFMA's own C layout — D[row 4 kq + i][column 4 uq + g]; the staging rows of the previous step have long been taken
                    if (!rm2_wait<2>(flags, FL_PF(bi), (unsigned)step, a.err, 0xDEAD0042u)) return;         // PF and SF are adjacent words
                    f32x4 acc[2];                                             // (two interleaved chains; four accumulators spilled)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int i = 0; i < 4; ++i) acc[j][i] = prel_of(bi)[(4 * kq + i) * GLD + g4 * GS + ul0 + j];
                    if (wave == 0) RM2_STAMP(7);
                    if (step > 0) {
                        if (!rm2_wait<2>(flags, FL_PR(bi, 0), (unsigned)step, a.err, 0xDEAD0041u)) return;
                        if (wave == 0) RM2_STAMP(3);
                        const unsigned* ar = planes_of(bi) + r16 * PLD + kq * 4;
                        auto load_a = [&](int ks) {
                            PsPlanes<8> A;
#pragma unroll
                            for (int pl = 0; pl < 3; ++pl) {
                                const ps_u32x4 q4 = *reinterpret_cast<const ps_u32x4*>(ar + pl * PLANE + ks * 16);
                                A.p[pl][0] = q4[0]; A.p[pl][1] = q4[1]; A.p[pl][2] = q4[2]; A.p[pl][3] = q4[3];
                            }
                            return A;
                        };
                        PsPlanes<8> A = load_a(0);
#pragma unroll
                        for (int ks = 0; ks < KS; ++ks) {
                            PsPlanes<8> An = A;
                            if (ks + 1 < KS) An = load_a(ks + 1);
                            acc[0] = ps_mfma6<8>(A, Wp[0][ks], acc[0]);
                            acc[1] = ps_mfma6<8>(A, Wp[1][ks], acc[1]);
                            A = An;
                        }
                        asm volatile("" ::: "memory");
                        rm2_post(flags, FL_MF(bi, wave), (unsigned)step, lane);
                    }
                    if (wave == 0) RM2_STAMP(4);
                    // ---- the two cells: (i, f, g, o) of (sequence cs, unit ul0 + j) after the quad transpose; stage the stash for the S wave
                    float hh[2];
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        f32x4 sg = acc[j];
                        quad_transpose(sg, g4);
                        const float ig = sigmoidf_acc(sg[0]);
                        const float fg = sigmoidf_acc(sg[1]);
                        const float gg = tanhf_acc(sg[2]);
                        const float og = sigmoidf_acc(sg[3]);
                        c[bi][j] = fg * c[bi][j] + ig * gg;
                        hh[j] = og * tanhf_acc(c[bi][j]);
                        float* s4 = stg4_of(bi) + cs * GLD + ul0 + j;
                        s4[0] = ig; s4[GS] = fg; s4[2 * GS] = gg; s4[3 * GS] = og;
                        float* s2 = stg2_of(bi) + cs * SLD + ul0 + j;
                        s2[0] = c[bi][j]; s2[GS] = hh[j];
                    }
                    // publish: the pair's three bf16 plane dwords (a NaN is canonicalised first: its terms must not look like the sentinel); the
                    // words of slot step + 2 go back to the sentinel (their consumers have read them: they have published h of step - 1 since,
                    // which this workgroup consumed before this cell ran)
                    unsigned p3[3];
                    ps_split_pair(__uint_as_float(pub_bits(hh[0])), __uint_as_float(pub_bits(hh[1])), p3[0], p3[1], p3[2]);
                    const bool pub = cs < nvalid[bi];
                    const bool reset = pub && step + 2 < T;
                    const unsigned off = (unsigned)(cs * RROW + ((u0 + ul0) >> 1));
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) {
                        const unsigned oh = pub ? (unsigned)((bi * 4 + (step & 3)) * RSLOT + pl * RPLANE) + off : dump_off;
                        const unsigned os = reset ? (unsigned)((bi * 4 + ((step + 2) & 3)) * RSLOT + pl * RPLANE) + off : dump_off;
                        if constexpr (L2X) {
                            __hip_atomic_store(ring + oh, p3[pl], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            __hip_atomic_store(ring + os, PS_SENT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        } else {
                            __hip_atomic_store(ring + oh, p3[pl], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            __hip_atomic_store(ring + os, PS_SENT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                    }
                    rm2_post(flags, FL_CD(bi, wave), (unsigned)(step + 1), lane);
                    if (wave == 0) RM2_STAMP(5);
                }
            }
        };
        if (l2x) m_role(std::true_type{}); else m_role(std::false_type{});
    } else if (wave < RM2_MW + 2) {
        // ================================================================================================ L: the h planes, ring -> LDS
        const int ll = tid - RM2_MW * 64, lw = wave - RM2_MW;
        for (int step = 1; step < T; ++step) {
#pragma unroll
            for (int bi = 0; bi < NBM; ++bi) {
                if (nvalid[bi] == 0) continue;
                // the plane buffer of this batch is free once every M wave has read the previous step's planes
                if (!rm2_wait<RM2_MW>(flags, FL_MF(bi, 0), (unsigned)(step - 1), a.err, 0xDEAD0043u)) return;
                if (lw == 0) RM2_STAMP(0);
                const unsigned* slot = ring + (size_t)(bi * 4 + ((step - 1) & 3)) * RSLOT;
                // 3 planes x 16 rows x 32 float4 = 1536 float4: twelve per lane; rows without a sequence are never written (row 0 stands in)
                const float* src[12];
#pragma unroll
                for (int j = 0; j < 12; ++j) {
                    const int f = ll + 128 * j, pl = f >> 9, row = (f >> 5) & 15, c4 = f & 31;
                    src[j] = reinterpret_cast<const float*>(slot) + opaque((unsigned)(pl * RPLANE + (row < nvalid[bi] ? row : 0) * RROW + c4 * 4));
                }
                // agent-scope loads in both placements: ring slots are reused every four steps, an ordinary load could hit a stale copy of
                // the line in this CU's L1 or a foreign L2 (same XCD: the producers' plain stores sit in the shared L2, read past the L1)
                f32x4 v[12];
                unsigned spins = 0;
                for (;;) {
                    rm2_ld12(src, v);
                    unsigned m = 0u;            // the sentinel is the largest unsigned value: one running maximum finds it
#pragma unroll
                    for (int j = 0; j < 12; ++j)
                        m = max(max(m, max(__float_as_uint(v[j][0]), __float_as_uint(v[j][1]))), max(__float_as_uint(v[j][2]), __float_as_uint(v[j][3])));
                    if (!__any(m == PS_SENT)) break;
                    if (spin_expired(spins, a.err, 0xDEAD0044u)) return;
                }
                if (lw == 0) RM2_STAMP(1);
                unsigned* dstb = planes_of(bi);
#pragma unroll
                for (int j = 0; j < 12; ++j) {
                    const int f = ll + 128 * j, pl = f >> 9, row = (f >> 5) & 15, c4 = f & 31;
                    *reinterpret_cast<f32x4*>(dstb + pl * PLANE + row * PLD + c4 * 4) = v[j];
                }
                rm2_post(flags, FL_PR(bi, lw), (unsigned)step, lane);
                if (lw == 0) RM2_STAMP(2);
            }
        }
    } else if (wave == RM2_MW + 2) {
        // ================================================================================================ F: feed the cells
        // Item k of a lane: sequence 2 k + hs, gate g, units 4 quad .. +3 — a wave-uniform base plus a small lane offset; eight wide loads
        // per batch-step stay far below the 63 outstanding vector-memory operations of a wave.
        const int hs = lane >> 5, g = (lane & 31) >> 3, quad = lane & 7;
        f32x4 pn[NBM][8];
        const int t0 = dir ? T - 1 : 0, t1 = T > 1 ? (dir ? T - 2 : 1) : t0;
        const unsigned lane_g = (unsigned)hs * (unsigned)T * 4u * H + g * H + quad * 4;      // lane part of an index into the (2, B, T, 4H) gate array
        const int lds_o = hs * GLD + g * GS + quad * 4;
#pragma unroll
        for (int bi = 0; bi < NBM; ++bi)
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const bool live = 2 * k + hs < nvalid[bi];
                const float* gb = a.gates + ((long)(dir * B + bbase[bi] + 2 * k) * T) * 4 * H + u0;      // (uniform; lanes without a sequence do not load)
                f32x4 p0 = {0.f, 0.f, 0.f, 0.f};
                pn[bi][k] = p0;
                if (live) {
                    p0 = ld4p(gb + (long)t0 * 4 * H + lane_g);
                    pn[bi][k] = ld4p(gb + (long)t1 * 4 * H + lane_g);
                }
                if (nvalid[bi] != 0) *reinterpret_cast<f32x4*>(prel_of(bi) + 2 * k * GLD + lds_o) = p0;       // step 0's values
            }
        rm2_post(flags, FL_INIT, 1u, lane);                                 // (FL_PF >= 0 holds from the start: step 0 has its own flag)
        for (int step = 0; step < T; ++step) {
            const int t = dir ? T - 1 - step : step;
            const int tn2 = step + 2 < T ? (dir ? t - 2 : t + 2) : t;
#pragma unroll
            for (int bi = 0; bi < NBM; ++bi) {
                if (nvalid[bi] == 0) continue;
                // the M waves have read this step's input half (their cells of this step are applied)
                if (!rm2_wait<RM2_MW>(flags, FL_CD(bi, 0), (unsigned)(step + 1), a.err, 0xDEAD0045u)) return;
#pragma unroll
                for (int k = 0; k < 8; ++k) *reinterpret_cast<f32x4*>(prel_of(bi) + 2 * k * GLD + lds_o) = pn[bi][k];
                rm2_post(flags, FL_PF(bi), (unsigned)(step + 1), lane);
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const float* gb = a.gates + ((long)(dir * B + bbase[bi] + 2 * k) * T + tn2) * 4 * H + u0;
                    if (2 * k + hs < nvalid[bi]) pn[bi][k] = ld4p(gb + lane_g);
                }
            }
        }
    } else {
        // ================================================================================================ S: store what the cells made
        // `out` and the stash of the backward pass from the LDS staging arrays, 16 bytes per lane (14 wide stores per batch-step).
        // Stores only: nothing here ever waits for memory.
        const int hs = lane >> 5, g = (lane & 31) >> 3, quad = lane & 7, s8 = lane >> 3;
        f32x4 hlast[NBM][2];
#pragma unroll
        for (int bi = 0; bi < NBM; ++bi) hlast[bi][0] = hlast[bi][1] = f32x4{0.f, 0.f, 0.f, 0.f};
        const unsigned lane_g = (unsigned)hs * (unsigned)T * 4u * H + g * H + quad * 4;      // gate items: sequence 2 k + hs, gate g, units 4 quad .. +3
        const int lds_g = hs * GLD + g * GS + quad * 4;
        const unsigned lane_h = (unsigned)s8 * (unsigned)T * H + quad * 4;                  // c / h items: sequence 8 k + s8, units 4 quad .. +3
        const unsigned lane_o = (unsigned)s8 * (unsigned)T * 2u * H + quad * 4;
        const int lds_h = s8 * SLD + quad * 4;
        for (int step = 0; step < T; ++step) {
            const int t = dir ? T - 1 - step : step;
#pragma unroll
            for (int bi = 0; bi < NBM; ++bi) {
                if (nvalid[bi] == 0) continue;
                if (!rm2_wait<RM2_MW>(flags, FL_CD(bi, 0), (unsigned)(step + 1), a.err, 0xDEAD0046u)) return;
                f32x4 gq[8], cq[2], hq[2];
#pragma unroll
                for (int k = 0; k < 8; ++k) gq[k] = *reinterpret_cast<const f32x4*>(stg4_of(bi) + 2 * k * GLD + lds_g);
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    cq[k] = *reinterpret_cast<const f32x4*>(stg2_of(bi) + 8 * k * SLD + lds_h);
                    hq[k] = *reinterpret_cast<const f32x4*>(stg2_of(bi) + 8 * k * SLD + GS + lds_h);
                }
                rm2_post(flags, FL_SF(bi), (unsigned)(step + 1), lane);
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    if (8 * k + s8 < nvalid[bi]) {
                        const long seq = (long)(dir * B + bbase[bi] + 8 * k);               // (uniform; this lane's sequence is seq + s8)
                        *reinterpret_cast<f32x4*>(a.out + ((long)(bbase[bi] + 8 * k) * T + t) * 2 * H + dir * H + u0 + lane_o) = hq[k];
                        if (STASH) {
                            *reinterpret_cast<f32x4*>(a.hprev + (seq * T + t) * H + u0 + lane_h) = hlast[bi][k];
                            *reinterpret_cast<f32x4*>(a.cbuf + (seq * T + t) * H + u0 + lane_h) = cq[k];
                        }
                    }
                    hlast[bi][k] = hq[k];
                }
                if (STASH) {
#pragma unroll
                    for (int k = 0; k < 8; ++k)
                        if (2 * k + hs < nvalid[bi])
                            *reinterpret_cast<f32x4*>(a.gates + ((long)(dir * B + bbase[bi] + 2 * k) * T + t) * 4 * H + u0 + lane_g) = gq[k];
                }
            }
        }
    }
}

}  // namespace

int rec_fwd_mfma2(float* gates, const float* w_hh_f, const float* w_hh_r, float* out, float* cbuf, float* hprev, int B, int T, int H,
                  int stash, unsigned long long* xbuf, unsigned* err, hipStream_t stream) {
    LAS_REQUIRE(H == 256, "rec_fwd_mfma2 shape");
    LAS_REQUIRE(err != nullptr && xbuf != nullptr && (!stash || (cbuf && hprev)), "rec_fwd_mfma2 buffers");
    using C = RecMfma2<256>;
    int dev = 0, cus = 0;
    LAS_HIP_CHECK(hipGetDevice(&dev));
    LAS_HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    const int groups_max = std::min(16, cus / (2 * C::G));            // groups per direction that are resident at once (ring and id slots: 32 groups)
    if (groups_max < 1) return fail(LAS_ERR_UNSUPPORTED, "rec_fwd_mfma2: %s%ld compute units are too few", "", (long)cus);
    const size_t smem = sizeof(float) * C::LDS_FLOATS;
    if (stash) LAS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&rec_fwd_mfma2_kernel<256, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    else LAS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&rec_fwd_mfma2_kernel<256, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    // one batch of 16 sequences per group while that covers the launch; beyond it two or three batches per group, pipelined through the roles
    const int nbat = std::min(RM2_NBMAX, std::max(1, (B + groups_max * RM_NB - 1) / (groups_max * RM_NB)));
    // equal launches rather than full ones and a remainder (B = 2048: 3 x 688 utterances instead of 768 + 768 + 512)
    const int chunk_max = groups_max * RM_NB * nbat;
    const int nlaunch = (B + chunk_max - 1) / chunk_max;
    const int chunk = std::min(chunk_max, ((B + nlaunch - 1) / nlaunch + RM_NB - 1) / RM_NB * RM_NB);
    float* ring = reinterpret_cast<float*>(reinterpret_cast<char*>(xbuf) + REC_MFMA_RING_OFFSET);       // (rec_xbuf_bytes makes room for it)
    constexpr size_t ring_group_dwords = (size_t)RM2_NBMAX * 4 * 3 * RM_NB * (256 / 2) + 1024;        // bf16 planes + dump words (= RGROUP in the kernel)
    for (int b0 = 0; b0 < B; b0 += chunk) {
        const int Bc = std::min(chunk, B - b0);
        const int ngroups = 2 * ((Bc + nbat * RM_NB - 1) / (nbat * RM_NB));
        const int grid = (ngroups + 7) / 8 * 8 * C::G;        // padded to whole rounds of eight groups (XCD-local placement)
        LAS_HIP_CHECK(hipMemsetAsync(xbuf, 0, sizeof(unsigned long long) * 32 * (size_t)ngroups, stream));
        LAS_HIP_CHECK(hipMemsetAsync(ring, 0xFF, sizeof(float) * (size_t)ngroups * ring_group_dwords, stream));
        RecMfmaArgs a{gates, w_hh_f, w_hh_r, out, cbuf, hprev, B, T, b0, Bc, err, nbat, xbuf, (int)opt_get(OPT_REC_AGENT_HANDOFF), ring, (int)opt_get(OPT_REC_TRACE)};
        if (stash) {
            if (!persistent_launch_fits(rec_fwd_mfma2_kernel<256, true>, RM2_THREADS, smem, grid))
                return fail(LAS_ERR_UNSUPPORTED, "rec_fwd_mfma2: %s%ld workgroups cannot all be resident", "", (long)grid);
            hipLaunchKernelGGL((rec_fwd_mfma2_kernel<256, true>), dim3(grid), dim3(RM2_THREADS), smem, stream, a);
        } else {
            if (!persistent_launch_fits(rec_fwd_mfma2_kernel<256, false>, RM2_THREADS, smem, grid))
                return fail(LAS_ERR_UNSUPPORTED, "rec_fwd_mfma2: %s%ld workgroups cannot all be resident", "", (long)grid);
            hipLaunchKernelGGL((rec_fwd_mfma2_kernel<256, false>), dim3(grid), dim3(RM2_THREADS), smem, stream, a);
        }
        LAS_LAUNCH_CHECK();
        path_note(PATH_REC_FWD, "rec_fwd_mfma2");
    }
    return LAS_OK;
}

}  // namespace las
