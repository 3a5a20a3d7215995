// Multi-utterance forward recurrence of a BiLSTM layer on the MATRIX pipe (gfx950): the large-batch form of pblstm_rec.hip.
//
// Replaces the recurrent half of nn.LSTM(bidirectional=True) behind the reference's pBLSTMLayer (model/las_model.py:72-79,90)
// when a launch carries enough utterances to fill MFMA tiles (B >= 64 at H = 256); cell equations, gate order, stash layout and
// the reverse direction are those of pblstm_rec.hip (rec_fwd_generic is the executable specification).
//
// Why: with one utterance per group the recurrent product is a mat-vec and lives on the VALU (64 FMAs per lane and step).  The
// multi-utterance kernels of pblstm_rec.hip amortise the hand-off over NB utterances but still do NB separate VALU mat-vecs
// (21 ns per utterance-step at B = 512 against a 3.3 ns FMA floor).  Sixteen utterances of one direction ARE a 16-row MFMA tile:
//   * a GROUP of G = H / 32 workgroups (one per CU, 1024 threads) owns 16 sequences for all T steps; workgroup m owns hidden units
//     [32 m, 32 m + 32): its 128 gate rows of W_hh (row order unit*4 + gate) are split ONCE into three bf16 planes and stay in
//     registers (48 per lane at H = 256: wave = (pair of N-tiles of 16 gate rows, K quarter), two 32-deep k-steps each);
//   * per step: h_{t-1} of the 16 sequences (16 x H floats, 16 KB at H = 256) is pulled once per workgroup (one float4 per lane),
//     split ONCE into its three bf16 planes on the way into LDS (the exact three-way split of persist_common.h: fp32-faithful;
//     splitting in the multiplying waves repeated the same VALU work in every N-tile's wave: 1.8 us per step), every wave reads its
//     16 x 32 operand blocks as planes and issues six v_mfma_f32_16x16x32_bf16 per block and N-tile, the four K quarters meet in
//     LDS, 512 lanes apply the cell (c in a register) and publish h_t;
//   * hand-off = a four-slot ring per (group, batch), slot = step % 4, [16 sequences][H]: sentinel-prefilled once, producers store h_t
//     (a half wave writes one whole 128-byte line: 32 units of one sequence) and reset their words of slot step + 2 themselves;
//     consumers poll their own float4 of the tile (same XCD) or watch one dword per producer first (otherwise), always with
//     agent-scope loads, and check every word against the sentinel (persist_common.h).  `out` receives h_t as an ordinary store.
// All spins are bounded and report through the device error word.
// From two batches per group (B > 256) the forward runs as the wave-specialised pipeline of pblstm_rec_mfma2.hip instead (same groups,
// same arithmetic; rec_fwd_mfma below dispatches); the backward kernel of this file serves every large batch.
#include "las_common.h"
#include "las_kernels.h"
#include "options.h"
#include "persist_common.h"
#include "rec_mfma_common.h"
#include <algorithm>

namespace las {

namespace {

template <int H>
struct RecMfma {
    static constexpr int G = H / RM_UW;                 // workgroups (CUs) per group
    static constexpr int KQ = H / 4;                    // K range of a wave (four K quarters)
    static constexpr int KS = KQ / 32;                  // 32-deep k-steps per wave
    static constexpr int PLD = H / 2 + 4;               // LDS row stride (dwords = bf16 pairs) of one plane of the h tile
    static constexpr int PLANE = RM_NB * PLD;           // dwords of one plane
    static constexpr int RLD = 20;                      // row stride of a partial 16x16 tile
    static constexpr int RED = 4 * 8 * 16 * RLD;        // [K quarter][N-tile][16 rows][RLD]
    static constexpr int PREL = 2 * 512 * 4;            // pre-activations of the coming cell phase: [batch][cell lane][gate]
    static constexpr int LDS_FLOATS = 3 * PLANE + RED + 16 + PREL;
    static_assert(H == 256, "register budget (2 N-tiles x KS x 12 plane registers per lane) and one tile float4 per lane");
};

#ifdef RM_TRACE      // debug build: phase stamps of workgroup 0 into the id buffer behind the id slots (tools/ubench_rec_mfma.py TRACE=1)
#define RM_STAMP(k) do { if (blockIdx.x == 0 && threadIdx.x == 0 && step < 256) a.idbuf[4096 + step * 8 + (k)] = wall_clock64(); } while (0)
#else
#define RM_STAMP(k) do { } while (0)
#endif

template <int H, bool STASH>
__global__ __launch_bounds__(RM_THREADS) void rec_fwd_mfma_kernel(RecMfmaArgs a) {
    using C = RecMfma<H>;
    constexpr int G = C::G, KS = C::KS, PLD = C::PLD, PLANE = C::PLANE, RLD = C::RLD;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    unsigned* hp3 = reinterpret_cast<unsigned*>(smem);  // [plane][16 sequences][PLD] bf16 pairs
    float* red = smem + 3 * PLANE;                      // [K quarter][N-tile][16 rows][RLD]
    volatile unsigned* cflags = reinterpret_cast<volatile unsigned*>(red + C::RED);
    float* prel = red + C::RED + 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int T = a.T, B = a.B;
    // a group steps NBAT batches of 16 sequences alternately (a.nbat = 2 when the launch carries more sequences than one batch per
    // group covers): while one batch's h travels between the CUs, the other batch is multiplied — the hand-off leaves the chain
    const int nbat = a.nbat;
    const int per_group = nbat * RM_NB;
    const int ngb = (a.Bc + per_group - 1) / per_group; // groups per direction
    const int ngroups = 2 * ngb;
    // group / member: XCD-local groups under round-robin dispatch (block b -> XCD b % 8) when the group count allows it
    int group, member;
    {
        const int bid = blockIdx.x;
        // XCD-local groups under round-robin dispatch (block b -> XCD b % 8, verified at run time below): the grid is padded to whole
        // rounds of eight groups and the workgroups of the missing groups leave at once
        const int q = bid >> 3;
        member = q % G; group = (q / G) * 8 + (bid & 7);
        if (group >= ngroups) return;
    }
    const int dir = group >= ngb ? 1 : 0;
    const float* __restrict__ w_hh = dir ? a.w_hh_r : a.w_hh_f;
    const int u0 = member * RM_UW;

    // ---- resident weights: wave = (N-tile pair np, K quarter kq4); tile column c -> unit 4 nt + c / 4, gate c % 4
    const int np = wave & 3, kq4 = wave >> 2;
    const int r16 = lane & 15, kq = lane >> 4;
    PsPlanes<8> Wp[2][KS];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int nt = 2 * np + j;
        const long wrow = (long)(r16 & 3) * H + u0 + 4 * nt + (r16 >> 2);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const float* src = w_hh + wrow * H + kq4 * C::KQ + ks * 32 + kq * 8;
            const f32x4 w0 = ld4p(src), w1 = ld4p(src + 4);
            const float v[8] = {w0[0], w0[1], w0[2], w0[3], w1[0], w1[1], w1[2], w1[3]};
            Wp[j][ks] = ps_split<8>(v);
        }
    }
    // ---- per batch: cell lanes (sequence cs, unit cu) for tid < 512; a half wave = the 32 units of one sequence = one 128-byte
    //      line of `out`
    //      The input half of the gates (x W_ih^T + b, written by the projection GEMM: a long-latency load per step) reaches the cell lanes
    //      through LDS: lane 512 + i of waves 8-15 loads what cell lane i needs a whole step ahead.  A wave's loads and stores retire
    //      through ONE in-order counter, so in the cell lanes those loads stood between the tile poll and its completion (the poll
    //      waited for them and for the stash stores' acknowledgements: 0.65 us per step on the chain).
    const int cs = (tid & 511) >> 5, cu = tid & 31;
    int bbase[2], nvalid[2];
    bool cell[2], feeder[2];
    int seqr[2];                                        // row base in the (2, B, T, *) stash arrays: 32-bit, the addresses are re-derived per
                                                        // use behind opaque() — as hoisted 64-bit loop invariants they pushed a pre-activation
                                                        // into scratch, whose spill WAITED for its load inside the cell phase (0.5 us per step)
    float c[2] = {0.f, 0.f}, hlast[2] = {0.f, 0.f};
    float pre[2][4];
#pragma unroll
    for (int bi = 0; bi < 2; ++bi) {
        bbase[bi] = a.b0 + ((group - dir * ngb) * nbat + bi) * RM_NB;
        nvalid[bi] = bi < nbat ? max(0, min(RM_NB, a.b0 + a.Bc - bbase[bi])) : 0;
        cell[bi] = tid < RM_NB * RM_UW && cs < nvalid[bi];
        feeder[bi] = tid >= RM_NB * RM_UW && cs < nvalid[bi];
        const int cb = bbase[bi] + (cs < nvalid[bi] ? cs : 0);
        seqr[bi] = (dir * B + (cs < nvalid[bi] ? cb : a.b0)) * T;
#pragma unroll
        for (int g = 0; g < 4; ++g) pre[bi][g] = 0.f;
        if (feeder[bi]) {       // step 0's values straight into LDS, step 1's into registers (in flight)
            const float* gb = a.gates + (long)seqr[bi] * 4 * H + u0 + cu;
            const int t0 = dir ? T - 1 : 0, t1 = T > 1 ? (dir ? T - 2 : 1) : t0;
            f32x4 p0;
#pragma unroll
            for (int g = 0; g < 4; ++g) p0[g] = gb[(long)t0 * 4 * H + g * H];
            *reinterpret_cast<f32x4*>(prel + (bi * 512 + (tid - 512)) * 4) = p0;
#pragma unroll
            for (int g = 0; g < 4; ++g) pre[bi][g] = gb[(long)t1 * 4 * H + g * H];
        }
    }
    // ---- tile loader: lane -> (row lr, float4 column lc) of the 16 x H tile: exactly one float4 per lane at H = 256
    constexpr int F4_PER_ROW = H / 4, TILE_F4 = RM_NB * F4_PER_ROW;
    unsigned cep = 0;
    if (tid < 4) cflags[tid] = 0u;
    lds_barrier();
    const bool l2x = !a.force_agent && rm_same_xcd<G>(a.idbuf + (size_t)group * 32, member, a.err, cflags + 4);

    static_assert(TILE_F4 == RM_THREADS, "one float4 of the tile per lane");
    const int tlr = tid / F4_PER_ROW, tlc = tid % F4_PER_ROW;
    // h travels through a four-slot ring per (group, batch) — slot = step % 4, [16 sequences][H] — not through `out`: a slot is reset to
    // the sentinel by its producer lanes two steps before its next use (the producer has then consumed every consumer's NEXT
    // publication, so they have finished reading it), and the host fills 128 KB per group instead of the whole output buffer
    // (105 MB at B = 128, 420 MB at B = 512: 2-4 % of the launch)
    constexpr int RSLOT = RM_NB * H;
    float* ring = a.ring + (size_t)group * 2 * 4 * RSLOT;
    auto tile_src = [&](int bi, int sprev) {
        return ring + (size_t)(bi * 4 + (sprev & 3)) * RSLOT + opaque((unsigned)((tlr < nvalid[bi] ? tlr : 0) * H + tlc * 4));      // (rows without a sequence are never written)
    };
    unsigned pf[4] = {0u, 0u, 0u, 0u};
    bool have_pf = false;

    for (int step = 0; step < T; ++step) {
        const int t = dir ? T - 1 - step : step;
#pragma unroll
        for (int bi = 0; bi < 2; ++bi) {
            if (nvalid[bi] == 0) continue;          // (uniform) no second batch, or an empty one
            f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
            if (bi == 0) RM_STAMP(0);
            if (step > 0) {
                // ---- h_{t-1} of the 16 sequences -> three bf16 planes in LDS
                // Same XCD: every lane polls its own float4 straight away (the polls hit this XCD's L2: no fabric traffic, and one round
                // trip less than canary-then-load).  Otherwise: canary first — wave 0, lane m watches producer m's last unit of the last
                // valid row — so that the memory system does not carry 16 KB of polls per workgroup and round.
                if (!l2x) {
                    const int lrow = nvalid[bi] - 1;
                    const unsigned* cp = reinterpret_cast<const unsigned*>(ring + (size_t)(bi * 4 + ((step - 1) & 3)) * RSLOT + lrow * H) + (lane < G ? lane * RM_UW + RM_UW - 1 : 0);
                    wg_canary_wait(cflags, ++cep, 1, wave, lane, cp, lane < G, a.err, 0xDEAD0031u);
                }
                if (bi == 0) RM_STAMP(1);
                {
                    // this lane's float4 of the tile: prefetched during the other batch's half-round when two batches alternate
                    f32x4 v;
                    const float* src = tile_src(bi, step - 1);
                    if (have_pf) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) v[k] = __uint_as_float(pf[k]);
                    } else {
                        // agent-scope loads in both modes: ring slots are REUSED every four steps, so an ordinary load could hit a stale copy
                        // of the line in this CU's L1 or this XCD's L2 (same XCD: the producers' plain stores sit in the shared L2, read past
                        // the L1 only; otherwise the canary above has seen the write-through stores land)
                        v = ld4_agent(src);
                    }
                    if (has_sentinel(v)) {          // raced ahead of a producer's lines: re-read past the L1 / L2
                        unsigned spins = 0;
                        for (;;) {
                            v = ld4_agent(src);
                            if (!has_sentinel(v)) break;
                            if (spin_expired(spins, a.err, 0xDEAD0032u)) break;
                        }
                    }
                    unsigned p0[3], p1[3];
                    ps_split_pair(v[0], v[1], p0[0], p0[1], p0[2]);
                    ps_split_pair(v[2], v[3], p1[0], p1[1], p1[2]);
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) {
                        ps_u32x2 w2 = {p0[pl], p1[pl]};
                        *reinterpret_cast<ps_u32x2*>(hp3 + pl * PLANE + tlr * PLD + tlc * 2) = w2;
                    }
                }
                lds_barrier();
                if (bi == 0) RM_STAMP(2);
                have_pf = false;
                if (feeder[bi]) {       // this step's pre-activations (loaded a step ago) -> LDS; the next step's -> registers
                    const f32x4 pv = {pre[bi][0], pre[bi][1], pre[bi][2], pre[bi][3]};
                    *reinterpret_cast<f32x4*>(prel + (bi * 512 + (tid - 512)) * 4) = pv;
                    const float* gb = a.gates + (long)opaque((unsigned)seqr[bi]) * 4 * H + u0 + cu;
                    const int tn = step + 1 < T ? (dir ? t - 1 : t + 1) : t;
#pragma unroll
                    for (int g = 0; g < 4; ++g) pre[bi][g] = gb[(long)tn * 4 * H + g * H];
                }
                // ---- this wave's part of G_t = h_{t-1} W_hh^T: 16 sequences x 2 x 16 gate rows over its K quarter
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    PsPlanes<8> A;
                    const unsigned* ar = hp3 + r16 * PLD + (kq4 * C::KQ + ks * 32 + kq * 8) / 2;
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) {
                        const ps_u32x4 q = *reinterpret_cast<const ps_u32x4*>(ar + pl * PLANE);
                        A.p[pl][0] = q[0]; A.p[pl][1] = q[1]; A.p[pl][2] = q[2]; A.p[pl][3] = q[3];
                    }
                    acc[0] = ps_mfma6<8>(A, Wp[0][ks], acc[0]);
                    acc[1] = ps_mfma6<8>(A, Wp[1][ks], acc[1]);
                }
            }
            if (nbat == 2) {
                // the NEXT half-round's tile (the other batch: this step's for batch 1, the next step's for batch 0) was published a
                // whole half-round ago: fetch it now, under the reduction / cell phases (four agent-scope dword loads, no wait here)
                const int nb_i = bi ^ 1;
                const int nstep = bi == 0 ? step : step + 1;
                if (nvalid[nb_i] != 0 && nstep > 0 && nstep < T) {
                    const unsigned* src = reinterpret_cast<const unsigned*>(tile_src(nb_i, nstep - 1));
#pragma unroll
                    for (int k = 0; k < 4; ++k) pf[k] = __hip_atomic_load(src + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    have_pf = true;
                }
            }
            if (bi == 0) RM_STAMP(3);
            // ---- the four K quarters meet in LDS: D[row 4 kq + i][column r16]
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                float* rt = red + ((kq4 * 8 + 2 * np + j) * 16) * RLD;
#pragma unroll
                for (int i = 0; i < 4; ++i) rt[(kq * 4 + i) * RLD + r16] = acc[j][i];
            }
            lds_barrier();
            if (bi == 0) RM_STAMP(4);
            if (cell[bi]) {
                const long srow = (long)opaque((unsigned)seqr[bi]);
                float* gb = a.gates + srow * 4 * H + u0 + cu;
                const float* r0 = red + (((cu >> 2) * 16) + cs) * RLD + (cu & 3) * 4;
                f32x4 sg = *reinterpret_cast<const f32x4*>(r0);
#pragma unroll
                for (int q = 1; q < 4; ++q) {
                    const f32x4 sq = *reinterpret_cast<const f32x4*>(r0 + q * 8 * 16 * RLD);
                    sg[0] += sq[0]; sg[1] += sq[1]; sg[2] += sq[2]; sg[3] += sq[3];
                }
                {
                    const f32x4 pv = *reinterpret_cast<const f32x4*>(prel + (bi * 512 + tid) * 4);
#pragma unroll
                    for (int g = 0; g < 4; ++g) sg[g] += pv[g];
                }
                const float ig = sigmoidf_acc(sg[0]);
                const float fg = sigmoidf_acc(sg[1]);
                const float gg = tanhf_acc(sg[2]);
                const float og = sigmoidf_acc(sg[3]);
                c[bi] = fg * c[bi] + ig * gg;
                const float h = og * tanhf_acc(c[bi]);
                const unsigned cb = opaque((unsigned)(bbase[bi] + cs));
                float* hp = ring + (size_t)(bi * 4 + (step & 3)) * RSLOT + opaque((unsigned)(cs * H + u0 + cu));
                if (l2x) __hip_atomic_store(reinterpret_cast<unsigned*>(hp), pub_bits(h), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                else st1_agent(hp, h);
                a.out[((long)cb * T + t) * 2 * H + dir * H + u0 + cu] = h;
                if (step + 2 < T) {         // its slot of step + 2 back to the sentinel
                    unsigned* sp = reinterpret_cast<unsigned*>(ring + (size_t)(bi * 4 + ((step + 2) & 3)) * RSLOT + opaque((unsigned)(cs * H + u0 + cu)));
                    if (l2x) __hip_atomic_store(sp, PS_SENT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    else __hip_atomic_store(sp, PS_SENT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                if (STASH) {
                    a.hprev[(srow + t) * H + u0 + cu] = hlast[bi];
                    a.cbuf[(srow + t) * H + u0 + cu] = c[bi];
                    float* gp = gb + (long)t * 4 * H;
                    gp[0] = ig; gp[H] = fg; gp[2 * H] = gg; gp[3 * H] = og;
                }
                hlast[bi] = h;
            }
            if (bi == 0) RM_STAMP(5);
            // (the next tile's barrier separates these reads of `red` from its next writes)
        }
    }
}


// ------------------------------------------------------------------------------------------------ backward (BPTT)
// dh_{t-1} = dG_t W_hh has K = 4H: cutting it over the OUTPUT units (as the forward does) would make every workgroup pull the whole
// 16 x 4H dG tile — 64 KB per step (first version: 4.5 us per step, 1.1 us of it the tile, 0.45 its split).  It is cut over K instead:
// workgroup m of a group owns hidden units [32 m, 32 m + 32), applies their cell backward, and multiplies ITS OWN 128 gate-gradient
// columns (16 sequences x 128, never leaving the CU: LDS planes) by its 128 rows of W_hh (split once into bf16 planes, 48 registers
// per lane: wave = one N-tile of 16 output units, the four gates are its four 32-deep k-steps) — a PARTIAL dh for all H output units.
// The partial sums travel: workgroup m publishes, per consumer c, a block [32 units of c][16 sequences] (one float4 per lane, a wave
// writes 1 KB) into a four-slot ring; every lane of consumer c polls ONE float4 (producer, unit, sequence quad) — 16 KB in per
// workgroup and step, like the forward — drops it into LDS, and 512 cell lanes sum the eight producers' parts.  A slot is
// re-filled with the sentinel by its producer two steps before it is written again (its consumer has provably read it: the
// producer has seen that consumer's NEXT publication), so only the ring (512 KB per group) is sentinel-filled by the host, not the
// gradient buffer.  The stash of the coming step (gates, c_t, c_{t-1}, dout: seven HBM reads per unit) is fetched a step ahead by
// the lanes 512 above the cell lanes, turned into the cell backward's factors there and handed over through LDS.
#ifdef RM_TRACE
__device__ unsigned long long rm_bwd_trace[256 * 8];
#define RB_STAMP(k) do { if (blockIdx.x == 0 && threadIdx.x == 0 && step < 256) rm_bwd_trace[step * 8 + (k)] = wall_clock64(); } while (0)
#else
#define RB_STAMP(k) do { } while (0)
#endif

template <int H>
struct RecMfmaBwd {
    static constexpr int G = H / RM_UW;
    static constexpr int KL = 4 * RM_UW;                // local contraction length: this workgroup's gate rows
    static constexpr int KS = KL / 32;                  // = 4: one 32-deep k-step per gate
    static constexpr int PLD = KL / 2 + 4;              // LDS row stride (dwords = bf16 pairs) of one plane of the local dG tile
    static constexpr int PLANE = RM_NB * PLD;
    static constexpr int XLD = RM_NB + 4;               // row stride of the partial-sum exchange: [producer][unit][16 sequences + pad]
    static constexpr int XS = G * RM_UW * XLD;
    static constexpr int FACL = 2 * 512 * 8;            // cell-backward factors: [step parity][cell lane][8] (written a step ahead)
    static constexpr int LDS_FLOATS = 3 * PLANE + XS + 16 + FACL;
    static constexpr int SLOT = G * G * RM_UW * RM_NB;  // floats of one ring slot: [consumer][producer][32 units][16 sequences]
    static constexpr int RING = 4 * SLOT;               // floats per group
    static_assert(H == 256 && G * 128 == RM_THREADS, "one float4 of the incoming partial sums per lane; 48 plane registers per lane");
};

struct RecMfmaBwdArgs {
    const float* dout; const float* gates; const float* cbuf; const float* w_hh_t; float* dgates;
    float* db_f; float* db_r;
    float* ring;                                        // [group][4 slots][SLOT], sentinel-prefilled
    int B, T, b0, Bc;
    unsigned* err;
    unsigned long long* idbuf;
    int force_agent;
};

template <int H>
__global__ __launch_bounds__(RM_THREADS) void rec_bwd_mfma_kernel(RecMfmaBwdArgs a) {
    using C = RecMfmaBwd<H>;
    constexpr int G = C::G, KS = C::KS, PLD = C::PLD, PLANE = C::PLANE, XLD = C::XLD, K4 = 4 * H;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    unsigned* gp3 = reinterpret_cast<unsigned*>(smem);  // [plane][16 sequences][PLD] bf16 pairs of this workgroup's dG columns
    float* xs = smem + 3 * PLANE;                       // [producer][unit][XLD] partial sums of dh
    volatile unsigned* cflags = reinterpret_cast<volatile unsigned*>(xs + C::XS);
    float* facl = xs + C::XS + 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int T = a.T, B = a.B;
    const int ngb = (a.Bc + RM_NB - 1) / RM_NB;         // groups per direction
    const int ngroups = 2 * ngb;
    int group, member;
    {
        const int bid = blockIdx.x;
        // XCD-local groups under round-robin dispatch (block b -> XCD b % 8, verified at run time below): the grid is padded to whole
        // rounds of eight groups and the workgroups of the missing groups leave at once
        const int q = bid >> 3;
        member = q % G; group = (q / G) * 8 + (bid & 7);
        if (group >= ngroups) return;
    }
    const int dir = group >= ngb ? 1 : 0;
    const int u0 = member * RM_UW;
    const int bbase = a.b0 + (group - dir * ngb) * RM_NB;
    const int nvalid = max(0, min(RM_NB, a.b0 + a.Bc - bbase));
    float* ring = a.ring + (size_t)group * C::RING;

    // ---- resident weights: wave w = N-tile of output units 16 w .. 16 w + 15; B operand column c -> unit n = 16 w + c; k-step ks = gate,
    //      its 8 k-slots = own units u0 + 8 kq .. + 7: W_hh[ks H + u0 + 8 kq + e][n] = w_hh_t[n][ks H + u0 + 8 kq + e]
    const int r16 = lane & 15, kq = lane >> 4;
    PsPlanes<8> Wp[KS];
    {
        const float* wrow = a.w_hh_t + ((long)dir * H + 16 * wave + r16) * K4 + u0 + kq * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const f32x4 w0 = ld4p(wrow + ks * H), w1 = ld4p(wrow + ks * H + 4);
            const float v[8] = {w0[0], w0[1], w0[2], w0[3], w1[0], w1[1], w1[2], w1[3]};
            Wp[ks] = ps_split<8>(v);
        }
    }
    // ---- cell lanes (sequence cs, unit cu) for tid < 512; lane 512 + i feeds cell lane i
    const int cs = (tid & 511) >> 5, cu = tid & 31;
    const bool valid = cs < nvalid;
    const bool cell = tid < 512 && valid, feeder = tid >= 512 && valid;
    const int seqr = (dir * B + bbase + (valid ? cs : 0)) * T;             // row base in the (2, B, T, *) arrays
    const int dor = (bbase + (valid ? cs : 0)) * T;                         // row base in dout (B, T, 2H)
    float dc = 0.f, bs[4] = {0.f, 0.f, 0.f, 0.f};
    float p[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto load_step = [&](int st) {                      // raw stash of processing step st (clamped: the last steps re-read their own row)
        const int sc = st < T ? st : T - 1;
        const int t = dir ? sc : T - 1 - sc;
        const long row = (long)opaque((unsigned)seqr) + t;
        const float* gp = a.gates + row * 4 * H + u0 + cu;
        p[0] = gp[0]; p[1] = gp[H]; p[2] = gp[2 * H]; p[3] = gp[3 * H];
        p[4] = a.cbuf[row * H + u0 + cu];
        const int tp = dir ? t + 1 : t - 1;             // the time processed before t in the FORWARD pass
        const int tpc = (tp >= 0 && tp < T) ? tp : t;
        const float cp = a.cbuf[((long)opaque((unsigned)seqr) + tpc) * H + u0 + cu];
        p[5] = (tp >= 0 && tp < T) ? cp : 0.f;
        p[6] = a.dout[((long)opaque((unsigned)dor) + t) * 2 * H + dir * H + u0 + cu];
    };
    auto prepare_to_lds = [&](int par) {                // p -> factors: beta, a_i, a_f, a_g | a_o, f, dout, -
        const float tc = tanhf_acc(p[4]);
        const f32x4 lo = {p[3] * (1.f - tc * tc), p[2] * p[0] * (1.f - p[0]), p[5] * p[1] * (1.f - p[1]), p[0] * (1.f - p[2] * p[2])};
        const f32x4 hi = {tc * p[3] * (1.f - p[3]), p[1], p[6], 0.f};
        *reinterpret_cast<f32x4*>(facl + (par * 512 + (tid - 512)) * 8) = lo;
        *reinterpret_cast<f32x4*>(facl + (par * 512 + (tid - 512)) * 8 + 4) = hi;
    };
    if (feeder) { load_step(0); prepare_to_lds(0); load_step(1); }
    if (tid < 4) cflags[tid] = 0u;
    lds_barrier();
    const bool l2x = !a.force_agent && rm_same_xcd<G>(a.idbuf + (size_t)group * 32, member, a.err, cflags + 4);

    // incoming partial sums: lane -> (producer pm, unit pu, sequence quad pq) of the block [this consumer][pm]
    const int pm = tid >> 7, pu = (tid >> 2) & 31, pq = tid & 3;
    const unsigned in_off = ((unsigned)(member * G + pm) * RM_UW + pu) * RM_NB + pq * 4;
    // outgoing: wave w holds output units 16 w .. + 15 = consumer w / 2, its units 16 (w & 1) + r16; a lane's acc = sequences 4 kq .. + 3
    const unsigned out_off = ((unsigned)((wave >> 1) * G + member) * RM_UW + (wave & 1) * 16 + r16) * RM_NB + kq * 4;
    unsigned short* gp16 = reinterpret_cast<unsigned short*>(gp3);

    for (int step = 0; step < T; ++step) {
        const int t = dir ? step : T - 1 - step;        // reverse of the forward processing order
        if (nvalid == 0) break;                         // (uniform) an empty group
        RB_STAMP(0);
        if (step > 0) {
            // ---- the eight producers' partial sums of dh for this workgroup's units: one float4 per lane, polled in place
            const float* src = at_bytes(ring + (size_t)((step - 1) & 3) * C::SLOT, 4u * opaque(in_off));
            f32x4 v = ld4_agent(src);
            if (has_sentinel(v)) {
                unsigned spins = 0;
                for (;;) {
                    v = ld4_agent(src);
                    if (!has_sentinel(v)) break;
                    if (spin_expired(spins, a.err, 0xDEAD0035u)) break;
                }
            }
            RB_STAMP(1);
            *reinterpret_cast<f32x4*>(xs + (pm * RM_UW + pu) * XLD + pq * 4) = v;
            lds_barrier();
            RB_STAMP(2);
        }
        if (feeder) {           // the NEXT step's factors (its stash was loaded a step ago) -> the other half of `facl` (this step's half is
            prepare_to_lds((step + 1) & 1);     // being read by the cell lanes right now); the stash of the step after it -> registers
            load_step(step + 2);
        }
        if (cell) {
            float carry = 0.f;
            if (step > 0) {
#pragma unroll
                for (int m = 0; m < G; ++m) carry += xs[(m * RM_UW + cu) * XLD + cs];
            }
            const f32x4 lo = *reinterpret_cast<const f32x4*>(facl + ((step & 1) * 512 + tid) * 8);
            const f32x4 hi = *reinterpret_cast<const f32x4*>(facl + ((step & 1) * 512 + tid) * 8 + 4);
            const float dh = hi[2] + carry;
            const float dct = dc + dh * lo[0];
            dc = dct * hi[1];
            const float g4[4] = {dct * lo[1], dct * lo[2], dct * lo[3], dh * hi[0]};
            // own dG columns -> bf16 planes in LDS (column gate * 32 + unit: a 16-bit store per plane and gate)
            unsigned pa[3], pb[3];
            ps_split_pair(g4[0], g4[1], pa[0], pa[1], pa[2]);
            ps_split_pair(g4[2], g4[3], pb[0], pb[1], pb[2]);
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                unsigned short* row = gp16 + ((size_t)pl * PLANE + cs * PLD) * 2 + cu;
                row[0] = (unsigned short)(pa[pl] & 0xffffu); row[32] = (unsigned short)(pa[pl] >> 16);
                row[64] = (unsigned short)(pb[pl] & 0xffffu); row[96] = (unsigned short)(pb[pl] >> 16);
            }
            float* dp = a.dgates + ((long)opaque((unsigned)seqr) + t) * K4 + u0 + cu;
#pragma unroll
            for (int g = 0; g < 4; ++g) { dp[g * H] = g4[g]; bs[g] += g4[g]; }
        }
        RB_STAMP(3);
        lds_barrier();
        RB_STAMP(4);
        if (step + 1 < T) {
            // ---- partial dh_{t-1} of ALL output units from this workgroup's gate gradients: 16 sequences x 16 units per wave
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                PsPlanes<8> A;
                const unsigned* ar = gp3 + r16 * PLD + (ks * 32 + kq * 8) / 2;
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    const ps_u32x4 q = *reinterpret_cast<const ps_u32x4*>(ar + pl * PLANE);
                    A.p[pl][0] = q[0]; A.p[pl][1] = q[1]; A.p[pl][2] = q[2]; A.p[pl][3] = q[3];
                }
                acc = ps_mfma6<8>(A, Wp[ks], acc);
            }
            RB_STAMP(5);
            f32x4 pubv;
#pragma unroll
            for (int i = 0; i < 4; ++i) pubv[i] = __uint_as_float(pub_bits(acc[i]));
            float* dst = at_bytes(ring + (size_t)(step & 3) * C::SLOT, 4u * opaque(out_off));
            if (l2x) *reinterpret_cast<f32x4*>(dst) = pubv; else st4_agent(dst, pubv);
            if (step + 2 < T) {     // the slot of step + 2 (last used at step - 2, read at step - 1 by a consumer whose step - 1 publication
                                    // this workgroup has consumed): back to the sentinel
                const f32x4 sent = {__uint_as_float(PS_SENT), __uint_as_float(PS_SENT), __uint_as_float(PS_SENT), __uint_as_float(PS_SENT)};
                float* sd = at_bytes(ring + (size_t)((step + 2) & 3) * C::SLOT, 4u * opaque(out_off));
                if (l2x) *reinterpret_cast<f32x4*>(sd) = sent; else st4_agent_raw(sd, sent);
            }
        }
        RB_STAMP(6);
        // (the barrier after the next poll separates these reads of the planes / `xs` / `facl` from their next writes)
    }
    float* db = dir ? a.db_r : a.db_f;
    if (db != nullptr && cell) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            atomicAdd(db + g * H + u0 + cu, bs[g]);                  // b_ih
            atomicAdd(db + 4 * H + g * H + u0 + cu, bs[g]);          // b_hh receives the same gradient
        }
    }
}

}  // namespace

// Eligibility: the shapes this form pays for.  Measured on MI355X, layer-0 forward at H = 256, T = 400 (tools/ubench_rec_mfma.py):
// B = 48 0.80 ms (the VALU multi-utterance kernels: 0.77), 64 0.80 (0.86), 96 0.82 (1.16), 128 0.84 (1.28), 200 1.04 (2.22), 256 1.07 (2.33), 512 2.10 (4.38),
// 2048 8.0 (17.1); one layer's backward incl. its GEMMs at B = 48 1.69 (1.80), 96 2.04 (2.73), 128 2.33 (3.07), 512 7.3 (15.2).
bool rec_fwd_mfma_eligible(int B, int H) {
    if (opt_get(OPT_REC_MFMA) == 0) return false;
    return H == 256 && B >= 64;
}

int rec_fwd_mfma(float* gates, const float* w_hh_f, const float* w_hh_r, float* out, float* cbuf, float* hprev, int B, int T, int H,
                 int stash, unsigned long long* xbuf, unsigned* err, hipStream_t stream) {
    LAS_REQUIRE(H == 256, "rec_fwd_mfma shape");
    LAS_REQUIRE(err != nullptr && xbuf != nullptr && (!stash || (cbuf && hprev)), "rec_fwd_mfma buffers");
    // From two batches of 16 sequences per group (B > 256 on 256 CUs) the wave-specialised pipeline (pblstm_rec_mfma2.hip) is the faster form:
    // measured layer-0 forward, T = 400: B = 512 1.93 against 2.11 ms, B = 768 2.79 against 3.82; with ONE batch per group its longer
    // per-step chain loses (B = 128: 1.24 against 0.82 ms).  REC_MFMA = 2 forces the pipeline, 3 forces this file's form (A/B).
    {
        int dev0 = 0, cus0 = 0;
        LAS_HIP_CHECK(hipGetDevice(&dev0));
        LAS_HIP_CHECK(hipDeviceGetAttribute(&cus0, hipDeviceAttributeMultiprocessorCount, dev0));
        const long mode = opt_get(OPT_REC_MFMA);
        const int one_batch = std::min(16, cus0 / 16) * RM_NB;          // sequences of one direction that one batch per group covers
        if (mode == 2 || (mode == 1 && B > one_batch))
            return rec_fwd_mfma2(gates, w_hh_f, w_hh_r, out, cbuf, hprev, B, T, H, stash, xbuf, err, stream);
    }
    using C = RecMfma<256>;
    int dev = 0, cus = 0;
    LAS_HIP_CHECK(hipGetDevice(&dev));
    LAS_HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    const int groups_max = std::min(16, cus / (2 * C::G));            // groups per direction that are resident at once (ring and id slots: 32 groups)
    if (groups_max < 1) return fail(LAS_ERR_UNSUPPORTED, "rec_fwd_mfma: %s%ld compute units are too few", "", (long)cus);
    const size_t smem = sizeof(float) * C::LDS_FLOATS;
    if (stash) LAS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&rec_fwd_mfma_kernel<256, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    else LAS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&rec_fwd_mfma_kernel<256, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    // one batch of 16 sequences per group while that covers the launch; beyond it two batches per group, stepped alternately
    const int nbat = B > groups_max * RM_NB ? 2 : 1;
    const int chunk = groups_max * RM_NB * nbat;
    float* ring = reinterpret_cast<float*>(reinterpret_cast<char*>(xbuf) + REC_MFMA_RING_OFFSET);       // (rec_xbuf_bytes makes room for it)
    for (int b0 = 0; b0 < B; b0 += chunk) {
        const int Bc = std::min(chunk, B - b0);
        const int ngroups = 2 * ((Bc + nbat * RM_NB - 1) / (nbat * RM_NB));
        const int grid = (ngroups + 7) / 8 * 8 * C::G;        // padded to whole rounds of eight groups (XCD-local placement)
        // id slots of the placement check: 32 per group, zeroed per launch (rec_xbuf_bytes covers 2 (B + 15) groups)
        LAS_HIP_CHECK(hipMemsetAsync(xbuf, 0, sizeof(unsigned long long) * 32 * (size_t)ngroups, stream));
        // the hand-off ring: four sentinel-filled slots of 16 x H floats per (group, batch)
        LAS_HIP_CHECK(hipMemsetAsync(ring, 0xFF, sizeof(float) * (size_t)ngroups * 2 * 4 * RM_NB * H, stream));
        RecMfmaArgs a{gates, w_hh_f, w_hh_r, out, cbuf, hprev, B, T, b0, Bc, err, nbat, xbuf, (int)opt_get(OPT_REC_AGENT_HANDOFF), ring, 0};
        if (stash) {
            if (!persistent_launch_fits(rec_fwd_mfma_kernel<256, true>, RM_THREADS, smem, grid))
                return fail(LAS_ERR_UNSUPPORTED, "rec_fwd_mfma: %s%ld workgroups cannot all be resident", "", (long)grid);
            hipLaunchKernelGGL((rec_fwd_mfma_kernel<256, true>), dim3(grid), dim3(RM_THREADS), smem, stream, a);
        } else {
            if (!persistent_launch_fits(rec_fwd_mfma_kernel<256, false>, RM_THREADS, smem, grid))
                return fail(LAS_ERR_UNSUPPORTED, "rec_fwd_mfma: %s%ld workgroups cannot all be resident", "", (long)grid);
            hipLaunchKernelGGL((rec_fwd_mfma_kernel<256, false>), dim3(grid), dim3(RM_THREADS), smem, stream, a);
        }
        LAS_LAUNCH_CHECK();
        path_note(PATH_REC_FWD, "rec_fwd_mfma");
    }
    return LAS_OK;
}


bool rec_bwd_mfma_eligible(int B, int H) {
    if (opt_get(OPT_REC_MFMA) == 0) return false;
    return H == 256 && B >= 64;
}

// floats of the partial-sum ring behind the xbuf region of a backward workspace (shape only: the caller sizes its workspace with it)
size_t rec_bwd_mfma_ring_floats(int B, int H) {
    if (!(H == 256 && B >= 64)) return 0;
    return (size_t)32 * RecMfmaBwd<256>::RING;          // 32 groups (256 CUs / 8) is the most one launch carries
}

int rec_bwd_mfma(const float* dout, const float* gates, const float* cbuf, const float* w_hh_t, float* dgates, int B, int T, int H,
                 unsigned long long* xbuf, unsigned* err, float* db_f, float* db_r, hipStream_t stream) {
    LAS_REQUIRE(H == 256, "rec_bwd_mfma shape");
    LAS_REQUIRE(err != nullptr && xbuf != nullptr, "rec_bwd_mfma buffers");
    using C = RecMfmaBwd<256>;
    int dev = 0, cus = 0;
    LAS_HIP_CHECK(hipGetDevice(&dev));
    LAS_HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    const int groups_max = std::min(16, cus / (2 * C::G));            // groups per direction that are resident at once (ring: 32 groups)
    if (groups_max < 1) return fail(LAS_ERR_UNSUPPORTED, "rec_bwd_mfma: %s%ld compute units are too few", "", (long)cus);
    const size_t smem = sizeof(float) * C::LDS_FLOATS;
    LAS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&rec_bwd_mfma_kernel<256>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    const int chunk = groups_max * RM_NB;
    {   // residency of the largest launch before anything is written
        const int Bc = std::min(chunk, B);
        const int grid = (2 * ((Bc + RM_NB - 1) / RM_NB) + 7) / 8 * 8 * C::G;
        if (!persistent_launch_fits(rec_bwd_mfma_kernel<256>, RM_THREADS, smem, grid))
            return fail(LAS_ERR_UNSUPPORTED, "rec_bwd_mfma: %s%ld workgroups cannot all be resident", "", (long)grid);
    }
    float* ring = reinterpret_cast<float*>(reinterpret_cast<char*>(xbuf) + rec_xbuf_bytes(B, H));      // (PblstmBwdLayout carves it there)
    for (int b0 = 0; b0 < B; b0 += chunk) {
        const int Bc = std::min(chunk, B - b0);
        const int ngroups = 2 * ((Bc + RM_NB - 1) / RM_NB);
        const int grid = (ngroups + 7) / 8 * 8 * C::G;          // padded to whole rounds of eight groups (XCD-local placement)
        LAS_HIP_CHECK(hipMemsetAsync(xbuf, 0, sizeof(unsigned long long) * 32 * (size_t)ngroups, stream));
        LAS_HIP_CHECK(hipMemsetAsync(ring, 0xFF, sizeof(float) * (size_t)ngroups * C::RING, stream));
        RecMfmaBwdArgs a{dout, gates, cbuf, w_hh_t, dgates, db_f, db_r, ring, B, T, b0, Bc, err, xbuf, (int)opt_get(OPT_REC_AGENT_HANDOFF)};
        hipLaunchKernelGGL((rec_bwd_mfma_kernel<256>), dim3(grid), dim3(RM_THREADS), smem, stream, a);
        LAS_LAUNCH_CHECK();
        path_note(PATH_REC_BWD, "rec_bwd_mfma");
    }
    return LAS_OK;
}

#ifdef RM_TRACE
extern "C" void las_debug_rm_bwd_trace(unsigned long long* host_out) {
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(rm_bwd_trace), sizeof(unsigned long long) * 256 * 8);
}
#endif

// extra bytes of a recurrence workspace for the matrix-pipe kernels (shape only): the forward's hand-off ring behind the id slots
size_t rec_mfma_xbuf_extra_bytes(int B, int H) {
    if (!(H == 256 && B >= 64)) return 0;
    return REC_MFMA_RING_OFFSET + sizeof(float) * (size_t)32 * (3 * 4 * 3 * RM_NB * 128 + 1024);     // 32 groups x (up to 3 batches (pipeline form) x 4 slots x 24 KB (three bf16 planes) + dump words)
}

}  // namespace las
