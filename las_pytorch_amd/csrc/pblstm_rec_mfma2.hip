// Multi-utterance forward recurrence of a BiLSTM layer on the matrix pipe, second form: a WAVE-SPECIALISED pipeline.
//
// Same job, group geometry, hand-off ring and arithmetic as pblstm_rec_mfma.hip (nn.LSTM(bidirectional=True) behind the reference's
// pBLSTMLayer, model/las_model.py:72-79,90; 16 sequences of one direction per group of G = H / 32 workgroups, exact three-way bf16
// operand split, fp32 accumulation) — what changes is WHO does what inside a workgroup.  In the first form all 16 waves walk through
// the phases of a step together (tile in + split | MFMA | K reduction | cell), separated by workgroup barriers: every phase uses one
// resource (memory / matrix pipe / LDS / VALU) while the others idle, and two alternating batches only hide the hand-off latency,
// not the phases themselves (2.6 us of local time per batch-step against 0.73 us of matrix-pipe time).  Here a workgroup has 8 waves
// with fixed roles that meet only through LDS flags (monotonic step counters; no workgroup barrier in the step loop):
//   * waves 0-3  M  (one per SIMD): wave w owns N-tiles 2w, 2w+1 (32 gate rows = 8 hidden units x 4 gates) over the WHOLE K = H:
//                   its W_hh rows as three bf16 planes in 192 registers, four independent accumulators; per batch-step 24 ds_read_b128
//                   of the h planes and 96 v_mfma_f32_16x16x32_bf16; no K split, hence no reduction through LDS;
//   * waves 4-5  L  : poll the 16 x H tile of h_{t-1} in the ring (eight float4 per lane, agent-scope loads, all in flight at once),
//                   split it once into bf16 planes, store them to this batch's plane buffer;
//   * waves 6-7  C  : 512 cells per batch-step (four per lane): gate sums from LDS + the input half of the gates (fetched a step ahead),
//                   activations, c / h update, publish h_t into the ring and `out`, stash for the backward pass.
// With two batches per group the M waves multiply batch 1 while batch 0's h is applied, published, travels and is split again: the
// matrix pipe, the VALU, the LDS and the memory path work at the same time, on different batches.
// A wave's vector loads and stores retire through ONE in-order counter: the polling waves (L) issue no stores, and the C waves issue
// the next step's pre-activation loads BEFORE their stores (pblstm_rec_mfma.hip found both the hard way).
// All spins are bounded and report through the device error word.
#include "las_common.h"
#include "las_kernels.h"
#include "options.h"
#include "persist_common.h"
#include "rec_mfma_common.h"
#include <algorithm>

namespace las {

namespace {

constexpr int RM2_THREADS = 512;

template <int H>
struct RecMfma2 {
    static constexpr int G = H / RM_UW;                 // workgroups (CUs) per group
    static constexpr int KS = H / 32;                   // 32-deep k-steps (whole K per M wave)
    static constexpr int PLD = H / 2 + 4;               // LDS row stride (dwords = bf16 pairs) of one plane of the h tile
    static constexpr int PLANE = RM_NB * PLD;           // dwords of one plane
    static constexpr int GLD = 4 * RM_UW + 4;           // row stride (floats) of the recurrent gate sums [16 sequences][32 units x 4 gates]
    static constexpr int GBUF = RM_NB * GLD;
    static constexpr int NFLAGS = 64;
    static constexpr int LDS_FLOATS = 2 * 3 * PLANE + 2 * GBUF + NFLAGS;
    static_assert(H == 256, "register budget of the M waves (2 N-tiles x KS x 12 plane registers) and eight tile float4 per L lane");
};

// flag words (LDS, monotonic step counters)
__device__ __forceinline__ int FL_PR(int b, int w) { return b * 2 + w; }            // L wave w stored batch b's planes for step s: s
__device__ __forceinline__ int FL_MF(int b, int w) { return 4 + b * 4 + w; }        // M wave w finished reading them: s
__device__ __forceinline__ int FL_GR(int b, int w) { return 12 + b * 4 + w; }       // M wave w wrote its gate sums of step s: s + 1
__device__ __forceinline__ int FL_GF(int b, int w) { return 20 + b * 2 + w; }       // C wave w took them: s + 1

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
        if (++tries < 64u) continue;                    // an LDS round trip per try: the partner role is usually a fraction of a microsecond away
        if (spin_expired(spins, err, code)) return false;
    }
    asm volatile("" ::: "memory");
    return true;
}
__device__ __forceinline__ void rm2_post(volatile unsigned* flags_generic, int idx, unsigned value, int lane) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // this wave's LDS traffic before the flag (LDS operations of a wave complete in order)
    if (lane == 0) ((volatile rm2_lds_u32*)flags_generic)[idx] = value;
}

// eight 16-byte agent-scope loads in flight, one wait
__device__ __forceinline__ void rm2_ld8(const float* const (&p)[8], f32x4 (&v)[8]) {
    asm volatile("global_load_dwordx4 %0, %8, off sc1\n\t"
                 "global_load_dwordx4 %1, %9, off sc1\n\t"
                 "global_load_dwordx4 %2, %10, off sc1\n\t"
                 "global_load_dwordx4 %3, %11, off sc1\n\t"
                 "global_load_dwordx4 %4, %12, off sc1\n\t"
                 "global_load_dwordx4 %5, %13, off sc1\n\t"
                 "global_load_dwordx4 %6, %14, off sc1\n\t"
                 "global_load_dwordx4 %7, %15, off sc1\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
                 : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]), "v"(p[5]), "v"(p[6]), "v"(p[7]) : "memory");
}

// phase stamps (option REC_TRACE; tools/ubench_rec_mfma.py TRACE=1): workgroup 0, batch 0, 100 MHz wall clock
#define RM2_STAMP(k) do { if (a.trace && blockIdx.x == 0 && bi == 0 && lane == 0 && step < 256) a.idbuf[4096 + step * 8 + (k)] = wall_clock64(); } while (0)

template <int H, bool STASH>
__global__ __launch_bounds__(RM2_THREADS) void rec_fwd_mfma2_kernel(RecMfmaArgs a) {
    using C = RecMfma2<H>;
    constexpr int G = C::G, KS = C::KS, PLD = C::PLD, PLANE = C::PLANE, GLD = C::GLD;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    unsigned* hp3 = reinterpret_cast<unsigned*>(smem);                      // [batch][plane][16 sequences][PLD] bf16 pairs
    float* gbuf = smem + 2 * 3 * PLANE;                                     // [batch][16 sequences][GLD]
    volatile unsigned* flags = reinterpret_cast<volatile unsigned*>(gbuf + 2 * C::GBUF);
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
    int bbase[2], nvalid[2];
#pragma unroll
    for (int bi = 0; bi < 2; ++bi) {
        bbase[bi] = a.b0 + ((group - dir * ngb) * nbat + bi) * RM_NB;
        nvalid[bi] = bi < nbat ? max(0, min(RM_NB, a.b0 + a.Bc - bbase[bi])) : 0;
    }
    if (tid < C::NFLAGS) flags[tid] = 0u;
    __syncthreads();
    const bool l2x = !a.force_agent && rm_same_xcd<G>(a.idbuf + (size_t)group * 32, member, a.err, flags + 32);
    constexpr int RSLOT = RM_NB * H;
    float* ring = a.ring + (size_t)group * 2 * 4 * RSLOT;

    if (wave < 4) {
        // ================================================================================================ M: the recurrent product
        const int r16 = lane & 15, kq = lane >> 4;
        PsPlanes<8> Wp[2][KS];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int nt = 2 * wave + j;                                    // tile column c -> unit 4 nt + c / 4, gate c % 4
            const long wrow = (long)(r16 & 3) * H + u0 + 4 * nt + (r16 >> 2);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const float* src = w_hh + wrow * H + ks * 32 + kq * 8;
                const f32x4 w0 = ld4p(src), w1 = ld4p(src + 4);
                const float v[8] = {w0[0], w0[1], w0[2], w0[3], w1[0], w1[1], w1[2], w1[3]};
                Wp[j][ks] = ps_split<8>(v);
            }
        }
        for (int step = 0; step < T; ++step) {
#pragma unroll
            for (int bi = 0; bi < 2; ++bi) {
                if (nvalid[bi] == 0) continue;                              // (uniform)
                f32x4 acc[2][2] = {{{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}};
                if (step > 0) {
                    if (!rm2_wait<2>(flags, FL_PR(bi, 0), (unsigned)step, a.err, 0xDEAD0041u)) return;
                    if (wave == 0) RM2_STAMP(3);
                    const unsigned* ar = hp3 + bi * 3 * PLANE + r16 * PLD + opaque((unsigned)(kq * 4));
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
                        acc[0][ks & 1] = ps_mfma6<8>(A, Wp[0][ks], acc[0][ks & 1]);
                        acc[1][ks & 1] = ps_mfma6<8>(A, Wp[1][ks], acc[1][ks & 1]);
                        A = An;
                    }
                    asm volatile("" ::: "memory");
                    rm2_post(flags, FL_MF(bi, wave), (unsigned)step, lane);
                    if (wave == 0) RM2_STAMP(4);
                }
                // the C waves have taken the previous step's sums of this batch
                if (!rm2_wait<2>(flags, FL_GF(bi, 0), (unsigned)step, a.err, 0xDEAD0042u)) return;
                float* gb = gbuf + bi * C::GBUF;
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int i = 0; i < 4; ++i) gb[(kq * 4 + i) * GLD + (2 * wave + j) * 16 + r16] = acc[j][0][i] + acc[j][1][i];
                rm2_post(flags, FL_GR(bi, wave), (unsigned)(step + 1), lane);
                if (wave == 0) RM2_STAMP(5);
            }
        }
    } else if (wave < 6) {
        // ================================================================================================ L: h tile -> bf16 planes
        const int ll = tid - 256, lw = wave - 4;
        for (int step = 1; step < T; ++step) {
#pragma unroll
            for (int bi = 0; bi < 2; ++bi) {
                if (nvalid[bi] == 0) continue;
                // the plane buffer of this batch is free once every M wave has read the previous step's planes
                if (!rm2_wait<4>(flags, FL_MF(bi, 0), (unsigned)(step - 1), a.err, 0xDEAD0043u)) return;
                if (lw == 0) RM2_STAMP(0);
                const float* slot = ring + (size_t)(bi * 4 + ((step - 1) & 3)) * RSLOT;
                const float* src[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int f = ll + 128 * j, row = f >> 6, c4 = f & 63;
                    src[j] = slot + opaque((unsigned)((row < nvalid[bi] ? row : 0) * H + c4 * 4));      // (rows without a sequence are never written)
                }
                // agent-scope loads in both placements: ring slots are reused every four steps, an ordinary load could hit a stale copy of
                // the line in this CU's L1 or a foreign L2 (same XCD: the producers' plain stores sit in the shared L2, read past the L1)
                f32x4 v[8];
                unsigned spins = 0;
                for (;;) {
                    rm2_ld8(src, v);
                    bool bad = false;
#pragma unroll
                    for (int j = 0; j < 8; ++j) bad = bad || has_sentinel(v[j]);
                    if (!__any(bad)) break;
                    if (spin_expired(spins, a.err, 0xDEAD0044u)) return;
                }
                if (lw == 0) RM2_STAMP(1);
                unsigned* dstb = hp3 + bi * 3 * PLANE;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int f = ll + 128 * j, row = f >> 6, c4 = f & 63;
                    unsigned p0[3], p1[3];
                    ps_split_pair(v[j][0], v[j][1], p0[0], p0[1], p0[2]);
                    ps_split_pair(v[j][2], v[j][3], p1[0], p1[1], p1[2]);
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) {
                        ps_u32x2 w2 = {p0[pl], p1[pl]};
                        *reinterpret_cast<ps_u32x2*>(dstb + pl * PLANE + row * PLD + c4 * 2) = w2;
                    }
                }
                rm2_post(flags, FL_PR(bi, lw), (unsigned)step, lane);
                if (lw == 0) RM2_STAMP(2);
            }
        }
    } else {
        // ================================================================================================ C: cells, publish, stash
        const int cl = tid - 384, cw = wave - 6;
        const int cu = cl & 31, q = cl >> 5;                                // cell j of this lane: sequence q + 4 j, unit cu
        float c[2][4], hlast[2][4], pre[2][4][4];
#pragma unroll
        for (int bi = 0; bi < 2; ++bi)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                c[bi][j] = 0.f; hlast[bi][j] = 0.f;
                const int cs = q + 4 * j;
                const bool live = cs < nvalid[bi];
                const long srow = (long)(dir * B + bbase[bi] + (live ? cs : 0)) * T;
                const int t0 = dir ? T - 1 : 0;
#pragma unroll
                for (int g = 0; g < 4; ++g) pre[bi][j][g] = live ? a.gates[(srow + t0) * 4 * H + g * H + u0 + cu] : 0.f;
            }
        for (int step = 0; step < T; ++step) {
            const int t = dir ? T - 1 - step : step;
            const int tn = step + 1 < T ? (dir ? t - 1 : t + 1) : t;
#pragma unroll
            for (int bi = 0; bi < 2; ++bi) {
                if (nvalid[bi] == 0) continue;
                // the input half of the NEXT step's gates first: these loads must not queue behind this step's stores
                float pn[4][4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int cs = q + 4 * j;
                    const bool live = cs < nvalid[bi];
                    const long srow = (long)(dir * B + bbase[bi] + (live ? cs : 0)) * T;
#pragma unroll
                    for (int g = 0; g < 4; ++g) pn[j][g] = live ? a.gates[(srow + tn) * 4 * H + g * H + u0 + cu] : 0.f;
                }
                if (!rm2_wait<4>(flags, FL_GR(bi, 0), (unsigned)(step + 1), a.err, 0xDEAD0045u)) return;
                if (cw == 0) RM2_STAMP(6);
                f32x4 sg[4];
                const float* gb = gbuf + bi * C::GBUF;
#pragma unroll
                for (int j = 0; j < 4; ++j) sg[j] = *reinterpret_cast<const f32x4*>(gb + (q + 4 * j) * GLD + cu * 4);
                rm2_post(flags, FL_GF(bi, cw), (unsigned)(step + 1), lane);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int cs = q + 4 * j;
                    if (cs >= nvalid[bi]) continue;
                    const long srow = (long)(dir * B + bbase[bi] + cs) * T;
                    const float ig = sigmoidf_acc(sg[j][0] + pre[bi][j][0]);
                    const float fg = sigmoidf_acc(sg[j][1] + pre[bi][j][1]);
                    const float gg = tanhf_acc(sg[j][2] + pre[bi][j][2]);
                    const float og = sigmoidf_acc(sg[j][3] + pre[bi][j][3]);
                    c[bi][j] = fg * c[bi][j] + ig * gg;
                    const float h = og * tanhf_acc(c[bi][j]);
                    float* hp = ring + (size_t)(bi * 4 + (step & 3)) * RSLOT + cs * H + u0 + cu;
                    if (l2x) __hip_atomic_store(reinterpret_cast<unsigned*>(hp), pub_bits(h), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    else st1_agent(hp, h);
                    a.out[((long)(bbase[bi] + cs) * T + t) * 2 * H + dir * H + u0 + cu] = h;
                    if (step + 2 < T) {         // this lane's word of slot step + 2 back to the sentinel (its consumers have read it: they have
                                                // published h of step - 1 since, which this workgroup consumed before this cell ran)
                        unsigned* sp = reinterpret_cast<unsigned*>(ring + (size_t)(bi * 4 + ((step + 2) & 3)) * RSLOT + cs * H + u0 + cu);
                        if (l2x) __hip_atomic_store(sp, PS_SENT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        else __hip_atomic_store(sp, PS_SENT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    if (STASH) {
                        a.hprev[(srow + t) * H + u0 + cu] = hlast[bi][j];
                        a.cbuf[(srow + t) * H + u0 + cu] = c[bi][j];
                        float* gp = a.gates + (srow + t) * 4 * H + u0 + cu;
                        gp[0] = ig; gp[H] = fg; gp[2 * H] = gg; gp[3 * H] = og;
                    }
                    hlast[bi][j] = h;
                }
                if (cw == 0) RM2_STAMP(7);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int g = 0; g < 4; ++g) pre[bi][j][g] = pn[j][g];
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
    // one batch of 16 sequences per group while that covers the launch; beyond it two batches per group, pipelined through the roles
    const int nbat = B > groups_max * RM_NB ? 2 : 1;
    const int chunk = groups_max * RM_NB * nbat;
    float* ring = reinterpret_cast<float*>(reinterpret_cast<char*>(xbuf) + REC_MFMA_RING_OFFSET);       // (rec_xbuf_bytes makes room for it)
    for (int b0 = 0; b0 < B; b0 += chunk) {
        const int Bc = std::min(chunk, B - b0);
        const int ngroups = 2 * ((Bc + nbat * RM_NB - 1) / (nbat * RM_NB));
        const int grid = (ngroups + 7) / 8 * 8 * C::G;        // padded to whole rounds of eight groups (XCD-local placement)
        LAS_HIP_CHECK(hipMemsetAsync(xbuf, 0, sizeof(unsigned long long) * 32 * (size_t)ngroups, stream));
        LAS_HIP_CHECK(hipMemsetAsync(ring, 0xFF, sizeof(float) * (size_t)ngroups * 2 * 4 * RM_NB * H, stream));
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
    }
    return LAS_OK;
}

}  // namespace las
