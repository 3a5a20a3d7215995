// Persistent backward of the teacher-forced Speller decode loop for gfx950: ONE launch walks all U steps in reverse.
//
// Replaces the per-step launch chain attn_step_bwd -> smallm_gemm_nn (layer 1) -> smallm_gemm_nn (layer 0) of
// las_speller_bwd (autograd of the reference's Speller.forward / Attention.forward, model/las_model.py:178-238,
// 275-297, driven by solver/solver.py:95) when the shapes allow it; the per-step kernels remain the general path.
// Loop-invariant-shaped work stays outside exactly as before: dz W_c for all steps (one GEMM before the loop) and every
// weight gradient / dfeat / dK contraction (GEMMs after the loop over the per-step gradients this kernel stashes).
//
// Per step s (descending) the dependent chain is
//     A: attention backward of utterance b         -> decoder-state gradient part (ns = 2, 4 or 8 workgroups per
//                                                     utterance, each owning a slice of the T' frames; the softmax-
//                                                     backward statistic sum_t a_t da_t equals ctx_s . dctx, so the
//                                                     slices never talk to each other)
//     Y: top-layer cell backward (pointwise)       -> dG1_s
//     X: dh0 = dG1_s W_ih1 (MFMA), bottom-layer cell backward -> dG0_s
//     Y: dctx = dG0_s W_ctx (MFMA)                 -> context gradient carried to step s-1's attention backward
// and, off that chain, the recurrent carries dG0_s W_hh0 (X itself, W_hh0 in LDS) and dG1_s W_hh1 (a fourth role R that
// starts once X has consumed dG1_s and hands its tile to Y) for step s-1.
// X, Y and R workgroups own 16 hidden units x 16 utterances (one 16x16 MFMA tile, K = 4Hs split over the 16 waves) and
// keep the columns of ONE weight matrix in VGPRs (32 floats per lane at Hs=512) for the whole launch — two resident
// matrices spilled; cell-state gradients never leave them.
// PRE variant, round 3 (speller_persist_bwd_pre_kernel: roles X, R, AttnBwdPre2Role — the training case): the attention workgroups
// contract the gate-gradient row with their P rows (no dG0 W_ctx product on the chain), exchange their parts of dq inside the utterance
// and apply the TOP cell's backward for a unit slice themselves: chain A -> X -> A(s-1); Y / RY do not exist there; X and R of a tile
// share the chain product (ProdPre2Role: half of K each, R's partial sum reaches X through the L2 of the XCD they share).
// Hand-off uses the protocol of speller_persist.hip (persist_common.h): sentinel-prefilled per-step slabs, agent-scope
// producers writing whole cache lines, one canary wave per consumer workgroup watching one dword per producer
// workgroup, L2-shared plain loads for the big tiles with the MFMA product started while they land.
#include "las_common.h"
#include "las_kernels.h"
#include "persist_common.h"
#include "options.h"
#include <algorithm>

namespace las {

struct PersistBwdArgs {
    const float* w_ih0; long ldw0; int V;         // (4Hs, V+Hs): the context columns start at V
    const float* w_hh0; const float* w_ih1; const float* w_hh1;      // (4Hs, Hs)
    const float* w_phi;                           // (M, Hs)
    const float* feat; const float* keys; const float* att; const float* q_all; const float* ctx_all;
    const float* gates_all; const float* c_all; const float* dcat_all;
    float* dG_all; float* dctx_all; float* de_all; float* dqpre_part;
    float* dx0; long ldx0;
    float* dhA;      // [U][B][ns][Hs]           decoder-state gradient parts of the attention workgroups of an utterance
    float* dGx;      // [2][U][Hs/16][32][64]    tiled gate gradients (layer, step, unit tile, row, unit*4+gate)
    float* dcx;      // [U][Hs/16][32][16]       context gradient carried to the previous step
    float* dhc;      // [U][Hs/16][32][16]       top-layer recurrent carry (R -> Y)
    float* dhp;      // [U+1][2 Hs/16][256]      PRE variant: R's half of the chain product dG1 W_ih1 (R -> X, same XCD), sentinel-prefilled;
                     //                          the extra step holds the XCC ids of the placement check
    // PRE variant (the forward ran speller_persist_fwd_pre_kernel): P = feat . W_ctx^T (B*T', 4Hs, columns unit*4+gate), the
    // forward's published sums gxf[s][b] = sum_t a_t P_t, and e0[s][b][t] = dcat_ctx[s][b] . feat[b][t] (one batched GEMM)
    const float* pctx; const float* gxf; const float* e0;
    float* dqx;      // [U][B][ns][M]            PRE variant: the frame slices' query-gradient parts, exchanged between the ns attention
                     //                          workgroups of an utterance (sentinel-prefilled)
    float* dqpre_all;   // (U*B, M)              PRE variant: the summed parts (what dW_phi needs), written by slice 0
    int B, Tp, U, relu;
    int ns;          // attention-backward workgroups per utterance (each owns ceil(T'/ns) frames)
    int NH;          // PRE variant, multi-head: ns = NH * (frame slices per head); workgroup `part` = head * (ns / NH) + slice owns that head's
                     // frame slice and Hs / ns top-layer units; pctx (B*Tp, NH*4Hs), gxf / q_all [U][B][NH][.], att / e0 / de_all [U][NH][B][Tp]
    unsigned* err;
    unsigned long long* trace;
};

#define PB_STAMP(role, s, k) do { if (a.trace && first_wg && threadIdx.x == 0) a.trace[((size_t)(role) * a.U + (s)) * 8 + (k)] = wall_clock64(); } while (0)

// ------------------------------------------------------------------------------------------------ X / Y workgroups
template <int HS>
struct ProdRole {
    static constexpr int NJ = HS / 16;                     // 16-unit tiles
    static constexpr int KB = (4 * HS / 16) / PS_NW;       // 16-wide k-blocks per wave of the 4Hs-long contraction
    static constexpr int JPW = KB / 4;                     // unit tiles per wave (a tile is 64 k = 4 blocks)
    static constexpr size_t GXS = (size_t)NJ * 32 * 64;    // floats of one tiled dG slab
    static constexpr size_t CXS = (size_t)NJ * 32 * 16;    // floats of one dcx slab
    static constexpr int RED = PS_NW * 16 * 17;
    static constexpr int LDS_FLOATS = RED + PS_NW * KB * 64 * 4 + 4;   // reduction buffer + (X role) the columns of W_hh0 + canary flags

    struct Lane {
        unsigned x[JPW];       // byte offset of this lane's float4 of the first k-block of unit tile i inside a dG slab
                               // (the other three blocks of the tile are +64, +128, +192 bytes: instruction offsets)
        unsigned canary;       // byte offset of the producer dword this lane watches when its wave is the canary wave
        bool cact;             // ... and whether that producer exists
        bool ok;               // its utterance row exists
        // PRE variant: a dG1 tile is produced by the attention workgroups — (utterance row, unit slice) — not by one workgroup per
        // unit tile: producer p = row * ns + slice, watched by lane p % 64 of canary wave p / 64
        unsigned canaryA; bool cactA; int npwA;
    };
    static __device__ __forceinline__ Lane lane_addr(int B, int mt, int wave, int lane, int ns = 0) {
        Lane t;
        const int r = lane & 15, kq = lane >> 4;
        t.ok = mt * 16 + r < B;
        const int row = t.ok ? mt * 16 + r : mt * 16;
#pragma unroll
        for (int i = 0; i < JPW; ++i) t.x[i] = 4u * (((wave * JPW + i) * 32 + row) * 64 + 4 * kq);
        // one dword per producer workgroup (unit tile `lane`, same M-tile): the last gate of its last row
        t.cact = lane < NJ;
        t.canary = 4u * (((t.cact ? lane : 0) * 32 + mt * 16 + min(15, B - 1 - mt * 16)) * 64 + 63);
        t.canaryA = 0u; t.cactA = false; t.npwA = 1;
        if (ns > 0) {
            const int p = wave * 64 + lane, prow = p / ns, pq = p % ns;     // the last gate of the last unit tile of slice pq, row prow
            t.npwA = (16 * ns + 63) / 64;
            t.cactA = p < 16 * ns && mt * 16 + prow < B;
            t.canaryA = 4u * ((((t.cactA ? pq + 1 : 1) * (NJ / ns) - 1) * 32 + mt * 16 + (t.cactA ? prow : 0)) * 64 + 63);
        }
        return t;
    }
    // resident MFMA B operand: W[(gate e)*HS + unit(k)][cb + 16 j + n] for this wave's k-blocks, k = unit*4 + gate
    static __device__ __forceinline__ void load_w(const float* w, long ld, int cb, int j, int wave, int lane, float (&W)[KB][4]) {
        const int r = lane & 15, kq = lane >> 4;
#pragma unroll
        for (int blk = 0; blk < KB; ++blk) {
            const int unit = (wave * JPW + blk / 4) * 16 + (blk % 4) * 4 + kq;
#pragma unroll
            for (int e = 0; e < 4; ++e) W[blk][e] = w[((long)e * HS + unit) * ld + cb + 16 * j + r];
        }
    }
    // tile of dG rows (this M-tile, all 4Hs gate columns): canary, L2-shared plain loads, per-word check, slow path
    // Returns mul(tile).  The product is started speculatively as the tile's loads land (load and MFMA time overlap);
    // the sentinel check comes afterwards and, if a word had not been published yet, the tile is repaired with
    // L2-bypassing loads and the product redone.
    template <bool FROM_A = false, class Mul>
    static __device__ __forceinline__ f32x4 poll_mul(const float* base, const Lane& t, f32x4 (&x)[KB], unsigned* err,
                                                     volatile unsigned* flags, unsigned& ep, Mul mul) {
        unsigned spins = 0;
        {
            const unsigned* cp = reinterpret_cast<const unsigned*>(at_bytes(base, opaque(FROM_A ? t.canaryA : t.canary)));
            wg_canary_wait(flags, ++ep, FROM_A ? t.npwA : 1, __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), threadIdx.x & 63, cp, FROM_A ? t.cactA : t.cact, err, 0xDEAD0021u);
        }
        asm volatile("" ::: "memory");
        bool need[KB];
        bool bad = false;
        unsigned xo[JPW];
#pragma unroll
        for (int i = 0; i < JPW; ++i) xo[i] = opaque(t.x[i]);
        // rows past B hold the sentinel for ever: they only feed output rows nobody reads (MFMA rows are independent)
#pragma unroll
        for (int blk = 0; blk < KB; ++blk)
            x[blk] = *reinterpret_cast<const f32x4*>(at_bytes(base, xo[blk / 4]) + 16 * (blk % 4));
        f32x4 acc = mul(x);
#pragma unroll
        for (int blk = 0; blk < KB; ++blk) {
            need[blk] = __any(t.ok && has_sentinel(x[blk]));
            bad |= need[blk];
        }
        asm volatile("" ::: "memory");
        const bool redo = bad;
        while (bad) {
            if (spin_expired(spins, err, 0xDEAD0022u)) break;
            bad = false;
#pragma unroll
            for (int blk = 0; blk < KB; ++blk) {
                if (need[blk]) {
                    x[blk] = ld4_agent(at_bytes(base, opaque(t.x[blk / 4])) + 16 * (blk % 4));
                    need[blk] = __any(t.ok && has_sentinel(x[blk]));
                    bad |= need[blk];
                }
            }
        }
        if (redo) acc = mul(x);
        return acc;
    }
    static __device__ __forceinline__ f32x4 mfma_tile(const f32x4 (&x)[KB], const float (&W)[KB][4]) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int blk = 0; blk < KB; ++blk)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(x[blk][e], W[blk][e], acc, 0, 0, 0);
        return acc;
    }
    // sum the 16 waves' partial 16x16 tiles; the result for (utterance m, unit n) lands in thread m*16+n (< 256)
    // ENTRY: the buffer is reused by consecutive calls, so a barrier must first retire the previous call's readers (X role,
    // whose LDS is taken by W_hh0); Y and R alternate between two buffers instead and need only the one barrier.
    template <bool ENTRY>
    static __device__ __forceinline__ float reduce_tile(float* red, const f32x4 acc, int wave, int lane, int tid) {
        const int r = lane & 15, kq = lane >> 4;
        if (ENTRY) lds_barrier();
#pragma unroll
        for (int i = 0; i < 4; ++i) red[(wave * 16 + kq * 4 + i) * 17 + r] = acc[i];
        lds_barrier();
        float s = 0.f;
        if (tid < 256) {
#pragma unroll
            for (int w = 0; w < PS_NW; ++w) s += red[(w * 16 + (tid >> 4)) * 17 + (tid & 15)];
        }
        return s;
    }

    // cell backward of one (utterance, unit): reference semantics = autograd of nn.LSTM's cell
    struct CellIn { float ig, fg, gg, og, c, cp; };
    static __device__ __forceinline__ CellIn load_cell(const PersistBwdArgs& a, int layer, int s, unsigned o4, unsigned o1) {
        const size_t slab = ((size_t)layer * a.U + s) * (size_t)a.B * HS;
        CellIn ci;
        const float* gp = at_bytes(a.gates_all + 4 * slab, o4);
        ci.ig = gp[0]; ci.fg = gp[HS]; ci.gg = gp[2 * HS]; ci.og = gp[3 * HS];
        ci.c = *at_bytes(a.c_all + slab, o1);
        ci.cp = s > 0 ? *at_bytes(a.c_all + slab - (size_t)a.B * HS, o1) : 0.f;
        return ci;
    }
    static __device__ __forceinline__ f32x4 cell_bwd(const CellIn& ci, float dh, float& dc) {
        const float tc = tanhf_acc(ci.c);
        const float dct = dc + dh * ci.og * (1.f - tc * tc);
        f32x4 g;
        g[0] = dct * ci.gg * ci.ig * (1.f - ci.ig);
        g[1] = dct * ci.cp * ci.fg * (1.f - ci.fg);
        g[2] = dct * ci.ig * (1.f - ci.gg * ci.gg);
        g[3] = dh * tc * ci.og * (1.f - ci.og);
        dc = dct * ci.fg;
        return g;
    }
    static __device__ __forceinline__ void stash_dG(const PersistBwdArgs& a, int layer, int s, unsigned o4, const f32x4 g) {
        float* dp = at_bytes(a.dG_all + 4 * (((size_t)layer * a.U + s) * (size_t)a.B * HS), o4);
        dp[0] = g[0]; dp[HS] = g[1]; dp[2 * HS] = g[2]; dp[3 * HS] = g[3];
    }

    // ROLE 0 (X): dh0 = dG1 W_ih1 -> bottom cell backward -> dG0 ; recurrent carry dG0 W_hh0 (own, off the chain, W in LDS)
    // ROLE 1 (Y): top cell backward -> dG1 ; dctx = dG0 W_ctx
    // ROLE 2 (R): recurrent carry of the top layer dG1 W_hh1, handed to the Y workgroup of the same tile (off the chain)
    // ROLE 3 (RY, PRE variant only): top cell backward AND the top layer's recurrent carry in one workgroup (the carry never
    // leaves it).  In the PRE variant there is no Y product: the attention workgroups take dG0 themselves.
    template <int ROLE, bool PREV = false>
    static __device__ void run(const PersistBwdArgs& a, float* smem, const int widx) {
        constexpr bool IS_X = ROLE == 0, IS_Y = ROLE == 1, IS_R = ROLE == 2, IS_RY = ROLE == 3;
        const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave: scalar register
        const int j = widx >> 1, mt = widx & 1;
        const int B = a.B, U = a.U;
        if (mt * 16 >= B) return;                         // no utterance in this M-tile: nobody waits for its rows
        const bool first_wg = widx == 0;
        float* red = smem;
        // one weight matrix resident in VGPRs per role (32 floats per lane at Hs=512); X also keeps W_hh0 in LDS
        float Wc[KB][4];
        if (IS_X) load_w(a.w_ih1, HS, 0, j, wave, lane, Wc);
        else if (IS_Y) load_w(a.w_ih0, a.ldw0, a.V, j, wave, lane, Wc);
        else load_w(a.w_hh1, HS, 0, j, wave, lane, Wc);           // R and RY
        float* wlds = smem + RED + (wave * KB * 64 + lane) * 4;              // [wave][block][lane][gate]
        if (IS_X) {
            float Wt[KB][4];
            load_w(a.w_hh0, HS, 0, j, wave, lane, Wt);
#pragma unroll
            for (int blk = 0; blk < KB; ++blk) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = Wt[blk][e];
                *reinterpret_cast<f32x4*>(wlds + blk * 256) = v;
            }
        }
        auto mfma_lds = [&](const f32x4 (&xt)[KB]) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int blk = 0; blk < KB; ++blk) {
                const f32x4 w = *reinterpret_cast<const f32x4*>(wlds + blk * 256);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xt[blk][e], w[e], acc, 0, 0, 0);
            }
            return acc;
        };
        auto mul_reg = [&](const f32x4 (&xt)[KB]) { return mfma_tile(xt, Wc); };
        const Lane la = lane_addr(B, mt, wave, lane, PREV ? a.ns : 0);
        volatile unsigned* cflags = reinterpret_cast<volatile unsigned*>(smem + RED + PS_NW * KB * 64 * 4);
        unsigned cep = 0;
        if (tid < 4) cflags[tid] = 0u;
        lds_barrier();
        // cell lane: (utterance pb, unit pu) of this workgroup's tile
        const int pb = mt * 16 + (tid >> 4), pu = tid & 15;
        const bool pw = tid < 256 && pb < B;
        const unsigned o1 = 4u * ((unsigned)pb * HS + 16 * j + pu);           // byte offset in a (B,Hs) slab
        const unsigned o4 = 4u * ((unsigned)pb * 4 * HS + 16 * j + pu);       // ... in a (B,4Hs) slab
        const unsigned ox = 4u * ((((unsigned)j * 32 + pb) * 16 + pu) * 4);   // its float4 in a tiled dG slab
        const unsigned oc = 4u * (((unsigned)j * 32 + pb) * 16 + pu);         // its dword in a dcx / dhc slab
        const int layer = (IS_Y || IS_RY) ? 1 : 0;
        const float* dG1x = a.dGx + (size_t)U * GXS;
        const float* dG0x = a.dGx;
        float* dGx_out = a.dGx + (size_t)layer * U * GXS;
        float dc = 0.f, dh_carry = 0.f;
        f32x4 x[KB];
        for (int s = U - 1; s >= 0; --s) {
            if (IS_R) {
                // ---- recurrent carry of the top layer for step s-1: dG1_s W_hh1 -> Y workgroup (j, mt)
                if (s == 0) break;
                PB_STAMP(3, s, 0);
                if (!PREV) {   // start once the X workgroup of the same tile has published dG0_s, i.e. consumed dG1_s: the chain's
                    // consumers get the fabric and the L2 to themselves (this carry is only needed a whole step later).  PRE variant:
                    // the chain is short enough that the carry has to start right away to be back in time.
                    const unsigned* cp = reinterpret_cast<const unsigned*>(dG0x + (size_t)s * GXS) + (((unsigned)j * 32 + mt * 16) * 64 + 63);
                    unsigned spins = 0;
                    while (__hip_atomic_load(cp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == PS_SENT)
                        if (spin_expired(spins, a.err, 0xDEAD0026u)) break;
                }
                const f32x4 accr = poll_mul<PREV>(dG1x + (size_t)s * GXS, la, x, a.err, cflags, cep, mul_reg);
                PB_STAMP(3, s, 1);
                const float v = reduce_tile<false>(red + (s & 1) * RED, accr, wave, lane, tid);
                if (pw) st1_agent(at_bytes(a.dhc + (size_t)s * CXS, opaque(oc)), v);
                PB_STAMP(3, s, 2);
                continue;
            }
            CellIn ci;
            float dcat = 0.f;
            if (pw) {
                ci = load_cell(a, layer, s, opaque(o4), opaque(o1));
                if (IS_Y || IS_RY) dcat = *at_bytes(a.dcat_all + (size_t)s * B * 2 * HS, opaque(4u * ((unsigned)pb * 2 * HS + 16 * j + pu)));
            }
            if (IS_RY) {
                // ---- decoder-state gradient parts of the attention workgroups + own carry -> top cell backward -> dG1_s
                PB_STAMP(1, s, 0);
                if (pw) {
                    const unsigned* p0 = reinterpret_cast<const unsigned*>(
                        at_bytes(a.dhA + (size_t)s * B * a.ns * HS, opaque(4u * ((unsigned)pb * a.ns * HS + 16 * j + pu))));
                    unsigned spins = 0;
                    float parts = 0.f;
                    for (;;) {
                        bool ok = true;
                        parts = 0.f;
                        for (int k = 0; k < a.ns; ++k) {
                            const unsigned v = __hip_atomic_load(p0 + k * HS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            ok &= v != PS_SENT;
                            parts += __uint_as_float(v);
                        }
                        if (ok) break;
                        if (spin_expired(spins, a.err, 0xDEAD0027u)) break;
                    }
                    PB_STAMP(1, s, 1);
                    const f32x4 g = cell_bwd(ci, dcat + parts + dh_carry, dc);
                    st4_agent(at_bytes(dGx_out + (size_t)s * GXS, opaque(ox)), g);
                    stash_dG(a, 1, s, opaque(o4), g);
                }
                PB_STAMP(1, s, 2);
                // ---- (off the chain) recurrent carry for step s-1: dG1_s W_hh1, once the X workgroup of the same tile has
                // consumed dG1_s (published dG0_s): the chain's consumers get the fabric and the L2 to themselves
                if (s > 0) {
                    {
                        const unsigned* cp = reinterpret_cast<const unsigned*>(dG0x + (size_t)s * GXS) + (((unsigned)j * 32 + mt * 16) * 64 + 63);
                        unsigned spins = 0;
                        while (__hip_atomic_load(cp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == PS_SENT)
                            if (spin_expired(spins, a.err, 0xDEAD0028u)) break;
                    }
                    dh_carry = reduce_tile<true>(red, poll_mul(dG1x + (size_t)s * GXS, la, x, a.err, cflags, cep, mul_reg), wave, lane, tid);
                }
                PB_STAMP(1, s, 3);
                continue;
            }
            if (IS_Y) {
                // ---- Y1: decoder-state gradient parts of the two attention halves (+ carry from R) -> top cell backward -> dG1_s
                PB_STAMP(1, s, 0);
                if (pw) {
                    const unsigned* p0 = reinterpret_cast<const unsigned*>(
                        at_bytes(a.dhA + (size_t)s * B * a.ns * HS, opaque(4u * ((unsigned)pb * a.ns * HS + 16 * j + pu))));
                    const unsigned* pc = reinterpret_cast<const unsigned*>(at_bytes(a.dhc + (size_t)(s + 1) * CXS, opaque(oc)));
                    unsigned spins = 0, vc = 0u;
                    float parts = 0.f;
                    for (;;) {
                        bool ok = true;
                        parts = 0.f;
                        for (int k = 0; k < a.ns; ++k) {
                            const unsigned v = __hip_atomic_load(p0 + k * HS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            ok &= v != PS_SENT;
                            parts += __uint_as_float(v);
                        }
                        if (s < U - 1) vc = __hip_atomic_load(pc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (ok && vc != PS_SENT) break;
                        if (spin_expired(spins, a.err, 0xDEAD0023u)) break;
                    }
                    PB_STAMP(1, s, 1);
                    const float dh = dcat + parts + __uint_as_float(vc);
                    const f32x4 g = cell_bwd(ci, dh, dc);
                    st4_agent(at_bytes(dGx_out + (size_t)s * GXS, opaque(ox)), g);
                    stash_dG(a, 1, s, opaque(o4), g);
                }
                PB_STAMP(1, s, 2);
                // ---- Y3: context gradient of step s-1's attention = dG0_s W_ctx
                PB_STAMP(1, s, 3);
                const f32x4 accy = poll_mul(dG0x + (size_t)s * GXS, la, x, a.err, cflags, cep, mul_reg);
                PB_STAMP(1, s, 4);
                const float dctx = reduce_tile<false>(red + (s & 1) * RED, accy, wave, lane, tid);
                if (pw) {
                    st1_agent(at_bytes(a.dcx + (size_t)s * CXS, opaque(oc)), dctx);
                    if (s == 0) a.dx0[(size_t)pb * a.ldx0 + a.V + 16 * j + pu] = dctx;
                }
                PB_STAMP(1, s, 5);
            } else {
                // ---- X1: dh0 = dG1_s W_ih1 -> bottom-layer cell backward -> dG0_s
                PB_STAMP(0, s, 0);
                const f32x4 accx = poll_mul<PREV>(dG1x + (size_t)s * GXS, la, x, a.err, cflags, cep, mul_reg);
                if (a.trace && first_wg && tid == 0) { asm volatile("s_nop 0" :: "v"(accx[0])); }
                PB_STAMP(0, s, 1);
                PB_STAMP(0, s, 6);
                const float dh0 = reduce_tile<true>(red, accx, wave, lane, tid);
                PB_STAMP(0, s, 7);
                if (pw) {
                    const f32x4 g = cell_bwd(ci, dh0 + dh_carry, dc);
                    st4_agent(at_bytes(dGx_out + (size_t)s * GXS, opaque(ox)), g);
                    stash_dG(a, 0, s, opaque(o4), g);
                }
                PB_STAMP(0, s, 2);
                // ---- X2 (off the chain): recurrent carry of the bottom layer for step s-1
                if (s > 0) {
                    if (!PREV) {   // start once the Y workgroup of the same tile has consumed dG0_s: both would pull the same 8 MB
                        const unsigned* cp = reinterpret_cast<const unsigned*>(a.dcx + (size_t)s * CXS) + ((unsigned)j * 32 + mt * 16) * 16;
                        unsigned spins = 0;
                        while (__hip_atomic_load(cp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == PS_SENT)
                            if (spin_expired(spins, a.err, 0xDEAD0025u)) break;
                    }
                    dh_carry = reduce_tile<true>(red, poll_mul(dG0x + (size_t)s * GXS, la, x, a.err, cflags, cep, mfma_lds), wave, lane, tid);
                }
                PB_STAMP(0, s, 3);
            }
        }
    }
};


// ------------------------------------------------------------------------------------------------ X / R workgroups, PRE variant (round 3, last change)
// What the X product waits for is bytes into ONE CU: its 16 x 4Hs tile of dG1 is 128 KB (1.7 us at the ~32 B/clk a CU takes in
// freshly written lines), and halving it needs twice the X workgroups — the launch has none to spare, but the R workgroups (the top
// layer's recurrent carry, off the chain) idle ~8 of every 11 us.  X (j, mt) and R (j, mt) therefore SHARE the chain product
// dh0 = dG1_s W_ih1: X takes the first half of K (64 KB in, half the weights: 16 registers), R the second half, and R hands its
// 16 x 16 partial sum to X.  The two sit on the same XCD (block indices NXY apart, NXY a multiple of 8; verified at run time), so that
// hand-over is an ordinary store into the shared L2 and an L1-bypassing load (~0.5 us) instead of a trip through memory.  R then
// pulls the other half of the tile for its own product (dG1_s W_hh1 over the whole K), as before off the chain.
template <int HS>
struct ProdPre2Role {
    using P = ProdRole<HS>;
    static constexpr int NJ = P::NJ, KB = P::KB, RED = P::RED;
    // the split of K: X takes KX of a wave's KB blocks, R the other KR.  Even halves are the measured best (5 : 3 in X's favour, meant to
    // cover the ~0.6 us R's partial sum needs to reach X: 10.3 instead of 9.75 us per step)
    static constexpr int KX = KB / 2, KR = KB - KX;
    static constexpr size_t GXS = P::GXS, CXS = P::CXS;
    static constexpr int NWG = 2 * NJ;                       // X (or R) workgroups
    static constexpr size_t HPS = (size_t)NWG * 256;         // floats of one step of the partial-sum slab
    static_assert(KX >= 1 && KR >= 1 && NJ * 4 == PS_NW * KB, "K = NJ unit tiles of four 16-wide blocks, KB blocks per wave");

    // part 0 = the first 16 KX blocks of K (X's share), part 1 = the rest; N = blocks per wave of that part
    template <int N> struct Part { unsigned x[N]; };         // this lane's float4 offsets (bytes) inside a tiled dG slab
    template <int PART, int N>
    static __device__ __forceinline__ int block_of(int wave, int i) { return PART == 0 ? wave * N + i : PS_NW * KX + wave * N + i; }
    template <int PART, int N>
    static __device__ __forceinline__ Part<N> part_addr(int row, int wave, int lane) {
        Part<N> h;
        const int kq = lane >> 4;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const int gb = block_of<PART, N>(wave, i), tile = gb >> 2, blk = gb & 3;
            h.x[i] = 4u * ((tile * 32 + row) * 64 + 16 * blk + 4 * kq);
        }
        return h;
    }
    // the matching columns of a weight matrix: W[(gate e) HS + unit(k)][16 j + n]
    template <int PART, int N>
    static __device__ __forceinline__ void load_wh(const float* w, int j, int wave, int lane, float (&W)[N][4]) {
        const int r = lane & 15, kq = lane >> 4;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const int gb = block_of<PART, N>(wave, i), tile = gb >> 2, blk = gb & 3;
            const int unit = tile * 16 + blk * 4 + kq;
#pragma unroll
            for (int e = 0; e < 4; ++e) W[i][e] = w[((long)e * HS + unit) * HS + 16 * j + r];
        }
    }
    // a part of the tile: ordinary (L2-shared) loads after the canary, every word checked, incomplete slots re-read past the caches
    template <int N>
    static __device__ __forceinline__ void load_half(const float* base, const Part<N>& h, bool ok, f32x4 (&x)[N], unsigned* err) {
#pragma unroll
        for (int i = 0; i < N; ++i) x[i] = *reinterpret_cast<const f32x4*>(at_bytes(base, opaque(h.x[i])));
        bool bad = false;
#pragma unroll
        for (int i = 0; i < N; ++i) bad |= __any(ok && has_sentinel(x[i]));
        unsigned spins = 0;
        while (bad) {
            if (spin_expired(spins, err, 0xDEAD002Du)) break;
            bad = false;
#pragma unroll
            for (int i = 0; i < N; ++i) {
                if (__any(ok && has_sentinel(x[i]))) {
                    x[i] = ld4_agent(at_bytes(base, opaque(h.x[i])));
                    bad |= __any(ok && has_sentinel(x[i]));
                }
            }
        }
    }
    template <int N>
    static __device__ __forceinline__ f32x4 mfma_half(const f32x4 (&x)[N], const float (&W)[N][4], f32x4 acc) {
#pragma unroll
        for (int i = 0; i < N; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(x[i][e], W[i][e], acc, 0, 0, 0);
        return acc;
    }

    template <bool IS_X>
    static __device__ void run(const PersistBwdArgs& a, float* smem, const int widx) {
        const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int j = widx >> 1, mt = widx & 1;
        const int B = a.B, U = a.U;
        if (mt * 16 >= B) return;                         // no utterance in this M-tile (X and R of the tile leave together)
        const bool first_wg = widx == 0;
        float* red = smem;
        const int r = lane & 15;
        const bool ok = mt * 16 + r < B;
        const int row = ok ? mt * 16 + r : mt * 16;
        const Part<KX> h0 = part_addr<0, KX>(row, wave, lane);
        const Part<KR> h1 = part_addr<1, KR>(row, wave, lane);
        const typename P::Lane la = P::lane_addr(B, mt, wave, lane, a.ns);
        volatile unsigned* cflags = reinterpret_cast<volatile unsigned*>(smem + RED + PS_NW * KB * 64 * 4);
        unsigned cep = 0;
        if (tid < 4) cflags[tid] = 0u;
        // ---- placement check: X and R of a tile publish their XCC ids in the extra step of the partial-sum slab
        bool l2x;
        {
            unsigned xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcc));
            unsigned* ids = reinterpret_cast<unsigned*>(a.dhp + (size_t)U * HPS + (size_t)widx * 256);
            volatile int* flag = reinterpret_cast<volatile int*>(smem + RED);
            if (tid == 0) { *flag = 1; __hip_atomic_store(ids + (IS_X ? 0 : 1), 0xC0DE0000u | xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
            lds_barrier();
            if (tid == 0) {
                unsigned spins = 0, v;
                for (;;) {
                    v = __hip_atomic_load(ids + (IS_X ? 1 : 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (v != PS_SENT) break;
                    if (spin_expired(spins, a.err, 0xDEAD002Eu)) break;
                }
                if (v != (0xC0DE0000u | xcc)) *flag = 0;
            }
            lds_barrier();
            l2x = *flag != 0;
            lds_barrier();
        }
        const int pb = mt * 16 + (tid >> 4), pu = tid & 15;
        const bool pw = tid < 256 && pb < B;
        const unsigned o1 = 4u * ((unsigned)pb * HS + 16 * j + pu);
        const unsigned o4 = 4u * ((unsigned)pb * 4 * HS + 16 * j + pu);
        const unsigned ox = 4u * ((((unsigned)j * 32 + pb) * 16 + pu) * 4);
        const unsigned oc = 4u * (((unsigned)j * 32 + pb) * 16 + pu);
        const unsigned op = 4u * ((unsigned)widx * 256 + (unsigned)(tid & 255));       // its word of the tile's partial sum
        const float* dG1x = a.dGx + (size_t)U * GXS;
        const float* dG0x = a.dGx;
        auto canary = [&](int s) {
            const unsigned* cp = reinterpret_cast<const unsigned*>(at_bytes(dG1x + (size_t)s * GXS, opaque(la.canaryA)));
            wg_canary_wait(cflags, ++cep, la.npwA, wave, lane, cp, la.cactA, a.err, 0xDEAD0021u);
            asm volatile("" ::: "memory");
        };

        if (IS_X) {
            // ---- X: first half of dh0 = dG1 W_ih1, + R's half -> bottom cell backward -> dG0; recurrent carry dG0 W_hh0 (own, off the chain)
            float Wx[KX][4];
            load_wh<0, KX>(a.w_ih1, j, wave, lane, Wx);
            float* wlds = smem + RED + (wave * KB * 64 + lane) * 4;              // W_hh0 columns: [wave][block][lane][gate]
            {
                float Wt[KB][4];
                P::load_w(a.w_hh0, HS, 0, j, wave, lane, Wt);
#pragma unroll
                for (int blk = 0; blk < KB; ++blk) {
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = Wt[blk][e];
                    *reinterpret_cast<f32x4*>(wlds + blk * 256) = v;
                }
            }
            auto mfma_lds = [&](const f32x4 (&xt)[KB]) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int blk = 0; blk < KB; ++blk) {
                    const f32x4 w = *reinterpret_cast<const f32x4*>(wlds + blk * 256);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xt[blk][e], w[e], acc, 0, 0, 0);
                }
                return acc;
            };
            lds_barrier();
            float dc = 0.f, dh_carry = 0.f;
            f32x4 x[KB];
            for (int s = U - 1; s >= 0; --s) {
                typename P::CellIn ci;
                if (pw) ci = P::load_cell(a, 0, s, opaque(o4), opaque(o1));
                PB_STAMP(0, s, 0);
                canary(s);
                f32x4 xh[KX];
                load_half(dG1x + (size_t)s * GXS, h0, ok, xh, a.err);
                const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
                const f32x4 accx = mfma_half(xh, Wx, zero);
                PB_STAMP(0, s, 1);
                // R's half of the sum: one word per cell lane, from the shared L2 (or, if the pair does not share an XCD, from memory); a first
                // look goes out before the reduction so that its latency hides under it
                unsigned pv = PS_SENT;
                if (pw) pv = __hip_atomic_load(reinterpret_cast<const unsigned*>(at_bytes(a.dhp + (size_t)s * HPS, opaque(op))), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                PB_STAMP(0, s, 6);
                float dh0 = P::template reduce_tile<true>(red, accx, wave, lane, tid);
                PB_STAMP(0, s, 7);
                if (pw) {
                    const unsigned* pp = reinterpret_cast<const unsigned*>(at_bytes(a.dhp + (size_t)s * HPS, opaque(op)));
                    unsigned spins = 0, v = pv;
                    while (v == PS_SENT) {
                        v = __hip_atomic_load(pp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (v != PS_SENT) break;
                        if (spin_expired(spins, a.err, 0xDEAD002Fu)) break;
                    }
                    dh0 += __uint_as_float(v);
                    const f32x4 g = P::cell_bwd(ci, dh0 + dh_carry, dc);
                    st4_agent(at_bytes(a.dGx + (size_t)s * GXS, opaque(ox)), g);
                    P::stash_dG(a, 0, s, opaque(o4), g);
                }
                PB_STAMP(0, s, 2);
                // ---- (off the chain) recurrent carry of the bottom layer for step s-1
                if (s > 0) dh_carry = P::template reduce_tile<true>(red, P::poll_mul(dG0x + (size_t)s * GXS, la, x, a.err, cflags, cep, mfma_lds), wave, lane, tid);
                PB_STAMP(0, s, 3);
            }
        } else {
            // ---- R: second half of the chain product for X (first), then the top layer's recurrent carry dG1 W_hh1 over the whole K
            float Wi[KR][4], Wr0[KX][4], Wr1[KR][4];
            load_wh<1, KR>(a.w_ih1, j, wave, lane, Wi);
            load_wh<0, KX>(a.w_hh1, j, wave, lane, Wr0);
            load_wh<1, KR>(a.w_hh1, j, wave, lane, Wr1);
            lds_barrier();
            for (int s = U - 1; s >= 0; --s) {
                PB_STAMP(3, s, 0);
                canary(s);
                // (the other half is pulled AFTER the partial sum is out: issued together, the two halves share the CU's intake and the
                // half the chain waits for arrives later — 10.8 instead of 9.7 us per step)
                f32x4 x1[KR];
                load_half(dG1x + (size_t)s * GXS, h1, ok, x1, a.err);
                const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
                const float part = P::template reduce_tile<true>(red, mfma_half(x1, Wi, zero), wave, lane, tid);
                if (tid < 256) {
                    float* dst = at_bytes(a.dhp + (size_t)s * HPS, opaque(op));
                    if (l2x) __hip_atomic_store(reinterpret_cast<unsigned*>(dst), pub_bits(part), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    else st1_agent(dst, part);
                }
                PB_STAMP(3, s, 1);
                if (s == 0) break;
                f32x4 x0[KX];
                load_half(dG1x + (size_t)s * GXS, h0, ok, x0, a.err);
                const f32x4 accr = mfma_half(x1, Wr1, mfma_half(x0, Wr0, zero));
                const float v = P::template reduce_tile<true>(red, accr, wave, lane, tid);
                if (pw) st1_agent(at_bytes(a.dhc + (size_t)s * CXS, opaque(oc)), v);
                PB_STAMP(3, s, 2);
            }
        }
    }
};

// ------------------------------------------------------------------------------------------------ attention backward
template <int HS>
struct AttnBwdRole {
    static constexpr int D = HS;
    static constexpr int CW = HS / 16, NF4 = CW / 4;        // columns per lane (16 lanes per frame)
    static constexpr int RP = 512 / HS;                     // frames per lane group
    static constexpr int TH = 64 * RP;                      // frames per workgroup
    static constexpr int NMH = PS_THREADS / HS, MPT = PS_M / NMH;   // dh = W_phi^T dqpre: (column, m-slice) per thread
    static __host__ __device__ constexpr int lds_floats() { return 3 * HS + PS_M + TH + PS_M + TH * PS_KLD; }

    static __device__ void run(const PersistBwdArgs& a, float* smem, const int widx) {
        const int ns = a.ns;
        const int b = widx / ns, half = widx % ns;          // `half`: which slice of the T' frames
        const bool first_wg = widx == 0;
        const int tid = threadIdx.x;
        const int B = a.B, U = a.U, Tp = a.Tp;
        const int th = (Tp + ns - 1) / ns, t0 = half * th, nt = max(0, min(th, Tp - t0));     // this workgroup's frames
        float* dctx = smem;
        float* ctxs = dctx + HS;
        float* qs = ctxs + HS;
        float* de = qs + PS_M;
        float* dqpre = de + TH;
        float* dhl = dqpre + PS_M;
        float* ks = dhl + HS;
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};

        // ---- resident operands: listener features of its frames (registers), keys (LDS), W_phi columns (registers)
        const int tl = tid >> 4, l16 = tid & 15;
        f32x4 fr[RP][NF4];
#pragma unroll
        for (int p = 0; p < RP; ++p) {
            const int t = tl + 64 * p;
#pragma unroll
            for (int i = 0; i < NF4; ++i) {
                const f32x4 v = ld4p(a.feat + ((size_t)b * Tp + (t < nt ? t0 + t : 0)) * D + 4 * (l16 + 16 * i));
                fr[p][i] = t < nt ? v : zero;
            }
        }
        for (int idx = tid; idx < nt * (PS_M / 4); idx += PS_THREADS) {
            const int t = idx / (PS_M / 4), m4 = idx % (PS_M / 4);
            *reinterpret_cast<f32x4*>(ks + t * PS_KLD + m4 * 4) = ld4p(a.keys + ((size_t)b * Tp + t0 + t) * PS_M + m4 * 4);
        }
        const int pmh = tid % NMH, pc = tid / NMH;        // the NMH lanes of a column are adjacent: DPP sum, no LDS
        float wph[MPT];
#pragma unroll
        for (int i = 0; i < MPT; ++i) wph[i] = a.w_phi[(size_t)(pmh * MPT + i) * HS + pc];
        lds_barrier();

        for (int s = U - 1; s >= 0; --s) {
            // ---- stash operands of this step (plain loads, issued before the wait)
            float at[RP];
#pragma unroll
            for (int p = 0; p < RP; ++p) {
                const int t = tl + 64 * p;
                at[p] = t < nt ? a.att[((size_t)s * B + b) * Tp + t0 + t] : 0.f;
            }
            if (tid < PS_M) qs[tid] = a.q_all[((size_t)s * B + b) * PS_M + tid];
            f32x4 dcv = zero;
            if (tid < HS / 4) {
                *reinterpret_cast<f32x4*>(ctxs + tid * 4) = ld4p(a.ctx_all + ((size_t)(s + 1) * B + b) * D + tid * 4);
                dcv = ld4p(a.dcat_all + ((size_t)s * B + b) * 2 * HS + HS + tid * 4);
            }
            PB_STAMP(2, s, 0);
            // ---- total context gradient = character-distribution part + what step s+1's bottom cell passed back
            if (tid < HS / 4) {
                f32x4 v = zero;
                if (s < U - 1) {
                    const float* p = a.dcx + (size_t)(s + 1) * ((size_t)(HS / 16) * 32 * 16) + ((size_t)(tid >> 2) * 32 + b) * 16 + (tid & 3) * 4;
                    unsigned spins = 0;
                    for (;;) {
                        v = ld4_agent(p);
                        if (!__any(has_sentinel(v))) break;
                        if (spin_expired(spins, a.err, 0xDEAD0024u)) break;
                    }
                }
                v[0] += dcv[0]; v[1] += dcv[1]; v[2] += dcv[2]; v[3] += dcv[3];
                *reinterpret_cast<f32x4*>(dctx + tid * 4) = v;
                if (half == 0) *reinterpret_cast<f32x4*>(a.dctx_all + ((size_t)s * B + b) * D + tid * 4) = v;
            }
            PB_STAMP(2, s, 1);
            lds_barrier();
            // ---- da_t = feat_t . dctx ; softmax backward de_t = a_t (da_t - ctx . dctx)   (16 lanes per frame)
            {
                float sd = 0.f, da[RP];
#pragma unroll
                for (int p = 0; p < RP; ++p) da[p] = 0.f;
#pragma unroll
                for (int i = 0; i < NF4; ++i) {
                    // lane l16 owns the float4 columns l16, l16+16, ...: consecutive lanes hit consecutive LDS banks
                    const f32x4 dv = *reinterpret_cast<const f32x4*>(dctx + 4 * (l16 + 16 * i));
                    sd = dot4p(*reinterpret_cast<const f32x4*>(ctxs + 4 * (l16 + 16 * i)), dv, sd);
#pragma unroll
                    for (int p = 0; p < RP; ++p) da[p] = dot4p(fr[p][i], dv, da[p]);
                }
                sd = gsum<16>(sd);
#pragma unroll
                for (int p = 0; p < RP; ++p) {
                    const float v = at[p] * (gsum<16>(da[p]) - sd);
                    const int t = tl + 64 * p;
                    if (l16 == 0) {
                        de[t] = v;                                           // zero beyond nt (at = 0)
                        if (t < nt) a.de_all[((size_t)s * B + b) * Tp + t0 + t] = v;
                    }
                }
            }
            lds_barrier();
            PB_STAMP(2, s, 2);
            // ---- dq[m] = sum_t de_t keys[t][m] over this slice's frames: 16 adjacent lanes per m, DPP row sum
            {
                const int m = tid >> 4, tg = tid & 15;
                float acc = 0.f;
                for (int t = tg; t < nt; t += 16) acc = fmaf(de[t], ks[t * PS_KLD + m], acc);
                acc = gsum<16>(acc);
                if (tg == 0) {
                    acc *= act_grad(qs[m], a.relu);      // (qs: the POST-activation query; relu -> 0 / 1, tanh -> 1 - q^2, sigmoid -> q (1 - q))
                    dqpre[m] = acc;
                    a.dqpre_part[(((size_t)half * U + s) * B + b) * PS_M + m] = acc;
                }
            }
            lds_barrier();
            PB_STAMP(2, s, 3);
            // ---- decoder-state gradient part W_phi^T dqpre: NMH adjacent lanes per column, published straight from registers
            {
                float acc = 0.f;
#pragma unroll
                for (int i = 0; i < MPT; ++i) acc = fmaf(wph[i], dqpre[pmh * MPT + i], acc);
                acc = gsum<NMH>(acc);
                if (NMH == 2) {          // a wave's 32 columns are one whole 128-byte line
                    if (pmh == 0) st1_agent(a.dhA + (((size_t)s * B + b) * ns + half) * HS + pc, acc);
                } else {                 // gather through LDS so that the publication still moves whole lines
                    if (pmh == 0) dhl[pc] = acc;
                    lds_barrier();
                    if (tid < HS / 4)
                        st4_agent(a.dhA + (((size_t)s * B + b) * ns + half) * HS + tid * 4, *reinterpret_cast<const f32x4*>(dhl + tid * 4));
                }
            }
            PB_STAMP(2, s, 4);
        }
    }
};


// ------------------------------------------------------------------------------------------------ attention backward, PRE variant
// The context gradient never materialises on the chain: with P = feat . W_ctx^T resident (its frames x all 4Hs gate columns,
// 32 lanes per frame), da_t = dcat_ctx . feat_t + dG0_{s+1} . P_t — the first term is a batched GEMM before the launch (e0),
// the second a register-resident contraction with the 4Hs-long gate gradient row of this utterance that the X workgroups
// just published.  The softmax-backward statistic ctx_s . dctx_s = sum_t a_t e0_t + dG0_{s+1} . gx_s uses the forward's
// hand-off slab gx_s = sum_t a_t P_t as one more "frame", so the four frame slices of an utterance still never talk.
template <int HS>
struct AttnBwdPreRole {
    static constexpr int GC = 4 * HS;                      // length of a P row / of a gate-gradient row
    static constexpr int LPS = HS / 16;                    // lanes per frame slot: 32 (Hs=512) or 16 (Hs=256)
    static constexpr int NSLOT = PS_THREADS / LPS;         // 32 or 64 slots; the last one holds the forward's gx row
    static constexpr int NC4 = GC / LPS / 4;               // float4 per lane: 16 (64 VGPRs of P)
    static constexpr int TH = NSLOT - 1;                   // frames per workgroup: 31 or 63
    static constexpr int MAXTP = 512;                      // rows of attention weights / e0 kept in LDS (T' <= 448)
    static constexpr int NJ = HS / 16;                     // producer tiles of a gate-gradient row (256 B each)
    static constexpr int NMH = PS_THREADS / HS, MPT = PS_M / NMH;
    static constexpr int WLD = PS_M + 4;                   // LDS row stride of a W_phi column (conflict-free 16-byte reads)
    static_assert(LPS == 16 || LPS == 32, "slot layout");
    static __host__ __device__ constexpr int lds_floats() { return GC + 2 * MAXTP + PS_M + 64 + PS_M + HS + 64 + TH * PS_KLD + HS * WLD; }

    static __device__ void run(const PersistBwdArgs& a, float* smem, const int widx) {
        const int NS = a.ns;                                // frame slices (workgroups) per utterance: 4, 8 or 16
        const int b = widx / NS, part = widx % NS;
        const bool first_wg = widx == 0;
        const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave: scalar register
        const int B = a.B, U = a.U, Tp = a.Tp;
        const int th = (Tp + NS - 1) / NS, t0 = part * th, nt = max(0, min(th, Tp - t0));     // this workgroup's frames
        float* dg = smem;
        float* attr = dg + GC;
        float* e0r = attr + MAXTP;
        float* qs = e0r + MAXTP;
        float* de = qs + PS_M;
        float* dqpre = de + 64;
        float* dhl = dqpre + PS_M;
        float* slotv = dhl + HS;
        float* ks = slotv + 64;
        float* wl = ks + TH * PS_KLD;            // W_phi, column-major: the P rows take the registers
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};

        // ---- resident operands: P rows of its frames (registers), keys of its frames and W_phi (LDS)
        const int slot = tid / LPS, l32 = tid % LPS;
        f32x4 pr[NC4];
#pragma unroll
        for (int i = 0; i < NC4; ++i) {
            const f32x4 v = ld4p(a.pctx + ((size_t)b * Tp + (slot < nt ? t0 + slot : 0)) * GC + 4 * (l32 + LPS * i));
            pr[i] = slot < nt ? v : zero;
        }
        for (int idx = tid; idx < nt * (PS_M / 4); idx += PS_THREADS) {
            const int t = idx / (PS_M / 4), m4 = idx % (PS_M / 4);
            *reinterpret_cast<f32x4*>(ks + t * PS_KLD + m4 * 4) = ld4p(a.keys + ((size_t)b * Tp + t0 + t) * PS_M + m4 * 4);
        }
        const int pmh = tid % NMH, pc = tid / NMH;        // the NMH lanes of a column are adjacent: DPP sum, no LDS
        for (int idx = tid; idx < PS_M * HS; idx += PS_THREADS) wl[(idx % HS) * WLD + idx / HS] = a.w_phi[idx];
        lds_barrier();

        for (int s = U - 1; s >= 0; --s) {
            // ---- stash operands of this step (plain loads, issued before the wait)
            // (wave-uniform base + opaque 32-bit lane offset: otherwise one 64-bit address pair per access is hoisted out
            //  of the step loop and spilled — the P rows leave no registers for that)
            const size_t sb = (size_t)s * B + b;
            if (tid < MAXTP) {
                const unsigned o = opaque(4u * (unsigned)(tid < Tp ? tid : 0));
                const float av = *at_bytes(a.att + sb * Tp, o), ev = *at_bytes(a.e0 + sb * Tp, o);
                attr[tid] = tid < Tp ? av : 0.f;
                e0r[tid] = tid < Tp ? ev : 0.f;
            }
            if (tid >= MAXTP && tid < MAXTP + PS_M) qs[tid - MAXTP] = *at_bytes(a.q_all + sb * PS_M, opaque(4u * (unsigned)(tid - MAXTP)));
            if (slot == NSLOT - 1) {
                const float* gp = at_bytes(a.gxf + sb * GC, opaque(16u * (unsigned)l32));
#pragma unroll
                for (int i = 0; i < NC4; ++i) pr[i] = ld4p(gp + 4 * LPS * i);
            }
            PB_STAMP(2, s, 0);
            // ---- the gate gradients of step s+1's bottom cell, row b: one 256-byte piece per X workgroup of this M-tile
            if (s < U - 1) {
                const float* slab = a.dGx + (size_t)(s + 1) * ((size_t)NJ * 32 * 64);
#ifndef PB_DIRECT_POLL      // -DPB_DIRECT_POLL: the eight loading waves poll the 8 KB row itself (one fabric round trip less, eight
                           // times the polling traffic) — measured not faster: 4 265 vs 4 280 utt/s
                if (wave == 0) {
                    const unsigned* cp = reinterpret_cast<const unsigned*>(at_bytes(slab, opaque(4u * (((unsigned)(lane < NJ ? lane : 0) * 32 + b) * 64 + 63))));
                    unsigned spins = 0;
                    for (;;) {
                        const unsigned v = __hip_atomic_load(cp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (!__any(lane < NJ && v == PS_SENT)) break;
                        if (spin_expired(spins, a.err, 0xDEAD0029u)) break;
                    }
                }
                lds_barrier();
#endif
                if (tid < GC / 4) {
                    const float* src = at_bytes(slab, opaque(4u * (((unsigned)(tid >> 4) * 32 + b) * 64 + 4 * (tid & 15))));
                    unsigned spins = 0;
                    f32x4 v;
                    for (;;) {
                        v = ld4_agent(src);
                        if (!__any(has_sentinel(v))) break;
                        if (spin_expired(spins, a.err, 0xDEAD002Au)) break;
                    }
                    *reinterpret_cast<f32x4*>(dg + 4 * tid) = v;
                }
            } else if (tid < GC / 4) {
                *reinterpret_cast<f32x4*>(dg + 4 * tid) = zero;
            }
            PB_STAMP(2, s, 1);
            lds_barrier();
            // ---- dG0 . P_t for its frames (and dG0 . gx_s in slot 31): 32 lanes per slot
            {
                float acc = 0.f;
#pragma unroll
                for (int i = 0; i < NC4; ++i) acc = dot4p(pr[i], *reinterpret_cast<const f32x4*>(dg + 4 * (l32 + LPS * i)), acc);
                acc = gsum<16>(acc);
                if (LPS == 32) {
                    const float s0 = lane_f(acc, 0) + lane_f(acc, 16), s1 = lane_f(acc, 32) + lane_f(acc, 48);
                    if (lane == 0) { slotv[2 * wave] = s0; slotv[2 * wave + 1] = s1; }
                } else if ((lane & 15) == 0) {
                    slotv[4 * wave + (lane >> 4)] = acc;
                }
            }
            lds_barrier();
            // ---- softmax backward de_t = a_t (e0_t + dG0 . P_t - ctx . dctx) for its frames (one wave)
            if (wave == 0) {
                float cp = 0.f;
#pragma unroll
                for (int k = 0; k < MAXTP / 64; ++k) cp = fmaf(attr[lane + 64 * k], e0r[lane + 64 * k], cp);
                const float c0 = wsum(cp);
                const float sd = c0 + slotv[NSLOT - 1];
                {
                    const bool valid = lane < nt;
                    const int tt = valid ? t0 + lane : 0;
                    const float v = valid ? attr[tt] * (e0r[tt] + slotv[lane] - sd) : 0.f;
                    de[lane] = v;
                    if (valid) *at_bytes(a.de_all + sb * Tp, opaque(4u * (unsigned)tt)) = v;
                }
            }
            lds_barrier();
            PB_STAMP(2, s, 2);
            // ---- dq[m] = sum_t de_t keys[t][m] over this slice's frames: 16 adjacent lanes per m, DPP row sum
            {
                const int m = tid >> 4, tg = tid & 15;
                float acc = 0.f;
                for (int t = tg; t < nt; t += 16) acc = fmaf(de[t], ks[t * PS_KLD + m], acc);
                acc = gsum<16>(acc);
                if (tg == 0) {
                    acc *= act_grad(qs[m], a.relu);      // (qs: the POST-activation query; relu -> 0 / 1, tanh -> 1 - q^2, sigmoid -> q (1 - q))
                    dqpre[m] = acc;
                    *at_bytes(a.dqpre_part + (((size_t)part * U + s) * B + b) * PS_M, opaque(4u * (unsigned)m)) = acc;
                }
            }
            lds_barrier();
            PB_STAMP(2, s, 3);
            // ---- decoder-state gradient part W_phi^T dqpre: NMH adjacent lanes per column, published straight from registers
            {
                float acc = 0.f;
                const float* wc = wl + pc * WLD + pmh * MPT;
#pragma unroll
                for (int i = 0; i < MPT; i += 4)
                    acc = dot4p(*reinterpret_cast<const f32x4*>(wc + i), *reinterpret_cast<const f32x4*>(dqpre + pmh * MPT + i), acc);
                acc = gsum<NMH>(acc);
                if (NMH == 2) {          // a wave's 32 columns are one whole 128-byte line
                    if (pmh == 0) st1_agent(at_bytes(a.dhA + (sb * NS + part) * HS, opaque(4u * (unsigned)pc)), acc);
                } else {                 // gather through LDS so that the publication still moves whole lines
                    if (pmh == 0) dhl[pc] = acc;
                    lds_barrier();
                    if (tid < HS / 4)
                        st4_agent(at_bytes(a.dhA + (sb * NS + part) * HS, opaque(16u * (unsigned)tid)), *reinterpret_cast<const f32x4*>(dhl + tid * 4));
                }
            }
            PB_STAMP(2, s, 4);
        }
    }
};

// ------------------------------------------------------------------------------------------------ attention backward, PRE variant, round 3
// The attention workgroups also own the TOP cell's pointwise backward, which removes a role (RY), a 64-producer fan-in and a hop
// from the chain:  A -> X -> A  instead of  A -> RY -> X -> A.
//   stage 1 (frame slice, as AttnBwdPreRole): gate-gradient row of step s+1 -> dG0 . P_t, softmax backward, de_t, this slice's part
//           of dq (masked: the relu mask is linear) -> exchanged between the ns workgroups of the utterance (256 bytes each);
//   stage 2 (UNIT slice: Hs / ns hidden units of the top layer, one per lane): dq = sum of the parts, decoder-state gradient
//           W_phi^T dq for its units (its W_phi columns live in LDS), + dz W_c part + recurrent carry (R workgroups) -> top cell
//           backward (cell-state gradient in a register) -> its piece of the tiled dG1 slab, whole 256-byte rows, straight to X.
template <int HS, bool MH = false>
struct AttnBwdPre2Role {
    static constexpr int GC = 4 * HS;                      // length of a P row / of a gate-gradient row
    static constexpr int LPS = HS / 16;                    // lanes per frame slot: 32 (Hs=512) or 16 (Hs=256)
    static constexpr int NSLOT = PS_THREADS / LPS;         // 32 or 64 slots; the last one holds the forward's gx row
    static constexpr int NC4 = GC / LPS / 4;               // float4 per lane: 16 (64 VGPRs of P)
    static constexpr int TH = NSLOT - 1;                   // frames per workgroup: 31 or 63
    static constexpr int MAXTP = HS == 256 ? 1024 : 512;   // rows of attention weights / e0 kept in LDS (T' <= 448; Hs = 256: <= 896, the frame-split forward's range)
    static constexpr int QBASE = MAXTP < PS_THREADS ? MAXTP : 0;      // first of the PS_M lanes that fetch the step's query
    static constexpr int NJ = HS / 16;                     // producer tiles of a gate-gradient row (256 B each)
    static constexpr int MAXUN = HS / 4;                   // units per workgroup at ns = 4 (fewer with more slices)
    static constexpr int WLD = PS_M + 4;                   // LDS row stride of a unit's W_phi column (16-byte aligned, bank spread)
    static constexpr int MAXNH = MH ? 4 : 1;               // heads whose summed dq a workgroup keeps (multi-head: its units need all of them)
    static_assert(LPS == 16 || LPS == 32, "slot layout");
    static __host__ __device__ constexpr int lds_floats() {
        return GC + 2 * MAXTP + PS_M + 64 + PS_M + 64 + TH * PS_KLD + WLD * MAXUN + 16 * PS_M + MAXNH * PS_M + 9 * MAXUN;
    }

    static __device__ void run(const PersistBwdArgs& a, float* smem, const int widx) {
        const int NS = a.ns;                                // workgroups per utterance: 4, 8 or 16
        // The NS workgroups of an utterance exchange their parts of dq in every step: they are placed on ONE XCD when the batch allows it
        // (block index mod 8 = XCD under round-robin dispatch; this role starts at a multiple of 8), so that the exchange can travel
        // through that XCD's L2 (~0.5 us) instead of through memory (~1 us): verified at run time below, never assumed
        const bool xmap = (a.B & 7) == 0;
        const int b = xmap ? ((widx >> 3) / NS) * 8 + (widx & 7) : widx / NS, part = xmap ? (widx >> 3) % NS : widx % NS;
        const bool first_wg = widx == 0;
        const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave: scalar register
        const int B = a.B, U = a.U, Tp = a.Tp;
        // multi-head: the NS workgroups of an utterance are NH heads x NSF frame slices; a workgroup works on ITS head's attention and on
        // Hs / NS top-layer units, for which it needs the summed dq of every head
        const int NH = MH ? a.NH : 1, NSF = MH ? NS / NH : NS;
        const int hd = MH ? part / NSF : 0, fsl = MH ? part % NSF : part;
        const int WLDH = MH ? NH * PS_M + 4 : WLD;          // LDS row stride of a unit's W_phi columns (all heads)
        const int th = (Tp + NSF - 1) / NSF, t0 = fsl * th, nt = max(0, min(th, Tp - t0));     // this workgroup's frames
        const int UN = HS / NS, u0 = part * UN;             // ... and its hidden units of the top layer
        float* dg = smem;
        float* attr = dg + GC;
        float* e0r = attr + MAXTP;
        float* qs = e0r + MAXTP;
        float* de = qs + PS_M;
        float* dqpre = de + 64;
        float* slotv = dqpre + PS_M;
        float* ks = slotv + 64;
        float* wps = ks + TH * PS_KLD;           // W_phi columns of its units: [unit][WLD]
        float* dqp = wps + WLD * MAXUN;          // the ns parts of dq as they arrive: [part][m]
        float* dqf = dqp + 16 * PS_M;            // their sum (per head)
        float* stl = dqf + MAXNH * PS_M;                 // stash of its units for this step: [i, f, g, o, c, c_prev, dz W_c part, dc][MAXUN]
        float* dhl = stl + 8 * MAXUN;            // W_phi^T dq of its units
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};

        // ---- run-time placement check: every workgroup of the utterance publishes its XCC id (agent scope) in the unused last step of
        //      the carry slab (sentinel-prefilled) and reads the others'
        bool l2x = false;
        if (xmap) {
            unsigned xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcc));
            unsigned* ids = reinterpret_cast<unsigned*>(a.dhc + (size_t)U * ((size_t)(HS / 16) * 32 * 16)) + b * 16;
            volatile int* flag = reinterpret_cast<volatile int*>(dqf);
            if (tid == 0) { *flag = 1; __hip_atomic_store(ids + part, 0xC0DE0000u | xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
            lds_barrier();
            if (tid < NS) {
                unsigned spins = 0, v;
                for (;;) {
                    v = __hip_atomic_load(ids + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (v != PS_SENT) break;
                    if (spin_expired(spins, a.err, 0xDEAD002Cu)) break;
                }
                if (v != (0xC0DE0000u | xcc)) *flag = 0;
            }
            lds_barrier();
            l2x = *flag != 0;
            lds_barrier();
        }
        // ---- resident operands: P rows of its frames (registers), keys of its frames and the W_phi columns of its units (LDS)
        const int slot = tid / LPS, l32 = tid % LPS;
        f32x4 pr[NC4];
#pragma unroll
        for (int i = 0; i < NC4; ++i) {
            const f32x4 v = ld4p(a.pctx + (((size_t)b * Tp + (slot < nt ? t0 + slot : 0)) * NH + hd) * GC + 4 * (l32 + LPS * i));
            pr[i] = slot < nt ? v : zero;
        }
        for (int idx = tid; idx < nt * (PS_M / 4); idx += PS_THREADS) {
            const int t = idx / (PS_M / 4), m4 = idx % (PS_M / 4);
            *reinterpret_cast<f32x4*>(ks + t * PS_KLD + m4 * 4) = ld4p(a.keys + ((size_t)b * Tp + t0 + t) * PS_M + m4 * 4);
        }
        for (int idx = tid; idx < NH * PS_M * UN; idx += PS_THREADS) wps[(idx % UN) * WLDH + idx / UN] = a.w_phi[(size_t)(idx / UN) * HS + u0 + idx % UN];
        lds_barrier();
        constexpr size_t GXS = (size_t)NJ * 32 * 64, CXS = (size_t)NJ * 32 * 16;
        const bool ulane = tid < UN;                         // stage-2 lane: hidden unit u0 + tid of utterance b
        if (ulane) stl[7 * MAXUN + tid] = 0.f;                  // cell-state gradient of that (utterance, unit): lives in LDS between the steps

        for (int s = U - 1; s >= 0; --s) {
            // ---- stash operands of this step (plain loads, issued before the wait)
            const size_t sb = (size_t)s * B + b;
            const size_t sbh = MH ? ((size_t)s * NH + hd) * B + b : sb;       // [U][NH][B] rows (attention weights, e0, de)
            const size_t sbq = MH ? sb * NH + hd : sb;                         // [U][B][NH] rows (queries, gx)
            if (tid < MAXTP) {
                const unsigned o = opaque(4u * (unsigned)(tid < Tp ? tid : 0));
                const float av = *at_bytes(a.att + sbh * Tp, o), ev = *at_bytes(a.e0 + sbh * Tp, o);
                attr[tid] = tid < Tp ? av : 0.f;
                e0r[tid] = tid < Tp ? ev : 0.f;
            }
            if (tid >= QBASE && tid < QBASE + PS_M) qs[tid - QBASE] = *at_bytes(a.q_all + sbq * PS_M, opaque(4u * (unsigned)(tid - QBASE)));
            if (slot == NSLOT - 1) {
                const float* gp = at_bytes(a.gxf + sbq * GC, opaque(16u * (unsigned)l32));
#pragma unroll
                for (int i = 0; i < NC4; ++i) pr[i] = ld4p(gp + 4 * LPS * i);
            }
            if (ulane) {      // top-layer stash of its units -> LDS (the workgroup is about to wait for the gate-gradient row anyway)
                const unsigned u = opaque((unsigned)tid);
                const size_t slab = ((size_t)U + s) * (size_t)B * HS;
                const float* gp = at_bytes(a.gates_all + 4 * slab, 4u * ((unsigned)b * 4 * HS + u0 + u));
                stl[0 * MAXUN + u] = gp[0]; stl[1 * MAXUN + u] = gp[HS]; stl[2 * MAXUN + u] = gp[2 * HS]; stl[3 * MAXUN + u] = gp[3 * HS];
                stl[4 * MAXUN + u] = *at_bytes(a.c_all + slab, 4u * ((unsigned)b * HS + u0 + u));
                stl[5 * MAXUN + u] = s > 0 ? *at_bytes(a.c_all + slab - (size_t)B * HS, 4u * ((unsigned)b * HS + u0 + u)) : 0.f;
                stl[6 * MAXUN + u] = *at_bytes(a.dcat_all + sb * 2 * HS, 4u * (unsigned)(u0 + u));
            }
            PB_STAMP(2, s, 0);
            // ---- the gate gradients of step s+1's bottom cell, row b: one 256-byte piece per X workgroup of this M-tile
            if (s < U - 1) {
                const float* slab = a.dGx + (size_t)(s + 1) * GXS;
                if (wave == 0) {
                    const unsigned* cp = reinterpret_cast<const unsigned*>(at_bytes(slab, opaque(4u * (((unsigned)(lane < NJ ? lane : 0) * 32 + b) * 64 + 63))));
                    unsigned spins = 0;
                    for (;;) {
                        const unsigned v = __hip_atomic_load(cp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (!__any(lane < NJ && v == PS_SENT)) break;
                        if (spin_expired(spins, a.err, 0xDEAD0029u)) break;
                    }
                }
                lds_barrier();
                if (tid < GC / 4) {
                    const float* src = at_bytes(slab, opaque(4u * (((unsigned)(tid >> 4) * 32 + b) * 64 + 4 * (tid & 15))));
                    unsigned spins = 0;
                    f32x4 v;
                    for (;;) {
                        v = ld4_agent(src);
                        if (!__any(has_sentinel(v))) break;
                        if (spin_expired(spins, a.err, 0xDEAD002Au)) break;
                    }
                    *reinterpret_cast<f32x4*>(dg + 4 * tid) = v;
                }
            } else if (tid < GC / 4) {
                *reinterpret_cast<f32x4*>(dg + 4 * tid) = zero;
            }
            PB_STAMP(2, s, 1);
            lds_barrier();
            // ---- dG0 . P_t for its frames (and dG0 . gx_s in the last slot)
            {
                // two partial sums, the row read four float4 at a time (the allocator otherwise funnels all sixteen reads through ONE
                // register quad: read, wait, four FMAs, sixteen times in a row)
                float acc = 0.f, acc1 = 0.f;
#pragma unroll
                for (int i = 0; i < NC4; i += 4) {
                    const f32x4 d0 = *reinterpret_cast<const f32x4*>(dg + 4 * (l32 + LPS * i));
                    const f32x4 d1 = *reinterpret_cast<const f32x4*>(dg + 4 * (l32 + LPS * (i + 1)));
                    const f32x4 d2 = *reinterpret_cast<const f32x4*>(dg + 4 * (l32 + LPS * (i + 2)));
                    const f32x4 d3 = *reinterpret_cast<const f32x4*>(dg + 4 * (l32 + LPS * (i + 3)));
                    acc = dot4p(pr[i], d0, acc); acc1 = dot4p(pr[i + 1], d1, acc1);
                    acc = dot4p(pr[i + 2], d2, acc); acc1 = dot4p(pr[i + 3], d3, acc1);
                }
                acc = gsum<16>(acc + acc1);
                if (LPS == 32) {
                    const float s0 = lane_f(acc, 0) + lane_f(acc, 16), s1 = lane_f(acc, 32) + lane_f(acc, 48);
                    if (lane == 0) { slotv[2 * wave] = s0; slotv[2 * wave + 1] = s1; }
                } else if ((lane & 15) == 0) {
                    slotv[4 * wave + (lane >> 4)] = acc;
                }
            }
            // the recurrent carry of its units (R workgroups, published ~2 us ago): one agent-scope dword per stage-2 lane, in flight
            // under the softmax backward / dq phases
            unsigned carry_bits = 0u;
            auto carry_src = [&]() {       // re-derived where it is used: as a loop-invariant pointer it would hold two registers for the whole step
                const unsigned un = (unsigned)u0 + (ulane ? opaque((unsigned)tid) : 0u);
                return reinterpret_cast<const unsigned*>(at_bytes(a.dhc + (size_t)(s + 1 < U ? s + 1 : s) * CXS, 4u * (((un >> 4) * 32 + b) * 16 + (un & 15))));
            };
            if (ulane && s < U - 1) carry_bits = __hip_atomic_load(carry_src(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            lds_barrier();
            // ---- softmax backward de_t = a_t (e0_t + dG0 . P_t - ctx . dctx) for its frames (one wave)
            if (wave == 0) {
                float cp = 0.f;
                if (HS == 256) {      // (a 1024-row table: walk the rows that exist)
                    const int kmax = (Tp + 63) >> 6;
                    for (int k = 0; k < kmax; ++k) cp = fmaf(attr[lane + 64 * k], e0r[lane + 64 * k], cp);
                } else {
#pragma unroll
                    for (int k = 0; k < MAXTP / 64; ++k) cp = fmaf(attr[lane + 64 * k], e0r[lane + 64 * k], cp);
                }
                const float c0 = wsum(cp);
                const float sd = c0 + slotv[NSLOT - 1];
                {
                    const bool valid = lane < nt;
                    const int tt = valid ? t0 + lane : 0;
                    const float v = valid ? attr[tt] * (e0r[tt] + slotv[lane] - sd) : 0.f;
                    de[lane] = v;
                    if (valid) *at_bytes(a.de_all + sbh * Tp, opaque(4u * (unsigned)tt)) = v;
                }
            }
            lds_barrier();
            PB_STAMP(2, s, 2);
            // ---- this slice's part of dq[m] = sum_t de_t keys[t][m] (masked): 16 adjacent lanes per m, DPP row sum
            {
                const int m = tid >> 4, tg = tid & 15;
                float acc = 0.f;
                for (int t = tg; t < nt; t += 16) acc = fmaf(de[t], ks[t * PS_KLD + m], acc);
                acc = gsum<16>(acc);
                if (tg == 0) {
                    acc *= act_grad(qs[m], a.relu);      // (qs: the POST-activation query; relu -> 0 / 1, tanh -> 1 - q^2, sigmoid -> q (1 - q))
                    dqpre[m] = acc;
                }
            }
            lds_barrier();
            // ---- exchange: publish its part (256 bytes, two whole lines), collect all ns parts of this utterance
            float* xs = a.dqx + sb * NS * PS_M;
            if (tid < PS_M / 4) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(dqpre + tid * 4);
                float* dst = at_bytes(xs + part * PS_M, opaque(16u * (unsigned)tid));
                if (l2x) {      // same XCD: an ordinary store reaches the shared L2, where the partners' L1-bypassing polls find it
                    f32x4 c;
#pragma unroll
                    for (int i = 0; i < 4; ++i) c[i] = __uint_as_float(pub_bits(v[i]));
                    *reinterpret_cast<f32x4*>(dst) = c;
                } else {
                    st4_agent(dst, v);
                }
            }
            PB_STAMP(2, s, 3);
            if (tid < NS * (PS_M / 4)) {      // lane = (part p, float4 i): ns * 16 lanes <= 256
                const float* src = at_bytes(xs, opaque(16u * (unsigned)tid));
                unsigned spins = 0;
                f32x4 v;
                for (;;) {
                    v = ld4_agent(src);
                    if (!__any(has_sentinel(v))) break;
                    if (spin_expired(spins, a.err, 0xDEAD002Bu)) break;
                }
                *reinterpret_cast<f32x4*>(dqp + tid * 4) = v;
            }
            lds_barrier();
            if (tid < NH * PS_M) {      // (multi-head: head tid / M sums its own NSF frame slices)
                const int h2 = MH ? tid / PS_M : 0, m = MH ? tid % PS_M : tid;
                float acc = 0.f;
                for (int p = 0; p < NSF; ++p) acc += dqp[(h2 * NSF + p) * PS_M + m];
                dqf[tid] = acc;
                if (part == 0) *at_bytes(a.dqpre_all + sb * (NH * PS_M), opaque(4u * (unsigned)tid)) = acc;
            }
            lds_barrier();
            PB_STAMP(2, s, 4);
            // ---- stage 2: decoder-state gradient of its units (W_phi^T dq: 8 adjacent lanes per unit, 8 terms each, DPP row sum) ...
            if (tid < 8 * UN) {
                const unsigned t8 = opaque((unsigned)tid), un = t8 >> 3, ms = t8 & 7;
                const float* wr = wps + un * WLDH + ms * 8;
                const float* qr = dqf + ms * 8;
                float acc = dot4p(*reinterpret_cast<const f32x4*>(wr), *reinterpret_cast<const f32x4*>(qr), 0.f);
                acc = dot4p(*reinterpret_cast<const f32x4*>(wr + 4), *reinterpret_cast<const f32x4*>(qr + 4), acc);
                if (MH) {
                    for (int h2 = 1; h2 < NH; ++h2) {
                        acc = dot4p(*reinterpret_cast<const f32x4*>(wr + h2 * PS_M), *reinterpret_cast<const f32x4*>(qr + h2 * PS_M), acc);
                        acc = dot4p(*reinterpret_cast<const f32x4*>(wr + h2 * PS_M + 4), *reinterpret_cast<const f32x4*>(qr + h2 * PS_M + 4), acc);
                    }
                }
                acc = gsum<8>(acc);
                if (ms == 0) dhl[un] = acc;
            }
            lds_barrier();
            // ... top cell backward, its piece of dG1_s
            if (ulane) {
                const unsigned u = opaque((unsigned)tid);
                float dh = stl[6 * MAXUN + u] + dhl[u];
                PB_STAMP(2, s, 6);
                if (s < U - 1) {
                    unsigned spins = 0;
                    while (__any(carry_bits == PS_SENT)) {
                        if (spin_expired(spins, a.err, 0xDEAD002Cu)) break;
                        carry_bits = __hip_atomic_load(carry_src(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    dh += __uint_as_float(carry_bits);
                }
                PB_STAMP(2, s, 7);
                typename ProdRole<HS>::CellIn ci;
                ci.ig = stl[0 * MAXUN + u]; ci.fg = stl[1 * MAXUN + u]; ci.gg = stl[2 * MAXUN + u]; ci.og = stl[3 * MAXUN + u];
                ci.c = stl[4 * MAXUN + u]; ci.cp = stl[5 * MAXUN + u];
                float dc1 = stl[7 * MAXUN + u];
                const f32x4 g = ProdRole<HS>::cell_bwd(ci, dh, dc1);
                stl[7 * MAXUN + u] = dc1;
                const unsigned un = (unsigned)u0 + u;
                st4_agent(at_bytes(a.dGx + ((size_t)U + s) * GXS, 4u * ((((un >> 4) * 32 + b) * 16 + (un & 15)) * 4)), g);
                float* dp = at_bytes(a.dG_all + 4 * (((size_t)U + s) * (size_t)B * HS), 4u * ((unsigned)b * 4 * HS + un));
                dp[0] = g[0]; dp[HS] = g[1]; dp[2 * HS] = g[2]; dp[3 * HS] = g[3];
            }
            PB_STAMP(2, s, 5);
        }
    }
};

template <int HS, bool MH = false>
__global__ __launch_bounds__(PS_THREADS) void speller_persist_bwd_pre_kernel(PersistBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NXY = (HS / 16) * 2;
    const int bx = blockIdx.x;
    if (bx < NXY) ProdPre2Role<HS>::template run<true>(a, smem, bx);
    else if (bx < 2 * NXY) ProdPre2Role<HS>::template run<false>(a, smem, bx - NXY);
    else AttnBwdPre2Role<HS, MH>::run(a, smem, bx - 2 * NXY);
}

template <int HS>
__global__ __launch_bounds__(PS_THREADS) void speller_persist_bwd_kernel(PersistBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NXY = (HS / 16) * 2;
    const int bx = blockIdx.x;
#if defined(PB_ONLY_PROD)
    ProdRole<HS>::template run<PB_ONLY_PROD>(a, smem, bx);
#elif defined(PB_ONLY_ATTN)
    AttnBwdRole<HS>::run(a, smem, bx);
#else
    if (bx < NXY) ProdRole<HS>::template run<0>(a, smem, bx);
    else if (bx < 2 * NXY) ProdRole<HS>::template run<1>(a, smem, bx - NXY);
    else if (bx < 3 * NXY) ProdRole<HS>::template run<2>(a, smem, bx - 2 * NXY);
    else AttnBwdRole<HS>::run(a, smem, bx - 3 * NXY);
#endif
}

// ------------------------------------------------------------------------------------------------ host side
static unsigned long long* g_persist_bwd_trace = nullptr;
void speller_persist_bwd_set_trace(unsigned long long* dev_buf) { g_persist_bwd_trace = dev_buf; }

// attention-backward workgroups per utterance: the smallest count whose 64*(512/Hs)-frame slices cover T' and that leaves
// every workgroup resident at once (one per CU); 0 = not applicable.  cus < 0: shape check only (workspace sizing).
static int persist_bwd_ns(int B, int Tp, int Hs, int cus) {
    for (int ns = 2; ns <= 8; ns *= 2)
        if (Tp <= ns * 64 * (512 / Hs) && (cus < 0 || 3 * (Hs / 16) * 2 + ns * B <= cus)) return ns;
    return 0;
}
static bool persist_bwd_shape(int B, int Hs, int D, int M, int L, int heads, int use_mlp) {
    return L == 2 && heads == 1 && use_mlp && M == PS_M && D == Hs && (Hs == 256 || Hs == 512) && B >= 1 && B <= 32;
}

bool speller_persist_bwd_eligible(int B, int Tp, int Hs, int D, int M, int V, int L, int heads, int use_mlp) {
    (void)V;
    if (!persist_bwd_shape(B, Hs, D, M, L, heads, use_mlp)) return false;
    int cus = 0, dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
        return false;
    return persist_bwd_ns(B, Tp, Hs, cus) != 0;
}

// attention-backward workgroups per utterance of the PRE variant: the smallest of 4 / 8 / 16 whose frame slots (31 at Hs=512,
// 63 at Hs=256) cover ceil(T'/ns) frames and that leaves every workgroup resident; 0 = n/a.  cus < 0: shape only.
static int persist_bwd_pre_ns(int B, int Tp, int Hs, int cus) {
    const int th = 1024 / (Hs / 16) - 1;
    for (int ns = 4; ns <= 16; ns *= 2)
        if ((Tp + ns - 1) / ns <= th && (cus < 0 || 2 * (Hs / 16) * 2 + ns * B <= cus)) return ns;
    return 0;
}
// floats of the hand-off slabs + the attention workgroups' dqpre parts; 0 when the shape is not covered.  The PRE variant
// (4 attention workgroups per utterance, no dcx / dhc slabs, plus the e0 rows) fits in the same block.
// multi-head form of the PRE variant: frame slices per head (0 = n/a) — B * heads takes B's place in the workgroup budget, every (head, slice)
// workgroup also owns Hs / (heads * slices) top-layer units (whole 16-unit tiles, at most 16 workgroups per utterance)
static int persist_bwd_pre_mh_nsf(int B, int Tp, int Hs, int heads, int cus) {
    const int nsf = persist_bwd_pre_ns(B * heads, Tp, Hs, cus);
    if (nsf == 0 || nsf * heads > 16 || (Hs / 16) % (nsf * heads) != 0) return 0;
    return nsf;
}
size_t speller_persist_bwd_mh_workspace_floats(int B, int Tp, int U, int Hs, int M, int heads) {
    if ((Hs != 256 && Hs != 512) || heads < 2 || heads > 4 || Tp > 448) return 0;
    const size_t nsf = (size_t)persist_bwd_pre_mh_nsf(B, Tp, Hs, heads, -1);
    if (nsf == 0) return 0;
    // [e0 (per head) | sentinel-prefilled slabs: dq exchange (heads * nsf parts) | tiled dG (2 layers) | recurrent carry of the top layer | R's partial sums]
    return (size_t)U * B * Tp * heads + 4 + nsf * heads * U * B * M + (size_t)2 * U * (Hs / 16) * 32 * 64 + (size_t)(U + 1) * (Hs / 16) * 32 * 16 +
           (size_t)(U + 1) * 2 * (Hs / 16) * 256;
}
bool speller_persist_bwd_pre_mh_eligible(int B, int Tp, int Hs, int D, int M, int V, int L, int heads, int use_mlp) {
    if (opt_get(OPT_SPELLER_PERSIST_BWD) == 0 || opt_get(OPT_SPELLER_PRE_BWD) == 0 || opt_get(OPT_SPELLER_PRE_MH) == 0 || Tp > 448) return false;
    if (!speller_persist_pre_mh_shape(B, Tp, Hs, D, M, V, L, heads, use_mlp)) return false;      // (the forward left P and gx for exactly these shapes)
    int cus = 0, dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
        return false;
    return persist_bwd_pre_mh_nsf(B, Tp, Hs, heads, cus) != 0;
}
size_t speller_persist_bwd_workspace_floats(int B, int Tp, int U, int Hs, int M) {
    if (Hs != 256 && Hs != 512) return 0;
    const int ns = persist_bwd_ns(B, Tp, Hs, -1);
    if (ns == 0) return 0;
    const size_t classic = (size_t)ns * U * B * M + (size_t)U * B * ns * Hs + (size_t)2 * U * (Hs / 16) * 32 * 64 + (size_t)2 * (U + 1) * (Hs / 16) * 32 * 16;
    const size_t nsp = (size_t)persist_bwd_pre_ns(B, Tp, Hs, -1);
    // PRE variant: [e0 | sentinel-prefilled slabs: dq exchange (ns parts) | tiled dG (2 layers) | recurrent carry of the top layer]
    const size_t pre = (nsp && Tp <= (Hs == 256 ? 896 : 448)) ? (size_t)U * B * Tp + 4 + nsp * U * B * M + (size_t)2 * U * (Hs / 16) * 32 * 64 + (size_t)(U + 1) * (Hs / 16) * 32 * 16 +
                                            (size_t)(U + 1) * 2 * (Hs / 16) * 256 : 0;
    return std::max(classic, pre);
}
bool speller_persist_bwd_pre_eligible(int B, int Tp, int Hs, int D, int M, int V, int L, int heads, int use_mlp) {
    const bool on = opt_get(OPT_SPELLER_PRE_BWD) != 0;
    if (!on || !persist_bwd_shape(B, Hs, D, M, L, heads, use_mlp) || Tp > (Hs == 256 ? 896 : 448)) return false;      // (Hs = 256: the frame-split forward's range)
    if (!speller_persist_pre_eligible(B, Tp, Hs, D, M, V, L, heads, use_mlp)) return false;      // the forward must have produced P and gx
    int cus = 0, dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
        return false;
    return persist_bwd_pre_ns(B, Tp, Hs, cus) != 0;
}

template <int HS, bool MH = false>
static int launch_persist_bwd_pre(const PersistBwdArgs& a, int grid, hipStream_t stream) {
    const size_t smem = sizeof(float) * (size_t)std::max(ProdRole<HS>::LDS_FLOATS, AttnBwdPre2Role<HS, MH>::lds_floats());
    LAS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&speller_persist_bwd_pre_kernel<HS, MH>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    if (!persistent_launch_fits(speller_persist_bwd_pre_kernel<HS, MH>, PS_THREADS, smem, grid))
        return fail(LAS_ERR_UNSUPPORTED, "persistent decode backward: %s%ld workgroups cannot all be resident", "", (long)grid);
    {
        KernelTimer timer(TIMED_DECODE_BWD, stream);
        hipLaunchKernelGGL((speller_persist_bwd_pre_kernel<HS, MH>), dim3(grid), dim3(PS_THREADS), smem, stream, a);
    }
    LAS_LAUNCH_CHECK();
    path_note(PATH_DECODE_BWD, MH ? "persist_pre_mh" : "persist_pre");
    return LAS_OK;
}

template <int HS>
static int launch_persist_bwd(const PersistBwdArgs& a, int grid, hipStream_t stream) {
    const size_t smem = sizeof(float) * (size_t)std::max(ProdRole<HS>::LDS_FLOATS, AttnBwdRole<HS>::lds_floats());
    LAS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&speller_persist_bwd_kernel<HS>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    if (!persistent_launch_fits(speller_persist_bwd_kernel<HS>, PS_THREADS, smem, grid))
        return fail(LAS_ERR_UNSUPPORTED, "persistent decode backward: %s%ld workgroups cannot all be resident", "", (long)grid);
    {
        KernelTimer timer(TIMED_DECODE_BWD, stream);
        hipLaunchKernelGGL((speller_persist_bwd_kernel<HS>), dim3(grid), dim3(PS_THREADS), smem, stream, a);
    }
    LAS_LAUNCH_CHECK();
    path_note(PATH_DECODE_BWD, "persist");
    return LAS_OK;
}

// Multi-head form of the PRE backward (heads 2..4; the forward left P (B*Tp, NH*4Hs) and gx [U][B][NH][4Hs]): the same roles, the attention
// workgroups of an utterance being NH heads x nsf frame slices.  p.dctxcat: U*B*NH*D floats of scratch (dz W_c's context part through dim_reduce).
static int speller_persist_bwd_mh(const PersistBwd& p, hipStream_t stream) {
    LAS_REQUIRE(p.err && p.xbuf && p.dqpre_all && p.pctx && p.gxf && p.w_dr && p.dctxcat, "persistent speller backward (multi-head) buffers");
    LAS_REQUIRE(speller_persist_pre_mh_shape(p.B, p.Tp, p.Hs, p.Hs, PS_M, p.V, 2, p.NH, 1), "persistent speller backward (multi-head) shape");
    int cus = 0, dev = 0;
    LAS_HIP_CHECK(hipGetDevice(&dev));
    LAS_HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    const int NH = p.NH, nsf = persist_bwd_pre_mh_nsf(p.B, p.Tp, p.Hs, NH, cus);
    if (nsf == 0) return fail(LAS_ERR_UNSUPPORTED, "persistent decode backward (multi-head): %s%ld heads do not fit the chip", "", (long)NH);
    PersistBwdArgs a;
    a.pctx = p.pctx; a.gxf = p.gxf; a.NH = NH; a.ns = nsf * NH;
    const size_t nq = (size_t)p.U * p.B * PS_M;
    float* e0 = p.xbuf;
    float* slabs = e0 + (((size_t)p.U * p.B * p.Tp * NH + 3) & ~(size_t)3);
    a.e0 = e0;
    a.dqx = slabs;
    a.dGx = a.dqx + (size_t)a.ns * nq;
    a.dhc = a.dGx + (size_t)2 * p.U * (p.Hs / 16) * 32 * 64;
    a.dhp = a.dhc + (size_t)(p.U + 1) * (p.Hs / 16) * 32 * 16;
    a.dcx = nullptr; a.dhA = nullptr; a.dqpre_part = nullptr;
    a.dqpre_all = p.dqpre_all;
    a.w_ih0 = p.w_ih0; a.ldw0 = p.V + p.Hs; a.V = p.V;
    a.w_hh0 = p.w_hh0; a.w_ih1 = p.w_ih1; a.w_hh1 = p.w_hh1; a.w_phi = p.w_phi;
    a.feat = p.feat; a.keys = p.keys; a.att = p.att; a.q_all = p.q_all; a.ctx_all = p.ctx_all;
    a.gates_all = p.gates_all; a.c_all = p.c_all; a.dcat_all = p.dcat_all;
    a.dG_all = p.dG_all; a.dctx_all = p.dctx_all; a.de_all = p.de_all;
    a.dx0 = p.dx0; a.ldx0 = p.V + p.Hs;
    a.B = p.B; a.Tp = p.Tp; a.U = p.U; a.relu = p.relu; a.err = p.err; a.trace = g_persist_bwd_trace;
    const size_t slab_floats = (size_t)a.ns * nq + (size_t)2 * p.U * (p.Hs / 16) * 32 * 64 + (size_t)(p.U + 1) * (p.Hs / 16) * 32 * 16 +
                               (size_t)(p.U + 1) * 2 * (p.Hs / 16) * 256;
    LAS_HIP_CHECK(hipMemsetAsync(slabs, 0xFF, sizeof(float) * slab_floats, stream));
    {   // the character distribution's context gradient through dim_reduce: (U*B, D) x (D, NH*D)
        GemmDesc g;
        g.A = p.dcat_all + p.Hs; g.lda = 2 * p.Hs; g.a_kc = true;
        g.B = p.w_dr; g.ldb = (long)NH * p.Hs; g.b_kc = false;
        g.C = p.dctxcat; g.ldc = (long)NH * p.Hs; g.M = p.U * p.B; g.N = NH * p.Hs; g.K = p.Hs; g.splitk = 1;
        LAS_TRY(gemm_f32(g, stream));
    }
    for (int hd = 0; hd < NH; ++hd) {   // e0[s][hd][b][t] = (that gradient's head block)[s][b] . feat[b][t]
        GemmDesc g;
        g.A = p.dctxcat + (size_t)hd * p.Hs; g.lda = (long)p.B * NH * p.Hs; g.a_kc = true; g.sA = (long)NH * p.Hs;
        g.B = p.feat; g.ldb = p.Hs; g.b_kc = true; g.sB = (long)p.Tp * p.Hs;
        g.C = e0 + (size_t)hd * p.B * p.Tp; g.ldc = (long)NH * p.B * p.Tp; g.sC = p.Tp;
        g.M = p.U; g.N = p.Tp; g.K = p.Hs; g.batch = p.B; g.splitk = 1;
        LAS_TRY(gemm_f32(g, stream));
    }
    const int grid = 2 * (p.Hs / 16) * 2 + a.ns * p.B;
    if (p.Hs == 512) LAS_TRY((launch_persist_bwd_pre<512, true>(a, grid, stream)));
    else LAS_TRY((launch_persist_bwd_pre<256, true>(a, grid, stream)));
    return LAS_OK;
}

int speller_persist_bwd(const PersistBwd& p, hipStream_t stream) {
    if (p.NH > 1) return speller_persist_bwd_mh(p, stream);
    LAS_REQUIRE(speller_persist_bwd_eligible(p.B, p.Tp, p.Hs, p.Hs, PS_M, p.V, 2, 1, 1), "persistent speller backward shape");
    LAS_REQUIRE(p.err != nullptr && p.xbuf != nullptr && p.dqpre_all != nullptr, "persistent speller backward buffers");
    int cus = 0, dev = 0;
    LAS_HIP_CHECK(hipGetDevice(&dev));
    LAS_HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    PersistBwdArgs a;
    a.NH = 1;
    a.pctx = p.pctx; a.gxf = p.gxf; a.e0 = nullptr;
    a.ns = p.pctx ? persist_bwd_pre_ns(p.B, p.Tp, p.Hs, cus) : persist_bwd_ns(p.B, p.Tp, p.Hs, cus);
    a.w_ih0 = p.w_ih0; a.ldw0 = p.V + p.Hs; a.V = p.V;
    a.w_hh0 = p.w_hh0; a.w_ih1 = p.w_ih1; a.w_hh1 = p.w_hh1; a.w_phi = p.w_phi;
    a.feat = p.feat; a.keys = p.keys; a.att = p.att; a.q_all = p.q_all; a.ctx_all = p.ctx_all;
    a.gates_all = p.gates_all; a.c_all = p.c_all; a.dcat_all = p.dcat_all;
    a.dG_all = p.dG_all; a.dctx_all = p.dctx_all; a.de_all = p.de_all;
    a.dx0 = p.dx0; a.ldx0 = p.V + p.Hs;
    // carve the workspace: [dqpre parts | sentinel-prefilled slabs: dhA | tiled dG (2 layers) | dcx | dhc]
    const size_t nq = (size_t)p.U * p.B * PS_M;
    a.dqpre_part = p.xbuf;
    if (p.pctx) {
        // PRE variant: [e0 | sentinel-prefilled slabs: dq exchange (ns parts) | tiled dG (2 layers) | top-layer recurrent carry]
        LAS_REQUIRE(p.gxf && a.ns != 0, "persistent speller backward (pre) buffers");
        float* e0 = p.xbuf;
        float* slabs = e0 + (((size_t)p.U * p.B * p.Tp + 3) & ~(size_t)3);
        a.e0 = e0;
        a.dqx = slabs;
        a.dGx = a.dqx + (size_t)a.ns * nq;
        a.dhc = a.dGx + (size_t)2 * p.U * (p.Hs / 16) * 32 * 64;
        a.dhp = a.dhc + (size_t)(p.U + 1) * (p.Hs / 16) * 32 * 16;
        a.dcx = nullptr; a.dhA = nullptr; a.dqpre_part = nullptr;
        a.dqpre_all = p.dqpre_all;
        a.w_ih0 = p.w_ih0; a.ldw0 = p.V + p.Hs; a.V = p.V;
        a.w_hh0 = p.w_hh0; a.w_ih1 = p.w_ih1; a.w_hh1 = p.w_hh1; a.w_phi = p.w_phi;
        a.feat = p.feat; a.keys = p.keys; a.att = p.att; a.q_all = p.q_all; a.ctx_all = p.ctx_all;
        a.gates_all = p.gates_all; a.c_all = p.c_all; a.dcat_all = p.dcat_all;
        a.dG_all = p.dG_all; a.dctx_all = p.dctx_all; a.de_all = p.de_all;
        a.dx0 = p.dx0; a.ldx0 = p.V + p.Hs;
        a.B = p.B; a.Tp = p.Tp; a.U = p.U; a.relu = p.relu; a.err = p.err; a.trace = g_persist_bwd_trace;
        // the hand-off slabs (92 MB of sentinel words at paper size) are filled on the side stream, beside the e0 GEMM
        const size_t slab_floats = (size_t)a.ns * nq + (size_t)2 * p.U * (p.Hs / 16) * 32 * 64 + (size_t)(p.U + 1) * (p.Hs / 16) * 32 * 16 +
                                   (size_t)(p.U + 1) * 2 * (p.Hs / 16) * 256;
        SideStream& side = side_stream();
        const bool side_fill = side.ok(stream);
        SideJoinGuard side_guard;
        if (side_fill) {
            LAS_TRY(side.fork(stream));
            side_guard.arm(side, stream);
            LAS_HIP_CHECK(hipMemsetAsync(slabs, 0xFF, sizeof(float) * slab_floats, side.s));
        }
        {   // e0[s][b][t] = dcat_ctx[s][b] . feat[b][t]: one batched GEMM over the utterances
            GemmDesc g;
            g.A = p.dcat_all + p.Hs; g.lda = (long)p.B * 2 * p.Hs; g.a_kc = true; g.sA = 2 * p.Hs;
            g.B = p.feat; g.ldb = p.Hs; g.b_kc = true; g.sB = (long)p.Tp * p.Hs;
            g.C = e0; g.ldc = (long)p.B * p.Tp; g.sC = p.Tp;
            g.M = p.U; g.N = p.Tp; g.K = p.Hs; g.batch = p.B;
            // one 128x128 tile per utterance and 32 k-tiles: split K so that the launch covers the chip (atomics onto a zeroed e0)
            g.splitk = p.B * ((p.U + 127) / 128) * ((p.Tp + 127) / 128) < 128 ? std::max(1, std::min(8, p.Hs / 64)) : 1;
            if (g.splitk > 1) {
                LAS_HIP_CHECK(hipMemsetAsync(e0, 0, sizeof(float) * (size_t)p.U * p.B * p.Tp, stream));
                g.c_zeroed = true;
            }
            LAS_TRY(gemm_f32(g, stream));
        }
        if (side_fill) LAS_TRY(side_guard.join());
        else LAS_HIP_CHECK(hipMemsetAsync(slabs, 0xFF, sizeof(float) * slab_floats, stream));
        const int grid = 2 * (p.Hs / 16) * 2 + a.ns * p.B;
        if (p.Hs == 512) LAS_TRY(launch_persist_bwd_pre<512>(a, grid, stream));
        else LAS_TRY(launch_persist_bwd_pre<256>(a, grid, stream));
        return LAS_OK;       // dqpre_all was written by the kernel (slice 0 of every utterance: the sum of the masked parts)
    }
    float* slabs = p.xbuf + (size_t)a.ns * nq;
    a.dhA = slabs;
    a.dGx = a.dhA + (size_t)p.U * p.B * a.ns * p.Hs;
    a.dcx = a.dGx + (size_t)2 * p.U * (p.Hs / 16) * 32 * 64;
    a.dhc = a.dcx + (size_t)(p.U + 1) * (p.Hs / 16) * 32 * 16;
    const size_t slab_floats = (size_t)p.U * p.B * a.ns * p.Hs + (size_t)2 * p.U * (p.Hs / 16) * 32 * 64 + (size_t)2 * (p.U + 1) * (p.Hs / 16) * 32 * 16;
    a.B = p.B; a.Tp = p.Tp; a.U = p.U; a.relu = p.relu; a.err = p.err; a.trace = g_persist_bwd_trace;
    LAS_HIP_CHECK(hipMemsetAsync(slabs, 0xFF, sizeof(float) * slab_floats, stream));
    const int grid = 3 * (p.Hs / 16) * 2 + a.ns * p.B;
    if (p.Hs == 512) LAS_TRY(launch_persist_bwd<512>(a, grid, stream));
    else LAS_TRY(launch_persist_bwd<256>(a, grid, stream));
    // dqpre = sum of the attention workgroups' parts (the relu mask is linear in dq)
    LAS_TRY(sum_parts(p.dqpre_all, a.dqpre_part, (long)nq, (long)nq, a.ns, stream));
    return LAS_OK;
}

}  // namespace las
