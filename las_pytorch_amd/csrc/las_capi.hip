// C-ABI layer of liblas_hip.so: argument checking, workspace carving and kernel orchestration.
// Declarations and the reference lines each entry point replaces: include/las_hip.h.
#include "../../include/las_hip.h"
#include "las_common.h"
#include "las_kernels.h"
#include "options.h"
#include <algorithm>
#include <utility>
#include <vector>
#include <algorithm>

using namespace las;

namespace {

inline size_t r4(size_t n) { return (n + 3) & ~(size_t)3; }   // keep every carved section 16-byte aligned

struct PblstmLayout {
    size_t gates, cbuf, hprev, xbuf, total;
    PblstmLayout(int B, int T, int H, bool stash) {
        size_t o = 0;
        gates = o; o += r4((size_t)2 * B * T * 4 * H);
        cbuf = o; if (stash) o += r4((size_t)2 * B * T * H);
        hprev = o; if (stash) o += r4((size_t)2 * B * T * H);
        xbuf = o; o += r4(rec_xbuf_bytes(B, H) / sizeof(float));
        total = o;
    }
};

struct PblstmBwdLayout {
    size_t dgates, wt, xbuf, total;
    PblstmBwdLayout(int B, int T, int H) {
        size_t o = 0;
        dgates = o; o += r4((size_t)2 * B * T * 4 * H);
        wt = o; o += r4((size_t)2 * H * 4 * H);
        xbuf = o; o += r4(rec_xbuf_bytes(B, H) / sizeof(float) + rec_bwd_mfma_ring_floats(B, H));     // (+ the matrix-pipe backward's ring)
        total = o;
    }
};

struct SpellerLayout {
    size_t y_all, ctx_all, h_all, c_all, gates_all, q_all, w0p, ctxcat_all, hx, r0x, lgx, wperm, wyperm, bperm, yw, pctx, gx, bqp, bfl, blg, beg, qct, wyT, plx,
        mperm, pbias, p0, wcd, bcp, exs, total;
    bool pre_mh;               // ... its multi-head form (heads 2..4): P, gx per head, the folded matrices W_ctx W_dr[:, h]
    bool big;                  // room for the one-launch decode of the Hs = 1024 shape (speller_big.hip)
    bool pre;                  // room for the persistent decode kernel's pre-multiplied context variant
    int Vp;                    // label width padded to a multiple of 16: every cell operand is aligned and tail-free
    SpellerLayout(const las_speller_desc* d, int U) {
        size_t o = 0;
        const size_t B = d->B;
        Vp = (d->V + 15) & ~15;
        y_all = o; o += r4((size_t)(U + 1) * B * Vp);
        ctx_all = o; o += r4((size_t)(U + 1) * B * d->D);
        h_all = o; o += r4((size_t)d->L * U * B * d->Hs);
        c_all = o; o += r4((size_t)d->L * U * B * d->Hs);
        gates_all = o; o += r4((size_t)d->L * U * B * 4 * d->Hs);
        q_all = o; if (d->use_mlp) o += r4((size_t)U * B * d->M * d->multi_head);
        ctxcat_all = o; if (d->multi_head > 1) o += r4((size_t)U * B * d->multi_head * d->D);   // per-head contexts (dim_reduce input)
        w0p = o; o += r4((size_t)4 * d->Hs * (Vp + d->Hs));      // W_ih0 re-laid as [W_y | 0 | W_ctx], ld = Vp + Hs
        // pre-multiplied context variant: row-permuted W_ctx, feat . W_ctx^T, and the per-step hand-off slabs of its weighted sums
        // (r0x directly behind hx: one sentinel fill covers both)
        pre = speller_persist_pre_shape(d->B, d->Tp, d->Hs, d->D, d->M, d->V, d->L, d->multi_head, d->use_mlp);
        pre_mh = speller_persist_pre_mh_shape(d->B, d->Tp, d->Hs, d->D, d->M, d->V, d->L, d->multi_head, d->use_mlp);
        const bool anypre = pre || pre_mh;
        const size_t NHp = pre_mh ? d->multi_head : 1;
        hx = o; if (d->L == 2) o += r4((size_t)2 * U * 32 * d->Hs);   // hand-off copy of h for the persistent decode kernel
        r0x = o; if (anypre) o += r4((size_t)U * 32 * 4 * d->Hs);     // ... and the cell workgroups' part of the bottom-layer gates
        gx = o; if (anypre) o += r4((size_t)U * B * NHp * 4 * d->Hs + (pre_mh ? B * 16 * 4 : 0));      // (+ the placement check's XCC ids, multi-head)
        lgx = o; if (d->L == 2) o += r4((size_t)U * B * 8 * 32 * NHp);   // ... and its partial logits (free-running decode; multi-head: the heads' logit shares)
        wperm = o; if (anypre) o += r4((size_t)4 * d->Hs * d->Hs);
        wyperm = o; if (anypre) o += r4((size_t)4 * d->Hs * Vp);   // W_y rows and b_ih0 + b_hh0 in the same row order ...
        bperm = o; if (anypre) o += r4((size_t)4 * d->Hs);
        yw = o; if (anypre) o += r4((size_t)U * B * 4 * d->Hs);    // ... and y_s W_y^T + b for every step (label half of the bottom-layer gates)
        pctx = o; if (anypre) o += r4((size_t)B * d->Tp * NHp * 4 * d->Hs);
        // multi-head: (W_ctx W_dr[:, h]) stacked over the heads (NH*4Hs, D), the bias of P's head-0 block (W_ctx b_dr, then zeros), feat[:, 0] . W_ctx^T
        mperm = o; if (pre_mh) o += r4(NHp * 4 * d->Hs * d->D);
        pbias = o; if (pre_mh) o += r4(NHp * 4 * d->Hs);
        p0 = o; if (pre_mh) o += r4(B * 4 * d->Hs);
        // free-running form of the PRE kernel (arg-max feedback, no backward): Q^T = W_c[:, Hs:] feat^T, W_y^T (permuted columns), partial-logit slabs
        qct = o; if (anypre) o += r4((size_t)B * NHp * 32 * d->Tp);
        wyT = o; if (anypre) o += r4((size_t)Vp * 4 * d->Hs);
        plx = o; if (anypre) o += r4((size_t)U * (d->Hs / 4) * 512);
        // Hs = 256 with 16 attention workgroups per utterance (long T'): the frame slices' energies, exchanged every step
        exs = o; if (pre && speller_persist_pre_ws(d->B, d->Tp, d->Hs, -1) == 16) o += r4((size_t)U * B * 16 * 64 + B * 16);      // (Hs = 512: the free-running form only)
        // multi-head, free-running: W_c[:, Hs:] W_dr (V rows of NH*D, stored 32 rows) and b_c + W_c[:, Hs:] b_dr
        wcd = o; if (pre_mh) o += r4((size_t)32 * NHp * d->D);
        bcp = o; if (pre_mh) o += r4(32);
        // Hs = 1024 one-launch decode: label half of the bottom-layer gates, query slices and partial contexts (hand-off slabs, adjacent)
        big = speller_big_shape(d->B, d->Tp, d->Hs, d->D, d->M, d->V, d->L, d->multi_head, d->use_mlp);
        if (big && !pre) { yw = o; o += r4((size_t)U * B * 4 * d->Hs); }
        bqp = o; if (big) o += speller_big_qp_floats(d->B, U);
        bfl = o; if (big) o += r4(speller_big_flag_words(U));
        blg = o; if (big) o += r4(speller_big_greedy_floats(d->B, U));
        beg = o; if (big && d->Tp > 256) o += r4((size_t)U * B * 512);
        total = o;
    }
};

struct SpellerBwdLayout {
    size_t dG_all, dz_all, dctx_all, de_all, dqpre_all, dh_top, dh_below, dh_carry, dc_carry, dx0, dK, dcat_all, dctxcat_all,
        pxbuf, bxbuf, total;
    SpellerBwdLayout(const las_speller_desc* d, int U) {
        size_t o = 0;
        const size_t B = d->B;
        const size_t Mq = d->use_mlp ? d->M : d->Hs;
        dG_all = o; o += r4((size_t)d->L * U * B * 4 * d->Hs);
        dz_all = o; o += r4((size_t)U * B * d->V);
        o += r4(B * d->D);                       // headroom: the row block "step -1" of the deferred dG0 . W_ctx product (PRE variant)
        dctx_all = o; o += r4((size_t)U * B * d->D);
        de_all = o; o += r4((size_t)U * B * d->Tp * d->multi_head);
        dqpre_all = o; o += r4((size_t)U * B * d->M * d->multi_head);
        dh_top = o; o += r4((size_t)(d->multi_head + 1) * B * d->Hs);     // decoder-state gradient parts (one per head + W_c part)
        dh_below = o; o += r4(B * d->Hs);
        dh_carry = o; o += r4((size_t)d->L * B * d->Hs);
        dc_carry = o; o += r4((size_t)d->L * B * d->Hs);
        dx0 = o; o += r4(B * (d->V + d->D));
        dK = o; o += r4(B * d->Tp * Mq);
        dcat_all = o; o += r4((size_t)U * B * (d->Hs + d->D));    // dz W_c for every step (teacher forcing / mode 1)
        dctxcat_all = o; if (d->multi_head > 1) o += r4((size_t)U * B * d->multi_head * d->D);
        // persistent backward kernel: the attention workgroups' dqpre parts and the sentinel-prefilled hand-off slabs
        pxbuf = o;
        if (d->L == 2 && d->multi_head == 1 && d->use_mlp) o += r4(speller_persist_bwd_workspace_floats(d->B, d->Tp, U, d->Hs, d->M));
        else if (speller_persist_pre_mh_shape(d->B, d->Tp, d->Hs, d->D, d->M, d->V, d->L, d->multi_head, d->use_mlp))
            o += r4(speller_persist_bwd_mh_workspace_floats(d->B, d->Tp, U, d->Hs, d->M, d->multi_head));
        // Hs = 1024 one-launch backward (speller_big.hip): gate-gradient slabs, carries, slice triples, flags
        bxbuf = o;
        if (speller_big_shape(d->B, d->Tp, d->Hs, d->D, d->M, d->V, d->L, d->multi_head, d->use_mlp)) o += r4(speller_big_bwd_workspace_floats(d->B, U));
        total = o;
    }
};

// Gradient outputs that lie back to back in memory (the views of one flat gradient buffer, las_pytorch_amd/dp.py) are zeroed
// with ONE memset; the split-K GEMMs and column sums that fill them then skip theirs (a launch-bound ~5 us each).
// Returns false (and does nothing) when the outputs are separate allocations.
// LAS_FLAG_GRADS_ZEROED (a thread-local scope set by the backward entry points): the caller has zeroed the block already — the flat
// gradient buffer is cleared once per step (dp.FlatGradAllReducer.zero) — so the fill is skipped (one launch-bound ~5 us memset per
// backward entry point, four per training step).
static thread_local bool tl_grads_zeroed = false;
struct GradsZeroedScope {
    bool saved;
    explicit GradsZeroedScope(int flags) : saved(tl_grads_zeroed) { tl_grads_zeroed = (flags & LAS_FLAG_GRADS_ZEROED) != 0 && opt_get(OPT_TRUST_ZEROED_GRADS) != 0; }
    ~GradsZeroedScope() { tl_grads_zeroed = saved; }
};
static bool zero_if_contiguous(std::vector<std::pair<float*, size_t>> outs, hipStream_t stream) {
    std::sort(outs.begin(), outs.end(), [](const std::pair<float*, size_t>& a, const std::pair<float*, size_t>& b) { return a.first < b.first; });
    size_t total = 0;
    for (size_t i = 0; i < outs.size(); ++i) {
        if (outs[i].first == nullptr) return false;
        if (i + 1 < outs.size() && outs[i].first + outs[i].second != outs[i + 1].first) return false;
        total += outs[i].second;
    }
    if (tl_grads_zeroed) return true;
    return hipMemsetAsync(outs[0].first, 0, sizeof(float) * total, stream) == hipSuccess;
}

// attention-only entry points (keys, stand-alone attention): no LSTM / character-distribution weights needed
int check_attn_desc(const las_speller_desc* d) {
    LAS_REQUIRE(d != nullptr, "descriptor");
    LAS_REQUIRE(d->B > 0 && d->Tp > 0 && d->D > 0 && d->Hs > 0, "attention dims");
    LAS_REQUIRE(d->D == d->Hs, "decoder state width must equal the listener feature width (reference las_model.py:266)");
    LAS_REQUIRE(d->multi_head >= 1 && d->multi_head <= 16, "attention heads");
    LAS_REQUIRE(d->multi_head == 1 || (d->use_mlp && d->w_dr && d->b_dr),
                "multi-head attention needs the phi/psi MLP and dim_reduce (reference las_model.py:266-269)");
    LAS_REQUIRE(!d->use_mlp || (d->M > 0 && d->w_phi && d->b_phi && d->w_psi && d->b_psi), "attention MLP weights");
    LAS_REQUIRE(d->relu >= LAS_ACT_NONE && d->relu <= LAS_ACT_SIGMOID, "attention activation code");
    return LAS_OK;
}

int check_desc(const las_speller_desc* d) {
    LAS_REQUIRE(d != nullptr, "descriptor");
    LAS_REQUIRE(d->B > 0 && d->Tp > 0 && d->D > 0 && d->Hs > 0 && d->V > 0, "speller dims");
    LAS_REQUIRE(d->L >= 1 && d->L <= LAS_MAX_SPELLER_LAYERS, "speller layers");
    LAS_REQUIRE(d->D == d->Hs, "Speller hidden_size must equal 2*listener_hidden_size (reference las_model.py:198)");
    LAS_REQUIRE(d->multi_head >= 1 && d->multi_head <= 16, "attention heads");
    LAS_REQUIRE(d->multi_head == 1 || (d->use_mlp && d->w_dr && d->b_dr),
                "multi-head attention needs the phi/psi MLP and dim_reduce (reference las_model.py:266-269)");
    LAS_REQUIRE(!d->use_mlp || d->M > 0, "attention mlp dim");
    for (int l = 0; l < d->L; ++l) LAS_REQUIRE(d->w_ih[l] && d->w_hh[l] && d->b_ih[l] && d->b_hh[l], "speller LSTM weights");
    LAS_REQUIRE(d->w_c && d->b_c, "character distribution weights");
    LAS_REQUIRE(!d->use_mlp || (d->w_phi && d->b_phi && d->w_psi && d->b_psi), "attention MLP weights");
    return LAS_OK;
}



// The loop-invariant-shaped contractions of the attention backward, shared by the decode loop (U steps), one differentiable
// decode step and the stand-alone attention (U = 1):  dfeat (context path + psi path), dK -> dW_psi / db_psi, dW_phi / db_phi,
// dim_reduce gradients.  All pointers are step-major slabs as las_speller_bwd lays them out.
struct AttnDeferred {
    const float* feat; const float* keys; const float* att; const float* q_all; const float* de_all; const float* dqpre_all;
    const float* dctx_all; const float* dctxcat_all = nullptr; const float* ctxcat_all = nullptr; const float* h_top_all;
    float* dK; int U; bool zg;
    const float* dx0_ctx = nullptr; long ld_dx0 = 0;      // gradient of the FIRST decoder input's context = feat[:,0,:] (las_model.py:198)
    bool skip_dw_phi = false;
};
static int attention_deferred(const las_speller_desc* d, const AttnDeferred& x, const las_speller_grads* g, hipStream_t stream) {
    const int B = d->B, Hs = d->Hs, D = d->D, Tp = d->Tp, M = d->M, NH = d->multi_head, U = x.U, UB = U * B;
    const float* feat = x.feat; const float* keys = x.keys; const float* att = x.att; const float* q_all = x.q_all;
    const float* de_all = x.de_all; const float* dqpre_all = x.dqpre_all; const float* dctx_all = x.dctx_all;
    const float* dctxcat_all = x.dctxcat_all; const float* ctxcat_all = x.ctxcat_all; const float* h_top_all = x.h_top_all;
    float* dK = x.dK; const bool zg = x.zg;
    for (int hd = 0; hd < NH; ++hd) {   // dfeat[b] (+)= att_h[:,b,:]^T dctx_h[:,b,:]   (context path, las_model.py:293-297,307-313)
        GemmDesc q;
        q.A = att + (size_t)hd * B * Tp; q.lda = (long)NH * B * Tp; q.a_kc = false; q.sA = Tp;
        if (NH == 1) { q.B = dctx_all; q.ldb = (long)B * D; q.sB = D; }
        else { q.B = dctxcat_all + (size_t)hd * D; q.ldb = (long)B * NH * D; q.sB = (long)NH * D; }
        q.b_kc = false;
        q.C = g->dfeat; q.ldc = D; q.sC = (long)Tp * D; q.batch = B;
        q.M = Tp; q.N = D; q.K = U; q.splitk = 1; q.accumulate = hd > 0;
        LAS_TRY(gemm_f32(q, stream));
    }
    // first decoder input used feat[:,0,:] as context (las_model.py:198)
    if (x.dx0_ctx) LAS_TRY(copy2d(x.dx0_ctx, x.ld_dx0, g->dfeat, (long)Tp * D, B, D, 1, stream));
    const int Mq = d->use_mlp ? M : Hs;
    for (int hd = 0; hd < NH; ++hd) {   // dK[b] (+)= de_h[:,b,:]^T q_h[:,b,:]   (energy path)
        GemmDesc q;
        q.A = de_all + (size_t)hd * B * Tp; q.lda = (long)NH * B * Tp; q.a_kc = false; q.sA = Tp;
        q.B = d->use_mlp ? q_all + (size_t)hd * M : h_top_all; q.ldb = (long)B * Mq * NH; q.b_kc = false; q.sB = (long)Mq * NH;
        q.C = d->use_mlp ? dK : g->dfeat; q.ldc = Mq; q.sC = (long)Tp * Mq; q.batch = B;
        q.M = Tp; q.N = Mq; q.K = U; q.splitk = 1; q.accumulate = !d->use_mlp || hd > 0;
        LAS_TRY(gemm_f32(q, stream));
    }
    if (NH > 1) {   // dim_reduce gradients: dW_dr = dctx^T ctxcat, db_dr = sum dctx
        GemmDesc q;
        q.A = dctx_all; q.lda = D; q.a_kc = false; q.B = ctxcat_all; q.ldb = (long)NH * D; q.b_kc = false;
        q.C = g->dw_dr; q.ldc = (long)NH * D; q.M = D; q.N = NH * D; q.K = U * B; q.c_zeroed = zg;
        LAS_TRY(gemm_f32(q, stream));
        LAS_TRY(colsum(dctx_all, D, U * B, D, g->db_dr, zg, stream));
    }
    if (d->use_mlp) {
        const int BT = B * Tp;
        if (d->relu) LAS_TRY(act_bwd_inplace(dK, keys, (long)BT * M, d->relu, stream));
        {   // dW_psi = dKpre^T feat
            GemmDesc q;
            q.A = dK; q.lda = M; q.a_kc = false; q.B = feat; q.ldb = D; q.b_kc = false;
            q.C = g->dw_psi; q.ldc = D; q.M = M; q.N = D; q.K = BT; q.c_zeroed = zg;
            LAS_TRY(gemm_f32(q, stream));
        }
        {   // db_psi and db_phi in one launch
            ColsumJob cj[2] = {{dK, (long)M, BT, M, g->db_psi, nullptr}, {dqpre_all, (long)M * NH, UB, M * NH, g->db_phi, nullptr}};
            LAS_TRY(colsum_multi(cj, 2, zg, stream));
        }
        {   // dfeat += dKpre W_psi
            GemmDesc q;
            q.A = dK; q.lda = M; q.a_kc = true; q.B = d->w_psi; q.ldb = D; q.b_kc = false;
            q.C = g->dfeat; q.ldc = D; q.M = BT; q.N = D; q.K = M; q.splitk = 1; q.accumulate = true;
            LAS_TRY(gemm_f32(q, stream));
        }
        if (!x.skip_dw_phi) {   // dW_phi = dqpre^T h_top (decode loop, single head: part of its grouped launch instead)
            GemmDesc q;
            q.A = dqpre_all; q.lda = (long)M * NH; q.a_kc = false; q.B = h_top_all; q.ldb = Hs; q.b_kc = false;
            q.C = g->dw_phi; q.ldc = Hs; q.M = M * NH; q.N = Hs; q.K = UB; q.c_zeroed = zg;
            LAS_TRY(gemm_f32(q, stream));
        }
    }
    return LAS_OK;
}

}  // namespace

extern "C" {

int las_abi_version(void) { return LAS_ABI_VERSION; }
const char* las_last_error(void) { return g_err; }

// ---------------------------------------------------------------------------------------------- pBLSTM
size_t las_pblstm_reserve_floats(int B, int T_in, int H, int flags) {
    return PblstmLayout(B, T_in / 2, H, flags & LAS_FLAG_STASH).total;
}

int las_pblstm_fwd(const float* x, int B, int T_in, int D_in, int H, const float* w_ih_f, const float* w_hh_f,
                   const float* b_ih_f, const float* b_hh_f, const float* w_ih_r, const float* w_hh_r, const float* b_ih_r,
                   const float* b_hh_r, float* out, float* reserve, uint32_t* err_word, int flags, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    GemmArithScope arith_scope(flags, err_word);
    LAS_REQUIRE(B > 0 && T_in > 0 && D_in > 0 && H > 0, "pblstm dims");
    LAS_REQUIRE(T_in % 2 == 0, "pBLSTM needs an even number of frames (reference las_model.py:86-87)");
    LAS_REQUIRE(x && out && reserve && err_word, "pblstm pointers");
    LAS_REQUIRE(w_ih_f && w_hh_f && b_ih_f && b_hh_f && w_ih_r && w_hh_r && b_ih_r && b_hh_r, "pblstm weights");
    LAS_REQUIRE((uintptr_t)reserve % 16 == 0, "reserve alignment");
    const int T = T_in / 2, D = 2 * D_in;
    const bool stash = flags & LAS_FLAG_STASH;
    PblstmLayout lay(B, T, H, stash);
    float* gates = reserve + lay.gates;
    // K2: input projection for both directions (MFMA GEMM, biases fused)
    // One launch of two "batches": A is shared, the per-direction operands are addressed through element strides that
    // are simply the distance between the two parameter tensors.  2 x 400 tiles in flight fill the 256 CUs more evenly
    // than two launches of 400 (1.56 tiles per CU each).
    const bool batch_dirs = opt_get(OPT_GEMM_BATCH_DIRS) != 0;
    for (int dir = 0; dir < (batch_dirs ? 1 : 2); ++dir) {
        GemmDesc g;
        g.A = x; g.lda = D; g.a_kc = true;
        g.B = dir ? w_ih_r : w_ih_f; g.ldb = D; g.b_kc = true;
        g.C = gates + (size_t)dir * B * T * 4 * H; g.ldc = 4 * H;
        g.bias0 = dir ? b_ih_r : b_ih_f; g.bias1 = dir ? b_hh_r : b_hh_f;
        g.M = B * T; g.N = 4 * H; g.K = D; g.splitk = 1;
        if (batch_dirs) {
            g.batch = 2; g.sA = 0; g.sB = w_ih_r - w_ih_f; g.sC = (long)B * T * 4 * H;
            g.sBias0 = b_ih_r - b_ih_f; g.sBias1 = b_hh_r - b_hh_f;
        }
        LAS_TRY(gemm_f32(g, stream));
    }
    // K3: time recurrence
    return pblstm_rec_fwd(gates, w_hh_f, w_hh_r, out, stash ? reserve + lay.cbuf : nullptr, stash ? reserve + lay.hprev : nullptr,
                          B, T, H, stash, (unsigned long long*)(reserve + lay.xbuf), err_word,
                          flags & LAS_FLAG_FORCE_GENERIC, stream);
}

size_t las_pblstm_bwd_workspace_floats(int B, int T_in, int H) { return PblstmBwdLayout(B, T_in / 2, H).total; }

int las_pblstm_bwd(const float* x, const float* dout, int B, int T_in, int D_in, int H, const float* w_ih_f,
                   const float* w_hh_f, const float* w_ih_r, const float* w_hh_r, const float* reserve, float* workspace,
                   float* dx, float* dw_ih_f, float* dw_hh_f, float* db_ih_f, float* db_hh_f, float* dw_ih_r, float* dw_hh_r,
                   float* db_ih_r, float* db_hh_r, uint32_t* err_word, int flags, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    GemmArithScope arith_scope(flags, err_word);
    GradsZeroedScope zeroed_scope(flags);
    LAS_REQUIRE(B > 0 && T_in > 0 && D_in > 0 && H > 0 && T_in % 2 == 0, "pblstm dims");
    LAS_REQUIRE(x && dout && reserve && workspace && err_word, "pblstm bwd pointers");
    LAS_REQUIRE(dw_ih_f && dw_hh_f && db_ih_f && db_hh_f && dw_ih_r && dw_hh_r && db_ih_r && db_hh_r, "pblstm grad outputs");
    const int T = T_in / 2, D = 2 * D_in;
    PblstmLayout lay(B, T, H, true);
    PblstmBwdLayout wl(B, T, H);
    const float* gates = reserve + lay.gates;
    const float* cbuf = reserve + lay.cbuf;
    const float* hprev = reserve + lay.hprev;
    float* dgates = workspace + wl.dgates;
    float* wt = workspace + wl.wt;
    LAS_TRY(transpose2d(w_hh_f, wt, 4 * H, H, stream, w_hh_r, wt + (size_t)H * 4 * H));
    const int BT = B * T;
    const size_t n_ih = (size_t)4 * H * D, n_hh = (size_t)4 * H * H, n_b = (size_t)4 * H;
    const bool zg = zero_if_contiguous({{dw_ih_f, n_ih}, {dw_hh_f, n_hh}, {db_ih_f, n_b}, {db_hh_f, n_b},
                                        {dw_ih_r, n_ih}, {dw_hh_r, n_hh}, {db_ih_r, n_b}, {db_hh_r, n_b}}, stream);
    // the persistent backward recurrences sum the bias gradients themselves when each direction's [b_ih | b_hh] pair is one
    // zeroed 8H block (the layout of the flat gradient buffer); otherwise a column-sum kernel runs below
    const bool db_in_kernel = zg && db_hh_f == db_ih_f + n_b && db_hh_r == db_ih_r + n_b;
    int db_done = 0;
    // (deferral needs a next recurrence to hide under — dx != null: not the input layer — and a batch that leaves XCDs free)
    const int confine = ((flags & LAS_FLAG_DEFER_DW) && dx && !(flags & LAS_FLAG_FORCE_GENERIC) && opt_get(OPT_DEFER_DW) != 0) ? rec_confine_xcds(B, H) : 0;
    LAS_TRY(pblstm_rec_bwd(dout, gates, cbuf, wt, dgates, B, T, H, (unsigned long long*)(workspace + wl.xbuf), err_word,
                           flags & LAS_FLAG_FORCE_GENERIC, stream, db_in_kernel ? db_ih_f : nullptr, db_in_kernel ? db_ih_r : nullptr,
                           &db_done, confine));
    // the four weight-gradient contractions (few output tiles, K = B*T) go out as ONE grouped stream-K launch
    GemmDesc dw[4];
    for (int dir = 0; dir < 2; ++dir) {
        const float* dG = dgates + (size_t)dir * BT * 4 * H;
        const float* hp = hprev + (size_t)dir * BT * H;
        GemmDesc& gi = dw[2 * dir];          // dW_ih = dG^T X
        gi.A = dG; gi.lda = 4 * H; gi.a_kc = false;
        gi.B = x; gi.ldb = D; gi.b_kc = false;
        gi.C = dir ? dw_ih_r : dw_ih_f; gi.ldc = D; gi.M = 4 * H; gi.N = D; gi.K = BT; gi.c_zeroed = zg;
        GemmDesc& gh = dw[2 * dir + 1];      // dW_hh = dG^T H_prev
        gh.A = dG; gh.lda = 4 * H; gh.a_kc = false;
        gh.B = hp; gh.ldb = H; gh.b_kc = false;
        gh.C = dir ? dw_hh_r : dw_hh_f; gh.ldc = H; gh.M = 4 * H; gh.N = H; gh.K = BT; gh.c_zeroed = zg;
    }
    // LAS_FLAG_DEFER_DW: the weight-gradient group leaves the critical path — it goes to the library's side stream, restricted to the XCDs the
    // confined recurrence above (and the next layer's, launched by the next call) does not use, and is joined by las_join_deferred
    hipStream_t ws = stream;
    bool deferred = false;
    if (confine > 0) {
        if (hipStream_t s = defer_side().begin(stream)) { ws = s; deferred = true; }
    }
    if (dx) path_note(PATH_DW, deferred ? "deferred" : "inline");      // (the input layer has no recurrence after it to hide under: not noted)
    // Neither deferred nor the input layer: dX (the critical path: 200 - 400 tiles, which leave 20 - 60 % of the resident slots idle) and the
    // weight-gradient group run at the SAME time on two streams and are joined before the call returns — the group draws its runs from a counter,
    // so its workgroups take whatever CUs dX leaves (option DW_CONCURRENT; nothing runs beside the next layer's recurrence)
    hipStream_t cs = nullptr;
    if (!deferred && dx && opt_get(OPT_DW_CONCURRENT) != 0 && !(flags & LAS_FLAG_FORCE_GENERIC)) cs = defer_side().begin_joined(stream);
    if (!deferred && cs == nullptr) {
        LAS_TRY(gemm_f32_group(dw, 4, stream));
        for (int dir = 0; dir < 2 && !db_done; ++dir)
            LAS_TRY(colsum(dgates + (size_t)dir * BT * 4 * H, 4 * H, BT, 4 * H, dir ? db_ih_r : db_ih_f, zg, stream, dir ? db_hh_r : db_hh_f));
    }
    if (dx) {   // dX = [dG_f | dG_r] [W_ih_f ; W_ih_r]: both directions in one pass over K = 2 * 4H
        GemmDesc g;
        g.A = dgates; g.lda = 4 * H; g.a_kc = true;
        g.B = w_ih_f; g.ldb = D; g.b_kc = false;
        g.A2 = dgates + (size_t)BT * 4 * H; g.B2 = w_ih_r; g.K1 = 4 * H;
        g.C = dx; g.ldc = D; g.M = BT; g.N = D; g.K = 8 * H; g.splitk = 1;
        LAS_TRY(gemm_f32(g, stream));
    }
    if (cs != nullptr) {
        const int rc = gemm_f32_group(dw, 4, cs, 0, (int)opt_get(OPT_DW_CONCURRENT));
        for (int dir = 0; rc == LAS_OK && dir < 2 && !db_done; ++dir)
            (void)colsum(dgates + (size_t)dir * BT * 4 * H, 4 * H, BT, 4 * H, dir ? db_ih_r : db_ih_f, zg, cs, dir ? db_hh_r : db_hh_f);
        LAS_TRY(defer_side().end_joined(stream));      // (joined on every path: the side stream must not outlive the call)
        LAS_TRY(rc);
    }
    if (deferred) {      // (issued after dX so that the critical-path GEMM is first in the hardware queues; both start once the recurrence is done)
        LAS_TRY(gemm_f32_group(dw, 4, ws, confine));
        for (int dir = 0; dir < 2 && !db_done; ++dir)
            LAS_TRY(colsum(dgates + (size_t)dir * BT * 4 * H, 4 * H, BT, 4 * H, dir ? db_ih_r : db_ih_f, zg, ws, dir ? db_hh_r : db_hh_f));
        LAS_TRY(defer_side().end());
    }
    return LAS_OK;
}

// ---------------------------------------------------------------------------------------------- Speller
int las_attn_keys_fwd(const las_speller_desc* d, const float* feat, float* keys, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    LAS_TRY(check_attn_desc(d));
    LAS_REQUIRE(d->use_mlp, "keys are the listener features themselves when the attention MLP is off");
    LAS_REQUIRE(feat && keys, "keys pointers");
    GemmDesc g;
    g.A = feat; g.lda = d->D; g.a_kc = true;
    g.B = d->w_psi; g.ldb = d->D; g.b_kc = true;
    g.C = keys; g.ldc = d->M; g.bias0 = d->b_psi;
    g.M = d->B * d->Tp; g.N = d->M; g.K = d->D;
    // a 64-column output gives only B*T'/128 tiles: split K over the chip and apply the activation in a second tiny pass
    const long tiles = (long)cdiv(g.M, 128) * cdiv(g.N, 128);
    if (tiles < 64 && g.K >= 256 && opt_get(OPT_KEYS_SPLITK) != 0) {
        g.splitk = (int)std::min<long>(g.K / 64, std::max<long>(1, 256 / tiles));
        LAS_TRY(gemm_f32(g, stream));
        if (d->relu) LAS_TRY(act_inplace(keys, (long)g.M * g.N, d->relu, stream));
        return LAS_OK;
    }
    g.splitk = 1; g.relu = d->relu == LAS_ACT_RELU;          // relu rides in the GEMM epilogue, other activations in a second pass
    LAS_TRY(gemm_f32(g, stream));
    if (d->relu > LAS_ACT_RELU) LAS_TRY(act_inplace(keys, (long)g.M * g.N, d->relu, stream));
    return LAS_OK;
}

size_t las_speller_reserve_floats(const las_speller_desc* d, int U) { return SpellerLayout(d, U).total; }

int las_speller_decode_batch(const las_speller_desc* d, int teacher_forced, int decode_mode) {
    if (!d || check_desc(d) != LAS_OK) return 0;
    // the YAML sizes (speller_big.hip): 16 utterances per launch, teacher forcing only (relu / no activation; the Hs <= 512 kernels take
    // every activation code)
    if ((teacher_forced || decode_mode == 1) && d->relu <= LAS_ACT_RELU &&
        speller_big_eligible(16, d->Tp, d->Hs, d->D, d->M, d->V, d->L, d->multi_head, d->use_mlp, !teacher_forced))
        return 16;
    if (opt_get(OPT_SPELLER_PERSIST) == 0 || (!teacher_forced && decode_mode == 2)) return 0;
    constexpr int NB = 32;      // the persistent kernels' utterance limit (two 16-row M tiles)
    if (d->multi_head > 1) {    // one set of attention workgroups per (utterance, head): 32 / heads utterances per launch at most
        // (a batch of more than two such slices is faster on the per-step kernels, which take all of it at once: heads = 4, B = 32 measured
        // 17.3 ms per step in four slices against 15.1).  Long utterances need more attention workgroups per (utterance, head) — 16 at
        // T' = 375 — so fewer utterances fit one launch: the slice is halved until the launch is resident (heads = 2, B = 8, T' = 375: two
        // slices of 4), under the same two-slice rule.
        for (int nb = NB / d->multi_head; nb >= 1; nb >>= 1) {
            if (d->B > 2 * nb) break;
            const int bb = std::min(nb, d->B);
            const bool ok = teacher_forced ? speller_persist_pre_mh_eligible(bb, d->Tp, d->Hs, d->D, d->M, d->V, d->L, d->multi_head, d->use_mlp)
                                           : (decode_mode == 1 && speller_persist_pre_mh_greedy_eligible(bb, d->Tp, d->Hs, d->D, d->M, d->V, d->L, d->multi_head, d->use_mlp));
            if (ok) return nb;
        }
        return 0;
    }
    return speller_persist_eligible(NB, d->Tp, d->Hs, d->D, d->M, d->V, d->L, d->multi_head, d->use_mlp, !teacher_forced) ? NB : 0;
}

// Profiling aid (declared at the end of include/las_hip.h): per-phase shader-clock stamps of the persistent decode kernel.
extern "C" void las_debug_persist_trace(unsigned long long* dev_buf) { speller_persist_set_trace(dev_buf); }
extern "C" void las_debug_persist_bwd_trace(unsigned long long* dev_buf) { speller_persist_bwd_set_trace(dev_buf); }
extern "C" void las_debug_big_trace(unsigned long long* dev_buf) { speller_big_set_trace(dev_buf); }
extern "C" void las_debug_big_bwd_trace(unsigned long long* dev_buf) { speller_big_bwd_set_trace(dev_buf); }
#ifdef LAS_REC_TRACE
extern "C" void las_debug_rec_trace(unsigned long long* dev_buf) { rec_set_trace(dev_buf); }
#endif

int las_speller_fwd(const las_speller_desc* d, const float* feat, const float* keys, const int64_t* labels_onehot, int U_lab,
                    int U, int teacher_forced, int decode_mode, const float* sample_noise, float* logp, float* att,
                    int32_t* argmax, float* reserve, uint32_t* err_word, int flags, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    GemmArithScope arith_scope(flags, err_word);
    LAS_TRY(check_desc(d));
    LAS_REQUIRE(U > 0, "decode steps");
    LAS_REQUIRE(feat && logp && att && reserve, "speller pointers");
    LAS_REQUIRE(!d->use_mlp || keys, "attention keys");
    LAS_REQUIRE(!teacher_forced || (labels_onehot && U_lab >= U), "teacher forcing needs labels for every step");
    if (!teacher_forced && (decode_mode < 0 || decode_mode > 2))
        return fail(LAS_ERR_UNSUPPORTED, "decode_mode %s%ld does not exist (reference las_model.py:219-234)", "", (long)decode_mode);
    LAS_REQUIRE(teacher_forced || decode_mode != 2 || sample_noise, "decode_mode 2 needs the caller's Exp(1) draws (sample_noise)");
    LAS_REQUIRE((uintptr_t)reserve % 16 == 0, "reserve alignment");
    const int B = d->B, Hs = d->Hs, V = d->V, D = d->D, Tp = d->Tp, L = d->L;
    SpellerLayout lay(d, U);
    float* y_all = reserve + lay.y_all;
    float* ctx_all = reserve + lay.ctx_all;
    float* h_all = reserve + lay.h_all;
    float* c_all = reserve + lay.c_all;
    float* gates_all = reserve + lay.gates_all;
    float* q_all = d->use_mlp ? reserve + lay.q_all : nullptr;
    const size_t sH = (size_t)B * Hs;       // one (B,Hs) slab

    const int Vp = lay.Vp;
    float* w0p = reserve + lay.w0p;
    LAS_REQUIRE(Hs % 16 == 0, "speller hidden size must be a multiple of 16");
    // (labels -> y_all, ctx_{-1} = feat[:,0,:] (las_model.py:198) and the W_ih0 shadow below: ONE launch, speller_prologue)
    // 16-byte aligned, tail-free shadow of W_ih0: columns [0,V) = label part, [V,Vp) = 0, [Vp,Vp+Hs) = context part (rebuilt
    // every call: in training the parameters change every step, so there is nothing to cache across calls)
    const bool persist_on = opt_get(OPT_SPELLER_PERSIST) != 0;
    const bool persist = persist_on && err_word && !(flags & LAS_FLAG_FORCE_GENERIC) && (teacher_forced || decode_mode != 2) &&
                         speller_persist_eligible(B, Tp, Hs, D, d->M, V, L, d->multi_head, d->use_mlp, !teacher_forced);
    // the pre-multiplied context variant of the persistent kernel (teacher forcing): P = feat . W_ctx^T in the cell
    // workgroups' column order; the kernel then publishes sum_t a_t P_t instead of the context, which one batched GEMM
    // recovers afterwards (the backward pass and the character distribution need it)
    // (independent of the classic kernel's eligibility: at Hs = 256 and T' > 448 only the PRE variant, with the keys split by frames, applies)
    const bool pre = persist_on && err_word && !(flags & LAS_FLAG_FORCE_GENERIC) && teacher_forced && lay.pre &&
                     speller_persist_pre_eligible(B, Tp, Hs, D, d->M, V, L, d->multi_head, d->use_mlp);
    // What a teacher-forced forward leaves in `reserve` depends on the SHAPE only (lay.pre), never on switches, the error word
    // or the occupancy calculator: P = feat . W_ctx^T and the per-step sums gx_s = sum_t a_t P_t are always there, so that
    // las_speller_bwd (LAS_FLAG_TEACHER_FORCED) can rely on them whichever forward variant actually ran.
    // (Only a stashing forward has a backward: without LAS_FLAG_STASH the two GEMMs are skipped unless the PRE kernel itself needs P.)
    // ... and its free-running form (decode_mode 1: the reference's validation decode, train.py:149-169, and the free-running training steps of a
    // teacher-forcing schedule below 1): the character distribution moves into the attention workgroups, one launch at ~8 instead of ~13 us per
    // step.  A one-hot fed-back symbol carries no gradient, so the backward of such a pass IS the teacher-forced backward over the emitted
    // symbols: a stashing forward leaves the same P / gx in the reserve (tf_like) and las_speller_bwd takes its PRE path for it too.
    const bool tf_like = teacher_forced || decode_mode == 1;
    const bool preg = !teacher_forced && decode_mode == 1 && lay.pre && persist_on && err_word &&
                      !(flags & LAS_FLAG_FORCE_GENERIC) && logp &&
                      speller_persist_pre_greedy_eligible(B, Tp, Hs, D, d->M, V, L, d->multi_head, d->use_mlp);
    const bool pre_stash = tf_like && lay.pre && ((flags & LAS_FLAG_STASH) || pre || preg);
    // ... and its multi-head form (heads 2..4, teacher forcing): one set of attention workgroups per (utterance, head), dim_reduce folded into P
    const bool pre_mh = persist_on && err_word && !(flags & LAS_FLAG_FORCE_GENERIC) && teacher_forced && lay.pre_mh &&
                        speller_persist_pre_mh_eligible(B, Tp, Hs, D, d->M, V, L, d->multi_head, d->use_mlp);
    // (as for the single head: what a teacher-forced stashing forward leaves — P and gx per head — depends on the shape only, so that
    // las_speller_bwd can take its multi-head PRE path whichever forward kernels ran)
    // Free-running decode_mode 1 with several heads runs the per-step kernels forward, but its backward is the teacher-forced one over the emitted
    // symbols (tf_like, as for the single head): a stashing forward leaves P and gx for it too.
    const bool preg_mh = !teacher_forced && decode_mode == 1 && lay.pre_mh && persist_on && err_word && !(flags & LAS_FLAG_FORCE_GENERIC) &&
                         logp && speller_persist_pre_mh_greedy_eligible(B, Tp, Hs, D, d->M, V, L, d->multi_head, d->use_mlp);
    const bool pre_mh_stash = tf_like && lay.pre_mh && ((flags & LAS_FLAG_STASH) || pre_mh || preg_mh);
    bool mh_gx_written = false;
    LAS_TRY(speller_prologue(d->w_ih[0], w0p, Hs, V, Vp, (pre_stash || preg || pre_mh_stash) ? reserve + lay.wperm : nullptr, (pre || preg || pre_mh || preg_mh) ? reserve + lay.wyperm : nullptr,
                             (pre || preg || pre_mh || preg_mh) ? reserve + lay.bperm : nullptr, d->b_ih[0], d->b_hh[0],
                             teacher_forced ? (const long long*)labels_onehot : nullptr, y_all, B, U, U_lab, feat, (long)Tp * D, ctx_all, D, stream));
    // the PRE kernel's hand-off slabs (50 MB of sentinel words at paper size) are filled on the side stream, beside the two GEMMs below
    SideStream& side = side_stream();
    bool side_fill = false;
    SideJoinGuard side_guard;
    if (pre && side.ok(stream)) {
        PersistFwd pf;
        pf.hx = reserve + lay.hx; pf.r0x = reserve + lay.r0x; pf.U = U; pf.Hs = Hs;
        LAS_TRY(side.fork(stream));
        side_guard.arm(side, stream);
        LAS_TRY(speller_persist_fwd_fill(pf, side.s));
        side_fill = true;
    }
    if (pre_stash || preg) {
        GemmDesc g;
        g.A = feat; g.lda = D; g.a_kc = true;
        g.B = reserve + lay.wperm; g.ldb = Hs; g.b_kc = true;
        g.C = reserve + lay.pctx; g.ldc = 4 * Hs; g.M = B * Tp; g.N = 4 * Hs; g.K = D; g.splitk = 1;
        LAS_TRY(gemm_f32(g, stream));
    }
    bool persist_ran = persist || pre || preg;
    bool pre_ran = false;           // a PRE kernel ran AND the contexts of every step are wanted: they are recovered by one GEMM below
    bool gx_written = false;        // ... a PRE kernel ran: the per-step sums gx are in the reserve already
    if (pre_mh_stash) {
        // Multi-head attention on the PRE kernel (las_model.py:298-314).  context_s = W_dr cat_h(ctx^h_s) + b_dr enters the bottom cell as
        // W_ctx context_s = sum_h sum_t a^h_{s,t} (feat_t M_h^T) + W_ctx b_dr with M_h = W_ctx W_dr[:, h D:(h+1) D]: the same pre-multiplied
        // form as the single head, one P block per head; W_ctx b_dr rides as a bias of head 0's block (its weights sum to 1)
        const int NH = d->multi_head;
        float* wperm = reserve + lay.wperm; float* mperm = reserve + lay.mperm; float* pbias = reserve + lay.pbias; float* p0 = reserve + lay.p0;
        {
            GemmDesc g;      // M_h (4Hs x D) for every head: one batched launch
            g.A = wperm; g.lda = Hs; g.a_kc = true; g.sA = 0;
            g.B = d->w_dr; g.ldb = (long)NH * D; g.b_kc = false; g.sB = D;
            g.C = mperm; g.ldc = D; g.sC = (long)4 * Hs * D;
            g.M = 4 * Hs; g.N = D; g.K = Hs; g.batch = NH; g.splitk = 1;
            LAS_TRY(gemm_f32(g, stream));
        }
        LAS_HIP_CHECK(hipMemsetAsync(pbias, 0, sizeof(float) * (size_t)NH * 4 * Hs, stream));
        LAS_TRY(matvec_rows(wperm, Hs, d->b_dr, pbias, 4 * Hs, Hs, stream));
        {
            GemmDesc g;      // P (B*T', NH*4Hs)
            g.A = feat; g.lda = D; g.a_kc = true;
            g.B = mperm; g.ldb = D; g.b_kc = true; g.bias0 = pbias;
            g.C = reserve + lay.pctx; g.ldc = (long)NH * 4 * Hs; g.M = B * Tp; g.N = NH * 4 * Hs; g.K = D; g.splitk = 1;
            LAS_TRY(gemm_f32(g, stream));
        }
      if (pre_mh || preg_mh) {
        {
            GemmDesc g;      // step 0: the context is the first listener frame itself (las_model.py:198), no dim_reduce
            g.A = feat; g.lda = (long)Tp * D; g.a_kc = true;
            g.B = wperm; g.ldb = Hs; g.b_kc = true;
            g.C = p0; g.ldc = 4 * Hs; g.M = B; g.N = 4 * Hs; g.K = D; g.splitk = 1;
            LAS_TRY(gemm_f32(g, stream));
        }
        {
            GemmDesc g;      // label half of the bottom-layer gates for every step (free-running: <sos> and the biases alone, two step blocks)
            g.A = y_all; g.lda = Vp; g.a_kc = true;
            g.B = reserve + lay.wyperm; g.ldb = Vp; g.b_kc = true; g.bias0 = reserve + lay.bperm;
            g.C = reserve + lay.yw; g.ldc = 4 * Hs; g.M = (preg_mh ? std::min(U, 2) : U) * B; g.N = 4 * Hs; g.K = Vp; g.splitk = 1;
            LAS_TRY(gemm_f32(g, stream));
        }
        PersistFwd p;
        if (preg_mh) {
            // the context share of the logits through dim_reduce: W_c[:, Hs:] context = sum_h sum_t a^h_t (W_cd,h feat_t) + W_c[:, Hs:] b_dr,
            // W_cd = W_c[:, Hs:] W_dr (V x NH*D); Q^T[b][h] = W_cd,h feat[b]^T (V x T')
            float* wcd = reserve + lay.wcd; float* bcp = reserve + lay.bcp;
            {
                GemmDesc g;
                g.A = d->w_c + Hs; g.lda = Hs + D; g.a_kc = true;
                g.B = d->w_dr; g.ldb = (long)NH * D; g.b_kc = false;
                g.C = wcd; g.ldc = (long)NH * D; g.M = V; g.N = NH * D; g.K = D; g.splitk = 1;
                LAS_TRY(gemm_f32(g, stream));
            }
            for (int hd = 0; hd < NH; ++hd) {
                GemmDesc g;
                g.A = wcd + (size_t)hd * D; g.lda = (long)NH * D; g.a_kc = true;
                g.B = feat; g.ldb = D; g.b_kc = true; g.sB = (long)Tp * D;
                g.C = reserve + lay.qct + (size_t)hd * 32 * Tp; g.ldc = Tp; g.sC = (long)NH * 32 * Tp;
                g.M = V; g.N = Tp; g.K = D; g.batch = B; g.splitk = 1;
                LAS_TRY(gemm_f32(g, stream));
            }
            LAS_TRY(matvec_rows(d->w_c + Hs, Hs + D, d->b_dr, bcp, V, D, stream, d->b_c));
            LAS_TRY(transpose2d(reserve + lay.wyperm, reserve + lay.wyT, 4 * Hs, Vp, stream));
            p.mode = 1; p.w_c = d->w_c; p.b_c = bcp; p.logp = logp; p.argmax = argmax; p.lgx = reserve + lay.lgx;
            p.qct = reserve + lay.qct; p.wyT = reserve + lay.wyT; p.plx = reserve + lay.plx;
        }
        p.NH = NH; p.p0 = p0;
        p.pctx = reserve + lay.pctx; p.gx = reserve + lay.gx; p.r0x = reserve + lay.r0x; p.yw = reserve + lay.yw;
        p.w0p = w0p; p.Vp = Vp;
        p.w_hh0 = d->w_hh[0]; p.w_ih1 = d->w_ih[1]; p.w_hh1 = d->w_hh[1];
        p.b_ih0 = d->b_ih[0]; p.b_hh0 = d->b_hh[0]; p.b_ih1 = d->b_ih[1]; p.b_hh1 = d->b_hh[1];
        p.w_phi = d->w_phi; p.b_phi = d->b_phi;
        p.feat = feat; p.keys = keys; p.y_all = y_all;
        p.ctx_all = ctx_all; p.h_all = h_all; p.c_all = c_all; p.gates_all = gates_all; p.q_all = q_all; p.att = att;
        p.B = B; p.Tp = Tp; p.U = U; p.Hs = Hs; p.V = V; p.relu = d->relu;
        p.hx = reserve + lay.hx;
        p.err = err_word;
        const int rc = speller_persist_fwd(p, stream);
        if (rc != LAS_ERR_UNSUPPORTED) {
            LAS_TRY(rc);
            persist_ran = true;
            mh_gx_written = true;
          if (teacher_forced || (flags & LAS_FLAG_STASH)) {      // (a free-running decode without a backward needs no contexts)
            // per-head contexts (the dim_reduce input, stashed for the backward), then the reduced context of every step
            float* ctxcat = reserve + lay.ctxcat_all;
            for (int hd = 0; hd < NH; ++hd) {
                GemmDesc g;
                g.A = att + (size_t)hd * B * Tp; g.lda = (long)NH * B * Tp; g.a_kc = true; g.sA = Tp;
                g.B = feat; g.ldb = D; g.b_kc = false; g.sB = (long)Tp * D;
                g.C = ctxcat + (size_t)hd * D; g.ldc = (long)B * NH * D; g.sC = (long)NH * D;
                g.M = U; g.N = D; g.K = Tp; g.batch = B; g.splitk = 1;
                LAS_TRY(gemm_f32(g, stream));
            }
            GemmDesc q;
            q.A = ctxcat; q.lda = (long)NH * D; q.a_kc = true;
            q.B = d->w_dr; q.ldb = (long)NH * D; q.b_kc = true; q.bias0 = d->b_dr;
            q.C = ctx_all + (size_t)B * D; q.ldc = D; q.M = U * B; q.N = D; q.K = NH * D; q.splitk = 1;
            LAS_TRY(gemm_f32(q, stream));
          }
        }
      }
    }
    if (persist || pre || preg) {
        PersistFwd p;
        p.prefilled = side_fill;
        if (preg) {
            // Q^T[b] = W_c[:, Hs:] feat[b]^T (V x T' per utterance): the context share of the logits becomes sum_t a_t Q[:, t]
            GemmDesc g;
            g.A = d->w_c + Hs; g.lda = Hs + D; g.a_kc = true;
            g.B = feat; g.ldb = D; g.b_kc = true; g.sB = (long)Tp * D;
            g.C = reserve + lay.qct; g.ldc = Tp; g.sC = (long)32 * Tp;
            g.M = V; g.N = Tp; g.K = D; g.batch = B; g.splitk = 1;
            LAS_TRY(gemm_f32(g, stream));
            // W_y^T in the permuted gate-column order: the bottom cell's lanes fetch the fed-back symbol's row
            LAS_TRY(transpose2d(reserve + lay.wyperm, reserve + lay.wyT, 4 * Hs, Vp, stream));
            p.qct = reserve + lay.qct; p.wyT = reserve + lay.wyT; p.plx = reserve + lay.plx;
        }
        if (pre || preg) {
            // label half of the bottom-layer gates for every step, off the decode chain: yw[s][b] = y_s[b] W_y^T + b_ih0 + b_hh0
            GemmDesc g;
            g.A = y_all; g.lda = Vp; g.a_kc = true;
            g.B = reserve + lay.wyperm; g.ldb = Vp; g.b_kc = true; g.bias0 = reserve + lay.bperm;
            // (free-running: y_0 = <sos>, every later label half is the bias alone — two step blocks suffice, the kernel reads block min(s, 1))
            g.C = reserve + lay.yw; g.ldc = 4 * Hs; g.M = (preg ? std::min(U, 2) : U) * B; g.N = 4 * Hs; g.K = Vp; g.splitk = 1;
            LAS_TRY(gemm_f32(g, stream));
            p.pctx = reserve + lay.pctx; p.gx = reserve + lay.gx; p.r0x = reserve + lay.r0x; p.yw = reserve + lay.yw;
            p.ex = reserve + lay.exs;
        }
        p.w0p = w0p; p.Vp = Vp;
        p.w_hh0 = d->w_hh[0]; p.w_ih1 = d->w_ih[1]; p.w_hh1 = d->w_hh[1];
        p.b_ih0 = d->b_ih[0]; p.b_hh0 = d->b_hh[0]; p.b_ih1 = d->b_ih[1]; p.b_hh1 = d->b_hh[1];
        p.w_phi = d->w_phi; p.b_phi = d->b_phi;
        p.feat = feat; p.keys = keys; p.y_all = y_all;
        p.ctx_all = ctx_all; p.h_all = h_all; p.c_all = c_all; p.gates_all = gates_all; p.q_all = q_all; p.att = att;
        p.B = B; p.Tp = Tp; p.U = U; p.Hs = Hs; p.V = V; p.relu = d->relu;
        p.hx = reserve + lay.hx;
        p.err = err_word;
        if (!teacher_forced) {      // the kernel produces log-probabilities, arg-max and the fed-back inputs itself
            p.mode = decode_mode == 1 ? 1 : 2;
            p.w_c = d->w_c; p.b_c = d->b_c; p.logp = logp; p.argmax = argmax; p.lgx = reserve + lay.lgx;
        }
        if (side_fill) LAS_TRY(side_guard.join());
        const int rc = speller_persist_fwd(p, stream);
        if (rc == LAS_ERR_UNSUPPORTED) persist_ran = false;      // residency check failed: the per-step kernels below run instead
        else LAS_TRY(rc);
        gx_written = persist_ran && (pre || preg);
        pre_ran = persist_ran && (pre || (preg && (flags & LAS_FLAG_STASH)));      // (free-running: only a backward pass reads the contexts)
        if (preg && !persist_ran && !persist) {      // (residency check failed and the classic kernel does not take this shape either)
            path_note(PATH_DECODE_FWD, "stepwise");
        } else if (preg && !persist_ran) {           // fall back to the classic free-running kernel
            PersistFwd q = p;
            q.pctx = nullptr; q.gx = nullptr; q.r0x = nullptr; q.yw = nullptr; q.qct = nullptr; q.wyT = nullptr; q.plx = nullptr; q.prefilled = false;
            const int rc2 = speller_persist_fwd(q, stream);
            if (rc2 != LAS_ERR_UNSUPPORTED) { LAS_TRY(rc2); persist_ran = true; }
            pre_ran = false; gx_written = false;     // (the classic kernel wrote the contexts itself; gx is recovered below)
        }
    }
    if (pre_ran) {   // contexts of every step: ctx_all[1+s][b] = att[s][b] . feat[b], one batched GEMM over the utterances
        GemmDesc g;
        g.A = att; g.lda = (long)B * Tp; g.a_kc = true; g.sA = Tp;
        g.B = feat; g.ldb = D; g.b_kc = false; g.sB = (long)Tp * D;
        g.C = ctx_all + (size_t)B * D; g.ldc = (long)B * D; g.sC = D;
        g.M = U; g.N = D; g.K = Tp; g.batch = B; g.splitk = 1;
        LAS_TRY(gemm_f32(g, stream));
    }
    // the reference's shipped sizes (Speller 1024x2, B <= 16): all U steps in one launch with register-resident cell weights
    const bool big_greedy = !teacher_forced && decode_mode == 1;
    if (!persist_ran && (teacher_forced || big_greedy) && lay.big && err_word && !(flags & LAS_FLAG_FORCE_GENERIC) && d->relu <= LAS_ACT_RELU &&
        speller_big_eligible(B, Tp, Hs, D, d->M, V, L, d->multi_head, d->use_mlp, big_greedy)) {
        if (teacher_forced) {
            GemmDesc g;      // label half of the bottom-layer gates for every step, off the decode chain: yw[s][b] = y_s[b] W_y^T
            g.A = y_all; g.lda = Vp; g.a_kc = true;
            g.B = w0p; g.ldb = Vp + Hs; g.b_kc = true;
            g.C = reserve + lay.yw; g.ldc = 4 * Hs; g.M = U * B; g.N = 4 * Hs; g.K = Vp; g.splitk = 1;
            LAS_TRY(gemm_f32(g, stream));
        }
        BigFwd p;
        if (big_greedy) {      // the kernel produces log-probabilities, arg-max and the fed-back one-hot rows itself
            p.mode = 1; p.w_c = d->w_c; p.b_c = d->b_c; p.logp = logp; p.argmax = argmax; p.y_all = y_all; p.lgp = reserve + lay.blg;
        }
        p.w0p = w0p; p.Vp = Vp;
        p.w_hh0 = d->w_hh[0]; p.w_ih1 = d->w_ih[1]; p.w_hh1 = d->w_hh[1];
        p.b_ih0 = d->b_ih[0]; p.b_hh0 = d->b_hh[0]; p.b_ih1 = d->b_ih[1]; p.b_hh1 = d->b_hh[1];
        p.w_phi = d->w_phi; p.b_phi = d->b_phi; p.feat = feat; p.keys = keys; p.yw = reserve + lay.yw;
        p.ctx_all = ctx_all; p.h_all = h_all; p.c_all = c_all; p.gates_all = gates_all; p.q_all = q_all; p.att = att;
        p.hx = reserve + lay.hx; p.qp = reserve + lay.bqp; p.flags = reinterpret_cast<unsigned*>(reserve + lay.bfl);
        if (Tp > 256) p.eg = reserve + lay.beg;
        p.B = B; p.Tp = Tp; p.U = U; p.V = V; p.relu = d->relu; p.err = err_word;
        const int rc = speller_big_fwd(p, stream);
        if (rc != LAS_ERR_UNSUPPORTED) { LAS_TRY(rc); persist_ran = true; path_note(PATH_DECODE_FWD, "big"); }
    }
    if (!persist_ran) path_note(PATH_DECODE_FWD, "stepwise");
    for (int s = 0; s < (persist_ran ? 0 : U); ++s) {
        for (int l = 0; l < L; ++l) {
            CellSeg segs[3];
            int n = 0;
            if (l == 0) {
                segs[n].x = ctx_all + (size_t)s * B * D; segs[n].ldx = D; segs[n].w = w0p + Vp; segs[n].ldw = Vp + Hs; segs[n].K = D; ++n;
            } else {
                segs[n].x = h_all + ((size_t)(l - 1) * U + s) * sH; segs[n].ldx = Hs; segs[n].w = d->w_ih[l]; segs[n].ldw = Hs; segs[n].K = Hs; ++n;
            }
            if (s > 0) {
                segs[n].x = h_all + ((size_t)l * U + s - 1) * sH; segs[n].ldx = Hs; segs[n].w = d->w_hh[l]; segs[n].ldw = Hs; segs[n].K = Hs; ++n;
            }
            if (l == 0) {
                segs[n].x = y_all + (size_t)s * B * Vp; segs[n].ldx = Vp; segs[n].w = w0p; segs[n].ldw = Vp + Hs; segs[n].K = Vp; ++n;
            }
            const size_t o = ((size_t)l * U + s) * sH;
            LAS_TRY(lstm_cell_fwd(segs, n, d->b_ih[l], d->b_hh[l], s > 0 ? c_all + o - sH : nullptr, h_all + o, c_all + o,
                                  gates_all + 4 * o, B, Hs, stream));
        }
        const int NH = d->multi_head;
        AttnFwdArgs a;
        a.h_top = h_all + ((size_t)(L - 1) * U + s) * sH;
        a.feat = feat; a.keys = d->use_mlp ? keys : feat;
        a.w_phi = d->w_phi; a.b_phi = d->b_phi; a.w_c = d->w_c; a.b_c = d->b_c;
        a.q_out = q_all ? q_all + (size_t)s * B * d->M * NH : nullptr; a.ldq = (long)d->M * NH;
        a.att_out = att + (size_t)s * NH * B * Tp; a.att_hs = (long)B * Tp;
        a.logp_out = teacher_forced ? nullptr : logp + (size_t)s * B * V;   // teacher forcing: deferred to one GEMM below
        a.argmax_out = argmax ? argmax + (size_t)s * B : nullptr;
        a.y_next = teacher_forced ? nullptr : y_all + (size_t)(s + 1) * B * Vp; a.ldy = Vp;
        a.y_mode = decode_mode;
        a.sample_noise = (!teacher_forced && decode_mode == 2) ? sample_noise + (size_t)s * B * V : nullptr;
        a.B = B; a.Tp = Tp; a.D = D; a.M = d->M; a.V = V; a.Hs = Hs; a.use_mlp = d->use_mlp; a.relu = d->relu;
        a.heads = NH;
        float* ctx_next = ctx_all + (size_t)(s + 1) * B * D;
        if (NH == 1) {
            a.ctx_out = ctx_next; a.ldctx = D;
            LAS_TRY(attn_step_fwd(a, stream));
        } else {
            // per-head attention -> concatenated contexts -> dim_reduce (las_model.py:298-314) -> character distribution
            float* ctxcat = reserve + lay.ctxcat_all + (size_t)s * B * NH * D;
            a.ctx_out = ctxcat; a.ldctx = (long)NH * D; a.phases = 1;
            LAS_TRY(attn_step_fwd(a, stream));
            CellSeg sg; sg.x = ctxcat; sg.ldx = (long)NH * D; sg.w = d->w_dr; sg.ldw = (long)NH * D; sg.K = NH * D;
            LAS_TRY(smallm_linear_nt(&sg, 1, d->b_dr, ctx_next, D, B, D, stream));
            if (!teacher_forced) {
                a.phases = 2; a.ctx_in = ctx_next;
                LAS_TRY(attn_step_fwd(a, stream));
            }
        }
    }
    if (pre_stash && !gx_written && (flags & LAS_FLAG_STASH)) {   // the per-step / classic persistent kernels ran: gx_s[b] = att_s[b] . P[b], one batched GEMM
        GemmDesc g;
        g.A = att; g.lda = (long)B * Tp; g.a_kc = true; g.sA = Tp;
        g.B = reserve + lay.pctx; g.ldb = 4 * Hs; g.b_kc = false; g.sB = (long)Tp * 4 * Hs;
        g.C = reserve + lay.gx; g.ldc = (long)B * 4 * Hs; g.sC = 4 * Hs;
        g.M = U; g.N = 4 * Hs; g.K = Tp; g.batch = B; g.splitk = 1;
        LAS_TRY(gemm_f32(g, stream));
    }
    if (pre_mh_stash && !mh_gx_written && (flags & LAS_FLAG_STASH)) {   // the per-step kernels ran: gx[s][b][h] = att[s][h][b] . P[b][:, h], batched over the utterances
        const int NH = d->multi_head;
        for (int hd = 0; hd < NH; ++hd) {
            GemmDesc g;
            g.A = att + (size_t)hd * B * Tp; g.lda = (long)NH * B * Tp; g.a_kc = true; g.sA = Tp;
            g.B = reserve + lay.pctx + (size_t)hd * 4 * Hs; g.ldb = (long)NH * 4 * Hs; g.b_kc = false; g.sB = (long)Tp * NH * 4 * Hs;
            g.C = reserve + lay.gx + (size_t)hd * 4 * Hs; g.ldc = (long)B * NH * 4 * Hs; g.sC = (long)NH * 4 * Hs;
            g.M = U; g.N = 4 * Hs; g.K = Tp; g.batch = B; g.splitk = 1;
            LAS_TRY(gemm_f32(g, stream));
        }
    }
    if (teacher_forced) {
        // character distribution of all U steps at once (reference las_model.py:181-182, per step there):
        // logits = [h_top | ctx] W_c^T + b_c as two MFMA GEMMs over U*B rows, then a row-wise log-softmax
        GemmDesc q;
        q.A = h_all + (size_t)(L - 1) * U * sH; q.lda = Hs; q.a_kc = true;
        q.B = d->w_c; q.ldb = Hs + D; q.b_kc = true; q.bias0 = d->b_c;
        q.C = logp; q.ldc = V; q.M = U * B; q.N = V; q.K = Hs + D; q.splitk = 0;      // auto split-K: only U*B/128 output tiles
        q.A2 = ctx_all + (size_t)B * D; q.B2 = d->w_c + Hs; q.K1 = Hs;                   // [h | ctx] as two K sources (D == Hs)
        LAS_TRY(gemm_f32(q, stream));
        LAS_TRY(log_softmax_rows(logp, (long)U * B, V, stream));
    }
    return LAS_OK;
}

// One decode step with caller-managed state (reference Speller.forward_step, las_model.py:178-184). Inference only.
size_t las_speller_step_workspace_floats(const las_speller_desc* d) {
    const size_t Vp = (d->V + 15) & ~15;
    return r4((size_t)d->B * Vp) + r4((size_t)d->B * d->D) + r4((size_t)4 * d->Hs * (Vp + d->Hs)) +
           r4((size_t)d->B * d->multi_head * d->D);
}

namespace {
struct StepReserve {       // stash of one differentiable decode step: post-activation gates, queries, per-head contexts
    size_t gates, q, ctxcat, total;
    explicit StepReserve(const las_speller_desc* d) {
        size_t o = 0;
        gates = o; o += r4((size_t)d->L * d->B * 4 * d->Hs);
        q = o; o += r4((size_t)d->B * (d->use_mlp ? d->M : 0) * d->multi_head);
        ctxcat = o; o += r4(d->multi_head > 1 ? (size_t)d->B * d->multi_head * d->D : 0);
        total = o;
    }
};
}  // namespace

size_t las_speller_step_reserve_floats(const las_speller_desc* d) { return StepReserve(d).total; }

int las_speller_step_fwd(const las_speller_desc* d, const float* feat, const float* keys, const float* input_word,
                         const float* h_in, const float* c_in, float* logp, float* h_out, float* c_out, float* ctx, float* att,
                         float* workspace, float* reserve, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    LAS_TRY(check_desc(d));
    LAS_REQUIRE(feat && input_word && logp && h_out && c_out && ctx && att && workspace, "step pointers");
    LAS_REQUIRE(!d->use_mlp || keys, "attention keys");
    LAS_REQUIRE((h_in == nullptr) == (c_in == nullptr), "h/c state must both be given or both be NULL");
    const int B = d->B, Hs = d->Hs, V = d->V, D = d->D, Tp = d->Tp, L = d->L;
    LAS_REQUIRE(Hs % 16 == 0, "speller hidden size must be a multiple of 16");
    const int Vp = (V + 15) & ~15;
    float* y = workspace;
    float* cin = y + r4((size_t)B * Vp);
    float* w0p = cin + r4((size_t)B * D);
    const size_t sH = (size_t)B * Hs;
    // split the reference's concatenated input [y | ctx] (las_model.py:198,236) into aligned, padded operands
    LAS_HIP_CHECK(hipMemsetAsync(y, 0, sizeof(float) * (size_t)B * Vp, stream));
    LAS_TRY(copy2d(input_word, V + Hs, y, Vp, B, V, 0, stream));
    LAS_TRY(copy2d(input_word + V, V + Hs, cin, D, B, D, 0, stream));
    LAS_TRY(build_w0p(d->w_ih[0], w0p, Hs, V, Vp, stream));
    for (int l = 0; l < L; ++l) {
        CellSeg segs[3];
        int n = 0;
        if (l == 0) {
            segs[n].x = cin; segs[n].ldx = D; segs[n].w = w0p + Vp; segs[n].ldw = Vp + Hs; segs[n].K = D; ++n;
            segs[n].x = y; segs[n].ldx = Vp; segs[n].w = w0p; segs[n].ldw = Vp + Hs; segs[n].K = Vp; ++n;
        } else {
            segs[n].x = h_out + (size_t)(l - 1) * sH; segs[n].ldx = Hs; segs[n].w = d->w_ih[l]; segs[n].ldw = Hs; segs[n].K = Hs; ++n;
        }
        if (h_in) { segs[n].x = h_in + (size_t)l * sH; segs[n].ldx = Hs; segs[n].w = d->w_hh[l]; segs[n].ldw = Hs; segs[n].K = Hs; ++n; }
        LAS_TRY(lstm_cell_fwd(segs, n, d->b_ih[l], d->b_hh[l], c_in ? c_in + (size_t)l * sH : nullptr, h_out + (size_t)l * sH,
                              c_out + (size_t)l * sH, reserve ? reserve + StepReserve(d).gates + (size_t)l * 4 * sH : nullptr, B, Hs, stream));
    }
    const int NH = d->multi_head;
    float* ctxcat = (reserve && NH > 1) ? reserve + StepReserve(d).ctxcat : w0p + r4((size_t)4 * Hs * (Vp + Hs));
    AttnFwdArgs a;
    a.h_top = h_out + (size_t)(L - 1) * sH;
    a.feat = feat; a.keys = d->use_mlp ? keys : feat;
    a.w_phi = d->w_phi; a.b_phi = d->b_phi; a.w_c = d->w_c; a.b_c = d->b_c;
    a.q_out = (reserve && d->use_mlp) ? reserve + StepReserve(d).q : nullptr; a.ldq = (long)d->M * NH;
    a.att_out = att; a.att_hs = (long)B * Tp; a.logp_out = logp; a.argmax_out = nullptr; a.y_next = nullptr; a.ldy = 0;
    a.y_mode = 1;
    a.B = B; a.Tp = Tp; a.D = D; a.M = d->M; a.V = V; a.Hs = Hs; a.use_mlp = d->use_mlp; a.relu = d->relu;
    a.heads = NH;
    if (NH == 1) {
        a.ctx_out = ctx; a.ldctx = D;
        return attn_step_fwd(a, stream);
    }
    a.ctx_out = ctxcat; a.ldctx = (long)NH * D; a.phases = 1;
    LAS_TRY(attn_step_fwd(a, stream));
    CellSeg sg; sg.x = ctxcat; sg.ldx = (long)NH * D; sg.w = d->w_dr; sg.ldw = (long)NH * D; sg.K = NH * D;
    LAS_TRY(smallm_linear_nt(&sg, 1, d->b_dr, ctx, D, B, D, stream));
    a.phases = 2; a.ctx_in = ctx;
    return attn_step_fwd(a, stream);
}

size_t las_speller_bwd_workspace_floats(const las_speller_desc* d, int U) { return SpellerBwdLayout(d, U).total; }

int las_speller_bwd(const las_speller_desc* d, const float* feat, const float* keys, const float* logp, const float* att,
                    const float* dlogp, int U, int feedback_mode0, const float* reserve, float* workspace,
                    const las_speller_grads* g, uint32_t* err_word, int flags, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    GemmArithScope arith_scope(flags, err_word);
    GradsZeroedScope zeroed_scope(flags);
    LAS_TRY(check_desc(d));
    LAS_REQUIRE(U > 0 && feat && logp && att && dlogp && reserve && workspace && g, "speller bwd pointers");
    LAS_REQUIRE(!d->use_mlp || keys, "attention keys");
    LAS_REQUIRE(g->dfeat && g->dw_c && g->db_c, "speller grad outputs");
    LAS_REQUIRE(!d->use_mlp || (g->dw_phi && g->db_phi && g->dw_psi && g->db_psi), "attention grad outputs");
    LAS_REQUIRE(d->multi_head == 1 || (g->dw_dr && g->db_dr), "dim_reduce grad outputs");
    const int B = d->B, Hs = d->Hs, V = d->V, D = d->D, Tp = d->Tp, L = d->L, M = d->M;
    for (int l = 0; l < L; ++l) LAS_REQUIRE(g->dw_ih[l] && g->dw_hh[l] && g->db_ih[l] && g->db_hh[l], "LSTM grad outputs");
    SpellerLayout lay(d, U);
    SpellerBwdLayout wl(d, U);
    const float* y_all = reserve + lay.y_all;
    const float* ctx_all = reserve + lay.ctx_all;
    const float* h_all = reserve + lay.h_all;
    const float* c_all = reserve + lay.c_all;
    const float* gates_all = reserve + lay.gates_all;
    const float* q_all = d->use_mlp ? reserve + lay.q_all : nullptr;
    float* dG_all = workspace + wl.dG_all;
    float* dz_all = workspace + wl.dz_all;
    float* dctx_all = workspace + wl.dctx_all;
    float* de_all = workspace + wl.de_all;
    float* dqpre_all = workspace + wl.dqpre_all;
    float* dh_top = workspace + wl.dh_top;
    float* dh_carry = workspace + wl.dh_carry;
    float* dc_carry = workspace + wl.dc_carry;
    float* dx0 = workspace + wl.dx0;
    float* dK = workspace + wl.dK;
    const size_t sH = (size_t)B * Hs;
    const float* keys_eff = d->use_mlp ? keys : feat;
    const float* h_top_all = h_all + (size_t)(L - 1) * U * sH;

    float* dcat_all = workspace + wl.dcat_all;
    const bool hoist = !feedback_mode0;
    if (hoist) {
        LAS_TRY(log_softmax_bwd_rows(dlogp, logp, dz_all, (long)U * B, V, stream));
        GemmDesc q;      // [dh_top | dctx] contributions of the character distribution for every step
        q.A = dz_all; q.lda = V; q.a_kc = true; q.B = d->w_c; q.ldb = Hs + D; q.b_kc = false;
        q.C = dcat_all; q.ldc = Hs + D; q.M = U * B; q.N = Hs + D; q.K = V; q.splitk = 1;
        LAS_TRY(gemm_f32(q, stream));
    }
    auto cell_pw = [&](int l, int s, bool last) {
        CellPw pw;
        const size_t o = ((size_t)l * U + s) * sH;
        pw.gates = gates_all + 4 * o; pw.c = c_all + o; pw.c_prev = s > 0 ? c_all + o - sH : nullptr;
        pw.dh_carry = last ? nullptr : dh_carry + (size_t)l * sH;
        pw.dc_in = last ? nullptr : dc_carry + (size_t)l * sH;
        pw.dG = dG_all + 4 * o; pw.dc_out = dc_carry + (size_t)l * sH;
        return pw;
    };
    const int NH = d->multi_head;
    float* dctxcat_all = NH > 1 ? workspace + wl.dctxcat_all : nullptr;
    const float* ctxcat_all = NH > 1 ? reserve + lay.ctxcat_all : nullptr;
    const bool persist_on = opt_get(OPT_SPELLER_PERSIST_BWD) != 0;
    const bool persist = persist_on && hoist && err_word && !(flags & LAS_FLAG_FORCE_GENERIC) &&
                         speller_persist_bwd_eligible(B, Tp, Hs, D, M, V, L, NH, d->use_mlp);
    bool persist_ran = persist;
    float* dx0_ctx = dx0 + V;              // gradient of the initial context (step 0's context input) and its row stride
    long ld_dx0 = V + D;
    // multi-head (heads 2..4) on the PRE backward: the forward left P and gx per head for this shape (pre_mh_stash)
    const bool pre_mh = NH > 1 && hoist && err_word && !(flags & LAS_FLAG_FORCE_GENERIC) && lay.pre_mh &&
                        (flags & LAS_FLAG_TEACHER_FORCED) && speller_persist_bwd_pre_mh_eligible(B, Tp, Hs, D, M, V, L, NH, d->use_mlp);
    if (pre_mh) {
        PersistBwd p;
        p.w_ih0 = d->w_ih[0]; p.w_hh0 = d->w_hh[0]; p.w_ih1 = d->w_ih[1]; p.w_hh1 = d->w_hh[1]; p.w_phi = d->w_phi;
        p.feat = feat; p.keys = keys; p.att = att; p.q_all = q_all; p.ctx_all = ctx_all;
        p.gates_all = gates_all; p.c_all = c_all; p.dcat_all = dcat_all;
        p.dG_all = dG_all; p.dctx_all = dctx_all; p.de_all = de_all; p.dqpre_all = dqpre_all;
        p.dx0 = dx0; p.xbuf = workspace + wl.pxbuf;
        p.B = B; p.Tp = Tp; p.U = U; p.Hs = Hs; p.V = V; p.relu = d->relu; p.err = err_word;
        p.pctx = reserve + lay.pctx; p.gxf = reserve + lay.gx; p.NH = NH; p.w_dr = d->w_dr; p.dctxcat = dctxcat_all;
        const int rc = speller_persist_bwd(p, stream);
        if (rc != LAS_ERR_UNSUPPORTED) {
            LAS_TRY(rc);
            persist_ran = true;
            // total context gradient of every step, off the chain: dctx_s = dcat_ctx_s + dG0_{s+1} W_ctx (row block -1: the initial context's),
            // then through dim_reduce for the per-head contractions of attention_deferred
            float* dctx_m1 = dctx_all - (size_t)B * D;
            LAS_HIP_CHECK(hipMemsetAsync(dctx_m1, 0, sizeof(float) * (size_t)B * D, stream));
            LAS_TRY(copy2d(dcat_all + Hs, Hs + D, dctx_all, D, (long)U * B, D, 0, stream));
            GemmDesc q;
            q.A = dG_all; q.lda = 4 * Hs; q.a_kc = true;
            q.B = reserve + lay.w0p + lay.Vp; q.ldb = lay.Vp + Hs; q.b_kc = false;
            q.C = dctx_m1; q.ldc = D; q.M = U * B; q.N = D; q.K = 4 * Hs; q.accumulate = true; q.splitk = 1;
            LAS_TRY(gemm_f32(q, stream));
            dx0_ctx = dctx_m1; ld_dx0 = D;
            GemmDesc r;
            r.A = dctx_all; r.lda = D; r.a_kc = true;
            r.B = d->w_dr; r.ldb = (long)NH * D; r.b_kc = false;
            r.C = dctxcat_all; r.ldc = (long)NH * D; r.M = U * B; r.N = NH * D; r.K = D; r.splitk = 1;
            LAS_TRY(gemm_f32(r, stream));
        }
    }
    if (persist) {
        PersistBwd p;
        p.w_ih0 = d->w_ih[0]; p.w_hh0 = d->w_hh[0]; p.w_ih1 = d->w_ih[1]; p.w_hh1 = d->w_hh[1]; p.w_phi = d->w_phi;
        p.feat = feat; p.keys = keys; p.att = att; p.q_all = q_all; p.ctx_all = ctx_all;
        p.gates_all = gates_all; p.c_all = c_all; p.dcat_all = dcat_all;
        p.dG_all = dG_all; p.dctx_all = dctx_all; p.de_all = de_all; p.dqpre_all = dqpre_all;
        p.dx0 = dx0; p.xbuf = workspace + wl.pxbuf;
        p.B = B; p.Tp = Tp; p.U = U; p.Hs = Hs; p.V = V; p.relu = d->relu; p.err = err_word;
        // PRE variant: same decision as las_speller_fwd took (same shape, same switches, same device), so P and the gx slabs exist
        const bool pre = lay.pre && (flags & LAS_FLAG_TEACHER_FORCED) && speller_persist_bwd_pre_eligible(B, Tp, Hs, D, M, V, L, NH, d->use_mlp);
        if (pre) { p.pctx = reserve + lay.pctx; p.gxf = reserve + lay.gx; }
        const int rc = speller_persist_bwd(p, stream);
        if (rc == LAS_ERR_UNSUPPORTED) persist_ran = false;      // residency check failed: the per-step kernels below run instead
        else LAS_TRY(rc);
        if (persist_ran && pre) {
            // context gradients of every step, off the chain: dctx_s = dcat_ctx_s + dG0_{s+1} W_ctx; the row block of dG0_0 is the
            // gradient of the initial context feat[:,0,:] (the headroom block in front of dctx_all)
            float* dctx_m1 = dctx_all - (size_t)B * D;
            LAS_HIP_CHECK(hipMemsetAsync(dctx_m1, 0, sizeof(float) * (size_t)B * D, stream));
            LAS_TRY(copy2d(dcat_all + Hs, Hs + D, dctx_all, D, (long)U * B, D, 0, stream));
            GemmDesc g;
            g.A = dG_all; g.lda = 4 * Hs; g.a_kc = true;
            g.B = reserve + lay.w0p + lay.Vp; g.ldb = lay.Vp + Hs; g.b_kc = false;
            g.C = dctx_m1; g.ldc = D; g.M = U * B; g.N = D; g.K = 4 * Hs; g.accumulate = true; g.splitk = 1;
            LAS_TRY(gemm_f32(g, stream));
            dx0_ctx = dctx_m1; ld_dx0 = D;
        }
    }
    // the reference's shipped sizes (Speller 1024x2, B <= 16): the whole loop in one launch with column-resident weights
    if (!persist_ran && hoist && err_word && !(flags & LAS_FLAG_FORCE_GENERIC) && d->relu <= LAS_ACT_RELU &&
        speller_big_bwd_eligible(B, Tp, Hs, D, M, V, L, NH, d->use_mlp)) {
        BigBwd p;
        p.w_ih1 = d->w_ih[1]; p.w_hh1 = d->w_hh[1]; p.w_hh0 = d->w_hh[0]; p.w0p = reserve + lay.w0p; p.Vp = lay.Vp; p.w_phi = d->w_phi;
        p.feat = feat; p.keys = keys; p.att = att; p.q_all = q_all; p.gates_all = gates_all; p.c_all = c_all; p.dcat_all = dcat_all;
        p.dG_all = dG_all; p.dctx_all = dctx_all; p.de_all = de_all; p.dqpre_all = dqpre_all;
        p.xbuf = workspace + wl.bxbuf;
        p.B = B; p.Tp = Tp; p.U = U; p.V = V; p.relu = d->relu; p.err = err_word;
        const int rc = speller_big_bwd(p, stream);
        if (rc != LAS_ERR_UNSUPPORTED) {
            LAS_TRY(rc);
            persist_ran = true;
            path_note(PATH_DECODE_BWD, "big");
            dx0_ctx = const_cast<float*>(speller_big_bwd_dx0_ctx(p.xbuf, U)); ld_dx0 = Hs;
        }
    }
    if (!persist_ran) path_note(PATH_DECODE_BWD, "stepwise");
    for (int s = persist_ran ? -1 : U - 1; s >= 0; --s) {
        const bool last = (s == U - 1);
        AttnBwdArgs a;
        a.dlogp = dlogp + (size_t)s * B * V; a.logp = logp + (size_t)s * B * V;
        a.dcat_pre = hoist ? dcat_all + (size_t)s * B * (Hs + D) : nullptr;
        a.h_top = h_top_all + (size_t)s * sH; a.ctx = ctx_all + (size_t)(s + 1) * B * D;
        a.att = att + (size_t)s * NH * B * Tp; a.att_hs = (long)B * Tp;
        a.q = q_all ? q_all + (size_t)s * B * M * NH : nullptr; a.ldq = (long)M * NH;
        a.feat = feat; a.keys = keys_eff; a.w_phi = d->w_phi; a.w_c = d->w_c;
        a.dctx_carry = last ? nullptr : dx0 + V; a.ldc = V + D;
        a.dy_carry = (feedback_mode0 && !last) ? dx0 : nullptr; a.ldy = V + D;
        a.dz_out = dz_all + (size_t)s * B * V; a.dctx_out = dctx_all + (size_t)s * B * D;
        a.de_out = de_all + (size_t)s * NH * B * Tp; a.dqpre_out = dqpre_all + (size_t)s * B * M * NH;
        a.B = B; a.Tp = Tp; a.D = D; a.M = M; a.V = V; a.Hs = Hs; a.use_mlp = d->use_mlp; a.relu = d->relu;
        a.heads = NH;
        if (NH == 1) {
            a.dh_top_out = nullptr;
            a.pw = cell_pw(L - 1, s, last);                // top layer's pointwise backward fused into this kernel
            LAS_TRY(attn_step_bwd(a, stream));
        } else {
            // (1) character-distribution part -> total context gradient + decoder-state part 0
            a.phases = 1; a.dh_top_out = dh_top; a.dh_hs = 0;
            LAS_TRY(attn_step_bwd(a, stream));
            // (2) dim_reduce backward: gradient of the concatenated per-head contexts
            float* dcc = dctxcat_all + (size_t)s * B * NH * D;
            LAS_TRY(smallm_gemm_nn2(dctx_all + (size_t)s * B * D, D, B, D, d->w_dr, (long)NH * D, dcc, (long)NH * D, NH * D, nullptr, 0,
                                    nullptr, 0, 0, CellPw(), Hs, stream));
            // (3) per-head attention backward -> decoder-state parts 1..NH
            a.phases = 2; a.dctx_in = dcc; a.ld_dctx_in = (long)NH * D; a.dh_top_out = dh_top + sH; a.dh_hs = (long)sH;
            LAS_TRY(attn_step_bwd(a, stream));
            // (4) sum the parts and apply the top cell's pointwise backward
            const CellPw pw = cell_pw(L - 1, s, last);
            LAS_TRY(lstm_cell_bwd_pointwise(dh_top, NH + 1, (long)sH, pw.dh_carry, pw.dc_in, pw.gates, pw.c, pw.c_prev, pw.dG,
                                            pw.dc_out, B, Hs, stream));
        }
        for (int l = L - 1; l >= 0; --l) {
            const float* dGl = dG_all + ((size_t)l * U + s) * 4 * sH;
            if (l > 0) {   // dh of layer l-1 feeds that layer's pointwise step in the epilogue; dh_carry[l] for step s-1
                LAS_TRY(smallm_gemm_nn2(dGl, 4 * Hs, B, 4 * Hs, d->w_ih[l], Hs, nullptr, Hs, Hs, d->w_hh[l], Hs,
                                        dh_carry + (size_t)l * sH, Hs, Hs, cell_pw(l - 1, s, last), Hs, stream));
            } else {
                LAS_TRY(smallm_gemm_nn2(dGl, 4 * Hs, B, 4 * Hs, d->w_ih[0], V + Hs, dx0, V + D, V + D, d->w_hh[0], Hs, dh_carry,
                                        Hs, Hs, CellPw(), Hs, stream));
            }
        }
    }

    // ---- deferred (loop-invariant-shaped) contractions, all MFMA GEMMs -------------------------
    const int UB = U * B;
    bool zg;
    {
        std::vector<std::pair<float*, size_t>> outs;
        for (int l = 0; l < L; ++l) {
            outs.push_back({g->dw_ih[l], (size_t)4 * Hs * (l == 0 ? V + Hs : Hs)});
            outs.push_back({g->dw_hh[l], (size_t)4 * Hs * Hs});
            outs.push_back({g->db_ih[l], (size_t)4 * Hs});
            outs.push_back({g->db_hh[l], (size_t)4 * Hs});
        }
        if (d->use_mlp) {
            outs.push_back({g->dw_phi, (size_t)M * NH * Hs}); outs.push_back({g->db_phi, (size_t)M * NH});
            outs.push_back({g->dw_psi, (size_t)M * D}); outs.push_back({g->db_psi, (size_t)M});
        }
        if (NH > 1) { outs.push_back({g->dw_dr, (size_t)D * NH * D}); outs.push_back({g->db_dr, (size_t)D}); }
        outs.push_back({g->dw_c, (size_t)V * (Hs + D)}); outs.push_back({g->db_c, (size_t)V});
        zg = zero_if_contiguous(outs, stream);
    }
    {
        AttnDeferred x;
        x.feat = feat; x.keys = keys; x.att = att; x.q_all = q_all; x.de_all = de_all; x.dqpre_all = dqpre_all; x.dctx_all = dctx_all;
        x.dctxcat_all = dctxcat_all; x.ctxcat_all = ctxcat_all; x.h_top_all = h_top_all; x.dK = dK; x.U = U; x.zg = zg;
        x.dx0_ctx = dx0_ctx; x.ld_dx0 = ld_dx0; x.skip_dw_phi = NH == 1;       // single head: dW_phi rides in the grouped launch below
        LAS_TRY(attention_deferred(d, x, g, stream));
    }
    {   // every remaining weight gradient (K = U*B rows each) in ONE grouped stream-K launch
        // LAS_FLAG_DEFER_DW: off the critical path (side stream, XCD partition, joined by las_join_deferred) — dfeat above is what the caller waits for
        hipStream_t main_stream = stream;
        int xcd_lo = ((flags & LAS_FLAG_DEFER_DW) && !(flags & LAS_FLAG_FORCE_GENERIC) && opt_get(OPT_DEFER_DW) != 0) ? rec_confine_xcds(B, Hs / 2) : 0;
        bool deferred = false;
        if (xcd_lo > 0) {
            if (hipStream_t s = defer_side().begin(main_stream)) { stream = s; deferred = true; } else xcd_lo = 0;
        }
        path_note(PATH_DW, deferred ? "deferred" : "inline");
        GemmDesc gs[8];
        int n = 0;
        auto add = [&](const float* A, long lda, const float* Bm, long ldb, float* C, long ldc, int Mv, int Nv, int Kv) {
            GemmDesc& q = gs[n++];
            q.A = A; q.lda = lda; q.a_kc = false; q.B = Bm; q.ldb = ldb; q.b_kc = false; q.C = C; q.ldc = ldc; q.M = Mv; q.N = Nv; q.K = Kv;
            q.c_zeroed = zg;
        };
        auto flush = [&]() { const int rc = gemm_f32_group(gs, n, stream, xcd_lo); n = 0; return rc; };
        // dW_c = dz^T [h_top | ctx]
        add(dz_all, V, h_top_all, Hs, g->dw_c, Hs + D, V, Hs, UB);
        add(dz_all, V, ctx_all + (size_t)B * D, D, g->dw_c + Hs, Hs + D, V, D, UB);
        if (d->use_mlp && NH == 1) add(dqpre_all, M, h_top_all, Hs, g->dw_phi, Hs, M, Hs, UB);      // dW_phi = dqpre^T h_top
        for (int l = 0; l < L; ++l) {
            const float* dGl = dG_all + (size_t)l * U * 4 * sH;
            if (n + 3 > 8) LAS_TRY(flush());
            if (l == 0) {
                add(dGl, 4 * Hs, y_all, lay.Vp, g->dw_ih[0], V + Hs, 4 * Hs, V, UB);
                add(dGl, 4 * Hs, ctx_all, D, g->dw_ih[0] + V, V + Hs, 4 * Hs, D, UB);
            } else {
                add(dGl, 4 * Hs, h_all + (size_t)(l - 1) * U * sH, Hs, g->dw_ih[l], Hs, 4 * Hs, Hs, UB);
            }
            if (U > 1) {   // dW_hh = sum_{s>=1} dG_s^T h_{s-1}
                add(dGl + 4 * sH, 4 * Hs, h_all + (size_t)l * U * sH, Hs, g->dw_hh[l], Hs, 4 * Hs, Hs, (U - 1) * B);
            } else if (!zg) {
                LAS_HIP_CHECK(hipMemsetAsync(g->dw_hh[l], 0, sizeof(float) * 4 * Hs * Hs, stream));
            }
        }
        LAS_TRY(flush());
        ColsumJob cj[COLSUM_MAX_JOBS];
        int nj = 0;
        cj[nj++] = {dz_all, V, UB, V, g->db_c, nullptr};
        for (int l = 0; l < L; ++l) cj[nj++] = {dG_all + (size_t)l * U * 4 * sH, (long)4 * Hs, UB, 4 * Hs, g->db_ih[l], g->db_hh[l]};
        LAS_TRY(colsum_multi(cj, nj, zg, stream));       // every LSTM / character-distribution bias gradient in one launch
        if (deferred) LAS_TRY(defer_side().end());
    }
    return LAS_OK;
}


// ---------------------------------------------------------------------------------------------- differentiable decode step
namespace {
struct StepBwdLayout {
    size_t dG, dz, dctx, de, dqpre, dh_top, dh_carry, dc_carry, dK, dctxcat, zeros, total;
    explicit StepBwdLayout(const las_speller_desc* d) {
        size_t o = 0;
        const size_t B = d->B, NH = d->multi_head, Mq = d->use_mlp ? d->M : d->Hs;
        dG = o; o += r4((size_t)d->L * B * 4 * d->Hs);
        dz = o; o += r4(B * d->V);
        dctx = o; o += r4(B * d->D);
        de = o; o += r4(NH * B * d->Tp);
        dqpre = o; o += r4(B * d->M * NH + 4);
        dh_top = o; o += r4((NH + 1) * B * d->Hs);
        dh_carry = o; o += r4((size_t)d->L * B * d->Hs);
        dc_carry = o; o += r4((size_t)d->L * B * d->Hs);
        dK = o; o += r4(B * d->Tp * Mq);
        dctxcat = o; o += r4(NH > 1 ? B * NH * d->D : 0);
        zeros = o; o += r4(std::max<size_t>(B * d->V, B * d->D));
        total = o;
    }
};

// zero every gradient output that lies in one contiguous block with one memset (the views of a flat gradient buffer)
bool zero_speller_grads(const las_speller_desc* d, const las_speller_grads* g, bool lstm, hipStream_t stream) {
    const int Hs = d->Hs, V = d->V, D = d->D, M = d->M, NH = d->multi_head, L = d->L;
    std::vector<std::pair<float*, size_t>> outs;
    if (lstm) {
        for (int l = 0; l < L; ++l) {
            outs.push_back({g->dw_ih[l], (size_t)4 * Hs * (l == 0 ? V + Hs : Hs)});
            outs.push_back({g->dw_hh[l], (size_t)4 * Hs * Hs});
            outs.push_back({g->db_ih[l], (size_t)4 * Hs});
            outs.push_back({g->db_hh[l], (size_t)4 * Hs});
        }
        outs.push_back({g->dw_c, (size_t)V * (Hs + D)}); outs.push_back({g->db_c, (size_t)V});
    }
    if (d->use_mlp) {
        outs.push_back({g->dw_phi, (size_t)M * NH * Hs}); outs.push_back({g->db_phi, (size_t)M * NH});
        outs.push_back({g->dw_psi, (size_t)M * D}); outs.push_back({g->db_psi, (size_t)M});
    }
    if (NH > 1) { outs.push_back({g->dw_dr, (size_t)D * NH * D}); outs.push_back({g->db_dr, (size_t)D}); }
    return outs.empty() ? false : zero_if_contiguous(outs, stream);
}
}  // namespace

size_t las_speller_step_bwd_workspace_floats(const las_speller_desc* d) { return StepBwdLayout(d).total; }

int las_speller_step_bwd(const las_speller_desc* d, const float* feat, const float* keys, const float* input_word, const float* h_in,
                         const float* c_in, const float* logp, const float* h_out, const float* c_out, const float* ctx,
                         const float* att, const float* reserve, const float* dlogp, const float* dh_out, const float* dc_out,
                         const float* dctx_in, float* dinput_word, float* dh_in, float* dc_in, const las_speller_grads* g,
                         float* workspace, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    LAS_TRY(check_desc(d));
    LAS_REQUIRE(feat && input_word && logp && h_out && c_out && ctx && att && reserve && workspace && g, "step bwd pointers");
    LAS_REQUIRE((h_in == nullptr) == (c_in == nullptr), "h/c state must both be given or both be NULL");
    LAS_REQUIRE(dinput_word && dh_in && dc_in && g->dfeat && g->dw_c && g->db_c, "step bwd outputs");
    LAS_REQUIRE(!d->use_mlp || (keys && g->dw_phi && g->db_phi && g->dw_psi && g->db_psi), "attention grad outputs");
    LAS_REQUIRE(d->multi_head == 1 || (g->dw_dr && g->db_dr), "dim_reduce grad outputs");
    const int B = d->B, Hs = d->Hs, V = d->V, D = d->D, Tp = d->Tp, L = d->L, M = d->M, NH = d->multi_head;
    for (int l = 0; l < L; ++l) LAS_REQUIRE(g->dw_ih[l] && g->dw_hh[l] && g->db_ih[l] && g->db_hh[l], "LSTM grad outputs");
    StepReserve rl(d);
    StepBwdLayout wl(d);
    const size_t sH = (size_t)B * Hs;
    const float* gates = reserve + rl.gates;
    const float* q = d->use_mlp ? reserve + rl.q : nullptr;
    const float* ctxcat = NH > 1 ? reserve + rl.ctxcat : nullptr;
    float* dG = workspace + wl.dG; float* dz = workspace + wl.dz; float* dctx = workspace + wl.dctx; float* de = workspace + wl.de;
    float* dqpre = workspace + wl.dqpre; float* dh_top = workspace + wl.dh_top; float* dh_carry = workspace + wl.dh_carry;
    float* dc_carry = workspace + wl.dc_carry; float* dK = workspace + wl.dK; float* dctxcat = workspace + wl.dctxcat;
    float* zeros = workspace + wl.zeros;
    const float* h_top = h_out + (size_t)(L - 1) * sH;
    const float* keys_eff = d->use_mlp ? keys : feat;
    // carries = gradients flowing into this step's outputs (from the next step or from the caller's loss)
    LAS_HIP_CHECK(hipMemsetAsync(zeros, 0, sizeof(float) * std::max<size_t>((size_t)B * V, (size_t)B * D), stream));
    if (dh_out) LAS_HIP_CHECK(hipMemcpyAsync(dh_carry, dh_out, sizeof(float) * L * sH, hipMemcpyDeviceToDevice, stream));
    else LAS_HIP_CHECK(hipMemsetAsync(dh_carry, 0, sizeof(float) * L * sH, stream));
    if (dc_out) LAS_HIP_CHECK(hipMemcpyAsync(dc_carry, dc_out, sizeof(float) * L * sH, hipMemcpyDeviceToDevice, stream));
    else LAS_HIP_CHECK(hipMemsetAsync(dc_carry, 0, sizeof(float) * L * sH, stream));
    auto cell_pw = [&](int l) {
        CellPw pw;
        pw.gates = gates + (size_t)l * 4 * sH; pw.c = c_out + (size_t)l * sH; pw.c_prev = c_in ? c_in + (size_t)l * sH : nullptr;
        pw.dh_carry = dh_carry + (size_t)l * sH; pw.dc_in = dc_carry + (size_t)l * sH;
        pw.dG = dG + (size_t)l * 4 * sH; pw.dc_out = dc_carry + (size_t)l * sH;
        return pw;
    };
    AttnBwdArgs a;
    a.dlogp = dlogp ? dlogp : zeros; a.logp = logp; a.dcat_pre = nullptr;
    a.h_top = h_top; a.ctx = ctx; a.att = att; a.att_hs = (long)B * Tp;
    a.q = q; a.ldq = (long)M * NH;
    a.feat = feat; a.keys = keys_eff; a.w_phi = d->w_phi; a.w_c = d->w_c;
    a.dctx_carry = dctx_in; a.ldc = D; a.dy_carry = nullptr; a.ldy = 0;
    a.dz_out = dz; a.dctx_out = dctx; a.de_out = de; a.dqpre_out = dqpre;
    a.B = B; a.Tp = Tp; a.D = D; a.M = M; a.V = V; a.Hs = Hs; a.use_mlp = d->use_mlp; a.relu = d->relu; a.heads = NH;
    if (NH == 1) {
        a.dh_top_out = nullptr; a.pw = cell_pw(L - 1);
        LAS_TRY(attn_step_bwd(a, stream));
    } else {
        a.phases = 1; a.dh_top_out = dh_top; a.dh_hs = 0;
        LAS_TRY(attn_step_bwd(a, stream));
        LAS_TRY(smallm_gemm_nn2(dctx, D, B, D, d->w_dr, (long)NH * D, dctxcat, (long)NH * D, NH * D, nullptr, 0, nullptr, 0, 0, CellPw(), Hs, stream));
        a.phases = 2; a.dctx_in = dctxcat; a.ld_dctx_in = (long)NH * D; a.dh_top_out = dh_top + sH; a.dh_hs = (long)sH;
        LAS_TRY(attn_step_bwd(a, stream));
        const CellPw pw = cell_pw(L - 1);
        LAS_TRY(lstm_cell_bwd_pointwise(dh_top, NH + 1, (long)sH, pw.dh_carry, pw.dc_in, pw.gates, pw.c, pw.c_prev, pw.dG, pw.dc_out, B, Hs, stream));
    }
    for (int l = L - 1; l >= 0; --l) {
        const float* dGl = dG + (size_t)l * 4 * sH;
        if (l > 0) {
            LAS_TRY(smallm_gemm_nn2(dGl, 4 * Hs, B, 4 * Hs, d->w_ih[l], Hs, nullptr, Hs, Hs, d->w_hh[l], Hs, dh_carry + (size_t)l * sH, Hs, Hs,
                                    cell_pw(l - 1), Hs, stream));
        } else {
            LAS_TRY(smallm_gemm_nn2(dGl, 4 * Hs, B, 4 * Hs, d->w_ih[0], V + Hs, dinput_word, V + Hs, V + Hs, d->w_hh[0], Hs, dh_carry, Hs, Hs,
                                    CellPw(), Hs, stream));
        }
    }
    LAS_HIP_CHECK(hipMemcpyAsync(dh_in, dh_carry, sizeof(float) * L * sH, hipMemcpyDeviceToDevice, stream));
    LAS_HIP_CHECK(hipMemcpyAsync(dc_in, dc_carry, sizeof(float) * L * sH, hipMemcpyDeviceToDevice, stream));

    // ---- parameter gradients and dfeat of this one step (K = B rows)
    const bool zg = zero_speller_grads(d, g, true, stream);
    {
        AttnDeferred x;
        x.feat = feat; x.keys = keys; x.att = att; x.q_all = q; x.de_all = de; x.dqpre_all = dqpre; x.dctx_all = dctx;
        x.dctxcat_all = dctxcat; x.ctxcat_all = ctxcat; x.h_top_all = h_top; x.dK = dK; x.U = 1; x.zg = zg;
        LAS_TRY(attention_deferred(d, x, g, stream));
    }
    auto tn = [&](const float* A, long lda, const float* Bm, long ldb, float* C, long ldc, int Mv, int Nv) {
        GemmDesc q2;
        q2.A = A; q2.lda = lda; q2.a_kc = false; q2.B = Bm; q2.ldb = ldb; q2.b_kc = false; q2.C = C; q2.ldc = ldc; q2.M = Mv; q2.N = Nv; q2.K = B;
        q2.c_zeroed = zg; q2.splitk = 1;
        return gemm_f32(q2, stream);
    };
    LAS_TRY(tn(dz, V, h_top, Hs, g->dw_c, Hs + D, V, Hs));
    LAS_TRY(tn(dz, V, ctx, D, g->dw_c + Hs, Hs + D, V, D));
    LAS_TRY(colsum(dz, V, B, V, g->db_c, zg, stream));
    for (int l = 0; l < L; ++l) {
        const float* dGl = dG + (size_t)l * 4 * sH;
        if (l == 0) LAS_TRY(tn(dGl, 4 * Hs, input_word, V + Hs, g->dw_ih[0], V + Hs, 4 * Hs, V + Hs));
        else LAS_TRY(tn(dGl, 4 * Hs, h_out + (size_t)(l - 1) * sH, Hs, g->dw_ih[l], Hs, 4 * Hs, Hs));
        if (h_in) LAS_TRY(tn(dGl, 4 * Hs, h_in + (size_t)l * sH, Hs, g->dw_hh[l], Hs, 4 * Hs, Hs));
        else if (!zg) LAS_HIP_CHECK(hipMemsetAsync(g->dw_hh[l], 0, sizeof(float) * 4 * Hs * Hs, stream));
        LAS_TRY(colsum(dGl, 4 * Hs, B, 4 * Hs, g->db_ih[l], zg, stream, g->db_hh[l]));
    }
    return LAS_OK;
}

// ---------------------------------------------------------------------------------------------- stand-alone attention
size_t las_attention_reserve_floats(const las_speller_desc* d) {
    return r4((size_t)d->B * (d->use_mlp ? d->M : 0) * d->multi_head) + r4(d->multi_head > 1 ? (size_t)d->B * d->multi_head * d->D : 0);
}

int las_attention_fwd(const las_speller_desc* d, const float* feat, const float* keys, const float* decoder_state, float* att,
                      float* ctx, float* reserve, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    LAS_TRY(check_attn_desc(d));
    LAS_REQUIRE(feat && decoder_state && att && ctx && reserve, "attention pointers");
    LAS_REQUIRE(!d->use_mlp || keys, "attention keys");
    const int B = d->B, NH = d->multi_head, D = d->D;
    float* q = reserve;
    float* ctxcat = reserve + r4((size_t)B * (d->use_mlp ? d->M : 0) * NH);
    AttnFwdArgs a;
    a.h_top = decoder_state; a.feat = feat; a.keys = d->use_mlp ? keys : feat;
    a.w_phi = d->w_phi; a.b_phi = d->b_phi; a.w_c = d->w_c; a.b_c = d->b_c;
    a.q_out = d->use_mlp ? q : nullptr; a.ldq = (long)d->M * NH;
    a.att_out = att; a.att_hs = (long)B * d->Tp; a.logp_out = nullptr; a.argmax_out = nullptr; a.y_next = nullptr; a.ldy = 0; a.y_mode = 1;
    a.B = B; a.Tp = d->Tp; a.D = D; a.M = d->M; a.V = d->V; a.Hs = d->Hs; a.use_mlp = d->use_mlp; a.relu = d->relu; a.heads = NH;
    if (NH == 1) {
        a.ctx_out = ctx; a.ldctx = D;
        return attn_step_fwd(a, stream);
    }
    a.ctx_out = ctxcat; a.ldctx = (long)NH * D; a.phases = 1;
    LAS_TRY(attn_step_fwd(a, stream));
    CellSeg sg; sg.x = ctxcat; sg.ldx = (long)NH * D; sg.w = d->w_dr; sg.ldw = (long)NH * D; sg.K = NH * D;
    return smallm_linear_nt(&sg, 1, d->b_dr, ctx, D, B, D, stream);
}

size_t las_attention_bwd_workspace_floats(const las_speller_desc* d) {
    const size_t B = d->B, NH = d->multi_head, Mq = d->use_mlp ? d->M : d->Hs;
    return r4(NH * B * d->Tp) + r4(B * d->M * NH + 4) + r4(NH * B * d->Hs) + r4(B * d->Tp * Mq) + r4(NH > 1 ? B * NH * d->D : 0);
}

int las_attention_bwd(const las_speller_desc* d, const float* feat, const float* keys, const float* decoder_state, const float* att,
                      const float* reserve, const float* dctx, float* ddecoder_state, const las_speller_grads* g, float* workspace,
                      void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    LAS_TRY(check_attn_desc(d));
    LAS_REQUIRE(feat && decoder_state && att && reserve && dctx && ddecoder_state && g && g->dfeat && workspace, "attention bwd pointers");
    LAS_REQUIRE(!d->use_mlp || (keys && g->dw_phi && g->db_phi && g->dw_psi && g->db_psi), "attention grad outputs");
    LAS_REQUIRE(d->multi_head == 1 || (g->dw_dr && g->db_dr), "dim_reduce grad outputs");
    const int B = d->B, Hs = d->Hs, D = d->D, Tp = d->Tp, M = d->M, NH = d->multi_head;
    const size_t sH = (size_t)B * Hs;
    const size_t Mq = d->use_mlp ? M : Hs;
    const float* q = d->use_mlp ? reserve : nullptr;
    const float* ctxcat = reserve + r4((size_t)B * (d->use_mlp ? M : 0) * NH);
    float* de = workspace;
    float* dqpre = de + r4((size_t)NH * B * Tp);
    float* dh_parts = dqpre + r4((size_t)B * M * NH + 4);
    float* dK = dh_parts + r4((size_t)NH * sH);
    float* dctxcat = dK + r4((size_t)B * Tp * Mq);
    const float* dctx_heads = dctx;
    long ld_dctx = D;
    if (NH > 1) {      // dim_reduce backward: gradient of the concatenated per-head contexts
        LAS_TRY(smallm_gemm_nn2(dctx, D, B, D, d->w_dr, (long)NH * D, dctxcat, (long)NH * D, NH * D, nullptr, 0, nullptr, 0, 0, CellPw(), Hs, stream));
        dctx_heads = dctxcat; ld_dctx = (long)NH * D;
    }
    AttnBwdArgs a;
    a.dlogp = nullptr; a.logp = nullptr; a.dcat_pre = nullptr; a.h_top = decoder_state; a.ctx = nullptr;
    a.att = att; a.att_hs = (long)B * Tp; a.q = q; a.ldq = (long)M * NH;
    a.feat = feat; a.keys = d->use_mlp ? keys : feat; a.w_phi = d->w_phi; a.w_c = d->w_c;
    a.dctx_carry = nullptr; a.ldc = 0; a.dy_carry = nullptr; a.ldy = 0;
    a.dz_out = nullptr; a.dctx_out = nullptr; a.de_out = de; a.dqpre_out = dqpre;
    a.B = B; a.Tp = Tp; a.D = D; a.M = M; a.V = d->V; a.Hs = Hs; a.use_mlp = d->use_mlp; a.relu = d->relu; a.heads = NH;
    a.phases = 2; a.dctx_in = dctx_heads; a.ld_dctx_in = ld_dctx;
    a.dh_top_out = NH == 1 ? ddecoder_state : dh_parts; a.dh_hs = (long)sH;
    LAS_TRY(attn_step_bwd(a, stream));
    if (NH > 1) {
        LAS_HIP_CHECK(hipMemcpyAsync(ddecoder_state, dh_parts, sizeof(float) * sH, hipMemcpyDeviceToDevice, stream));
        for (int hd = 1; hd < NH; ++hd) LAS_TRY(add_inplace(ddecoder_state, dh_parts + (size_t)hd * sH, (long)sH, stream));
    }
    const bool zg = zero_speller_grads(d, g, false, stream);
    AttnDeferred x;
    x.feat = feat; x.keys = keys; x.att = att; x.q_all = q; x.de_all = de; x.dqpre_all = dqpre; x.dctx_all = dctx;
    x.dctxcat_all = dctxcat; x.ctxcat_all = ctxcat; x.h_top_all = decoder_state; x.dK = dK; x.U = 1; x.zg = zg;
    return attention_deferred(d, x, g, stream);
}

// ---------------------------------------------------------------------------------------------- caller-side contract
int las_ls_loss(const float* logp, int64_t stride_u, int64_t stride_b, const int64_t* labels_onehot, int U, int U_lab, int B, int V,
                float smoothing, float* loss, float* dlogp, int64_t dstride_u, int64_t dstride_b, float* scratch, void* stream) {
    LAS_REQUIRE(logp && labels_onehot && loss && scratch, "loss pointers");
    LAS_REQUIRE(U > 0 && U <= U_lab && B > 0 && V > 0, "loss dims");
    return ls_loss(logp, stride_u, stride_b, (const long long*)labels_onehot, U, U_lab, B, V, smoothing, scratch, loss, dlogp,
                   dstride_u, dstride_b, (hipStream_t)stream);
}

int las_letter_error_rate(const float* logp, int64_t stride_u, int64_t stride_b, const int64_t* labels_onehot, int U, int U_lab,
                          int B, int V, float* ler_out, int32_t* work, void* stream) {
    LAS_REQUIRE(logp && labels_onehot && ler_out, "LER pointers");      // work: unused since ABI 9 (may be NULL)
    LAS_REQUIRE(U > 0 && U <= U_lab && B > 0 && V > 0, "LER dims");
    return ler(logp, stride_u, stride_b, (const long long*)labels_onehot, U, U_lab, B, V, ler_out, work, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------- input side
int las_collate_pad(const float* packed_feat, const int64_t* feat_offsets, const int64_t* packed_labels,
                    const int64_t* label_offsets, int B, int T, int F, int U, int V, float* inputs, int64_t* targets,
                    void* stream) {
    LAS_REQUIRE(B > 0 && T > 0 && F > 0 && U > 0 && V > 0, "collate dims");
    LAS_REQUIRE(packed_feat && feat_offsets && packed_labels && label_offsets && inputs && targets, "collate pointers");
    return collate_pad(packed_feat, (const long long*)feat_offsets, (const long long*)packed_labels, (const long long*)label_offsets,
                       B, T, F, U, V, inputs, (long long*)targets, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------- building blocks
int las_gemm_f32(const float* A, const float* B, float* C, const float* bias0, const float* bias1, int M, int N, int K,
                 int64_t lda, int64_t ldb, int64_t ldc, int a_kc, int b_kc, int batch, int64_t sA, int64_t sB, int64_t sC,
                 int splitk, int accumulate, int relu, void* stream) {
    GemmDesc g;
    g.A = A; g.B = B; g.C = C; g.bias0 = bias0; g.bias1 = bias1; g.M = M; g.N = N; g.K = K;
    g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.a_kc = a_kc; g.b_kc = b_kc; g.batch = batch; g.sA = sA; g.sB = sB; g.sC = sC;
    g.splitk = splitk; g.accumulate = accumulate; g.relu = relu;
    return gemm_f32(g, (hipStream_t)stream);
}

size_t las_planes_bytes(int64_t rows, int64_t ld) { return planes_floats((size_t)rows, (size_t)ld) * sizeof(float); }
int las_split_planes(const float* src, int64_t ld_src, int R, int C, void* dst, int64_t ld_dst, void* stream) {
    return split_planes(src, ld_src, R, C, dst, ld_dst, (hipStream_t)stream);
}
int las_gemm_planes(const void* A, const void* B, float* C, const float* bias0, const float* bias1, int M, int N, int K,
                    int64_t lda, int64_t ldb, int64_t ldc, int a_kc, int b_kc, int batch, int64_t sA, int64_t sB, int64_t sC,
                    int splitk, int accumulate, int relu, void* stream) {
    GemmDesc g;
    g.A = reinterpret_cast<const float*>(A); g.B = reinterpret_cast<const float*>(B); g.C = C; g.bias0 = bias0; g.bias1 = bias1;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.a_kc = a_kc; g.b_kc = b_kc; g.batch = batch; g.sA = sA; g.sB = sB; g.sC = sC;
    g.splitk = splitk; g.accumulate = accumulate; g.relu = relu; g.planes = true;
    return gemm_f32(g, (hipStream_t)stream);
}

int las_gemm_get_arith(void) { return gemm_get_arith(); }
void las_gemm_set_arith(int mode) { gemm_set_arith(mode); }
void las_gemm_set_tuning(int key, int64_t value) { gemm_set_tuning(key, (long)value); }
int las_gemm_check(void) { return gemm_sk_check(); }

int las_gemm_f32_group(const las_gemm_desc* descs, int n, void* stream) {
    LAS_REQUIRE(descs != nullptr && n >= 1 && n <= 8, "gemm group");
    GemmDesc g[8];
    for (int i = 0; i < n; ++i) {
        const las_gemm_desc& d = descs[i];
        g[i].A = d.A; g[i].B = d.B; g[i].C = d.C; g[i].A2 = d.A2; g[i].B2 = d.B2; g[i].K1 = d.K1;
        g[i].M = d.M; g[i].N = d.N; g[i].K = d.K; g[i].lda = d.lda; g[i].ldb = d.ldb; g[i].ldc = d.ldc;
        g[i].a_kc = d.a_kc; g[i].b_kc = d.b_kc; g[i].accumulate = d.accumulate; g[i].c_zeroed = d.c_zeroed; g[i].planes = d.planes != 0;
    }
    return gemm_f32_group(g, n, (hipStream_t)stream);
}

size_t las_rec_xbuf_bytes(int B, int H) { return rec_xbuf_bytes(B, H); }

int las_pblstm_rec_fwd(float* gates, const float* w_hh_f, const float* w_hh_r, float* out, float* cbuf, float* hprev, int B,
                       int T, int H, void* xbuf, uint32_t* err_word, int flags, void* stream) {
    return pblstm_rec_fwd(gates, w_hh_f, w_hh_r, out, cbuf, hprev, B, T, H, flags & LAS_FLAG_STASH,
                          (unsigned long long*)xbuf, err_word, flags & LAS_FLAG_FORCE_GENERIC, (hipStream_t)stream);
}

}  // extern "C"
