// Internal C++ interface between the kernel translation units and the C-ABI layer (las_capi.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace las {

// ---- gemm_f32.hip ----------------------------------------------------------------------
struct GemmDesc {
    const float* A = nullptr;      // A(m,k) = a_kc ? A[m*lda + k] : A[k*lda + m]
    const float* B = nullptr;      // B(k,n) = b_kc ? B[n*ldb + k] : B[k*ldb + n]
    float* C = nullptr;            // C[m*ldc + n]
    const float* bias0 = nullptr;  // optional per-N bias (added once)
    const float* bias1 = nullptr;
    int M = 0, N = 0, K = 0;
    long lda = 0, ldb = 0, ldc = 0;
    bool a_kc = true, b_kc = true;
    int batch = 1;
    long sA = 0, sB = 0, sC = 0;   // batch strides (elements)
    long sBias0 = 0, sBias1 = 0;   // batch strides of the biases (elements; may be a pointer difference between two tensors)
    int splitk = 0;                // 0 = auto, 1 = none, >1 = forced (atomic accumulation)
    bool accumulate = false;       // C += ...
    bool c_zeroed = false;         // the caller has already zeroed C (split-K then skips its own memset)
    bool relu = false;
    // optional second source along K (C = [A | A2][B ; B2]): k >= K1 reads A2 / B2 at k - K1; same layouts and leading dimensions
    const float* A2 = nullptr; const float* B2 = nullptr; int K1 = 0;
    // pre-split operands: A / B / A2 / B2 point at P8x3 granule buffers (split_planes, or a producer kernel) of the fp32 matrices the other
    // fields describe; lda / ldb / sA / sB stay element counts.  Both operands or neither.  Takes the bf16-MFMA path whatever GEMM_ARITH says.
    bool planes = false;
};
int gemm_f32(const GemmDesc& d, hipStream_t stream);
// 256 x 256 tiles of the split-operand arithmetic (gemm_big.hip): LAS_ERR_UNSUPPORTED (no error text) when the shape does not fill such
// tiles / the chip or the arithmetic mode is not 1 — gemm_f32 / gemm_f32_group try these first and fall through to the 128-tile kernels
int gemm_big(const GemmDesc& d, hipStream_t stream);
int gemm_big_group(const GemmDesc* ds, int n, hipStream_t stream);
// fp32 R x C matrix (row stride ld_src) -> its P8x3 image (three bf16 terms per element, x = p1 + p2 + p3 exactly; 16-byte granule
// (r, c / 8, plane) at index (r (ld_dst / 8) + c / 8) 3 + plane).  C, ld_dst multiples of 8; 6 bytes per element.
int split_planes(const float* src, long ld_src, int R, int C, void* dst, long ld_dst, hipStream_t stream);
inline size_t planes_floats(size_t rows, size_t ld) { return (rows * ld * 6 + 15) / 16 * 4; }      // size of a P8x3 image in floats (16-byte multiple)
// n independent GEMMs of one operand layout, no bias / activation, outputs pre-zeroed (or accumulated onto): ONE launch, the
// k-iterations of all problems spread evenly over the resident workgroups.  Falls back to n launches when not groupable.
// xcd_lo > 0: the launch leaves XCDs [0, xcd_lo) alone (their workgroups exit at once; the 128-tile group kernel only): for a group that runs on a
// side stream beside a chain kernel confined to those XCDs (pblstm_rec_bwd's confine_nx).  Falls back to one launch per problem as before.
// drawn_runs > 0: the group's k-iterations are cut into that many equal runs which the workgroups DRAW from a counter (instead of one assigned run
// each): for a group that shares the chip with another GEMM on a second stream — whichever workgroups get CUs first do the work.
int gemm_f32_group(const GemmDesc* ds, int n, hipStream_t stream, int xcd_lo = 0, int drawn_runs = 0);
// LAS_ERR_DEVICE once after a stream-K fix-up wait ran into its spin limit (GEMM workgroups not all resident); clears the report
int gemm_sk_check();
// true when gemm_f32 on this stream takes the stream-K fix-up schedule for an M x N output (option on, enough output tiles, scratch there or
// creatable): callers that would otherwise split K by hand (zero fill + atomics + a separate activation pass) then leave the split to gemm_f32
bool gemm_sk_fixup_ready(hipStream_t stream, int M, int N);
// arithmetic of the interior tiles: 0 = v_mfma_f32_32x32x2_f32, 1 = exact three-way bf16 operand split on the bf16 MFMA pipe
// (six partial products, fp32 accumulate; as accurate as the fp32 MFMA, see gemm_f32.hip).  Process-wide; LAS_GEMM_ARITH.
int gemm_get_arith();
void gemm_set_arith(int mode);
void gemm_set_tuning(int key, long value);     // schedule knobs for tools/ubench_gemm_sched.py (see gemm_f32.hip)

// ---- pblstm_rec.hip --------------------------------------------------------------------
// Forward time recurrence of one bidirectional LSTM layer, both directions in one launch.
//   gates : (2, B, T, 4H)  in: x_t W_ih^T + b_ih + b_hh (PyTorch row order i,f,g,o) ; out (if stash): post-activation gates
//   out   : (B, T, 2H)     h_fwd(t) | h_bwd(t)
//   cbuf  : (2, B, T, H)   cell state c_t             (stash, may be null when !stash)
//   hprev : (2, B, T, H)   h used as recurrent input at time t (stash)
//   xbuf  : hand-off granules for the multi-CU variants, >= rec_xbuf_bytes(); err: device error word
int pblstm_rec_fwd(float* gates, const float* w_hh_f, const float* w_hh_r, float* out, float* cbuf, float* hprev,
                   int B, int T, int H, int stash, unsigned long long* xbuf, unsigned* err, int force_generic,
                   hipStream_t stream);
// Backward (BPTT) of the same recurrence.
//   dout  : (B, T, 2H) upstream gradient ; gates/cbuf/hprev: the forward stash
//   dgates: (2, B, T, 4H) out: dG_t (pre-activation gradients, PyTorch row order)
//   w_hh_t: (2, H, 4H) transposed recurrent weights (see transpose_w_hh)
// db_f / db_r (optional, pre-zeroed): (2, 4H) bias gradients [b_ih | b_hh] of each direction, summed inside the persistent
// kernels; *db_done reports whether that happened (0: the generic kernels ran — the caller column-sums dgates itself).
// confine_nx in {2, 4}: the one-utterance-per-group kernel keeps to XCDs [0, confine_nx) (grid over-subscribed 8 / confine_nx times, the
// workgroups dispatched to the other XCDs leave at once): a weight-gradient GEMM group on a side stream then has the other XCDs' CUs
// AND their L2s to itself.  rec_confine_xcds() says whether / how far a batch can be confined.
int pblstm_rec_bwd(const float* dout, const float* gates, const float* cbuf, const float* w_hh_t, float* dgates,
                   int B, int T, int H, unsigned long long* xbuf, unsigned* err, int force_generic,
                   hipStream_t stream, float* db_f = nullptr, float* db_r = nullptr, int* db_done = nullptr, int confine_nx = 0);
int rec_confine_xcds(int B, int H);      // 0: the backward recurrence of this batch needs the whole chip (or the device is not 8 x 32 CUs)
size_t rec_xbuf_bytes(int B, int H);
// pblstm_rec_mfma.hip: the forward recurrence for batches that fill MFMA tiles (16 utterances per group of H/32 workgroups, bf16
// matrix pipe with the exact three-way operand split).  pblstm_rec_fwd dispatches to it by itself when eligible.
bool rec_fwd_mfma_eligible(int B, int H);
bool rec_bwd_mfma_eligible(int B, int H);
size_t rec_bwd_mfma_ring_floats(int B, int H);
size_t rec_mfma_xbuf_extra_bytes(int B, int H);
int rec_bwd_mfma(const float* dout, const float* gates, const float* cbuf, const float* w_hh_t, float* dgates, int B, int T, int H,
                 unsigned long long* xbuf, unsigned* err, float* db_f, float* db_r, hipStream_t stream);
// second form of the forward (pblstm_rec_mfma2.hip: wave-specialised pipeline, 512-thread workgroups); same buffers and semantics
int rec_fwd_mfma2(float* gates, const float* w_hh_f, const float* w_hh_r, float* out, float* cbuf, float* hprev, int B, int T, int H,
                  int stash, unsigned long long* xbuf, unsigned* err, hipStream_t stream);
int rec_fwd_mfma(float* gates, const float* w_hh_f, const float* w_hh_r, float* out, float* cbuf, float* hprev, int B, int T, int H,
                 int stash, unsigned long long* xbuf, unsigned* err, hipStream_t stream);
#ifdef LAS_REC_TRACE
void rec_set_trace(unsigned long long* dev_buf);    // profiling build only, see tools/ubench_rec_trace.py
#endif
// dst[c][r] = src[r][c]; an optional second (src1, dst1) pair of the same shape rides in the same launch
int transpose2d(const float* src, float* dst, int rows, int cols, hipStream_t stream, const float* src1 = nullptr, float* dst1 = nullptr);

// ---- speller.hip -----------------------------------------------------------------------
struct CellSeg {            // one dense input segment of an LSTM cell step:  gates += x(B,K) * W(4Hs,K)^T
    const float* x = nullptr; long ldx = 0;
    const float* w = nullptr; long ldw = 0;
    int K = 0;
};
// One LSTM cell step for all B utterances (speller layer): gates = sum_seg x_seg W_seg^T + b_ih + b_hh, then the cell.
// gates_out (B,4Hs) post-activation stash (may be null); c_prev may be null (zero state).
int lstm_cell_fwd(const CellSeg* segs, int nseg, const float* b_ih, const float* b_hh, const float* c_prev, float* h_out,
                  float* c_out, float* gates_out, int B, int Hs, hipStream_t stream);
// Same small-M MFMA machinery as a plain linear layer: out(B,N) = sum_seg x_seg W_seg^T + bias  (W rows = outputs, N % 16 == 0)
int smallm_linear_nt(const CellSeg* segs, int nseg, const float* bias, float* out, long ldo, int B, int N, hipStream_t stream);
// Pointwise part of the cell backward: dh = dh_a + dh_b, dc_in, stash -> dG (B,4Hs), dc_prev (B,Hs)
int lstm_cell_bwd_pointwise(const float* dh_a, int nparts, long part_stride, const float* dh_b, const float* dc_in,
                            const float* gates, const float* c, const float* c_prev, float* dG, float* dc_prev, int B, int Hs,
                            hipStream_t stream);
// Operands of one cell's backward pointwise step, fused into the kernel that produces that cell's dh:
// dh = (produced value) + dh_carry ; stash -> dG (B,4Hs), dc_out (B,Hs).  dc_out may alias dc_in.
struct CellPw {
    const float* gates = nullptr;    // (B,4Hs) forward stash of this cell; null = fusion disabled
    const float* c = nullptr; const float* c_prev = nullptr;     // (B,Hs); c_prev null at step 0
    const float* dh_carry = nullptr; const float* dc_in = nullptr;  // (B,Hs), null at the last step
    float* dG = nullptr; float* dc_out = nullptr;
};
// Small-M dense products of the cell backward: out_i(B,N_i) = a(B,K) * W_i(K,N_i), i = 0,1 (W row-major, ld = ldw_i).
// If pw.gates is set, N0 must equal Hs and the epilogue of output 0 applies the NEXT LOWER cell's backward pointwise
// step to the dh it just produced (out0 may then be null).
int smallm_gemm_nn2(const float* a, long lda, int B, int K, const float* w0, long ldw0, float* out0, long ldo0, int N0,
                    const float* w1, long ldw1, float* out1, long ldo1, int N1, const CellPw& pw, int Hs, hipStream_t stream);

struct AttnFwdArgs {
    const float* h_top;    // (B,Hs) decoder state
    const float* feat;     // (B,Tp,D)   D = 2H = Hs
    const float* keys;     // (B,Tp,M)   relu(psi(feat)) (or feat itself when !use_mlp)
    const float* w_phi; const float* b_phi;     // (M,Hs),(M)
    const float* w_c; const float* b_c;         // (V,2Hs),(V)
    float* q_out;          // (B,M)   post-activation query (stash)
    float* att_out;        // (B,Tp)  attention weights
    float* ctx_out;        // (B,D)
    float* logp_out;       // (B,V); null = character distribution deferred to one GEMM after the loop (teacher forcing)
    int* argmax_out;       // (B) or null
    float* y_next; long ldy; // (B,ldy) or null: next-step input written on device (free-running decode)
    int y_mode;            // 0: feed log-probs back, 1: feed one-hot argmax, 2: feed a one-hot sample (needs sample_noise)
    const float* sample_noise = nullptr;   // (B,V) Exp(1) draws of this step (decode_mode 2)
    int B, Tp, D, M, V, Hs;
    int use_mlp, relu;
    // multi-head (reference las_model.py:298-314): grid (B, heads); head h uses rows [h*M,(h+1)*M) of phi and writes its
    // context to ctx_out[b*ldctx + h*D ...] (the concatenation fed to dim_reduce).  phases: bit0 attention, bit1
    // character distribution (with ctx_in: read the context from memory instead of computing it).
    int heads = 1; long ldq = 0, ldctx = 0, att_hs = 0;
    const float* ctx_in = nullptr;
    int phases = 3;
};
int attn_step_fwd(const AttnFwdArgs& a, hipStream_t stream);

// Persistent teacher-forced decode loop (speller_persist.hip): one launch for all U steps.  Pointers are the
// SpellerLayout stash arrays; eligibility (shape / residency) must be checked first.
// A side stream per (host thread, device) for work that is independent of the launches around it (the sentinel fills of the hand-off
// slabs, 50 - 90 MB of pure writes, beside MFMA-bound GEMMs): fork() makes the side stream wait for everything issued so far on the main
// stream, join() makes the main stream wait for the side stream.  Both are event record / wait pairs, so they are captured into a HIP
// graph like any other stream operation.  ok() is false when the handles could not be created (e.g. first use during a stream capture):
// the caller then does the work inline.  Option SIDE_FILLS = 0 switches the mechanism off (A/B).
struct SideStream {
    hipStream_t s = nullptr; hipEvent_t e_fork = nullptr, e_join = nullptr; bool tried = false;
    bool ok(hipStream_t main);
    int fork(hipStream_t main);
    int join(hipStream_t main);
};
SideStream& side_stream();
// Deferred weight-gradient work (LAS_FLAG_DEFER_DW): a second library-owned stream per DEVICE (shared by the host threads: the autograd worker
// thread defers, the caller's thread joins) whose launches are NOT joined
// before the entry point returns; las_join_deferred makes a stream wait for everything issued there.  begin() records the fork on `main`
// and returns the side stream (nullptr: unavailable — stream capture, creation failure — do the work on `main`); end() marks the work pending.
struct DeferSide {
    hipStream_t s = nullptr; hipEvent_t e_fork = nullptr, e_done = nullptr; bool tried = false, pending = false;
    hipStream_t begin(hipStream_t main);
    int end();
    int join(hipStream_t main);
    // the same side stream for work that is joined inside the call that issued it (fork ... join around two concurrent launches)
    hipStream_t begin_joined(hipStream_t main) { return begin(main); }
    int end_joined(hipStream_t main);
};
DeferSide& defer_side();
// placement probe (las_debug_xcd_probe): 2 x 1024 words; the XCD-confined recurrence writes XCC id + 1 of block b at [b], the XCD-partitioned
// GEMM group at [1024 + b] (b < 1024) — to check the round-robin block -> XCD assumption the two launches' SPEED rests on
unsigned* xcd_probe_ptr();
// fork() ... join() with every exit path covered: an early return between the two (a failing GEMM, a fall-back) still joins the side
// stream, so that its fill can never race with whatever the main stream does next with the same slabs
struct SideJoinGuard {
    SideStream* side = nullptr; hipStream_t main = nullptr;
    void arm(SideStream& s, hipStream_t m) { side = &s; main = m; }
    int join() { SideStream* s = side; side = nullptr; return s ? s->join(main) : 0; }
    ~SideJoinGuard() { if (side) (void)side->join(main); }
};

struct PersistFwd {
    const float* w0p; int Vp;                       // [W_y | 0 | W_ctx] shadow of W_ih0, ld = Vp + Hs
    const float* w_hh0; const float* w_ih1; const float* w_hh1;
    const float* b_ih0; const float* b_hh0; const float* b_ih1; const float* b_hh1;
    const float* w_phi; const float* b_phi;
    const float* feat; const float* keys; float* y_all;
    float* ctx_all; float* h_all; float* c_all; float* gates_all; float* q_all; float* att;
    float* hx;                                      // 2*U*32*Hs floats: tiled hand-off copy of h (see speller_persist.hip)
    int mode = 0;                                   // 0 teacher forcing, 1 feed one-hot arg-max, 2 feed log-probabilities
    const float* w_c = nullptr; const float* b_c = nullptr;
    float* logp = nullptr; int* argmax = nullptr;   // free-running outputs (U,B,V), (U,B)
    float* lgx = nullptr;                           // U*B*8*32 floats: partial logits of the attention workgroups
    // "pre-multiplied context" variant (teacher forcing only; speller_persist_pre_eligible): pctx = feat . W_ctx^T in the
    // cell workgroups' column order (B*Tp, 4Hs), gx = U*B*4Hs floats of hand-off slabs.  The kernel then leaves
    // ctx_all[1..U] to the caller (one batched GEMM att . feat after the launch).
    const float* pctx = nullptr; float* gx = nullptr;
    float* r0x = nullptr;                           // PRE variant: U*32*4Hs floats, the cell workgroups' part of the bottom-layer gates
    const float* yw = nullptr;                      // PRE variant: (U*B, 4Hs) label half + biases of the bottom-layer gates, permuted columns
    // PRE variant with mode 1 (free-running arg-max feedback, speller_persist_pre_greedy_eligible): Q^T = W_c[:, Hs:] feat^T (B, 32, Tp), W_y^T in
    // the permuted gate-column order (Vp, 4Hs), and U*(Hs/4)*512 floats of partial-logit slabs
    const float* qct = nullptr; const float* wyT = nullptr; float* plx = nullptr;
    // PRE variant with NH > 1 attention heads (speller_persist_pre_mh_eligible; teacher forcing): pctx is (B*Tp, NH*4Hs), gx U*B*NH*4Hs floats (stash and the
    // heads' exchange slab), p0 = feat[:, 0] . W_ctx^T (B, 4Hs); q_all is (U*B, NH*M), att [U][NH][B][Tp] as the per-step kernels lay them out
    int NH = 1; const float* p0 = nullptr;
    float* ex = nullptr;                            // PRE variant at Hs = 256 with 16 workgroups per utterance (speller_persist_pre_ws): U*B*16*64 floats, the frame slices' energies
    int B, Tp, U, Hs, V, relu;
    unsigned* err;
    bool prefilled = false;                         // the caller has sentinel-filled the hand-off slabs already (speller_persist_fwd_fill)
};
int speller_persist_fwd_fill(const PersistFwd& p, hipStream_t stream);      // (PRE variant: needs hx, r0x, U, Hs only)
bool speller_persist_eligible(int B, int Tp, int Hs, int D, int M, int V, int L, int heads, int use_mlp, int free_running);
int speller_persist_pre_ws(int B, int Tp, int Hs, int cus);   // attention workgroups per utterance of the PRE variant (0: n/a; cus < 0: shape only)
bool speller_persist_pre_shape(int B, int Tp, int Hs, int D, int M, int V, int L, int heads, int use_mlp);   // shape only (sizes the reserve)
bool speller_persist_pre_eligible(int B, int Tp, int Hs, int D, int M, int V, int L, int heads, int use_mlp);   // shape + residency
bool speller_persist_pre_greedy_eligible(int B, int Tp, int Hs, int D, int M, int V, int L, int heads, int use_mlp);   // ... of its free-running (mode 1) form
bool speller_persist_pre_mh_shape(int B, int Tp, int Hs, int D, int M, int V, int L, int heads, int use_mlp);      // ... of its multi-head form (heads 2..4): shape only
bool speller_persist_pre_mh_eligible(int B, int Tp, int Hs, int D, int M, int V, int L, int heads, int use_mlp);   // ... shape + switches + residency
bool speller_persist_pre_mh_greedy_eligible(int B, int Tp, int Hs, int D, int M, int V, int L, int heads, int use_mlp);   // ... of its free-running (mode 1) form
int speller_persist_fwd(const PersistFwd& p, hipStream_t stream);
void speller_persist_set_trace(unsigned long long* dev_buf);   // profiling aid, see tools/ubench_persist_trace.py

// One-launch teacher-forced decode forward for Hs = D = 1024, M = 64, B <= 16 (speller_big.hip): the reference's shipped YAML sizes.
struct BigFwd {
    const float* w0p; int Vp;                       // [W_y | 0 | W_ctx] shadow of W_ih0, ld = Vp + Hs
    const float* w_hh0; const float* w_ih1; const float* w_hh1;
    const float* b_ih0; const float* b_hh0; const float* b_ih1; const float* b_hh1;
    const float* w_phi; const float* b_phi;
    const float* feat; const float* keys;
    const float* yw;                                // (U*B, 4Hs): y_s W_y^T, rows in PyTorch gate order (no bias)
    float* ctx_all; float* h_all; float* c_all; float* gates_all; float* q_all; float* att;
    float* hx; float* qp; unsigned* flags;          // hand-off slabs: speller_big_hx_floats / _qp_floats / _flag_words (16-byte aligned)
    int mode = 0;                                   // 0 teacher forcing, 1 free-running greedy (feed the one-hot arg-max, reference decode_mode 1)
    const float* w_c = nullptr; const float* b_c = nullptr; float* logp = nullptr; int* argmax = nullptr; float* y_all = nullptr;
    float* lgp = nullptr;                           // greedy: speller_big_greedy_floats() (partial logits + fed-back symbols)
    float* eg = nullptr;                            // T' > 256: U*B*512 floats (energy rows exchanged by an utterance's workgroups)
    int B, Tp, U, V, relu;
    unsigned* err;
};
bool speller_big_shape(int B, int Tp, int Hs, int D, int M, int V, int L, int heads, int use_mlp);       // shape only (sizes the reserve)
bool speller_big_eligible(int B, int Tp, int Hs, int D, int M, int V, int L, int heads, int use_mlp, int greedy = 0);    // shape + switch + residency
size_t speller_big_greedy_floats(int B, int U);
size_t speller_big_hx_floats(int U);
size_t speller_big_qp_floats(int B, int U);
size_t speller_big_flag_words(int U);
int speller_big_fwd(const BigFwd& p, hipStream_t stream);
void speller_big_set_trace(unsigned long long* dev_buf);
// ... and its backward (speller_big.hip): all U steps of the decode loop's backward in one launch, weights resident column-wise
struct BigBwd {
    const float* w_ih1; const float* w_hh1; const float* w_hh0; const float* w0p; int Vp; const float* w_phi;
    const float* feat; const float* keys; const float* att; const float* q_all; const float* gates_all; const float* c_all;
    const float* dcat_all;                          // (U*B, Hs + D): dz W_c of every step
    float* dG_all; float* dctx_all; float* de_all; float* dqpre_all;      // per-step gradients for the deferred GEMMs
    float* xbuf;                                    // speller_big_bwd_workspace_floats(), 16-byte aligned
    int B, Tp, U, V, relu;
    unsigned* err;
};
bool speller_big_bwd_eligible(int B, int Tp, int Hs, int D, int M, int V, int L, int heads, int use_mlp);
size_t speller_big_bwd_workspace_floats(int B, int U);
int speller_big_bwd(const BigBwd& p, hipStream_t stream);
const float* speller_big_bwd_dx0_ctx(const float* xbuf, int U);      // (16, 1024) row-major: gradient of the initial context
void speller_big_bwd_set_trace(unsigned long long* dev_buf);

// Persistent backward of the teacher-forced decode loop (speller_persist_bwd.hip): one launch for all U steps.
struct PersistBwd {
    const float* w_ih0; const float* w_hh0; const float* w_ih1; const float* w_hh1; const float* w_phi;
    const float* feat; const float* keys; const float* att; const float* q_all; const float* ctx_all;
    const float* gates_all; const float* c_all; const float* dcat_all;      // forward stash + dz W_c of every step
    float* dG_all; float* dctx_all; float* de_all; float* dqpre_all;        // per-step gradients for the deferred GEMMs
    float* dx0;                                                             // (B,V+D): context part written at step 0
    float* xbuf;                                                            // speller_persist_bwd_workspace_floats()
    // PRE variant (speller_persist_bwd_pre_eligible; the forward ran its PRE variant): feat . W_ctx^T and the forward's gx slabs.
    // The kernel then leaves dctx_all and the context part of dx0 to the caller (one GEMM over the stashed dG0 afterwards).
    const float* pctx = nullptr; const float* gxf = nullptr;
    // ... its multi-head form (speller_persist_bwd_pre_mh_eligible): NH heads, dim_reduce weight (D, NH*D), U*B*NH*D floats of scratch
    int NH = 1; const float* w_dr = nullptr; float* dctxcat = nullptr;
    int B, Tp, U, Hs, V, relu;
    unsigned* err;
};
size_t speller_persist_bwd_mh_workspace_floats(int B, int Tp, int U, int Hs, int M, int heads);
bool speller_persist_bwd_pre_mh_eligible(int B, int Tp, int Hs, int D, int M, int V, int L, int heads, int use_mlp);
bool speller_persist_bwd_eligible(int B, int Tp, int Hs, int D, int M, int V, int L, int heads, int use_mlp);
size_t speller_persist_bwd_workspace_floats(int B, int Tp, int U, int Hs, int M);
bool speller_persist_bwd_pre_eligible(int B, int Tp, int Hs, int D, int M, int V, int L, int heads, int use_mlp);
int speller_persist_bwd(const PersistBwd& p, hipStream_t stream);
void speller_persist_bwd_set_trace(unsigned long long* dev_buf);

struct AttnBwdArgs {
    const float* dlogp;    // (B,V) upstream gradient of this step's log-probs (may include mode-0 feedback grad)
    const float* dcat_pre; // (B,Hs+D) or null: dz W_c precomputed for all steps by one GEMM (no gradient through y)
    const float* logp;     // (B,V)
    const float* h_top;    // (B,Hs)
    const float* ctx;      // (B,D)
    const float* att;      // (B,Tp)
    const float* q;        // (B,M)
    const float* feat; const float* keys;
    const float* w_phi; const float* w_c;
    const float* dctx_carry; long ldc;   // (B,D) gradient flowing into this step's context from step s+1's input (may be null)
    const float* dy_carry; long ldy;     // (B,V) gradient flowing into this step's log-probs from step s+1's input (decode_mode 0)
    float* dz_out;         // (B,V)   stash for dW_c / db_c
    float* dctx_out;       // (B,D)   total context gradient (stash for dfeat GEMM)
    float* de_out;         // (B,Tp)  energy gradient (stash for dK GEMM)
    float* dqpre_out;      // (B,M)   pre-activation query gradient (stash for dW_phi)
    float* dh_top_out;     // (B,Hs)  gradient wrt decoder state from this step's attention + char distribution (may be null)
    CellPw pw;             // top LSTM layer's backward pointwise step, fused (pw.gates == null: disabled)
    // multi-head: phases bit0 = character-distribution part (dz, dh_top part 0, total dctx), bit1 = per-head attention part
    // (grid (B, heads); dctx_in = this head's slice of the dim_reduce input gradient; dh_top_out + h*dh_hs receives its part)
    int heads = 1; int phases = 3; long ldq = 0, att_hs = 0, dh_hs = 0;
    const float* dctx_in = nullptr; long ld_dctx_in = 0;
    int B, Tp, D, M, V, Hs;
    int use_mlp, relu;
};
int attn_step_bwd(const AttnBwdArgs& a, hipStream_t stream);

// ---- misc.hip --------------------------------------------------------------------------
int colsum(const float* src, long ld, int rows, int cols, float* dst, int accumulate, hipStream_t stream,
           float* dst2 = nullptr);  // dst[c] (+)= sum_r src[r][c]; dst2 (optional) receives the same sums
// up to COLSUM_MAX_JOBS independent column sums in one launch (dst (+)= column sums of src; dst2 optional copy)
constexpr int COLSUM_MAX_JOBS = 8;
struct ColsumJob { const float* src; long ld; int rows; int cols; float* dst; float* dst2; };
int colsum_multi(const ColsumJob* jobs, int n, int zeroed, hipStream_t stream);
int act_bwd_inplace(float* grad, const float* act, long n, int code, hipStream_t stream);                    // grad *= act'(.) (LAS_ACT_* code)
int add_inplace(float* dst, const float* src, long n, hipStream_t stream);
int sum_parts(float* dst, const float* src, long n, long stride, int parts, hipStream_t stream);   // dst[i] = sum_k src[k*stride+i]
int act_inplace(float* x, long n, int code, hipStream_t stream);
int copy2d(const float* src, long lds, float* dst, long ldd, int rows, int cols, int accumulate, hipStream_t stream);
int log_softmax_rows(float* x, long rows, int V, hipStream_t stream);
int log_softmax_bwd_rows(const float* dlogp, const float* logp, float* dz, long rows, int V, hipStream_t stream);
int ls_loss(const float* logp, long sU, long sB, const long long* labels, int U, int U_lab, int B, int V, float eps, float* part,
            float* loss, float* dlogp, long dU, long dB, hipStream_t stream);
int ler(const float* logp, long sU, long sB, const long long* labels, int U, int U_lab, int B, int V, float* out, int* work,
        hipStream_t stream);
// wperm / wyperm / bperm (optional, together): W_ctx rows, W_y rows (4Hs, Vp) and b_ih0 + b_hh0 in the persistent decode kernel's
// unit*4 + gate row order
int build_w0p(const float* w_ih0, float* w0p, int Hs, int V, int Vp, hipStream_t stream, float* wperm = nullptr,
              float* wyperm = nullptr, float* bperm = nullptr, const float* b_ih0 = nullptr, const float* b_hh0 = nullptr);
int labels_to_y(const long long* labels, float* y_all, int B, int U, int V, int Vp, int u_lab, hipStream_t stream);
// build_w0p + labels_to_y + ctx_{-1} = feat[:,0,:] in one launch (the Speller forward's element-wise preparations)
int matvec_rows(const float* w, long ld, const float* x, float* out, int rows, int K, hipStream_t stream, const float* addend = nullptr);      // out[r] = w[r, :] . x (+ addend[r])
int speller_prologue(const float* w_ih0, float* w0p, int Hs, int V, int Vp, float* wperm, float* wyperm, float* bperm, const float* b_ih0,
                     const float* b_hh0, const long long* labels, float* y_all, int B, int U, int u_lab, const float* feat, long ldfeat,
                     float* ctx0, int D, hipStream_t stream);
int collate_pad(const float* packed, const long long* foff, const long long* plab, const long long* loff, int B, int T, int F, int U,
                int V, float* inputs, long long* targets, hipStream_t stream);

}  // namespace las
