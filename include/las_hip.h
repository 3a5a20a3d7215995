/* liblas_hip.so — C ABI of the MI355X-native LAS hot path (gfx950 only).
 *
 * The reference (jiwidi/las-pytorch) has NO native boundary for this path: its Listener/Speller call
 * torch.nn.LSTM / nn.Linear / torch.bmm directly (reference model/las_model.py).  These entry points
 * are what a ctypes binding inside the reference's model/las_model.py would call instead; each one
 * names the reference lines it replaces.  INTEGRATION.md shows that binding.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes; no torch types.  Every pointer is a DEVICE pointer into
 *     caller-owned memory (PyTorch-ROCm tensors: tensor.data_ptr()); fp32 contiguous unless stated.
 *   - All work is enqueued asynchronously on `stream` (a hipStream_t passed as void*); no hidden
 *     device synchronisation.  The library allocates nothing persistent.
 *   - Return value: 0 = ok; nonzero = error (1 argument/shape precondition, 2 HIP runtime error,
 *     3 unsupported configuration, 4 device-side failure).  las_last_error() returns a thread-local
 *     message.  Never throws, never exits.
 *   - `err_word` points at TWO caller-owned, zero-initialised device uint32: kernels that hand data between
 *     workgroups write a nonzero code into err_word[0] if a bounded spin expires (results are then invalid).
 *     err_word[1] is an optional spin budget: 0 keeps the built-in 2^18 spins (~42 ms, sized for callers that can re-run a
 *     step); a caller that cannot (no fused update to skip) stores a larger count there (2^21 = ~340 ms).  It is read only
 *     after the built-in budget is spent.
 *   - Re-entrant per device.  Process-wide state is limited to (a) the mutex-guarded per-device RCCL handles, (b) the option
 *     registry below (atomics, initialised once from LAS_<NAME> environment variables: A/B and profiling switches whose
 *     defaults are the measured best) and (c) the las_debug_* profiling hooks declared at the end of this header.  Nothing a
 *     call computes depends on another thread's call; the GEMM arithmetic can be chosen per call (LAS_FLAG_GEMM_F32).
 *   - The persistent kernels (Listener recurrences; the one-launch Speller decode loop, chosen automatically for
 *     2-layer single-head MLP-attention spellers with Hs in {256,512}, B <= 32) need every workgroup resident at
 *     once, up to all compute units of the device: do not run other kernels concurrently on other streams while a
 *     las_pblstm_* / las_speller_* call is in flight.  If the device exposes too few compute units the library
 *     falls back to the per-step kernels by itself; a violated assumption ends in a bounded-spin timeout reported
 *     through `err_word`, never in a hang.
 */
#ifndef LAS_HIP_H
#define LAS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LAS_ABI_VERSION 10
#define LAS_MAX_SPELLER_LAYERS 4

/* flags */
#define LAS_FLAG_STASH          1   /* keep what the backward pass needs (training) */
#define LAS_FLAG_FORCE_GENERIC  2   /* use the generic kernels (L2-streaming recurrence, per-step speller launches): A/B tests */
#define LAS_FLAG_TEACHER_FORCED 4   /* las_speller_bwd only: the las_speller_fwd call that filled `reserve` ran teacher-forced — or free-running
                                       with decode_mode 1, whose one-hot feedback carries no gradient: its backward is the teacher-forced
                                       one over the emitted symbols — with the same flags / err_word.  Lets the backward use what that
                                       forward left in the reserve
                                       (feat.W_ctx^T and its per-step attention-weighted sums); without it the backward
                                       recomputes nothing and runs its classic path — always correct, about 0.4 ms slower. */

#define LAS_FLAG_GEMM_F32       8   /* this call's MFMA GEMMs run on the fp32 matrix pipe (v_mfma_f32_32x32x2_f32) whatever the
                                       process-wide GEMM_ARITH option says: per-call, per-thread, re-entrant */

#define LAS_FLAG_GRADS_ZEROED   16  /* las_pblstm_bwd / las_speller_bwd: the gradient outputs lie in one block that the caller has ALREADY zeroed
                                       (the flat gradient buffer, cleared once per step): the entry point skips its own fill of that block
                                       (~5 us each, launch-bound).  Without the flag the outputs may hold anything. */

#define LAS_FLAG_DEFER_DW       32  /* las_pblstm_bwd / las_speller_bwd: the weight-gradient GEMM group (and bias column sums) of this call may be left
                                       running on a library-owned side stream when the call returns; the caller MUST call las_join_deferred(stream)
                                       before anything reads those gradients, and must keep every buffer it passed alive until then.  Taken only
                                       where it pays: batches whose backward recurrences fit 2 or 4 of the 8 XCDs (B <= 16 at H = 256) — those
                                       launches are then confined to these XCDs and the deferred group runs on the others (their CUs and L2s),
                                       hidden under the NEXT layer's recurrence.  Taken at 2 of 8 XCDs only (B <= 8 at H = 256: the long-utterance batches of
                                       BASELINE configs[4]; measured slower at 4).  Ignored otherwise (and during stream capture). */

int las_abi_version(void);
/* Make `stream` wait for the deferred work of earlier LAS_FLAG_DEFER_DW calls on the current device — issued by ANY host thread (PyTorch runs
 * backward on its autograd worker thread, the caller joins from its own) — no-op if none. */
int las_join_deferred(void* stream);
const char* las_last_error(void);

/* ------------------------------------------------------------------------------------------------
 * Run-time options (process-wide atomics; initial value = environment variable LAS_<NAME>, read once).  Keys are
 * case-insensitive, with or without the LAS_ prefix.  Returns 0, or 1 for an unknown key.
 *   GEMM_ARITH            1* split-operand bf16 MFMA (fp32-faithful, see las_gemm_f32), 0 fp32 MFMA
 *   GEMM_STREAMK, GEMM_SK_MIN_TILES, GEMM_SPLIT_BELOW, GEMM_SPLIT_TARGET   schedule thresholds of the GEMM (-1* = automatic)
 *   GEMM_SK_FIXUP 0*      1: stream-K with in-kernel fix-up for GEMMs of >= 64 output tiles: tiles that straddle workgroup runs are summed from
 *                         parked partial tiles by the workgroup owning the tile's first k-iteration (no atomics, no zeroing pass, any epilogue,
 *                         run-to-run deterministic); 2: also in the grouped launches (las_gemm_f32_group).  Off by default: inside the
 *                         training step it measured 0.05 ms SLOWER than the atomic forms (5.93 against 5.88 ms, alternating in one process)
 *   GEMM_SKF_MIN_KT, GEMM_SKF_MIN_RUN  fewest k-iterations per tile / per workgroup run for that schedule (-1* = automatic: 8 / 8)
 *   GEMM_SLOTS_PER_CU     resident GEMM workgroups per CU (2*; read at the first GEMM)
 *   GEMM_GROUP 1*, GEMM_XCD_SWZ 1*, GEMM_BATCH_DIRS 1*          grouped weight-gradient launches / XCD order / batched directions
 *   SPELLER_PERSIST 1*, SPELLER_PERSIST_BWD 1*                   one-launch decode loop forward / backward (0: per-step launches)
 *   SPELLER_PRE 1*, SPELLER_PRE_BWD 1*                           pre-multiplied-context variants of those kernels
 *   SPELLER_BIG 1*, SPELLER_BIG_BWD 1*                           one-launch decode loop (teacher-forced or greedy) forward / backward for the reference's
 *                                                                shipped sizes (Speller 1024x2, attention MLP 64, B <= 16, T' <= 512 / 480:
 *                                                                speller_big.hip; 0: per-step launches)
 *   SPELLER_BIG_TUNE 0*   poll pacing of those two kernels in units of 64 clocks: byte 0 / 2 / 3 before the first poll of the forward's h0 /
 *                         context / h1 hand-off, byte 1 between polls (A/B aid)
 *   REC_UW 0*, REC_NB 0*, REC_PIPE 1*, REC_AGENT_HANDOFF 0*, REC_MFMA 1*   Listener recurrence: units per workgroup, utterances per
 *                                                                group, pipelined halves, agent-scope hand-off, matrix-pipe form
 *                                                                (0 off, 1 automatic, 2 / 3: forward always as wave-specialised
 *                                                                pipeline / as the barrier-phased first form)
 *   REC_TRACE 0*          phase stamps of the pipeline form's first workgroup into its id buffer (tools/ubench_rec_mfma.py)
 *   CELL_MT 0*            M-tiles per workgroup of the per-step cell kernel
 *   SIDE_FILLS 0*         1: the sentinel fills of the decode kernels' hand-off slabs run on a library-owned side stream beside the GEMMs that
 *                         precede the launch (fork / join by events).  Off: measured 0.07 ms SLOWER per training step (the two cross-stream
 *                         joins cost more than the 23 us of fills they hide)
 *   TRUST_ZEROED_GRADS 1* honour LAS_FLAG_GRADS_ZEROED (0: fill the gradient blocks regardless; A/B)
 *   TIME_KERNELS 0*       record HIP events around the one-launch decode kernels on their launch stream (las_debug_kernel_ms)
 *   SPELLER_PRE_GREEDY 1  free-running (decode_mode 1) decode on the pre-multiplied-context kernel (0: classic persistent kernel; A/B)
 *   SPELLER_PRE_MH 1      multi-head attention (heads 2 or 4; teacher forcing and decode_mode 1) on the pre-multiplied-context kernels, forward and backward
 *                         (0: per-step kernels; A/B)
 *   GEMM_BIG 1            256 x 256-tile GEMM where it fills the chip (0 never, 2 whenever the shape allows; A/B)
 * Replaces nothing in the reference (pure Python, no switches).
 * ---------------------------------------------------------------------------------------------- */
int las_set_option(const char* key, int64_t value);
int las_get_option(const char* key, int64_t* value_out);

/* ------------------------------------------------------------------------------------------------
 * Listener: one pyramidal BiLSTM layer.
 * Replaces pBLSTMLayer.forward, reference model/las_model.py:81-91 (time-pair concat + nn.LSTM
 * bidirectional, batch_first) and its autograd (solver/solver.py:95).
 *   x (B, T_in, D_in) -> out (B, T_in/2, 2H);  T = T_in/2, D = 2*D_in.
 *   weights in PyTorch layout: w_ih (4H, D), w_hh (4H, H), b_ih (4H), b_hh (4H); *_r = reverse direction.
 * reserve: las_pblstm_reserve_floats() floats, 16-byte aligned; written by fwd, read by bwd.
 * ---------------------------------------------------------------------------------------------- */
size_t las_pblstm_reserve_floats(int B, int T_in, int H, int flags);
int las_pblstm_fwd(const float* x, int B, int T_in, int D_in, int H,
                   const float* w_ih_f, const float* w_hh_f, const float* b_ih_f, const float* b_hh_f,
                   const float* w_ih_r, const float* w_hh_r, const float* b_ih_r, const float* b_hh_r,
                   float* out, float* reserve, uint32_t* err_word, int flags, void* stream);

size_t las_pblstm_bwd_workspace_floats(int B, int T_in, int H);
/* dout (B,T,2H).  Gradient outputs are OVERWRITTEN.  dx (B,T_in,D_in) may be NULL (first layer). */
int las_pblstm_bwd(const float* x, const float* dout, int B, int T_in, int D_in, int H,
                   const float* w_ih_f, const float* w_hh_f, const float* w_ih_r, const float* w_hh_r,
                   const float* reserve, float* workspace,
                   float* dx,
                   float* dw_ih_f, float* dw_hh_f, float* db_ih_f, float* db_hh_f,
                   float* dw_ih_r, float* dw_hh_r, float* db_ih_r, float* db_hh_r,
                   uint32_t* err_word, int flags, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Speller: attention keys (loop-invariant), decode loop, backward.
 * Replaces Speller.forward / forward_step and Attention.forward, reference model/las_model.py:178-238,
 * 275-297 (single-head dot attention with optional phi/psi MLP + relu), CreateOnehotVariable /
 * TimeDistributed (utils/functions.py:54-63,72-77).
 * ---------------------------------------------------------------------------------------------- */
typedef struct las_speller_desc {
    int B;            /* utterances */
    int Tp;           /* encoder frames T' */
    int D;            /* listener feature dim 2H (== Hs, reference las_model.py:198) */
    int Hs;           /* speller hidden size */
    int V;            /* vocabulary (label_dim) */
    int M;            /* attention MLP dim (ignored when !use_mlp) */
    int L;            /* speller LSTM layers (1..LAS_MAX_SPELLER_LAYERS) */
    int use_mlp;      /* use_mlp_in_attention */
    int relu;         /* activation after phi/psi (mlp_activate_in_attention): 0 none, 1 relu, 2 tanh, 3 sigmoid */
    int multi_head;   /* attention heads (reference las_model.py:298-314); > 1 needs use_mlp and w_dr/b_dr */
    /* parameters, PyTorch layouts: w_ih[0] (4Hs, V+Hs), w_ih[l>0] (4Hs, Hs), w_hh (4Hs, Hs), biases (4Hs) */
    const float* w_ih[LAS_MAX_SPELLER_LAYERS];
    const float* w_hh[LAS_MAX_SPELLER_LAYERS];
    const float* b_ih[LAS_MAX_SPELLER_LAYERS];
    const float* b_hh[LAS_MAX_SPELLER_LAYERS];
    const float* w_phi; const float* b_phi;   /* (M*heads, Hs), (M*heads) */
    const float* w_psi; const float* b_psi;   /* (M, D), (M) */
    const float* w_c;   const float* b_c;     /* (V, 2Hs), (V) */
    const float* w_dr;  const float* b_dr;    /* dim_reduce (D, heads*D), (D): multi_head > 1 only, else NULL */
} las_speller_desc;

typedef struct las_speller_grads {           /* all OVERWRITTEN by las_speller_bwd */
    float* dw_ih[LAS_MAX_SPELLER_LAYERS];
    float* dw_hh[LAS_MAX_SPELLER_LAYERS];
    float* db_ih[LAS_MAX_SPELLER_LAYERS];
    float* db_hh[LAS_MAX_SPELLER_LAYERS];
    float* dw_phi; float* db_phi;
    float* dw_psi; float* db_psi;
    float* dw_c;   float* db_c;
    float* dw_dr;  float* db_dr;              /* multi_head > 1 only */
    float* dfeat;                             /* (B, Tp, D) */
} las_speller_grads;

/* keys (B,Tp,M) = act(psi(feat)): the reference recomputes this every decode step (las_model.py:279). */
int las_attn_keys_fwd(const las_speller_desc* d, const float* feat, float* keys, void* stream);

size_t las_speller_reserve_floats(const las_speller_desc* d, int U);

/* Utterances per launch of the one-launch decode kernels for this description: the largest batch for which
 * las_speller_fwd / las_speller_bwd take a one-launch path (today 32 for Hs <= 512, 16 for the Hs = 1024 kernels, 32 / heads with 2 or 4 attention heads, or 0
 * when the shape, the decode mode or a switch rules it out).  d->B is ignored except with several heads, where 0 is returned for a batch of more than two such slices
 * (the per-step kernels, which take the whole batch at once, are faster there).  A caller with a larger batch gains by running the Speller in slices of that many utterances — every slice then decodes
 * in one launch instead of U per-step launch chains (las_pytorch_amd/model/las_model.py::Speller._run does; the reference's loop
 * model/las_model.py:205-236 has no such notion).  Pure function of its arguments, the option registry and the device's CU count. */
int las_speller_decode_batch(const las_speller_desc* d, int teacher_forced, int decode_mode);
/* Decode U steps.
 *   labels_onehot : int64 (B, U_lab, V) one-hot ground truth as utils/data.py:141-143 delivers it, or NULL
 *   teacher_forced: 1 -> step s+1 is fed labels[:, s] (las_model.py:216-217); 0 -> free running with
 *   decode_mode 0 (feed log-probs, :220-221), 1 (feed one-hot argmax, :223-227) or 2 (feed a one-hot sample of
 *   Categorical(raw_pred), :229-234).  Mode 2 reproduces the reference exactly as torch evaluates it: the
 *   log-probabilities are renormalised as if they were probabilities (p_v = logp_v / sum_v logp_v, i.e. proportional
 *   to |log p|) and the sample is argmax_v p_v / q_v with q ~ Exp(1) (torch.multinomial's single-draw path);
 *   sample_noise (U,B,V) holds the caller's q draws (fp32, > 0) and may be NULL for the other modes.
 *   logp (U,B,V), att (U,heads,B,Tp), argmax (U,B) int32 or NULL.  keys may be NULL when !use_mlp.
 *   err_word: device uint32 (see las_pblstm_fwd) the persistent teacher-forced decode kernel reports hand-off timeouts
 *   through; NULL (or LAS_FLAG_FORCE_GENERIC) selects the per-step launch chain, which needs none. */
int las_speller_fwd(const las_speller_desc* d, const float* feat, const float* keys,
                    const int64_t* labels_onehot, int U_lab, int U, int teacher_forced, int decode_mode,
                    const float* sample_noise, float* logp, float* att, int32_t* argmax, float* reserve, uint32_t* err_word,
                    int flags, void* stream);

/* One decode step with caller-managed state: Speller.forward_step, reference model/las_model.py:178-184.
 *   input_word (B, V+Hs) = [y | context] as the reference concatenates it (:198,:236); h_in/c_in (L,B,Hs) or both NULL
 *   (zero state).  Outputs: logp (B,V), h_out/c_out (L,B,Hs), ctx (B,D), att (heads,B,Tp).
 *   reserve: NULL for inference; las_speller_step_reserve_floats() floats to keep what las_speller_step_bwd needs. */
size_t las_speller_step_workspace_floats(const las_speller_desc* d);
size_t las_speller_step_reserve_floats(const las_speller_desc* d);
int las_speller_step_fwd(const las_speller_desc* d, const float* feat, const float* keys, const float* input_word,
                         const float* h_in, const float* c_in, float* logp, float* h_out, float* c_out, float* ctx,
                         float* att, float* workspace, float* reserve, void* stream);
/* Backward of that step (the reference differentiates forward_step through autograd, solver/solver.py:95).
 *   dlogp (B,V), dh_out / dc_out (L,B,Hs), dctx (B,D): gradients wrt the step's outputs, each may be NULL (= zero).
 *   Outputs (overwritten): dinput_word (B,V+Hs), dh_in / dc_in (L,B,Hs), every member of g incl. g->dfeat (B,Tp,D)
 *   — the contribution of THIS step; a caller chaining steps sums them (autograd does). */
size_t las_speller_step_bwd_workspace_floats(const las_speller_desc* d);
int las_speller_step_bwd(const las_speller_desc* d, const float* feat, const float* keys, const float* input_word,
                         const float* h_in, const float* c_in, const float* logp, const float* h_out, const float* c_out,
                         const float* ctx, const float* att, const float* reserve, const float* dlogp, const float* dh_out,
                         const float* dc_out, const float* dctx, float* dinput_word, float* dh_in, float* dc_in,
                         const las_speller_grads* g, float* workspace, void* stream);

/* Attention.forward on its own (reference model/las_model.py:275-318): decoder_state (B,Hs) -> att (heads,B,Tp), ctx (B,D)
 * (after dim_reduce when multi_head > 1), and its backward from dctx (the attention scores are returned for inspection and
 * carry no gradient).  reserve: las_attention_reserve_floats(); g: dw_phi/db_phi/dw_psi/db_psi(/dw_dr/db_dr)/dfeat are
 * overwritten, the LSTM / character-distribution members are ignored. */
size_t las_attention_reserve_floats(const las_speller_desc* d);
int las_attention_fwd(const las_speller_desc* d, const float* feat, const float* keys, const float* decoder_state, float* att,
                      float* ctx, float* reserve, void* stream);
size_t las_attention_bwd_workspace_floats(const las_speller_desc* d);
int las_attention_bwd(const las_speller_desc* d, const float* feat, const float* keys, const float* decoder_state,
                      const float* att, const float* reserve, const float* dctx, float* ddecoder_state,
                      const las_speller_grads* g, float* workspace, void* stream);

size_t las_speller_bwd_workspace_floats(const las_speller_desc* d, int U);
/* dlogp (U,B,V): gradient of the loss wrt the returned log-probs.  feedback_mode0: the forward ran
 * free-running with decode_mode 0 (gradient flows through the fed-back log-probs).
 * err_word / flags: as las_speller_fwd (the persistent backward kernel needs the error word; NULL or
 * LAS_FLAG_FORCE_GENERIC selects the per-step launch chain).  Add LAS_FLAG_TEACHER_FORCED when the forward call that filled
 * `reserve` was teacher-forced (same flags, same err_word): the backward then reuses the feat.W_ctx^T product and the
 * per-step attention-weighted sums that forward left in `reserve` instead of multiplying dG0.W_ctx on the decode chain. */
int las_speller_bwd(const las_speller_desc* d, const float* feat, const float* keys,
                    const float* logp, const float* att, const float* dlogp, int U, int feedback_mode0,
                    const float* reserve, float* workspace, const las_speller_grads* g, uint32_t* err_word, int flags,
                    void* stream);

/* ------------------------------------------------------------------------------------------------
 * Caller-side contract on device (the code that CALLS the hot path, reference solver/solver.py).
 *   logp is addressed as logp[s*stride_u + b*stride_b + c] so both the (U,B,V) buffer of las_speller_fwd and the
 *   (B,U,V) tensor solver.py:68 builds can be passed; labels_onehot int64 (B, U_lab, V); U = min(U_lab, max_label_len).
 * las_ls_loss: label_smoothing_loss (solver.py:33-45) and, if dlogp != NULL, its gradient wrt logp (same addressing
 *   with dstride_*).  loss: 1 float; scratch: B floats.
 * las_letter_error_rate: LetterErrorRate (solver.py:11-24) of the argmax sequences; ler_out: B floats; U <= 4095.  One wave per
 *   utterance walks the anti-diagonals of the edit-distance table in registers; work is unused since ABI 9 (may be NULL; rounds 1-4
 *   kept the DP rows there: 4*B*(U+1) int32).
 * ---------------------------------------------------------------------------------------------- */
int las_ls_loss(const float* logp, int64_t stride_u, int64_t stride_b, const int64_t* labels_onehot, int U, int U_lab,
                int B, int V, float smoothing, float* loss, float* dlogp, int64_t dstride_u, int64_t dstride_b,
                float* scratch, void* stream);
int las_letter_error_rate(const float* logp, int64_t stride_u, int64_t stride_b, const int64_t* labels_onehot, int U,
                          int U_lab, int B, int V, float* ler_out, int32_t* work, void* stream);

/* Global-norm clip + Adam step on the flat gradient buffer in two launches: torch.nn.utils.clip_grad_norm_(params, max_norm)
 * followed by torch.optim.Adam.step() (amsgrad off, weight_decay 0) — reference solver/solver.py:96-97, train.py:82.
 *   params[t]            device pointer of parameter tensor t (fp32, contiguous), t < n_tensors   (HOST array)
 *   offsets[t]           element offset of tensor t inside grad_flat / exp_avg / exp_avg_sq; offsets[n_tensors] = total (HOST array)
 *   grad_flat            all gradients back to back (scaled in place when the clip is active, as clip_grad_norm_ does)
 *   exp_avg, exp_avg_sq  Adam moments in the same flat layout (caller-owned, zero before step 1)
 *   step                 1-based step count of THIS update (bias corrections 1 - beta^step)
 *   max_norm <= 0        no clipping;  norm_out (1 float, may be NULL) receives the total gradient norm before clipping
 *   workspace            las_clip_adam_workspace_floats() floats
 *   err_word             optional device error word of the step's persistent kernels: if nonzero when the update kernel runs,
 *                        parameters and moments are left UNCHANGED (the step's gradients are invalid; the host re-runs it) */
size_t las_clip_adam_workspace_floats(void);
int las_clip_adam(float* const* params, const int64_t* offsets, int n_tensors, float* grad_flat, float* exp_avg,
                  float* exp_avg_sq, float max_norm, double lr, double beta1, double beta2, double eps, int step,
                  float* norm_out, float* workspace, const uint32_t* err_word, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Input side: the collate contract on device (reference utils/data.py:116-149, collate_fn).
 *   packed_feat (sum_b len_b, F) fp32: the utterances' frames back to back; feat_offsets int64 (B+1) frame offsets.
 *   packed_labels int64 (sum_b n_b) character indices back to back; label_offsets int64 (B+1).
 *   inputs (B,T,F): frames then zero padding (np.pad constant 0, data.py:128); T is chosen by the caller (the reference
 *   rounds max len up to a multiple of 2**listener_layers = 32, data.py:124-125).
 *   targets int64 (B,U,V): one-hot rows, padding rows = onehot(PAD=0) (data.py:129,133).
 * ---------------------------------------------------------------------------------------------- */
int las_collate_pad(const float* packed_feat, const int64_t* feat_offsets, const int64_t* packed_labels,
                    const int64_t* label_offsets, int B, int T, int F, int U, int V, float* inputs, int64_t* targets,
                    void* stream);

/* ------------------------------------------------------------------------------------------------
 * Data-parallel gradient exchange: ONE all-reduce of the flat fp32 gradient per step over RCCL / xGMI.
 * Replaces nn.DataParallel's per-step broadcast / gather / reduce, reference train.py:76-78.
 * One communicator per device and process (mutex-guarded handle; librccl is opened on first use, so the library
 * itself has no link-time dependency on it).
 *   las_comm_uid   : rank 0 creates the 128-byte unique id and ships it to the other ranks by any host channel
 *   las_comm_init  : collective over all ranks; binds the communicator to the CURRENT device
 *   las_allreduce_f32 : in place on `buf` (count floats), enqueued on `stream`; average != 0 divides by the world size
 *                       inside the collective (ncclAvg)
 *   las_comm_destroy : releases the current device's communicator
 * ---------------------------------------------------------------------------------------------- */
int las_comm_uid(void* uid_out128);
int las_comm_init(int rank, int world, const void* uid128);
int las_allreduce_f32(float* buf, size_t count, int average, void* stream);
int las_comm_destroy(void);

/* ------------------------------------------------------------------------------------------------
 * Building blocks exported for tests / micro-benchmarks.
 * ---------------------------------------------------------------------------------------------- */
/* C[M,N] (+)= act(A(M,K) B(K,N) + bias0 + bias1); a_kc: A(m,k)=A[m*lda+k] else A[k*lda+m];
 * b_kc: B(k,n)=B[n*ldb+k] else B[k*ldb+n]. */
int las_gemm_f32(const float* A, const float* B, float* C, const float* bias0, const float* bias1,
                 int M, int N, int K, int64_t lda, int64_t ldb, int64_t ldc, int a_kc, int b_kc,
                 int batch, int64_t sA, int64_t sB, int64_t sC, int splitk, int accumulate, int relu, void* stream);
/* Stream-K fix-up bookkeeping: 0, or LAS_ERR_DEVICE (4) once after a fix-up wait of an earlier GEMM launch ran into its spin limit (its
 * workgroups were not all resident, e.g. a foreign kernel held the CUs: the tile it was waiting for is wrong).  Reading clears the
 * report.  Every las_gemm_* call and every entry point that issues GEMMs checks it too (one launch late); call it after a
 * synchronisation for a definite answer. */
int las_gemm_check(void);
/* n independent GEMMs C_i (+)= A_i B_i in ONE launch (stream-K across the concatenated k-iterations of all problems) when they
 * share the operand orientation (with GEMM_SK_FIXUP=0 every output must also be pre-zeroed (c_zeroed) or accumulated onto); otherwise
 * one launch each.
 * A2/B2/K1: optional second source along K (k >= K1 reads A2/B2 at k-K1), single problems only.  This is how the backward
 * passes issue their weight-gradient contractions; exported so that bench.py times exactly the launches a step makes. */
typedef struct las_gemm_desc {
    const float* A; const float* B; float* C; const float* A2; const float* B2;
    int M, N, K, K1;
    int64_t lda, ldb, ldc;
    int a_kc, b_kc, accumulate, c_zeroed;
    int planes;      /* A / B are P8x3 images (las_split_planes) of the fp32 matrices the other fields describe */
} las_gemm_desc;
int las_gemm_f32_group(const las_gemm_desc* descs, int n, void* stream);
/* Pre-split operands.  las_split_planes writes the "P8x3" image of an fp32 R x C matrix (row stride ld_src): every element as the exact sum
 * of three bf16 terms (x = p1 + p2 + p3, the split of arithmetic mode 1), 16-byte granule (r, c / 8, plane) = that term of elements
 * (r, 8 (c / 8) .. + 7) at 16-byte index (r (ld_dst / 8) + c / 8) 3 + plane; C and ld_dst multiples of 8, 6 bytes per element
 * (las_planes_bytes).  las_gemm_planes is las_gemm_f32 on two such images (lda / ldb / sA / sB remain ELEMENT counts of the logical fp32
 * matrices; K, the leading dimensions and every extent that runs along an operand's contiguous dimension must be multiples of 8):
 * the same six bf16 partial products and fp32 accumulation as mode 1 — bit-identical results — without the per-tile split arithmetic,
 * which every workgroup along the other tile dimension would repeat.  The training step splits each layer input, weight matrix and
 * gate-gradient slab once and feeds every GEMM that reads it from the image.  Replaces nothing in the reference (ATen's GEMMs,
 * model/las_model.py:90,174,279). */
size_t las_planes_bytes(int64_t rows, int64_t ld);
int las_split_planes(const float* src, int64_t ld_src, int R, int C, void* dst, int64_t ld_dst, void* stream);
int las_gemm_planes(const void* A_planes, const void* B_planes, float* C, const float* bias0, const float* bias1,
                    int M, int N, int K, int64_t lda, int64_t ldb, int64_t ldc, int a_kc, int b_kc,
                    int batch, int64_t sA, int64_t sB, int64_t sC, int splitk, int accumulate, int relu, void* stream);
/* Arithmetic of the MFMA GEMMs' interior tiles (= option GEMM_ARITH; LAS_FLAG_GEMM_F32 overrides it for one call):
 *   0  v_mfma_f32_32x32x2_f32 (fp32 operands on the fp32 matrix pipe)
 *   1  every fp32 operand split EXACTLY into three bf16 terms in registers, six of the nine partial products on
 *      v_mfma_f32_32x32x16_bf16 with fp32 accumulation; the dropped terms are below 2^-26 of |a*b|, i.e. the result is as
 *      accurate as mode 0 (tests/test_hip_kernels.py measures both against float64) at 2.67x its matrix-pipe roofline.
 *      Range edge: an operand that is +-Inf, or rounds to Inf in bf16 (|x| > ~3.39e38), yields NaN where mode 0 yields +-Inf
 *      (its residual is Inf - Inf); operands below ~2^-108 lose their second / third term to the matrix pipe's denormal flush.
 * Replaces nothing in the reference (ATen picks its own GEMM kernels, model/las_model.py:90,279). */
int las_gemm_get_arith(void);
void las_gemm_set_arith(int mode);
/* profiling aid: schedule thresholds of las_gemm_f32 (key 0 stream-K on/off, 1 fewest tiles for the persistent schedule,
 * 2 split K below this many tiles, 3 workgroup target of the split); value -1 restores the default */
void las_gemm_set_tuning(int key, int64_t value);
/* recurrence only: gates (2,B,T,4H) pre-activations in, see las_pblstm_fwd for the rest */
size_t las_rec_xbuf_bytes(int B, int H);
int las_pblstm_rec_fwd(float* gates, const float* w_hh_f, const float* w_hh_r, float* out, float* cbuf, float* hprev,
                       int B, int T, int H, void* xbuf, uint32_t* err_word, int flags, void* stream);


/* ------------------------------------------------------------------------------------------------
 * Profiling hooks (tools/ubench_persist*_trace.py): per-phase shader-clock stamps of workgroup 0 of each role of the
 * persistent decode kernels.  dev_buf: device buffer of 3*U*8 (forward) / 4*U*8 (backward) uint64; NULL switches the stamps off again.  Process-wide,
 * not for production use.  las_debug_rec_trace exists only in builds with -DLAS_REC_TRACE.
 * ---------------------------------------------------------------------------------------------- */
void las_debug_persist_trace(unsigned long long* dev_buf);
void las_debug_persist_bwd_trace(unsigned long long* dev_buf);
/* ... of workgroup 0 of the Hs = 1024 one-launch decode (speller_big.hip): 64 steps x 16 stamps. */
void las_debug_big_trace(unsigned long long* dev_buf);
/* ... of workgroups 0 / 64 / 128 / 192 (one per matrix role) of its backward: 4 x 64 steps x 16 stamps + 256 workgroups x 8. */
void las_debug_big_bwd_trace(unsigned long long* dev_buf);
/* With option TIME_KERNELS = 1: duration (HIP events on the launch stream) of the most recent launch of the one-launch decode
 * kernel, which = 0 forward (speller_persist_fwd*_kernel), 1 backward (speller_persist_bwd*_kernel).  Synchronises on that launch.
 * bench.py prices these two kernels against the roofline with it. */
int las_debug_kernel_ms(int which, float* ms_out);
/* Placement probe of the XCD-partitioned launches (LAS_FLAG_DEFER_DW): dev_buf = 2048 device uint32 (NULL: off).  The confined backward recurrence
 * writes XCC id + 1 of block b at [b], the partitioned GEMM group at [1024 + b]. */
void las_debug_xcd_probe(unsigned* dev_buf);
/* Name of the kernel family the most recent call launched for slot `which` (process-wide): 0 Listener recurrence forward
 * ("rec_fwd_fast" | "rec_fwd_multi" | "rec_fwd_mfma" | "rec_fwd_mfma2" | "rec_fwd_generic"), 1 its backward ("rec_bwd_fast" | "rec_bwd_multi" |
 * "rec_bwd_mfma" | "rec_bwd_generic"), 2 decode loop forward ("persist_pre" | "persist_pre_greedy" (free-running form) | "persist_pre_mh" |
 * "persist_pre_mh_greedy" (multi-head instantiations) | "persist" | "big" | "stepwise"), 3 decode loop backward ("persist_pre" | "persist_pre_mh" |
 * "persist" | "big" | "stepwise"), 5 the weight-gradient group of the most recent backward entry point ("deferred": left on the side stream under
 * LAS_FLAG_DEFER_DW | "inline"; las_join_deferred overwrites it with "joined" when it found pending work), 4 the most recent GEMM's operand path ("split" | "split256" (256 x 256 tiles, gemm_big.hip) | "f32" | "planes").  The parity tests assert it per fixture, so a silent
 * fall-back (e.g. LAS_ERR_UNSUPPORTED from a residency check) cannot leave a golden green on the wrong kernel.  Returns 0 / LAS_ERR_ARG. */
int las_debug_last_path(int which, char* out, int cap);

#ifdef __cplusplus
}
#endif
#endif /* LAS_HIP_H */
