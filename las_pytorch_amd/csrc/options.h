// Run-time options of liblas_hip.so: ONE registry behind las_set_option / las_get_option (include/las_hip.h).
// Every option is a process-wide atomic whose initial value comes from the environment variable LAS_<NAME> (read once, at first
// use) and otherwise from the built-in default; there is no other getenv in the library.  They are A/B and profiling aids — the
// defaults are the measured best — except GEMM_ARITH, which a single call can also override with LAS_FLAG_GEMM_F32.
#pragma once
#include <hip/hip_runtime.h>

namespace las {

enum Opt : int {
    OPT_GEMM_ARITH = 0,         // 1 split-operand bf16 MFMA (default), 0 fp32 MFMA
    OPT_GEMM_STREAMK,           // persistent data-parallel + stream-K schedule on/off            (-1 = automatic)
    OPT_GEMM_SK_MIN_TILES,      // fewest output tiles for the persistent schedule                (-1 = automatic)
    OPT_GEMM_SPLIT_BELOW,       // split K below this many output tiles                           (-1 = automatic)
    OPT_GEMM_SPLIT_TARGET,      // workgroup count the automatic split-K aims for                 (-1 = automatic)
    OPT_GEMM_SLOTS_PER_CU,      // resident GEMM workgroups per CU (default 2; read once)
    OPT_GEMM_GROUP,             // grouped weight-gradient launches on/off
    OPT_GEMM_XCD_SWZ,           // XCD-aware workgroup order in the grouped launch
    OPT_GEMM_BATCH_DIRS,        // both directions' input projections in one batched launch
    OPT_SPELLER_PERSIST,        // one-launch decode forward
    OPT_SPELLER_PERSIST_BWD,    // one-launch decode backward
    OPT_SPELLER_PRE,            // pre-multiplied-context variant of the decode kernels (forward + backward)
    OPT_SPELLER_PRE_BWD,        // ... of the backward only
    OPT_REC_UW,                 // hidden units per workgroup of the forward recurrence (0 = automatic)
    OPT_REC_AGENT_HANDOFF,      // agent-scope hand-off even when a recurrence group shares an XCD
    OPT_REC_NB,                 // minimum utterances per recurrence group (0 = automatic)
    OPT_REC_PIPE,               // pipelined halves in the multi-utterance forward recurrence
    OPT_REC_MFMA,               // multi-utterance recurrences on the matrix pipe where eligible (1: automatic; forward: 2 pipeline form always, 3 first form always)
    OPT_REC_TRACE,              // phase stamps of the pipeline form's workgroup 0 into its id buffer (profiling aid)
    OPT_CELL_MT,                // M-tiles per workgroup of the per-step cell kernel (0 = automatic)
    OPT_GEMM_SK_FIXUP,          // stream-K with in-kernel fix-up (parked partial sums, no atomics / zeroing) on/off
    OPT_GEMM_SKF_MIN_KT,        // fewest k-iterations per output tile for that schedule             (-1 = automatic)
    OPT_GEMM_SKF_MIN_RUN,       // fewest k-iterations per workgroup run                              (-1 = automatic)
    OPT_SIDE_FILLS,             // sentinel fills of the decode kernels' hand-off slabs on a side stream (off: measured slower, see las_hip.h)
    OPT_TRUST_ZEROED_GRADS,     // honour LAS_FLAG_GRADS_ZEROED (skip the gradient-block memsets of the backward entry points); 0: always fill (A/B)
    OPT_SPELLER_BIG,            // one-launch decode forward for Hs = 1024 (speller_big.hip)
    OPT_SPELLER_BIG_BWD,        // ... and its backward
    OPT_SPELLER_BIG_TUNE,       // ... its poll pacing in units of 64 clocks: byte 0 before the first h0 poll, byte 1 between polls, byte 2 / 3 before the first ctx / h1 poll
    OPT_TIME_KERNELS,           // record HIP events around the persistent decode kernels (las_debug_kernel_ms reads them)
    OPT_GEMM_BIG,               // 256 x 256 tiles (gemm_big.hip) where they fill the chip: 1 automatic, 0 never, 2 whenever the shape allows (A/B)
    OPT_SPELLER_PRE_GREEDY,     // free-running (arg-max feedback) decode without a backward pass on the pre-multiplied-context kernel
    OPT_SPELLER_PRE_MH,         // multi-head attention (heads 2..4, teacher forcing) on the pre-multiplied-context kernels
    OPT_REC_EPOCH_SCRATCH,      // hand-off granules of the register-resident recurrences in the library's persistent, epoch-tagged scratch (no fill per launch); 0: fill the caller's buffer (A/B)
    OPT_KEYS_SPLITK,            // psi keys GEMM (64 output columns: few tiles) split over K with a separate activation pass (1) or one pass with the activation in its epilogue (0)
    OPT_DEFER_DW,               // honour LAS_FLAG_DEFER_DW (weight-gradient groups on a side stream beside XCD-confined backward recurrences); 0: inline (A/B)
    OPT_DW_CONCURRENT,          // a Listener layer's weight-gradient group beside its dX GEMM on a second stream, joined inside the call: number of runs the group is cut into (0 = off, the default: measured 5.904 - 5.948 ms per step at 512 - 2048 runs against 5.918 on one stream)
    OPT_REC_EPOCH_SEED,         // test hook of the epoch-tagged hand-off scratch: the next launch's epoch base is at least this (exercises the wrap-around re-zero)
    OPT_COUNT
};

long opt_get(int opt);
void opt_set(int opt, long value);
int opt_find(const char* name);          // "GEMM_ARITH" / "gemm_arith" / "LAS_GEMM_ARITH" -> index, -1 if unknown
const char* opt_name(int opt);

// Per-call override of OPT_GEMM_ARITH for the calling thread (LAS_FLAG_GEMM_F32): RAII, nests.
struct GemmArithScope {
    int saved; unsigned* saved_err;
    // err_word: the calling entry point's device error word; a stream-K fix-up wait that times out inside one of this call's GEMMs raises
    // it too, so that las_clip_adam (which gates on it) skips the update of a step whose gradient holds a wrong tile
    explicit GemmArithScope(int flags, unsigned* err_word = nullptr);
    ~GemmArithScope();
};
unsigned* gemm_call_err_word();          // the enclosing entry point's device error word, or null
int gemm_arith_effective();              // thread override if any, else the option

// HIP-event timing of single kernels on their launch stream (OPT_TIME_KERNELS; bench.py's roofline blocks): RAII around the launch
enum : int { TIMED_DECODE_FWD = 0, TIMED_DECODE_BWD = 1, TIMED_COUNT = 2 };
struct KernelTimer {
    int which; hipStream_t stream; bool on; int dev = -1;
    KernelTimer(int which, hipStream_t stream);
    ~KernelTimer();
};
int kernel_timer_read(int which, float* ms_out);    // synchronises on the closing event

// Which kernel family a dispatcher actually launched (las_debug_last_path): every dispatcher notes its choice, so that a parity
// test can assert that a fixture really pinned the kernel it was written for — a silent fall-back to the generic path would otherwise
// leave every golden green.  Process-wide, most recent launch per slot (a debugging aid, like the trace hooks).
enum : int { PATH_REC_FWD = 0, PATH_REC_BWD = 1, PATH_DECODE_FWD = 2, PATH_DECODE_BWD = 3, PATH_GEMM = 4, PATH_DW = 5, PATH_COUNT = 6 };
void path_note(int which, const char* name);
int path_read(int which, char* out, int cap);

}  // namespace las
