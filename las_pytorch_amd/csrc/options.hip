// Option registry (options.h) and the C-ABI entry points las_set_option / las_get_option.
#include "../../include/las_hip.h"
#include "las_common.h"
#include "options.h"
#include "las_kernels.h"
#include <atomic>
#include <ctype.h>
#include <mutex>
#include <stdlib.h>

namespace las {

namespace {
struct OptDef { const char* name; long dflt; };
// order = enum Opt
const OptDef kDefs[OPT_COUNT] = {
    {"GEMM_ARITH", 1}, {"GEMM_STREAMK", -1}, {"GEMM_SK_MIN_TILES", -1}, {"GEMM_SPLIT_BELOW", -1}, {"GEMM_SPLIT_TARGET", -1},
    {"GEMM_SLOTS_PER_CU", 2}, {"GEMM_GROUP", 1}, {"GEMM_XCD_SWZ", 1}, {"GEMM_BATCH_DIRS", 1},
    {"SPELLER_PERSIST", 1}, {"SPELLER_PERSIST_BWD", 1}, {"SPELLER_PRE", 1}, {"SPELLER_PRE_BWD", 1},
    {"REC_UW", 0}, {"REC_AGENT_HANDOFF", 0}, {"REC_NB", 0}, {"REC_PIPE", 1}, {"REC_MFMA", 1}, {"REC_TRACE", 0}, {"CELL_MT", 0},
    {"GEMM_SK_FIXUP", 0}, {"GEMM_SKF_MIN_KT", -1}, {"GEMM_SKF_MIN_RUN", -1},
    {"SIDE_FILLS", 0}, {"TRUST_ZEROED_GRADS", 1}, {"SPELLER_BIG", 1}, {"SPELLER_BIG_BWD", 1}, {"SPELLER_BIG_TUNE", 0},
    {"TIME_KERNELS", 0}, {"GEMM_BIG", 1}, {"SPELLER_PRE_GREEDY", 1}, {"SPELLER_PRE_MH", 1},
    {"REC_EPOCH_SCRATCH", 1}, {"KEYS_SPLITK", 1}, {"DEFER_DW", 1}, {"DW_CONCURRENT", 0}, {"REC_EPOCH_SEED", 0},
};
std::atomic<long> g_val[OPT_COUNT];
std::atomic<int> g_init{0};

void init_once() {
    int st = g_init.load(std::memory_order_acquire);
    if (st == 2) return;
    int expect = 0;
    if (g_init.compare_exchange_strong(expect, 1, std::memory_order_acq_rel)) {
        for (int i = 0; i < OPT_COUNT; ++i) {
            char env[64];
            snprintf(env, sizeof(env), "LAS_%s", kDefs[i].name);
            const char* e = getenv(env);
            g_val[i].store(e ? atol(e) : kDefs[i].dflt, std::memory_order_relaxed);
        }
        g_init.store(2, std::memory_order_release);
    } else {
        while (g_init.load(std::memory_order_acquire) != 2) {}
    }
}

thread_local int tl_arith = -1;     // per-call override (GemmArithScope)
}  // namespace

long opt_get(int opt) { init_once(); return g_val[opt].load(std::memory_order_relaxed); }
void opt_set(int opt, long value) { init_once(); g_val[opt].store(value, std::memory_order_relaxed); }
const char* opt_name(int opt) { return kDefs[opt].name; }
int opt_find(const char* name) {
    if (!name) return -1;
    if ((name[0] == 'L' || name[0] == 'l') && (name[1] == 'A' || name[1] == 'a') && (name[2] == 'S' || name[2] == 's') && name[3] == '_') name += 4;
    for (int i = 0; i < OPT_COUNT; ++i) {
        const char* a = kDefs[i].name; const char* b = name;
        while (*a && *b && toupper((unsigned char)*b) == *a) { ++a; ++b; }
        if (!*a && !*b) return i;
    }
    return -1;
}

thread_local unsigned* tl_call_err = nullptr;
GemmArithScope::GemmArithScope(int flags, unsigned* err_word) : saved(tl_arith), saved_err(tl_call_err) {
    if (flags & LAS_FLAG_GEMM_F32) tl_arith = 0;
    if (err_word) tl_call_err = err_word;
}
GemmArithScope::~GemmArithScope() { tl_arith = saved; tl_call_err = saved_err; }
unsigned* gemm_call_err_word() { return tl_call_err; }
int gemm_arith_effective() { return tl_arith >= 0 ? tl_arith : (opt_get(OPT_GEMM_ARITH) ? 1 : 0); }

namespace {
// events per (device, kernel), guarded: two device threads with TIME_KERNELS=1 must not share or race on them
constexpr int TIMER_MAX_DEV = 16;
std::mutex g_ev_mu;
hipEvent_t g_ev[TIMER_MAX_DEV][TIMED_COUNT][2] = {};
bool g_ev_valid[TIMER_MAX_DEV][TIMED_COUNT] = {};
int timer_device() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= TIMER_MAX_DEV) return -1;
    return dev;
}
}  // namespace
KernelTimer::KernelTimer(int w, hipStream_t s) : which(w), stream(s), on(opt_get(OPT_TIME_KERNELS) != 0) {
    if (!on) return;
    dev = timer_device();
    if (dev < 0) { on = false; return; }
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) { on = false; return; }   // a graph replay records nothing
    std::lock_guard<std::mutex> lk(g_ev_mu);
    for (int k = 0; k < 2; ++k)
        if (!g_ev[dev][which][k] && hipEventCreate(&g_ev[dev][which][k]) != hipSuccess) { on = false; return; }
    g_ev_valid[dev][which] = false;
    if (hipEventRecord(g_ev[dev][which][0], stream) != hipSuccess) on = false;
}
KernelTimer::~KernelTimer() {
    if (!on) return;
    std::lock_guard<std::mutex> lk(g_ev_mu);
    if (hipEventRecord(g_ev[dev][which][1], stream) == hipSuccess) g_ev_valid[dev][which] = true;
}
SideStream& side_stream() {
    thread_local SideStream tl[16];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = 0;
    return tl[dev];
}
bool SideStream::ok(hipStream_t main) {
    if (opt_get(OPT_SIDE_FILLS) == 0) return false;
    if (s != nullptr) return true;
    if (tried) return false;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(main, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) return false;      // not now; try again on a later call
    tried = true;
    if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) { s = nullptr; (void)hipGetLastError(); return false; }
    if (hipEventCreateWithFlags(&e_fork, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&e_join, hipEventDisableTiming) != hipSuccess) {
        (void)hipGetLastError(); (void)hipStreamDestroy(s); s = nullptr; return false;
    }
    return true;
}
int SideStream::fork(hipStream_t main) {
    LAS_HIP_CHECK(hipEventRecord(e_fork, main));
    LAS_HIP_CHECK(hipStreamWaitEvent(s, e_fork, 0));
    return LAS_OK;
}
int SideStream::join(hipStream_t main) {
    LAS_HIP_CHECK(hipEventRecord(e_join, s));
    LAS_HIP_CHECK(hipStreamWaitEvent(main, e_join, 0));
    return LAS_OK;
}

static std::atomic<unsigned*> g_xcd_probe{nullptr};
unsigned* xcd_probe_ptr() { return g_xcd_probe.load(std::memory_order_relaxed); }
// One instance per DEVICE, not per host thread: PyTorch runs the backward Functions on its autograd worker thread while the caller joins from
// the thread that drives the step — work deferred by one thread must be visible to the other.  (One host thread per device at a time, as
// for every entry point; the mutex orders the hand-over between the autograd thread and the caller's.)
static std::mutex g_defer_mu;
DeferSide& defer_side() {
    static DeferSide inst[16];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = 0;
    return inst[dev];
}
hipStream_t DeferSide::begin(hipStream_t main) {
    std::lock_guard<std::mutex> lk(g_defer_mu);
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(main, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) return nullptr;      // a capture keeps everything on one stream
    if (s == nullptr) {
        if (tried) return nullptr;
        tried = true;
        int lo = 0, hi = 0;      // lowest priority: the critical-path GEMM on the caller's stream wins the CUs both could use
        if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) { lo = 0; (void)hipGetLastError(); }
        if (hipStreamCreateWithPriority(&s, hipStreamNonBlocking, lo) != hipSuccess) { s = nullptr; (void)hipGetLastError(); return nullptr; }
        if (hipEventCreateWithFlags(&e_fork, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&e_done, hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError(); (void)hipStreamDestroy(s); s = nullptr; return nullptr;
        }
    }
    if (hipEventRecord(e_fork, main) != hipSuccess || hipStreamWaitEvent(s, e_fork, 0) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return s;
}
int DeferSide::end() {
    std::lock_guard<std::mutex> lk(g_defer_mu);
    LAS_HIP_CHECK(hipEventRecord(e_done, s));
    pending = true;
    return LAS_OK;
}
int DeferSide::end_joined(hipStream_t main) {
    std::lock_guard<std::mutex> lk(g_defer_mu);
    hipEvent_t e = pending ? e_fork : e_done;      // (e_done may be what an earlier deferred launch is still known by: keep it)
    LAS_HIP_CHECK(hipEventRecord(e, s));
    LAS_HIP_CHECK(hipStreamWaitEvent(main, e, 0));
    return LAS_OK;
}
int DeferSide::join(hipStream_t main) {
    std::lock_guard<std::mutex> lk(g_defer_mu);
    if (!pending) return LAS_OK;
    pending = false;
    LAS_HIP_CHECK(hipStreamWaitEvent(main, e_done, 0));
    path_note(PATH_DW, "joined");      // (what a test asserts: the deferred work of the step was really waited for)
    return LAS_OK;
}

namespace {
std::mutex g_path_mu;
char g_path[PATH_COUNT][48] = {};
}  // namespace
void path_note(int which, const char* name) {
    if (which < 0 || which >= PATH_COUNT || !name) return;
    std::lock_guard<std::mutex> lk(g_path_mu);
    snprintf(g_path[which], sizeof(g_path[which]), "%s", name);
}
int path_read(int which, char* out, int cap) {
    if (which < 0 || which >= PATH_COUNT || !out || cap <= 0) return fail(LAS_ERR_ARG, "las_debug_last_path: slot %s%ld", "", (long)which);
    std::lock_guard<std::mutex> lk(g_path_mu);
    snprintf(out, (size_t)cap, "%s", g_path[which]);
    return LAS_OK;
}

int kernel_timer_read(int which, float* ms_out) {
    const int dev = timer_device();
    if (which < 0 || which >= TIMED_COUNT || !ms_out || dev < 0) return fail(LAS_ERR_ARG, "no timed launch of kernel %s%ld", "", (long)which);
    hipEvent_t e0, e1;
    {
        std::lock_guard<std::mutex> lk(g_ev_mu);
        if (!g_ev_valid[dev][which]) return fail(LAS_ERR_ARG, "no timed launch of kernel %s%ld", "", (long)which);
        e0 = g_ev[dev][which][0]; e1 = g_ev[dev][which][1];
    }
    LAS_HIP_CHECK(hipEventSynchronize(e1));
    LAS_HIP_CHECK(hipEventElapsedTime(ms_out, e0, e1));
    return LAS_OK;
}

}  // namespace las

using namespace las;

extern "C" {

int las_set_option(const char* key, int64_t value) {
    const int i = opt_find(key);
    if (i < 0) return fail(LAS_ERR_ARG, "unknown option %s", key ? key : "(null)");
    opt_set(i, (long)value);
    return LAS_OK;
}

void las_debug_xcd_probe(unsigned* dev_buf) { g_xcd_probe.store(dev_buf, std::memory_order_relaxed); }
int las_join_deferred(void* stream) { return defer_side().join((hipStream_t)stream); }

int las_debug_kernel_ms(int which, float* ms_out) { return kernel_timer_read(which, ms_out); }
int las_debug_last_path(int which, char* out, int cap) { return path_read(which, out, cap); }

int las_get_option(const char* key, int64_t* value_out) {
    const int i = opt_find(key);
    if (i < 0 || !value_out) return fail(LAS_ERR_ARG, "unknown option %s", key ? key : "(null)");
    *value_out = (int64_t)opt_get(i);
    return LAS_OK;
}

}  // extern "C"
