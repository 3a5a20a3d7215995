// Shared device/host helpers for the LAS hot-path kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

namespace las {

// ---- status / error string (thread-local; C ABI returns int) --------------------------
enum : int {
    LAS_OK = 0,
    LAS_ERR_ARG = 1,        // shape / pointer / alignment precondition violated
    LAS_ERR_HIP = 2,        // a HIP runtime call failed
    LAS_ERR_UNSUPPORTED = 3,// configuration not implemented by the kernels
    LAS_ERR_DEVICE = 4,     // a kernel reported a device-side failure (e.g. hand-off timeout)
};

extern thread_local char g_err[512];

inline int fail(int code, const char* fmt, const char* a = "", long x = 0, long y = 0) {
    snprintf(g_err, sizeof(g_err), fmt, a, x, y);
    return code;
}

#define LAS_HIP_CHECK(expr)                                                              \
    do {                                                                                 \
        hipError_t _e = (expr);                                                          \
        if (_e != hipSuccess) {                                                          \
            snprintf(::las::g_err, sizeof(::las::g_err), "%s failed: %s (%s:%d)", #expr, \
                     hipGetErrorString(_e), __FILE__, __LINE__);                         \
            return ::las::LAS_ERR_HIP;                                                   \
        }                                                                                \
    } while (0)

#define LAS_TRY(expr) do { int _rc = (expr); if (_rc != LAS_OK) return _rc; } while (0)
#define LAS_REQUIRE(cond, msg)                                                           \
    do {                                                                                 \
        if (!(cond)) {                                                                   \
            snprintf(::las::g_err, sizeof(::las::g_err), "precondition failed: %s [%s] (%s:%d)", msg, #cond, \
                     __FILE__, __LINE__);                                                \
            return ::las::LAS_ERR_ARG;                                                   \
        }                                                                                \
    } while (0)

#define LAS_LAUNCH_CHECK() LAS_HIP_CHECK(hipGetLastError())

// ---- accurate transcendental helpers (no fast-math: parity bar is 1e-3 rel through ~400
//      recurrent steps and identical argmax; SURVEY.md section 7 "Transcendentals parity") ----
// Two flavours.  LAS_ACCURATE_ACT=1 uses libm-grade expf/tanhf and IEEE division (~0.75 us per recurrent step of
// dependent latency in the persistent LSTM kernels).  The default uses the hardware transcendental units
// (v_exp_f32 / v_rcp_f32, ~1 ulp each; absolute error of sigma/tanh <= ~3e-7), which keeps the 1e-3 / identical-argmax
// parity bar with three orders of magnitude of margin (tests/test_hip_parity.py records the observed error).
#ifndef LAS_ACCURATE_ACT
#define LAS_ACCURATE_ACT 0
#endif
#if LAS_ACCURATE_ACT
__device__ __forceinline__ float sigmoidf_acc(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float tanhf_acc(float x) { return tanhf(x); }
#else
__device__ __forceinline__ float sigmoidf_acc(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
__device__ __forceinline__ float tanhf_acc(float x) {      // 1 - 2/(e^{2x}+1): saturates correctly at +-inf
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.8853900817779268f * x));
}
#endif

// Attention MLP activation codes (las_speller_desc::relu): the reference accepts any torch.nn.functional name
// (model/las_model.py:270-273); relu is what the configs use, tanh / sigmoid are provided with libm-accurate forms.
enum : int { LAS_ACT_NONE = 0, LAS_ACT_RELU = 1, LAS_ACT_TANH = 2, LAS_ACT_SIGMOID = 3 };
__device__ __forceinline__ float act_apply(float x, int code) {
    switch (code) {
        case LAS_ACT_RELU: return fmaxf(x, 0.f);
        case LAS_ACT_TANH: return tanhf(x);
        case LAS_ACT_SIGMOID: return 1.0f / (1.0f + expf(-x));
        default: return x;
    }
}
// derivative factor from the POST-activation value y = act(x)
__device__ __forceinline__ float act_grad(float y, int code) {
    switch (code) {
        case LAS_ACT_RELU: return y > 0.f ? 1.f : 0.f;
        case LAS_ACT_TANH: return 1.f - y * y;
        case LAS_ACT_SIGMOID: return y * (1.f - y);
        default: return 1.f;
    }
}

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// Co-residency check of a persistent launch whose workgroups spin on each other: the occupancy calculator must admit at
// least one workgroup of this kernel per CU with the launch's REAL dynamic LDS size (register / LDS footprint), and the
// grid must not exceed the CU count.  The caller falls back to kernels without inter-workgroup waits otherwise.
template <class Kern>
inline bool persistent_launch_fits(Kern kernel, int threads, size_t dyn_lds, int grid) {
    int dev = 0, cus = 0, nb = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return false;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, threads, dyn_lds) != hipSuccess) return false;
    return nb >= 1 && grid <= cus;
}

}  // namespace las
