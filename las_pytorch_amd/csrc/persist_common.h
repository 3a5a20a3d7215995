// Device-side helpers shared by the persistent Speller kernels (speller_persist.hip, speller_persist_bwd.hip):
// sentinel-as-flag hand-off, agent-scope accesses, LDS-only barrier, DPP reductions, bounded spins.
#pragma once
#include "las_common.h"
#include <type_traits>

namespace las {
namespace {

using u64 = unsigned long long;
constexpr int PS_THREADS = 1024, PS_NW = 16;
constexpr unsigned PS_SENT = 0xFFFFFFFFu;
// bounded spins: ~160 ns each, i.e. ~42 ms until a wait gives up (round 4: 1 << 21 = 336 ms of dead time per timeout; a hand-off normally
// lands within microseconds, so this is still four orders of magnitude of margin — and a false alarm only costs one step on the generic kernels).
// A caller that cannot re-run a step (no fused update to skip: a plain torch optimizer) extends the budget through the word BEHIND the error
// word (err[1], option HANDOFF_SPIN_LOG2 / las_pytorch_amd._cabi.set_handoff_spin_log2): it is read only once the built-in budget is spent.
constexpr unsigned PS_SPIN_LIMIT = 1u << 18;
__device__ __forceinline__ bool spin_budget_spent(unsigned spins, const unsigned* err, unsigned builtin) {
    if (spins <= builtin) return false;
    return spins > __hip_atomic_load(err + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // 0 (the default): the built-in budget stands
}
constexpr int PS_M = 64;           // attention MLP width handled by the persistent kernel
constexpr int PS_KLD = PS_M + 4;   // LDS row stride of the keys (bank spread)

__device__ __forceinline__ f32x4 ld4p(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
// Per-lane BYTE offset that the optimiser must treat as new in every loop iteration: without this it hoists one 64-bit
// (pointer + lane offset) pair per access out of the step loop, which costs ~30 VGPRs and ends in scratch spills on the
// critical path.  With it the access is `uniform base (SGPR pair) + 32-bit lane offset`.
__device__ __forceinline__ unsigned opaque(unsigned v) { asm volatile("" : "+v"(v)); return v; }
template <class T>
__device__ __forceinline__ T* at_bytes(T* base, unsigned byte_off) {
    return reinterpret_cast<T*>(reinterpret_cast<char*>(const_cast<typename std::remove_const<T>::type*>(base)) + byte_off);
}

// 16-byte agent-scope (sc1: L2 write-through / L2-bypassing) accesses as ONE instruction: a wave then moves whole 128 B
// lines.  (Two 8-byte atomics per lane make every line arrive at the memory side as two partial writes.)
__device__ __forceinline__ f32x4 ld4_agent(const float* p) {
    f32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ bool has_sentinel(const f32x4 v) {
    return __float_as_uint(v[0]) == PS_SENT || __float_as_uint(v[1]) == PS_SENT || __float_as_uint(v[2]) == PS_SENT ||
           __float_as_uint(v[3]) == PS_SENT;
}
// a value that is published must never look like the sentinel (only a NaN could): canonicalise NaNs
__device__ __forceinline__ unsigned pub_bits(float v) { return (v != v) ? 0x7FC00000u : __float_as_uint(v); }
__device__ __forceinline__ void st1_agent(float* p, float v) {
    __hip_atomic_store(reinterpret_cast<unsigned*>(p), pub_bits(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st4_agent(float* p, const f32x4 v) {
    f32x4 b;
    b[0] = __uint_as_float(pub_bits(v[0])); b[1] = __uint_as_float(pub_bits(v[1]));
    b[2] = __uint_as_float(pub_bits(v[2])); b[3] = __uint_as_float(pub_bits(v[3]));
    // The s_nop covers the store-data hazard (a VALU write to the data VGPRs of a >64-bit VMEM store within the next
    // wait states corrupts the stored value): the compiler protects its own stores but cannot see into inline asm.
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 2" : : "v"(p), "v"(b) : "memory");
}
// the same store without the NaN canonicalisation: for writing the SENTINEL itself back into a slot that is about to be reused
__device__ __forceinline__ void st4_agent_raw(float* p, const f32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 2" : : "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void lds_barrier() {     // orders LDS traffic only (does not wait for global stores)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
// bounded-spin bookkeeping shared by every poller; returns true when the caller must give up
__device__ __forceinline__ bool spin_expired(unsigned& spins, unsigned* err, unsigned code) {
    ++spins;
    // the error word is looked at rarely: that agent-scope load takes ~1 us, and a poller that is inside it when its data arrives
    // delays its whole workgroup (with a check every 128 spins one of a workgroup's 16 waves was caught in almost every wait)
    if ((spins & 8191u) == 0) {
        if (spin_budget_spent(spins, err, PS_SPIN_LIMIT)) { atomicExch(err, code); return true; }
        if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return true;
    }
    __builtin_amdgcn_s_sleep(1);
    return false;
}
// Workgroup-level canary.  Every consumer tile is guarded by one dword per PRODUCER workgroup; the first ceil(P/64)
// waves of the consumer watch them (one lane per producer, agent-scope loads) and post the verdict on an LDS flag that
// all 16 waves wait for.  Letting each wave watch its own producers multiplied the agent-scope poll traffic by the number
// of waves and rows (~10^5 32-byte transactions per microsecond over the chip — the memory system's whole transaction
// rate), which is what the hops were actually waiting for.  `ep` is a per-workgroup call counter (monotonic, so a wave
// that runs ahead cannot make a slower one miss its epoch).
typedef __attribute__((address_space(3))) unsigned ps_lds_u32;
__device__ __forceinline__ void wg_canary_wait(volatile unsigned* flags_generic, unsigned ep, int npw, int wave, int lane,
                                               const unsigned* cp, bool active, unsigned* err, unsigned code) {
    // the flags live in LDS: as an LDS-address-space pointer they cost one 32-bit register and ds_ instructions; as a generic pointer
    // they are a 64-bit VGPR pair behind flat_ loads / stores (which register allocation spilled to scratch — on the chain)
    volatile ps_lds_u32* flags = (volatile ps_lds_u32*)flags_generic;
    unsigned spins = 0;
    if (wave < npw) {
        for (;;) {
            const unsigned v = __hip_atomic_load(cp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (!__any(active && v == PS_SENT)) break;
            if (spin_expired(spins, err, code)) break;
        }
        if (lane == 0) flags[wave] = ep;
    }
    for (int k = 0; k < npw; ++k) {
        while ((int)(flags[k] - ep) < 0)
            if (spin_expired(spins, err, code)) return;
    }
}

// ---- fp32 products on the bf16 matrix pipe (same arithmetic as gemm_f32.hip's split-operand mode) -----------------------------
// Every fp32 value is the EXACT sum of three bf16 values (x1 = rne(x), x2 = rne(x - x1), x3 = x - x1 - x2); six of the nine partial
// products per operand pair are issued (the dropped ones are below 2^-26 |a b|), each exact in the MFMA's fp32 accumulator.  On gfx950
// v_mfma_f32_16x16x4_f32 occupies a SIMD's matrix pipe for 32 cycles per 4 k, v_mfma_f32_16x16x32_bf16 for 16 cycles per 32 k: six
// of the latter replace eight of the former for a lane's 8 k-slots — 2.67x less matrix-pipe time on the decode chain.
typedef __attribute__((ext_vector_type(8))) __bf16 ps_bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 ps_bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 ps_bf16x2;
typedef __attribute__((ext_vector_type(2))) float ps_f32x2;
typedef __attribute__((ext_vector_type(4))) unsigned ps_u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned ps_u32x2;
__device__ __forceinline__ unsigned ps_pk_bf16(float a, float b) {
    ps_f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, ps_bf16x2));      // v_cvt_pk_bf16_f32 (round to nearest even)
}
// (x0, x1) -> three packed bf16 pairs (low half = x0) with x = p1 + p2 + p3 exactly
__device__ __forceinline__ void ps_split_pair(float x0, float x1, unsigned& p1, unsigned& p2, unsigned& p3) {
    p1 = ps_pk_bf16(x0, x1);
    x0 -= __uint_as_float(p1 << 16); x1 -= __uint_as_float(p1 & 0xffff0000u);
    p2 = ps_pk_bf16(x0, x1);
    x0 -= __uint_as_float(p2 << 16); x1 -= __uint_as_float(p2 & 0xffff0000u);
    p3 = ps_pk_bf16(x0, x1);
}
// N8 consecutive k-slots of one lane (N8 = 1: 8 slots -> one 16x16x32 operand; the 4-slot form serves Hs = 256) as three bf16 planes
template <int NK> struct PsPlanes { unsigned p[3][NK / 2]; };      // NK = 8 or 4 k-slots
template <int NK>
__device__ __forceinline__ PsPlanes<NK> ps_split(const float (&v)[NK]) {
    PsPlanes<NK> o;
#pragma unroll
    for (int i = 0; i < NK / 2; ++i) ps_split_pair(v[2 * i], v[2 * i + 1], o.p[0][i], o.p[1][i], o.p[2][i]);
    return o;
}
__device__ __forceinline__ f32x4 ps_mfma(const unsigned (&a)[4], const unsigned (&b)[4], f32x4 c) {
    ps_u32x4 av = {a[0], a[1], a[2], a[3]}, bv = {b[0], b[1], b[2], b[3]};
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(ps_bf16x8, av), __builtin_bit_cast(ps_bf16x8, bv), c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 ps_mfma(const unsigned (&a)[2], const unsigned (&b)[2], f32x4 c) {
    ps_u32x2 av = {a[0], a[1]}, bv = {b[0], b[1]};
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(__attribute__((ext_vector_type(4))) short, av),
                                                     __builtin_bit_cast(__attribute__((ext_vector_type(4))) short, bv), c, 0, 0, 0);
}
// c += A . B for one lane-set of k-slots: six partial products, smallest terms first
template <int NK>
__device__ __forceinline__ f32x4 ps_mfma6(const PsPlanes<NK>& a, const PsPlanes<NK>& b, f32x4 c) {
    c = ps_mfma(a.p[1], b.p[1], c);
    c = ps_mfma(a.p[1], b.p[0], c);
    c = ps_mfma(a.p[0], b.p[1], c);
    c = ps_mfma(a.p[2], b.p[0], c);
    c = ps_mfma(a.p[0], b.p[2], c);
    c = ps_mfma(a.p[0], b.p[0], c);
    return c;
}

// Reductions with DPP row operations (1 VALU instruction per level) instead of __shfl_xor (a ds_bpermute, i.e. an LDS
// round trip, per level): these sit on the serial chain of every decode step.
__device__ __forceinline__ float dpp_f(float v, int ctrl) {
    switch (ctrl) {
        case 0: return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
        case 1: return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
        case 2: return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));   // row_half_mirror
        default: return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));  // row_mirror
    }
}
template <int W>
__device__ __forceinline__ float gsum(float v) {      // sum over aligned groups of W <= 16 lanes, result in every lane
    if (W >= 2) v += dpp_f(v, 0);
    if (W >= 4) v += dpp_f(v, 1);
    if (W >= 8) v += dpp_f(v, 2);
    if (W >= 16) v += dpp_f(v, 3);
    return v;
}
__device__ __forceinline__ float lane_f(float v, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l)); }
__device__ __forceinline__ float wsum(float v) {
    v = gsum<16>(v);
    return (lane_f(v, 0) + lane_f(v, 16)) + (lane_f(v, 32) + lane_f(v, 48));
}
__device__ __forceinline__ float wmax(float v) {
    v = fmaxf(v, dpp_f(v, 0)); v = fmaxf(v, dpp_f(v, 1)); v = fmaxf(v, dpp_f(v, 2)); v = fmaxf(v, dpp_f(v, 3));
    return fmaxf(fmaxf(lane_f(v, 0), lane_f(v, 16)), fmaxf(lane_f(v, 32), lane_f(v, 48)));
}
__device__ __forceinline__ float dot4p(const f32x4 a, const f32x4 b, float acc) {
    acc = fmaf(a[0], b[0], acc); acc = fmaf(a[1], b[1], acc); acc = fmaf(a[2], b[2], acc); acc = fmaf(a[3], b[3], acc);
    return acc;
}

}  // namespace

}  // namespace las
