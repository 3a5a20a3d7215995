// Shared by the matrix-pipe recurrences (pblstm_rec_mfma.hip: first form; pblstm_rec_mfma2.hip: wave-specialised pipeline): launch
// arguments, group geometry and the run-time placement check.
#pragma once
#include "las_common.h"
#include "persist_common.h"

namespace las {
namespace {

constexpr int RM_THREADS = 1024, RM_NB = 16, RM_UW = 32;
constexpr size_t REC_MFMA_RING_OFFSET = 64 * 1024;      // bytes from the start of xbuf: behind the id slots (and a trace build's stamps)

struct RecMfmaArgs {
    float* gates; const float* w_hh_f; const float* w_hh_r; float* out; float* cbuf; float* hprev;
    int B, T, b0, Bc;                                   // this launch covers utterances [b0, b0 + Bc)
    unsigned* err;
    int nbat;                                           // batches of 16 sequences per group: 1, or 2 stepped alternately
    unsigned long long* idbuf;                          // zeroed: 32 id slots per group (run-time placement check)
    int force_agent;                                    // A/B: agent-scope hand-off even when a group shares an XCD
    float* ring;                                        // [group][batch 2][slot 4][16 sequences][H], sentinel-prefilled: the hand-off slab
    int trace;                                          // pipeline form: phase stamps of workgroup 0 / batch 0 behind the id slots (option REC_TRACE)
};

// Run-time placement check (as pblstm_rec.hip::same_xcd_group): every member publishes its XCC id with agent-scope stores and reads
// all the others'.  True iff all G workgroups of the group run on one XCD — then h may travel through that XCD's L2 (plain stores,
// L1-bypassing loads: ~0.5 us per hop) instead of through memory (write-through stores, ~1.3 us + a fabric round trip per tile).
template <int G>
__device__ __forceinline__ bool rm_same_xcd(unsigned long long* idbuf, int member, unsigned* err, volatile unsigned* lds_flag) {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcc));
    const int tid = threadIdx.x;
    if (tid == 0) {
        *lds_flag = 1u;
        __hip_atomic_store(idbuf + member, (0xC0DE0002ull << 32) | (unsigned long long)xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (tid < G) {
        unsigned spins = 0;
        unsigned long long x;
        for (;;) {
            x = __hip_atomic_load(idbuf + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((unsigned)(x >> 32) == 0xC0DE0002u) break;
            if (spin_expired(spins, err, 0xDEAD0033u)) break;
        }
        if ((unsigned)x != xcc || (unsigned)(x >> 32) != 0xC0DE0002u) *lds_flag = 0u;
    }
    __syncthreads();
    return *lds_flag != 0u;
}


}  // namespace
}  // namespace las
