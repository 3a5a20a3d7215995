// Shared between the GEMM translation units (gemm_f32.hip: 128 x 128 tiles, every mode and schedule; gemm_big.hip: 256 x 256 tiles of the
// split-operand arithmetic): vector types, the exact three-way bf16 split, the kernel parameter blocks.
#pragma once
#include "las_common.h"

namespace las {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

static __device__ __forceinline__ unsigned pk_bf16(float a, float b) {
    f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));      // v_cvt_pk_bf16_f32 (round to nearest even)
}
// (x0, x1) -> three packed bf16 pairs (low half = x0) with x = p1 + p2 + p3 exactly
static __device__ __forceinline__ void split_pair(float x0, float x1, unsigned& p1, unsigned& p2, unsigned& p3) {
    p1 = pk_bf16(x0, x1);
    x0 -= __uint_as_float(p1 << 16); x1 -= __uint_as_float(p1 & 0xffff0000u);
    p2 = pk_bf16(x0, x1);
    x0 -= __uint_as_float(p2 << 16); x1 -= __uint_as_float(p2 & 0xffff0000u);
    p3 = pk_bf16(x0, x1);
}

struct GemmParams {
    const float* A; const float* B; float* C; const float* bias0; const float* bias1;
    int M, N, K;
    long lda, ldb, ldc;
    long sA, sB, sC, sBias0, sBias1;
    int splitk, kper;
    int accumulate, relu, atomic;
    int a_vec, b_vec;   // 16-byte vector loads legal for this operand
    // optional second source along K: k >= K1 reads A2 / B2 at k - K1 (same leading dimensions and layouts; K1 % BK == 0):
    // C = [A | A2] [B ; B2] in one pass instead of a second accumulating GEMM
    const float* A2; const float* B2; int K1;
    int gx, swz;
    // persistent (data-parallel + stream-K) schedule
    int persistent, gy, kt, dp_tiles, sk_atomic_whole, xcd_swz;
    long sk_iters, sk_per;
    // stream-K with in-kernel fix-up (no atomics, no zeroing pass): a workgroup that covers a tile's k-range only from k-iteration
    // it0 > 0 parks its 128x128 partial sum in sk_part[slot] and raises sk_flag[slot] = sk_id; the workgroup that owns k-iteration 0
    // of the tile (its LAST segment) adds the parked sums of the slots behind it and runs the epilogue (bias / accumulate / relu)
    float* sk_part; unsigned* sk_flag; unsigned* sk_err; unsigned sk_id;
    unsigned* call_err;      // device error word of the enclosing entry point (null outside one): raised together with sk_err
};

// what a segment does with its accumulators
enum : int { SEG_STORE = 0, SEG_ATOMIC = 1, SEG_PART = 2 };
struct SegRole {
    int kind;           // SEG_STORE: epilogue + plain store (after adding the partial sums of slots [c0, c1));  SEG_ATOMIC: atomicAdd onto C;
    int slot;           // SEG_PART: park the partial sum in sk_part[slot]
    int c0, c1;
    bool add_bias;
};
static __device__ __forceinline__ SegRole seg_store(bool add_bias = true) { return SegRole{SEG_STORE, 0, 0, 0, add_bias}; }
static __device__ __forceinline__ SegRole seg_atomic(bool add_bias) { return SegRole{SEG_ATOMIC, 0, 0, 0, add_bias}; }


constexpr int GROUP_MAX = 8;
struct GemmGroupParams {
    GemmParams prob[GROUP_MAX]; long first[GROUP_MAX + 1]; int n; int xcd_swz;
    unsigned* xcd_probe;
    unsigned* run_counter;      // drawn-runs form (xcd_lo > 0 or nruns > 0): zeroed before the launch; the workgroups draw their runs from it
    int nruns;                  // ... number of equal runs the whole problem is cut into (any number of workgroups may take part)
    int xcd_lo;      // > 0: the workgroups that find themselves on XCDs [0, xcd_lo) (actual XCC id) leave at once; the others draw the W' =
                     // gridDim.x / 8 * (8 - xcd_lo) equal runs of the whole problem from run_counter — correct for ANY placement of the blocks
                     // (XCD partition beside a latency-bound chain kernel confined to those XCDs)
    float* sk_part; unsigned* sk_flag; unsigned* sk_err; unsigned sk_id;      // stream-K fix-up (see GemmParams); null: atomics onto zeroed outputs
    unsigned* call_err;
};


}  // namespace las
