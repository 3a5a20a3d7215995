// fp32 MFMA GEMM for gfx950:  C[M,N] (+)= act( A(M,K) * B(K,N) + bias0[N] + bias1[N] )
//
// Replaces the dense contractions the reference delegates to ATen/cuDNN:
//   * LSTM input projection  X * W_ih^T + b_ih + b_hh   (nn.LSTM, reference model/las_model.py:90)
//   * psi(listener_feature)  relu(feat * W_psi^T + b)    (model/las_model.py:279, utils/functions.py:72-77)
//   * every dX / dW contraction of the backward pass (autograd of the above, solver/solver.py:95)
//
// Design (CDNA4): 128x128 block tile, BK=16, 256 threads = 4 wave64 in a 2x2 grid, each wave owns a
// 64x64 sub-tile = 2x2 32x32 MFMA accumulators (64 acc VGPRs).  Two arithmetic modes, both fp32 in / fp32 out / fp32 accumulate:
//   arith 1 (default): every operand is split exactly into three bf16 terms on its way to LDS and six partial products run on
//            v_mfma_f32_32x32x16_bf16 (16x the fp32 MFMA rate): fp32-faithful at 2.67x the fp32 matrix roofline, see split_pair
//   arith 0: v_mfma_f32_32x32x2_f32, fp32 operands on the fp32 matrix pipe (157 TF peak; there is no TF32 on gfx950);
//            operands go global -> registers -> LDS (K-major tiles, conflict-free ds_read_b32 fragments), double-buffered.
// Either operand may be K-contiguous or M/N-contiguous (all four transposition cases of the backward pass).
#include "las_common.h"
#include "las_kernels.h"
#include "options.h"
#include "gemm_common.h"
#include <stdlib.h>
#include <algorithm>
#include <atomic>
#include <mutex>
#include <type_traits>

namespace las {

#ifndef LAS_GEMM_BK
#define LAS_GEMM_BK 16
#endif
#ifndef LAS_GEMM_DMA
#define LAS_GEMM_DMA 0      // 1: M/N-contiguous operands via global_load_lds (correct, measured 8-10 % SLOWER than register staging)
#endif
#ifndef LAS_GEMM_PF
#define LAS_GEMM_PF 1
#endif
#ifndef LAS_SPLIT_ABL
#define LAS_SPLIT_ABL 0     // timing ablations of the split-operand loop (wrong results): 1 no global loads, 2 no split/store, 4 no fragment reloads
#endif
#ifndef LAS_PLANES_ABL
#define LAS_PLANES_ABL 0    // timing ablations of the pre-split-operand loop (wrong results): 1 no global loads, 2 no LDS stores, 4 no fragment reloads, 8 no barrier
#endif
#ifndef LAS_GEMM_ARITH_DEFAULT
#define LAS_GEMM_ARITH_DEFAULT 1
#endif
constexpr int BM = 128, BN = 128, BK = LAS_GEMM_BK, PAD = 4, GEMM_THREADS = 256, PF = LAS_GEMM_PF;
constexpr int NLD = BM * BK / 4 / GEMM_THREADS;     // float4 loads per thread per operand tile
constexpr int KQ = BK / 4;                           // float4 per K-contiguous row segment

typedef __attribute__((address_space(3))) void* lds_ptr_t;
static __device__ __forceinline__ lds_ptr_t to_lds(float* p) { return (lds_ptr_t)p; }     // generic -> LDS address space

// ---- split-operand arithmetic (LAS_GEMM_ARITH=1) ------------------------------------------------------------------------
// fp32 MFMA runs at 1/16 of the bf16 MFMA rate on gfx950.  Every fp32 value is the EXACT sum of three bf16 values
// (x = x1 + x2 + x3 with x1 = rne_bf16(x), x2 = rne_bf16(x - x1), x3 = x - x1 - x2: the residuals are exact in fp32 and the
// third one has at most 8 significant bits left), so a*b = sum_ij a_i*b_j exactly, every a_i*b_j is exact in the fp32
// accumulator of the bf16 MFMA (8 x 8 significant bits), and the three terms that are dropped (a2*b3, a3*b2, a3*b3) are below
// 2^-26 |a*b| — a quarter of the rounding error of ONE fp32 multiply-add.  Six v_mfma_f32_32x32x16_bf16 per 16 k-steps replace
// eight v_mfma_f32_32x32x2_f32 at 1/16 of the cost each: 2.67x the fp32-MFMA roofline at the accuracy of an fp32 GEMM
// (tests/test_hip_kernels.py compares both against float64).  The split happens in registers on the way from global memory to
// LDS; LDS holds three bf16 planes per operand as k-pairs ([plane][k/2][row] dwords), conflict-free for both store patterns and
// for the fragment reads (4 dwords = 8 consecutive k per lane and plane).
// Two LDS images of a bf16 plane (128 rows x 16 k, at most 1088 dwords), chosen by the operand's memory orientation so that
// stores AND fragment reads are wide and conflict-free:
//   K-contiguous operand  : [k-half (2)][row][4 dwords]  (SP_KH dwords per half)  store 8 B per thread and plane, read one b128
//   row-contiguous operand: [k-quad (4)][slot(row)][2 dwords]  (256 dwords per quad; a dword = one k-pair).  A thread holds ONE
//       k-pair of four rows, its partner 32 lanes up the other k-pair of the quad for the same rows: one v_permlane32_swap per
//       plane and row pair gives the lower lane both k-pairs of rows r, r+1 and the upper lane those of rows r+2, r+3, stored as
//       8 bytes each; slot(row) = (row % 4) * 32 + (row / 4 + 8 (row % 4)) % 32 makes those stores (16 lanes x 8 B of one
//       row class) and the fragment reads (32 consecutive rows = 4 classes x 8 slots, 64 distinct banks) conflict-free.
// either way a lane ends up with k = 8 (lane / 32) .. +7 of its row in increasing order.
constexpr int SP_LD = BM + 8;                    // (plane size; the k-pair-major image of the first version needed the padding)
constexpr int SP_KH = BM * 4 + 16;               // +16 dwords: the two k-halves of an 8-byte store (16 lanes, 32 banks) stay disjoint
constexpr int SP_KQ = BM * 2;                    // dwords per k-quad of a row-contiguous operand
constexpr int SP_PLANE = (16 / 2) * SP_LD;       // one bf16 plane of one operand tile (128 rows x 16 k)
static __device__ __forceinline__ int sp_slot(int row) { return (row & 3) * 32 + (((row >> 2) + 8 * (row & 3)) & 31); }
static_assert(SP_PLANE >= 2 * SP_KH && SP_PLANE >= 4 * SP_KQ && SP_KH % 4 == 0, "both plane images fit; 16-byte aligned k-halves");
constexpr int SP_OPER = 3 * SP_PLANE;
constexpr int SP_BUF = 2 * SP_OPER;              // A and B
constexpr int SP_NBUF = 3;                       // tile kt is multiplied while kt+1 is read into fragments and kt+2 is stored
constexpr int SP_SMEM_BYTES = SP_NBUF * SP_BUF * 4;

// registers of one operand tile (two 16-byte loads per thread) -> the three LDS planes, in two halves so that the caller can
// spread the work between its MFMA groups
//  KC : load i covers row (t + 256 i) / 4, k = 4 ((t + 256 i) % 4) .. +3         -> half i: two k-pair dwords per plane
//  !KC: loads 0/1 cover k = 2 (t / 32) and 2 (t / 32) + 1, rows 4 (t % 32) .. +3  -> half h: rows h and h+2, exchanged with the
//       partner lane so that each lane stores both k-pairs of one row
template <bool KC, int HALF>
static __device__ __forceinline__ void split_store_half(unsigned* S, const f32x4 (&reg)[NLD]) {
    static_assert(NLD == 2 && BK == 16, "split path is written for BK = 16");
    const int t = threadIdx.x;
    unsigned a[3], b[3];
    if constexpr (KC) {
        const int idx = t + HALF * GEMM_THREADS;
        const int row = idx >> 2, q = idx & 3;
        split_pair(reg[HALF][0], reg[HALF][1], a[0], a[1], a[2]);
        split_pair(reg[HALF][2], reg[HALF][3], b[0], b[1], b[2]);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            u32x2 v = {a[pl], b[pl]};
            *reinterpret_cast<u32x2*>(&S[pl * SP_PLANE + (q >> 1) * SP_KH + row * 4 + (q & 1) * 2]) = v;
        }
    } else {
        // rows rq + HALF (held as k-pair 2w by lanes < 32, as k-pair 2w+1 by their partners) and rq + HALF + 2
        const int rq = (t & 31) * 4 + HALF + ((t & 32) ? 2 : 0), kq = t >> 6;
        split_pair(reg[0][HALF], reg[1][HALF], a[0], a[1], a[2]);
        split_pair(reg[0][HALF + 2], reg[1][HALF + 2], b[0], b[1], b[2]);
        unsigned* dst = S + kq * SP_KQ + sp_slot(rq) * 2;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            auto r = __builtin_amdgcn_permlane32_swap(a[pl], b[pl], false, false);      // {[a.lo | b.lo], [a.hi | b.hi]}
            u32x2 v = {r[0], r[1]};                                                      // (even k-pair, odd k-pair) of this lane's row
            *reinterpret_cast<u32x2*>(dst + pl * SP_PLANE) = v;
        }
    }
}
// Guarded form of the split path's register image (edge tiles, a K range that is not a multiple of 16, operands that are not
// 16-byte aligned): same thread -> element map as the vector loads above, zero outside [0,R) x [k0,kend).
template <bool KC>
static __device__ __forceinline__ void split_load_guarded(const float* __restrict__ P, long ld, int R, int r0, int k0, int kend,
                                                          bool vec_ok, f32x4 (&reg)[NLD]) {
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if constexpr (KC) {
            const int idx = t + i * GEMM_THREADS;
            const int r = r0 + (idx >> 2), k = k0 + (idx & 3) * 4;
            if (r < R) {
                const float* q = P + (long)r * ld + k;
                if (vec_ok && k + 3 < kend) {
                    v = *reinterpret_cast<const f32x4*>(q);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) if (k + j < kend) v[j] = q[j];
                }
            }
        } else {
            const int k = k0 + 2 * (t >> 5) + i, r = r0 + (t & 31) * 4;
            if (k < kend) {
                const float* q = P + (long)k * ld + r;
                if (vec_ok && r + 3 < R) {
                    v = *reinterpret_cast<const f32x4*>(q);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) if (r + j < R) v[j] = q[j];
                }
            }
        }
        reg[i] = v;
    }
}
struct SplitFrag { u32x4 a[3][2], b[3][2]; };      // [plane][32-row tile]: 8 consecutive k of this lane's row as bf16

// ---- pre-split operands (MODE 2) ------------------------------------------------------------------------------------------
// The operand split above costs ~88 VALU instructions per k-tile and wave beside 24 MFMAs, and every workgroup along the other tile
// dimension repeats it (a weight tile is split by every M-tile's workgroup).  An operand that several GEMMs of a step read (layer
// outputs, weights, gate gradients) can instead be split ONCE, by its producer or by split_planes_kernel below, into the
// "P8x3" image: for a logical R x C fp32 matrix with row stride ld (elements, a multiple of 8)
//     granule(r, c / 8, plane) = 16 bytes = the bf16 term `plane` of elements (r, 8 (c / 8) .. + 7),  at 16-byte index (r (ld / 8) + c / 8) 3 + plane
// — the three planes of an octet are adjacent (48 bytes), so a K-contiguous reader takes 96 contiguous bytes per row and k-tile and a
// row-contiguous reader 768 contiguous bytes per k.  6 bytes per element instead of 4; x = p1 + p2 + p3 exactly as in split_pair.
// The GEMM then moves granules global -> registers -> LDS without arithmetic:
//   K-contiguous operand : a granule IS the fragment's 8 consecutive k: one ds_write_b128 per plane into the [k-half][row][4 dwords] image
//   row-contiguous       : a granule holds ONE k of 8 rows; the four lanes l, l+16, l+32, l+48 of a wave hold the four k of a quad for the
//       same row octet, a 4 x 4 dword transpose across them is two v_permlane32_swap + two v_permlane16_swap, four v_perm_b32 pair the
//       halves: every lane ends with rows (2i, 2i+1) x 4 k = two 8-byte items of the [k-quad][slot(row)][2 dwords] image.
//       pl_slot keeps those stores (16 lanes: one row class c, all 16 octets) and the fragment reads (32 consecutive rows) conflict-free.
typedef const u32x4* __restrict__ gran_ptr;
static __device__ __forceinline__ int pl_slot(int row) {
    const int a = row >> 5, b = (row >> 3) & 3, c = row & 7;
    return a * 32 + ((4 * c + b + 4 * a) & 31);
}
struct PlaneRegs { u32x4 g[3]; };          // the three planes of one granule position
// one thread's granule position inside a 128 x 16 operand tile, in 16-byte units relative to the tile origin (r0, k0):
//   KC: row t / 2, k-octet t % 2;   !KC: k = 4 (t / 64) + (t % 64) / 16, row octet t % 16
template <bool KC>
static __device__ __forceinline__ long plane_thread_offset(long ldo) {
    const int t = threadIdx.x;
    if constexpr (KC) return ((long)(t >> 1) * ldo + (t & 1)) * 3;
    else return ((long)(4 * (t >> 6) + ((t & 63) >> 4)) * ldo + (t & 15)) * 3;
}
template <bool KC>
static __device__ __forceinline__ void plane_store(unsigned* S, const PlaneRegs& r, int pl0, int pl1) {
    const int t = threadIdx.x;
    if constexpr (KC) {
        const int row = t >> 1, kh = t & 1;
#pragma unroll
        for (int pl = pl0; pl < pl1; ++pl) *reinterpret_cast<u32x4*>(&S[pl * SP_PLANE + kh * SP_KH + row * 4]) = r.g[pl];
    } else {
        const int l = t & 63, w = t >> 6, i = l >> 4, row = 8 * (l & 15) + 2 * i;
        unsigned* dst0 = S + w * SP_KQ + pl_slot(row) * 2;
        unsigned* dst1 = S + w * SP_KQ + pl_slot(row + 1) * 2;
#pragma unroll
        for (int pl = pl0; pl < pl1; ++pl) {
            // d_j = rows (2j, 2j+1) of this lane's k;  after the two stages lane i holds T_j = rows (2i, 2i+1) at k = j of the quad
            auto s02 = __builtin_amdgcn_permlane32_swap(r.g[pl][0], r.g[pl][2], false, false);
            auto s13 = __builtin_amdgcn_permlane32_swap(r.g[pl][1], r.g[pl][3], false, false);
            auto t01 = __builtin_amdgcn_permlane16_swap(s02[0], s13[0], false, false);
            auto t23 = __builtin_amdgcn_permlane16_swap(s02[1], s13[1], false, false);
            const unsigned T0 = t01[0], T1 = t01[1], T2 = t23[0], T3 = t23[1];
            u32x2 lo = {__builtin_amdgcn_perm(T1, T0, 0x05040100u), __builtin_amdgcn_perm(T3, T2, 0x05040100u)};
            u32x2 hi = {__builtin_amdgcn_perm(T1, T0, 0x07060302u), __builtin_amdgcn_perm(T3, T2, 0x07060302u)};
            *reinterpret_cast<u32x2*>(dst0 + pl * SP_PLANE) = lo;
            *reinterpret_cast<u32x2*>(dst1 + pl * SP_PLANE) = hi;
        }
    }
}

// 16-byte agent-scope accesses (sc1: L2 write-through / L2-bypassing), the hand-off idiom of the persistent kernels (persist_common.h)
static __device__ __forceinline__ void sk_st4(float* p, const f32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 2" : : "v"(p), "v"(v) : "memory");
}
// eight 16-byte agent-scope loads in flight, one wait: p + q * 4 KB, q = 0..7 (two accumulator blocks of a parked tile)
static __device__ __forceinline__ void sk_ld4x8(const float* p, f32x4 (&v)[8]) {
    asm volatile("global_load_dwordx4 %0, %8, off sc1\n\t"
                 "global_load_dwordx4 %1, %9, off sc1\n\t"
                 "global_load_dwordx4 %2, %10, off sc1\n\t"
                 "global_load_dwordx4 %3, %11, off sc1\n\t"
                 "global_load_dwordx4 %4, %12, off sc1\n\t"
                 "global_load_dwordx4 %5, %13, off sc1\n\t"
                 "global_load_dwordx4 %6, %14, off sc1\n\t"
                 "global_load_dwordx4 %7, %15, off sc1\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
                 : "v"(p), "v"(p + 1024), "v"(p + 2048), "v"(p + 3072), "v"(p + 4096), "v"(p + 5120), "v"(p + 6144), "v"(p + 7168) : "memory");
}
constexpr unsigned SK_SPIN_LIMIT = 1u << 24;


// Load one BK x BM(BN) operand tile into registers (NLD float4 per thread).
// KC = true : element(r, k) at P[r*ld + k]   (row index r is the M or N index)
// KC = false: element(r, k) at P[k*ld + r]
template <bool KC>
__device__ __forceinline__ void load_tile(const float* __restrict__ P, long ld, int R, int K, int r0, int k0, int kend,
                                          bool vec_ok, f32x4 (&reg)[NLD]) {
    const int t = threadIdx.x;
    // Interior tiles (the common case) take a branch-free path: per-element guards compile to a branch + wait per
    // load, which serialises the tile fetch into one exposed memory latency per load.
    const bool interior = vec_ok && (r0 + BM <= R) && (k0 + BK <= kend);      // workgroup-uniform
    if (interior) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int idx = t + i * GEMM_THREADS;
            if (KC) {
                const int row = idx / KQ, kq = (idx % KQ) * 4;
                reg[i] = *reinterpret_cast<const f32x4*>(P + (long)(r0 + row) * ld + k0 + kq);
            } else {
                const int krow = idx >> 5, rq = (idx & 31) * 4;
                reg[i] = *reinterpret_cast<const f32x4*>(P + (long)(k0 + krow) * ld + r0 + rq);
            }
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int idx = t + i * GEMM_THREADS;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (KC) {
            const int row = idx / KQ, kq = (idx % KQ) * 4;
            const int r = r0 + row, k = k0 + kq;
            if (r < R) {
                const float* p = P + (long)r * ld + k;
                if (vec_ok && k + 3 < kend) {
                    v = *reinterpret_cast<const f32x4*>(p);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) if (k + j < kend) v[j] = p[j];
                }
            }
        } else {
            const int krow = idx >> 5, rq = (idx & 31) * 4;
            const int k = k0 + krow, r = r0 + rq;
            if (k < kend) {
                const float* p = P + (long)k * ld + r;
                if (vec_ok && r + 3 < R) {
                    v = *reinterpret_cast<const f32x4*>(p);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) if (r + j < R) v[j] = p[j];
                }
            }
        }
        reg[i] = v;
    }
}

template <bool KC>
__device__ __forceinline__ void store_tile(float (*S)[BM + PAD], const f32x4 (&reg)[NLD]) {
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int idx = t + i * GEMM_THREADS;
        if (KC) {
            const int row = idx / KQ, kq = (idx % KQ) * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) S[kq + j][row] = reg[i][j];
        } else {
            const int krow = idx >> 5, rq = (idx & 31) * 4;
            *reinterpret_cast<f32x4*>(&S[krow][rq]) = reg[i];
        }
    }
}

// One output tile over the k-iterations [it0, it1) (BK each): main loop + epilogue.  `atomic`: this segment is one of several
// contributors to the tile (split-K / stream-K): accumulate with atomics onto a C that starts from zero (or from the
// value to accumulate onto); the contributor that owns k-iteration 0 adds the biases.
// MODE: 0 fp32 operands on the fp32 matrix pipe, 1 fp32 operands split in the kernel, 2 pre-split operands (P8x3 granules: A / B / A2 / B2
// point at granule buffers, lda / ldb / sA / sB stay ELEMENT counts of the logical fp32 matrices)
template <bool A_KC, bool B_KC, int MODE>
__device__ __forceinline__ void gemm_segment(const GemmParams& p, float (*As)[BK][BM + PAD], float (*Bs)[BK][BN + PAD], unsigned* sp, int bz,
                                             int m0, int n0, int kbeg, int kend, const SegRole role) {
    const bool atomic = role.kind == SEG_ATOMIC, add_bias = role.add_bias;
    // (MODE 2: A / B are granule buffers, 6 bytes per logical element)
    const float* A = MODE == 2 ? reinterpret_cast<const float*>(reinterpret_cast<const char*>(p.A) + (long)bz * p.sA * 6) : p.A + (long)bz * p.sA;
    const float* B = MODE == 2 ? reinterpret_cast<const float*>(reinterpret_cast<const char*>(p.B) + (long)bz * p.sB * 6) : p.B + (long)bz * p.sB;
    float* C = p.C + (long)bz * p.sC;
    const int ntiles = (kend - kbeg + BK - 1) / BK;
    // k-tile loader with the optional second K source (a k-tile never straddles K1)
    auto load_ab = [&](int k0, f32x4 (&ra_)[NLD], f32x4 (&rb_)[NLD]) {
        if (p.A2 != nullptr && k0 >= p.K1) {
            load_tile<A_KC>(p.A2, p.lda, p.M, p.K - p.K1, m0, k0 - p.K1, kend - p.K1, p.a_vec, ra_);
            load_tile<B_KC>(p.B2, p.ldb, p.N, p.K - p.K1, n0, k0 - p.K1, kend - p.K1, p.b_vec, rb_);
        } else {
            const int ke = (p.A2 != nullptr) ? min(kend, p.K1) : kend;
            load_tile<A_KC>(A, p.lda, p.M, p.K, m0, k0, ke, p.a_vec, ra_);
            load_tile<B_KC>(B, p.ldb, p.N, p.K, n0, k0, ke, p.b_vec, rb_);
        }
    };

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    const int lr = lane & 31, lk = lane >> 5;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // Interior segments (whole tile inside M x N, vector-aligned operands, k range a multiple of BK — every large GEMM of the
    // training step) take a loop without per-element guards: per-thread element offsets are computed once, a k-tile costs
    // four 16-byte loads, and the only branch is the loop itself.  Everything else (edge tiles, odd K, unaligned operands)
    // goes through the guarded loader below.
    const bool fast = p.a_vec && p.b_vec && (m0 + BM <= p.M) && (n0 + BN <= p.N) && ((kend - kbeg) % BK == 0) && ntiles > 0;
    constexpr bool SPLIT = MODE == 1;
    if constexpr (MODE == 2) {
      if (ntiles > 0) {
        // Pre-split operands: the pipeline of the split-operand loop below (three LDS buffers, one fragment set, one barrier per k-tile, the
        // same six MFMA groups and early fragment reloads) with the split arithmetic gone: per k-tile and thread three 16-byte granule loads
        // per operand, stored as they are (K-contiguous) or after the 4 x 4 lane transpose (row-contiguous).
        const int t = threadIdx.x;
        const long ldoA = p.lda >> 3, ldoB = p.ldb >> 3;
        // granule offsets (16-byte units) of this thread at k = 0; a k-tile advances by kA / kB
        const long oA = (A_KC ? (long)m0 * ldoA * 3 : (long)(m0 >> 3) * 3) + plane_thread_offset<A_KC>(ldoA);
        const long oB = (B_KC ? (long)n0 * ldoB * 3 : (long)(n0 >> 3) * 3) + plane_thread_offset<B_KC>(ldoB);
        const long kA = A_KC ? 3 : ldoA * 24, kB = B_KC ? 3 : ldoB * 24;          // per k-OCTET (KC) / per 8 k rows (!KC): granules
        // guards of this thread's granule position (whole octets: M, N, K are multiples of 8 wherever they index granules)
        const int rowA = A_KC ? m0 + (t >> 1) : m0 + 8 * (t & 15), rowB = B_KC ? n0 + (t >> 1) : n0 + 8 * (t & 15);
        const int kofA = A_KC ? 8 * (t & 1) : 4 * (t >> 6) + ((t & 63) >> 4), kofB = B_KC ? 8 * (t & 1) : 4 * (t >> 6) + ((t & 63) >> 4);
        const bool rokA = rowA < p.M, rokB = rowB < p.N;
        auto gload_any = [&](auto FASTC, int k0, PlaneRegs& ra_, PlaneRegs& rb_) {
            const bool second = p.A2 != nullptr && k0 >= p.K1;          // wave-uniform
            gran_ptr Ab = reinterpret_cast<gran_ptr>(second ? p.A2 : A);
            gran_ptr Bb = reinterpret_cast<gran_ptr>(second ? p.B2 : B);
            const long kk = second ? k0 - p.K1 : k0;
            gran_ptr pa = Ab + oA + (kk >> 3) * kA, pb = Bb + oB + (kk >> 3) * kB;
            if constexpr (decltype(FASTC)::value) {
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) ra_.g[pl] = pa[pl];
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) rb_.g[pl] = pb[pl];
            } else {
                const int ke = second ? kend - p.K1 : (p.A2 != nullptr ? min(kend, p.K1) : kend);
                const bool va = rokA && (int)kk + kofA < ke, vb = rokB && (int)kk + kofB < ke;
                const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) ra_.g[pl] = va ? pa[pl] : z;
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) rb_.g[pl] = vb ? pb[pl] : z;
            }
        };
        int fa[2], fb[2], fa2[2], fb2[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            fa[i] = A_KC ? lk * SP_KH + (wm + i * 32 + lr) * 4 : (2 * lk) * SP_KQ + pl_slot(wm + i * 32 + lr) * 2;
            fb[i] = B_KC ? lk * SP_KH + (wn + i * 32 + lr) * 4 : (2 * lk) * SP_KQ + pl_slot(wn + i * 32 + lr) * 2;
            fa2[i] = fa[i] + SP_KQ; fb2[i] = fb[i] + SP_KQ;
            if constexpr (!A_KC) asm volatile("" : "+v"(fa2[i]));
            if constexpr (!B_KC) asm volatile("" : "+v"(fb2[i]));
        }
        SplitFrag f;
        auto rd_a = [&](int buf, int pl) {
            const unsigned* pa = sp + buf * SP_BUF + pl * SP_PLANE;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if constexpr (A_KC) {
                    f.a[pl][i] = *reinterpret_cast<const u32x4*>(pa + fa[i]);
                } else {
                    const u32x2 lo = *reinterpret_cast<const u32x2*>(pa + fa[i]), hi = *reinterpret_cast<const u32x2*>(pa + fa2[i]);
                    f.a[pl][i] = u32x4{lo[0], lo[1], hi[0], hi[1]};
                }
            }
        };
        auto rd_b = [&](int buf, int pl) {
            const unsigned* pb = sp + buf * SP_BUF + SP_OPER + pl * SP_PLANE;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if constexpr (B_KC) {
                    f.b[pl][i] = *reinterpret_cast<const u32x4*>(pb + fb[i]);
                } else {
                    const u32x2 lo = *reinterpret_cast<const u32x2*>(pb + fb[i]), hi = *reinterpret_cast<const u32x2*>(pb + fb2[i]);
                    f.b[pl][i] = u32x4{lo[0], lo[1], hi[0], hi[1]};
                }
            }
        };
        auto grp = [&](int pa, int pb) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f.a[pa][i]),
                                                                         __builtin_bit_cast(bf16x8, f.b[pb][j]), acc[i][j], 0, 0, 0);
        };
        auto mainloop = [&](auto FASTC) {
        auto gload = [&](int k0, PlaneRegs& ra_, PlaneRegs& rb_) { gload_any(FASTC, k0, ra_, rb_); };
        PlaneRegs ra[3], rb[3];
        auto store_all = [&](int buf, const PlaneRegs& ra_, const PlaneRegs& rb_) {
            unsigned* sa = sp + buf * SP_BUF;
            plane_store<A_KC>(sa, ra_, 0, 3); plane_store<B_KC>(sa + SP_OPER, rb_, 0, 3);
        };
        gload(kbeg, ra[0], rb[0]);
        if (ntiles > 1) gload(kbeg + BK, ra[1], rb[1]);
        if (ntiles > 2) gload(kbeg + 2 * BK, ra[2], rb[2]);
        store_all(0, ra[0], rb[0]);
        if (ntiles > 3) gload(kbeg + 3 * BK, ra[0], rb[0]);
        if (ntiles > 1) store_all(1, ra[1], rb[1]);
        __syncthreads();
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) { rd_a(0, pl); rd_b(0, pl); }
        // same buffer / register-set rotation as the split-operand loop (see there)
        auto step = [&](auto RC, auto FULLC, int kt) {
            constexpr int R = decltype(RC)::value, R1 = (R + 1) % 3, R2 = (R + 2) % 3;
            constexpr bool FULL = decltype(FULLC)::value;
            const bool nxt = !(LAS_PLANES_ABL & 4) && (FULL || kt + 1 < ntiles), st = !(LAS_PLANES_ABL & 2) && (FULL || kt + 2 < ntiles);
            unsigned* sa = sp + R2 * SP_BUF;
            unsigned* sb = sa + SP_OPER;
            if (!(LAS_PLANES_ABL & 1) && (FULL || kt + 4 < ntiles)) gload(kbeg + (kt + 4) * BK, ra[R1], rb[R1]);
            grp(1, 1);
            if (st) plane_store<A_KC>(sa, ra[R2], 0, 2);
            __builtin_amdgcn_sched_barrier(0);
            grp(1, 0);
            if (nxt) rd_a(R1, 1);
            if (st) plane_store<A_KC>(sa, ra[R2], 2, 3);
            __builtin_amdgcn_sched_barrier(0);
            grp(0, 1);
            if (nxt) rd_b(R1, 1);
            if (st) plane_store<B_KC>(sb, rb[R2], 0, 2);
            __builtin_amdgcn_sched_barrier(0);
            grp(2, 0);
            if (nxt) rd_a(R1, 2);
            if (st) plane_store<B_KC>(sb, rb[R2], 2, 3);
            __builtin_amdgcn_sched_barrier(0);
            grp(0, 2);
            if (nxt) rd_b(R1, 2);
            __builtin_amdgcn_sched_barrier(0);
            grp(0, 0);
            if (!(LAS_PLANES_ABL & 8)) __syncthreads();
            if (nxt) { rd_a(R1, 0); rd_b(R1, 0); }
            __builtin_amdgcn_sched_barrier(0);
        };
        int kt = 0;
        for (; kt + 6 < ntiles; kt += 3) {       // all three steps unguarded
            step(std::integral_constant<int, 0>{}, std::true_type{}, kt);
            step(std::integral_constant<int, 1>{}, std::true_type{}, kt + 1);
            step(std::integral_constant<int, 2>{}, std::true_type{}, kt + 2);
        }
        for (; kt < ntiles; kt += 3) {
            step(std::integral_constant<int, 0>{}, std::false_type{}, kt);
            if (kt + 1 < ntiles) step(std::integral_constant<int, 1>{}, std::false_type{}, kt + 1);
            if (kt + 2 < ntiles) step(std::integral_constant<int, 2>{}, std::false_type{}, kt + 2);
        }
              };
        if (fast) mainloop(std::true_type{}); else mainloop(std::false_type{});
      }
    } else if constexpr (SPLIT) {
      if (ntiles > 0) {
        // Split-operand main loop (see the comment at split_pair).  Per k-tile of 16 and wave: 24 MFMAs in six groups of four (one
        // group = one pair of planes on the 2x2 accumulators).  Three LDS buffers, ONE fragment register set, one barrier per tile:
        // while tile kt is multiplied, its planes are replaced by those of tile kt+1 as soon as their last group has been issued
        // (group order x2*y2, x2*y1, x1*y2, x3*y1, x1*y3, x1*y1 frees x2, y2, x3, y3 early; x1 / y1 are reloaded after the barrier,
        // under the first group of the next tile, which does not need them), tile kt+2 is split in registers and stored to the
        // third buffer in four slices between the groups, and the global loads of tile kt+3 are issued.
        const int t = threadIdx.x;
        long oA[NLD], oB[NLD];
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int idx = t + i * GEMM_THREADS;
            oA[i] = A_KC ? (long)(m0 + (idx >> 2)) * p.lda + (idx & 3) * 4 : (long)(2 * (t >> 5) + i) * p.lda + m0 + (t & 31) * 4;
            oB[i] = B_KC ? (long)(n0 + (idx >> 2)) * p.ldb + (idx & 3) * 4 : (long)(2 * (t >> 5) + i) * p.ldb + n0 + (t & 31) * 4;
        }
        const long strideA = A_KC ? 1 : p.lda, strideB = B_KC ? 1 : p.ldb;
        auto gload_any = [&](auto FASTC, int k0, f32x4 (&ra_)[NLD], f32x4 (&rb_)[NLD]) {
            const bool second = p.A2 != nullptr && k0 >= p.K1;          // wave-uniform
            const float* Ab = second ? p.A2 : A;
            const float* Bb = second ? p.B2 : B;
            const long kk = second ? k0 - p.K1 : k0;
            if constexpr (decltype(FASTC)::value) {
#pragma unroll
                for (int i = 0; i < NLD; ++i) ra_[i] = *reinterpret_cast<const f32x4*>(Ab + oA[i] + kk * strideA);
#pragma unroll
                for (int i = 0; i < NLD; ++i) rb_[i] = *reinterpret_cast<const f32x4*>(Bb + oB[i] + kk * strideB);
            } else {
                const int ke = second ? kend - p.K1 : (p.A2 != nullptr ? min(kend, p.K1) : kend);
                split_load_guarded<A_KC>(Ab, p.lda, p.M, m0, (int)kk, ke, p.a_vec, ra_);
                split_load_guarded<B_KC>(Bb, p.ldb, p.N, n0, (int)kk, ke, p.b_vec, rb_);
            }
        };
        // fragment offsets (dwords) of this lane inside an operand plane.  The second k-quad of a row-contiguous operand gets an
        // offset the compiler cannot relate to the first, or it would merge the two 8-byte reads into one ds_read2_b64 (8 LDS
        // cycles instead of 2 + 2)
        int fa[2], fb[2], fa2[2], fb2[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            fa[i] = A_KC ? lk * SP_KH + (wm + i * 32 + lr) * 4 : (2 * lk) * SP_KQ + sp_slot(wm + i * 32 + lr) * 2;
            fb[i] = B_KC ? lk * SP_KH + (wn + i * 32 + lr) * 4 : (2 * lk) * SP_KQ + sp_slot(wn + i * 32 + lr) * 2;
            fa2[i] = fa[i] + SP_KQ; fb2[i] = fb[i] + SP_KQ;
            if constexpr (!A_KC) asm volatile("" : "+v"(fa2[i]));
            if constexpr (!B_KC) asm volatile("" : "+v"(fb2[i]));
        }
        SplitFrag f;
        auto rd_a = [&](int buf, int pl) {
            const unsigned* pa = sp + buf * SP_BUF + pl * SP_PLANE;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if constexpr (A_KC) {
                    f.a[pl][i] = *reinterpret_cast<const u32x4*>(pa + fa[i]);
                } else {
                    const u32x2 lo = *reinterpret_cast<const u32x2*>(pa + fa[i]), hi = *reinterpret_cast<const u32x2*>(pa + fa2[i]);
                    f.a[pl][i] = u32x4{lo[0], lo[1], hi[0], hi[1]};
                }
            }
        };
        auto rd_b = [&](int buf, int pl) {
            const unsigned* pb = sp + buf * SP_BUF + SP_OPER + pl * SP_PLANE;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if constexpr (B_KC) {
                    f.b[pl][i] = *reinterpret_cast<const u32x4*>(pb + fb[i]);
                } else {
                    const u32x2 lo = *reinterpret_cast<const u32x2*>(pb + fb[i]), hi = *reinterpret_cast<const u32x2*>(pb + fb2[i]);
                    f.b[pl][i] = u32x4{lo[0], lo[1], hi[0], hi[1]};
                }
            }
        };
        auto grp = [&](int pa, int pb) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f.a[pa][i]),
                                                                         __builtin_bit_cast(bf16x8, f.b[pb][j]), acc[i][j], 0, 0, 0);
        };
        // interior segments and guarded ones (edge tiles, odd K, unaligned operands) run the same loop with their own loader
        auto mainloop = [&](auto FASTC) {
        auto gload = [&](int k0, f32x4 (&ra_)[NLD], f32x4 (&rb_)[NLD]) { gload_any(FASTC, k0, ra_, rb_); };
        // register sets: tile j travels in set j % 3; its loads are issued two iterations before its split/store
        f32x4 ra[3][NLD], rb[3][NLD];
        auto sstore_all = [&](int buf, const f32x4 (&ra_)[NLD], const f32x4 (&rb_)[NLD]) {
            unsigned* sa = sp + buf * SP_BUF;
            split_store_half<A_KC, 0>(sa, ra_); split_store_half<A_KC, 1>(sa, ra_);
            split_store_half<B_KC, 0>(sa + SP_OPER, rb_); split_store_half<B_KC, 1>(sa + SP_OPER, rb_);
        };
        gload(kbeg, ra[0], rb[0]);
        if (ntiles > 1) gload(kbeg + BK, ra[1], rb[1]);
        if (ntiles > 2) gload(kbeg + 2 * BK, ra[2], rb[2]);
        sstore_all(0, ra[0], rb[0]);
        if (ntiles > 3) gload(kbeg + 3 * BK, ra[0], rb[0]);
        if (ntiles > 1) sstore_all(1, ra[1], rb[1]);
        __syncthreads();
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) { rd_a(0, pl); rd_b(0, pl); }
        // iteration kt with R = kt % 3: tile kt is in the fragment registers and LDS buffer R, tile kt+1 in buffer R+1, tile kt+2 in
        // register set R+2 (on its way to buffer R+2), tile kt+3 in flight into set R, tile kt+4 is requested into set R+1.
        // FULL: kt + 4 < ntiles, nothing to guard.
        auto step = [&](auto RC, auto FULLC, int kt) {
            constexpr int R = decltype(RC)::value, R1 = (R + 1) % 3, R2 = (R + 2) % 3;
            constexpr bool FULL = decltype(FULLC)::value;
            const bool nxt = !(LAS_SPLIT_ABL & 4) && (FULL || kt + 1 < ntiles), st = !(LAS_SPLIT_ABL & 2) && (FULL || kt + 2 < ntiles);
            unsigned* sa = sp + R2 * SP_BUF;
            unsigned* sb = sa + SP_OPER;
            if (!(LAS_SPLIT_ABL & 1) && (FULL || kt + 4 < ntiles)) gload(kbeg + (kt + 4) * BK, ra[R1], rb[R1]);
            grp(1, 1);
            if (st) split_store_half<A_KC, 0>(sa, ra[R2]);
            __builtin_amdgcn_sched_barrier(0);
            grp(1, 0);
            if (nxt) rd_a(R1, 1);
            if (st) split_store_half<A_KC, 1>(sa, ra[R2]);
            __builtin_amdgcn_sched_barrier(0);
            grp(0, 1);
            if (nxt) rd_b(R1, 1);
            if (st) split_store_half<B_KC, 0>(sb, rb[R2]);
            __builtin_amdgcn_sched_barrier(0);
            grp(2, 0);
            if (nxt) rd_a(R1, 2);
            if (st) split_store_half<B_KC, 1>(sb, rb[R2]);
            __builtin_amdgcn_sched_barrier(0);
            grp(0, 2);
            if (nxt) rd_b(R1, 2);
            __builtin_amdgcn_sched_barrier(0);
            grp(0, 0);
            __syncthreads();
            if (nxt) { rd_a(R1, 0); rd_b(R1, 0); }
            __builtin_amdgcn_sched_barrier(0);
        };
        int kt = 0;
        for (; kt + 6 < ntiles; kt += 3) {       // all three steps unguarded
            step(std::integral_constant<int, 0>{}, std::true_type{}, kt);
            step(std::integral_constant<int, 1>{}, std::true_type{}, kt + 1);
            step(std::integral_constant<int, 2>{}, std::true_type{}, kt + 2);
        }
        for (; kt < ntiles; kt += 3) {
            step(std::integral_constant<int, 0>{}, std::false_type{}, kt);
            if (kt + 1 < ntiles) step(std::integral_constant<int, 1>{}, std::false_type{}, kt + 1);
            if (kt + 2 < ntiles) step(std::integral_constant<int, 2>{}, std::false_type{}, kt + 2);
        }
              };
        if (fast) mainloop(std::true_type{}); else mainloop(std::false_type{});
      }
    } else if (fast) {
        long offA[NLD], offB[NLD];
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int idx = threadIdx.x + i * GEMM_THREADS;
            offA[i] = A_KC ? (long)(m0 + idx / KQ) * p.lda + (idx % KQ) * 4 : (long)(idx >> 5) * p.lda + m0 + (idx & 31) * 4;
            offB[i] = B_KC ? (long)(n0 + idx / KQ) * p.ldb + (idx % KQ) * 4 : (long)(idx >> 5) * p.ldb + n0 + (idx & 31) * 4;
        }
        const long strideA = A_KC ? 1 : p.lda, strideB = B_KC ? 1 : p.ldb;
        // M/N-contiguous operands (every operand of a weight-gradient GEMM) go global -> LDS directly (global_load_lds_dwordx4:
        // 1 KB per wave-instruction, destination = wave-uniform base + lane*16): a 128-float k-row is exactly what 32 lanes write,
        // so the K-major tile image is lane-linear with row stride 128 (no staging VGPRs, no ds_write pass, nothing to wait for
        // before the LDS stores).  K-contiguous operands keep the register-staged transposing store.
        constexpr bool A_DMA = !A_KC && LAS_GEMM_DMA, B_DMA = !B_KC && LAS_GEMM_DMA;
        constexpr int LDA_S = A_DMA ? BM : BM + PAD, LDB_S = B_DMA ? BN : BN + PAD;     // LDS row strides (floats)
        auto load_fast = [&](int k0, int buf, f32x4 (&ra_)[NLD], f32x4 (&rb_)[NLD]) {
            const bool second = p.A2 != nullptr && k0 >= p.K1;          // wave-uniform
            const float* Ab = second ? p.A2 : A;
            const float* Bb = second ? p.B2 : B;
            const long kk = second ? k0 - p.K1 : k0;
            const int wv = threadIdx.x >> 6;
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                if constexpr (A_DMA) {
                    __builtin_amdgcn_global_load_lds(Ab + offA[i] + kk * strideA,
                                                     to_lds(&As[buf][0][0] + (wv * 64 + i * GEMM_THREADS) * 4),
                                                     16, 0, 0);
                } else {
                    ra_[i] = *reinterpret_cast<const f32x4*>(Ab + offA[i] + kk * strideA);
                }
            }
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                if constexpr (B_DMA) {
                    __builtin_amdgcn_global_load_lds(Bb + offB[i] + kk * strideB,
                                                     to_lds(&Bs[buf][0][0] + (wv * 64 + i * GEMM_THREADS) * 4),
                                                     16, 0, 0);
                } else {
                    rb_[i] = *reinterpret_cast<const f32x4*>(Bb + offB[i] + kk * strideB);
                }
            }
        };
        auto stash = [&](int buf, const f32x4 (&ra_)[NLD], const f32x4 (&rb_)[NLD]) {
            if constexpr (!A_DMA) store_tile<A_KC>(As[buf], ra_);
            if constexpr (!B_DMA) store_tile<B_KC>(Bs[buf], rb_);
            if constexpr (A_DMA || B_DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the DMA pieces of this wave have landed
        };
        f32x4 ra[NLD], rb[NLD];
        load_fast(kbeg, 0, ra, rb);
        stash(0, ra, rb);
        __syncthreads();
        // Software pipeline across k-steps AND across the tile boundary, pinned with sched_barrier (left alone, hipcc sinks the
        // fragment reads of k-step kk+1 behind the MFMAs of kk and the matrix pipe idles for an LDS latency per k-step):
        //   k-step kk:  [ds_read fragments kk+1] [4 MFMAs of kk]
        //   kk = NK-2:  ... then the next tile's registers -> LDS (the stores issue while the MFMAs execute)
        //   kk = NK-1:  barrier FIRST, then [ds_read fragments 0 of the next tile] [4 MFMAs of NK-1]:
        //               the barrier wait hides behind the MFMAs of NK-2, the first LDS latency behind those of NK-1.
        constexpr int NK = BK / 2;
        float a[2], b[2], an[2], bn[2];
        auto frag = [&](int buf, int kk, float (&fa)[2], float (&fb)[2]) {
            const float* Ap = &As[buf][0][0] + (kk * 2 + lk) * LDA_S + wm + lr;
            const float* Bp = &Bs[buf][0][0] + (kk * 2 + lk) * LDB_S + wn + lr;
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[i] = Ap[i * 32];
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[j] = Bp[j * 32];
        };
        auto mma = [&]() {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        };
        frag(0, 0, a, b);
        for (int kt = 0; kt < ntiles; ++kt) {
            const int cur = kt & 1;
            const bool more = kt + 1 < ntiles;
            if (more) load_fast(kbeg + (kt + 1) * BK, cur ^ 1, ra, rb);
#pragma unroll
            for (int kk = 0; kk < NK - 1; ++kk) {
                frag(cur, kk + 1, an, bn);
                __builtin_amdgcn_sched_barrier(0);
                mma();
                __builtin_amdgcn_sched_barrier(0);
                a[0] = an[0]; a[1] = an[1]; b[0] = bn[0]; b[1] = bn[1];
            }
            if (more) stash(cur ^ 1, ra, rb);
            __syncthreads();
            if (more) frag(cur ^ 1, 0, an, bn);
            __builtin_amdgcn_sched_barrier(0);
            mma();                                   // k-step NK-1 of tile kt (its fragments were read before the barrier)
            __builtin_amdgcn_sched_barrier(0);
            a[0] = an[0]; a[1] = an[1]; b[0] = bn[0]; b[1] = bn[1];
        }
        __syncthreads();                             // the segment's last reads precede the next segment's first LDS writes
    } else {
    // Register-staged prefetch PF k-tiles ahead (the loads of tile kt+PF are issued before tile kt is multiplied): one tile
    // of MFMA work (2048 cycles per wave) does not cover an L2-miss round trip under load, and with two resident workgroups
    // per CU there are 256 VGPRs per lane to spend.  LDS stays double-buffered (tile kt+1 is written while kt is read).
    f32x4 ra[PF][NLD], rb[PF][NLD];
#pragma unroll
    for (int i = 0; i < PF; ++i)
        if (i < ntiles) load_ab(kbeg + i * BK, ra[i], rb[i]);
    if (ntiles > 0) {
        store_tile<A_KC>(As[0], ra[0]);
        store_tile<B_KC>(Bs[0], rb[0]);
    }
    __syncthreads();

    for (int kt0 = 0; kt0 < ntiles; kt0 += PF) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int kt = kt0 + u;
            if (kt >= ntiles) break;
            const int cur = kt & 1;
            if (kt + PF < ntiles) load_ab(kbeg + (kt + PF) * BK, ra[u], rb[u]);      // stage u held tile kt: already in LDS
            // fragments of k-step kk+1 are read from LDS before the MFMAs of k-step kk are issued (software pipeline):
            // a single wave per SIMD then keeps the matrix pipe busy instead of idling for an LDS latency per k-step
            float a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = As[cur][lk][wm + i * 32 + lr];
#pragma unroll
            for (int j = 0; j < 2; ++j) b[j] = Bs[cur][lk][wn + j * 32 + lr];
#pragma unroll
            for (int kk = 0; kk < BK / 2; ++kk) {
                float an[2] = {0.f, 0.f}, bn[2] = {0.f, 0.f};
                if (kk + 1 < BK / 2) {
#pragma unroll
                    for (int i = 0; i < 2; ++i) an[i] = As[cur][(kk + 1) * 2 + lk][wm + i * 32 + lr];
#pragma unroll
                    for (int j = 0; j < 2; ++j) bn[j] = Bs[cur][(kk + 1) * 2 + lk][wn + j * 32 + lr];
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
                a[0] = an[0]; a[1] = an[1]; b[0] = bn[0]; b[1] = bn[1];
            }
            if (kt + 1 < ntiles) {
                store_tile<A_KC>(As[cur ^ 1], ra[(u + 1) % PF]);
                store_tile<B_KC>(Bs[cur ^ 1], rb[(u + 1) % PF]);
            }
            __syncthreads();
        }
    }

    }
    // Stream-K fix-up (workgroup-uniform role).  Parked image: [i][j][quad q of the 16 accumulator registers][thread][4 floats] — every
    // thread of every workgroup holds the same tile positions, so the image needs no index arithmetic and a wave moves whole lines.
    // A 32x32 block of the wave's 64x64 sub-tile that lies wholly outside M x N is neither parked nor fetched (wave-uniform; the same
    // predicate on both sides): the vocabulary-sized problems (N = 30) move a quarter of the tile.
    auto blk_live = [&](int i, int j) { return m0 + wm + i * 32 < p.M && n0 + wn + j * 32 < p.N; };
    if (role.kind == SEG_PART) {
        float* dst = p.sk_part + (size_t)role.slot * (BM * BN) + threadIdx.x * 4;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (!blk_live(i, j)) continue;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 v = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
                    sk_st4(dst + ((i * 2 + j) * 4 + q) * (GEMM_THREADS * 4), v);
                }
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's part has reached the memory side
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(p.sk_flag + role.slot, p.sk_id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    for (int c = role.c0; c < role.c1; ++c) {
        if (threadIdx.x == 0) {
            unsigned spins = 0;
            while (__hip_atomic_load(p.sk_flag + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != p.sk_id) {
                if (++spins > SK_SPIN_LIMIT) {      // a contributor that is not resident: reported by the host, and the step's update skips itself
                    atomicExch(p.sk_err, 1u);
                    if (p.call_err) atomicExch(p.call_err, 0xDEAD0005u);
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
            }
        }
        __syncthreads();
        const float* src = p.sk_part + (size_t)c * (BM * BN) + threadIdx.x * 4;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const bool l0 = blk_live(i, 0), l1 = blk_live(i, 1);
            if (!l0 && !l1) continue;
            f32x4 v[8];
            sk_ld4x8(src + (i * 2 * 4) * (GEMM_THREADS * 4), v);         // (a dead block's words are stale data: fetched, not used)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (!(j ? l1 : l0)) continue;
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[i][j][4 * q + e] += v[j * 4 + q][e];
            }
        }
        __syncthreads();                                           // every wave has its part: the slot may be reused by a later launch
        if (threadIdx.x == 0) __hip_atomic_store(p.sk_flag + c, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // Epilogue.  C/D map of 32x32 MFMA: col(n) = lane&31, row(m) = (r&3) + 8*(r>>2) + 4*(lane>>5).
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + wn + j * 32 + lr;
        if (n >= p.N) continue;
        float bsum = 0.f;
        if (add_bias) {
            if (p.bias0) bsum += p.bias0[(long)bz * p.sBias0 + n];
            if (p.bias1) bsum += p.bias1[(long)bz * p.sBias1 + n];
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                if (m >= p.M) continue;
                float v = acc[i][j][r] + bsum;
                float* c = C + (long)m * p.ldc + n;
                if (atomic) {
                    atomicAdd(c, v);
                } else {
                    if (p.accumulate) v += *c;
                    if (p.relu) v = fmaxf(v, 0.f);
                    *c = v;
                }
            }
        }
    }
}

// Two schedules share the segment code:
//  * classic (p.persistent == 0): grid (tiles, 1, batch*splitk); every workgroup owns one tile and one k-slice.
//  * persistent (p.persistent == 1): grid = W resident workgroups.  The first p.dp_tiles tiles are done whole, round-robin
//    (data parallel, plain stores); the k-iterations of the remaining tiles — the tail that would leave most CUs idle for
//    a whole tile time — are cut into W equal runs (stream-K): a run may end one tile and start the next, partial
//    tiles are combined with atomics on a C window zeroed beforehand.
// (The parameter block is passed by value: ~100 of its SGPRs spill into VGPR lanes, which measured FASTER than fetching
// the fields from the kernel-argument segment inside the loop.)
// LDS of one workgroup: the fp32 kernels hold two K-major fp32 tiles per operand (static); the split-operand kernels take
// SP_SMEM_BYTES of dynamic LDS (three buffers of bf16 planes) and carve the fp32 tiles of their guarded path out of it.
template <int MODE> struct GemmSmem;
template <> struct GemmSmem<0> {
    float (*As)[BK][BM + PAD]; float (*Bs)[BK][BN + PAD]; unsigned* sp;
    __device__ __forceinline__ GemmSmem() {
        __shared__ __attribute__((aligned(16))) float as_[2][BK][BM + PAD];
        __shared__ __attribute__((aligned(16))) float bs_[2][BK][BN + PAD];
        As = as_; Bs = bs_; sp = nullptr;
    }
};
template <> struct GemmSmem<1> {
    float (*As)[BK][BM + PAD]; float (*Bs)[BK][BN + PAD]; unsigned* sp;
    __device__ __forceinline__ GemmSmem() {
        extern __shared__ __attribute__((aligned(16))) unsigned dyn_[];
        static_assert(SP_SMEM_BYTES >= (int)sizeof(float) * 2 * 2 * BK * (BM + PAD), "guarded path fits the dynamic LDS");
        sp = dyn_;
        As = reinterpret_cast<float (*)[BK][BM + PAD]>(dyn_);
        Bs = reinterpret_cast<float (*)[BK][BN + PAD]>(dyn_ + 2 * BK * (BM + PAD));
    }
};

template <> struct GemmSmem<2> : GemmSmem<1> {};
template <bool A_KC, bool B_KC, int MODE>
__global__ __launch_bounds__(GEMM_THREADS, 2) void gemm_f32_kernel(GemmParams p) {
    GemmSmem<MODE> sm;
    float (*As)[BK][BM + PAD] = sm.As; float (*Bs)[BK][BN + PAD] = sm.Bs; unsigned* sp = sm.sp;
    if (!p.persistent) {
        const int bz = blockIdx.z / p.splitk, kz = blockIdx.z % p.splitk;
        // XCD-aware tile order: workgroups are dispatched round-robin over the 8 XCDs (private L2 each); remap so that
        // each XCD works on a contiguous run of tiles (the N-tiles of one M-tile share the A panel in that XCD's L2).
        int tile = blockIdx.x;
        if (p.swz) tile = (tile & 7) * (gridDim.x >> 3) + (tile >> 3);
        const int kbeg = kz * p.kper;
        gemm_segment<A_KC, B_KC, MODE>(p, As, Bs, sp, bz, (tile / p.gx) * BM, (tile % p.gx) * BN, kbeg, min(p.K, kbeg + p.kper),
                                        p.atomic ? seg_atomic(kz == 0) : seg_store(kz == 0));
        return;
    }
    // workgroup -> slot: blocks are dispatched round-robin over the 8 XCDs (block b on XCD b % 8, observed; speed only), so
    // slot = (b % 8) * W/8 + b / 8 gives every XCD a CONTIGUOUS run of tiles / k-runs: the N-tiles of one M-panel share the
    // A panel through that XCD's L2 instead of every XCD streaming all of A
    const int W = gridDim.x;
    const int w = (W % 8 == 0 && p.xcd_swz) ? (int)(blockIdx.x & 7) * (W >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    const int per_batch = p.gx * p.gy;
    for (int tile = w; tile < p.dp_tiles; tile += W) {
        const int bz = tile / per_batch, t = tile % per_batch;
        gemm_segment<A_KC, B_KC, MODE>(p, As, Bs, sp, bz, (t / p.gx) * BM, (t % p.gx) * BN, 0, p.K, seg_store());
    }
    long i0 = (long)w * p.sk_per, i1 = min(i0 + p.sk_per, p.sk_iters);
    while (i0 < i1) {
        const int tile = p.dp_tiles + (int)(i0 / p.kt);
        const int it0 = (int)(i0 % p.kt), it1 = (int)min((long)p.kt, it0 + (i1 - i0));
        const int bz = tile / per_batch, t = tile % per_batch;
        const bool whole = it0 == 0 && it1 == p.kt;
        SegRole role;
        if (p.sk_part == nullptr) {
            role = (!whole || p.sk_atomic_whole) ? seg_atomic(it0 == 0) : seg_store();
        } else if (it0 > 0) {
            role = SegRole{SEG_PART, w, 0, 0, false};
        } else {
            // owner of the tile: the slots behind this one whose runs start inside the tile's k-range hold the rest of the sum
            const long tile_end = i0 + p.kt;
            const int c1 = (int)min((long)W, (min(tile_end, p.sk_iters) + p.sk_per - 1) / p.sk_per);
            role = SegRole{SEG_STORE, 0, w + 1, whole ? w + 1 : c1, true};
        }
        gemm_segment<A_KC, B_KC, MODE>(p, As, Bs, sp, bz, (t / p.gx) * BM, (t % p.gx) * BN, it0 * BK, min(p.K, it1 * BK), role);
        i0 += it1 - it0;
    }
}

// Several independent GEMMs of the same operand layout in ONE launch (the weight-gradient contractions of a backward pass:
// few output tiles each, long K).  The k-iterations of ALL problems are laid end to end and cut into W equal runs, one per
// resident workgroup (stream-K across the problems): a workgroup walks through its run — the tail of one tile, whole tiles,
// the head of the next, crossing from one problem into the following one — without any grid-wide synchronisation, so there
// is no tail between the GEMMs, a 16-tile problem no longer holds the chip, and a tile is split between as few workgroups
// as the balance allows (the atomic traffic is W + #tiles partial tiles, not W per problem).  Partial tiles accumulate with
// atomics onto buffers the caller has zeroed (the flat gradient buffer); whole tiles are stored (or added) plainly.
// PART: the XCD-partitioned form (g.xcd_lo > 0; instantiated for the weight-gradient operand layout only)
template <bool A_KC, bool B_KC, int MODE, bool PART = false>
__global__ __launch_bounds__(GEMM_THREADS, 2) void gemm_group_kernel(GemmGroupParams g) {
    GemmSmem<MODE> sm;
    float (*As)[BK][BM + PAD] = sm.As; float (*Bs)[BK][BN + PAD] = sm.Bs; unsigned* sp = sm.sp;
    int W = gridDim.x;
    int w = (W % 8 == 0 && g.xcd_swz) ? (int)(blockIdx.x & 7) * (W >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    // (mailbox of the run draw: the first word of the operand buffers, free between two runs — no second __shared__ object in front of the
    // dynamic LDS, whose 16-byte alignment the fragment reads rely on)
    volatile unsigned* mbox = MODE == 0 ? reinterpret_cast<volatile unsigned*>(&As[0][0][0]) : reinterpret_cast<volatile unsigned*>(sp);
    if (PART) {      // drawn runs (the atomic combination only): correct for any number and placement of the workgroups that take part
        if (g.xcd_lo > 0) {      // XCD partition: leave XCDs [0, xcd_lo) to the chain kernel confined there
            unsigned xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcc));
            if (g.xcd_probe && threadIdx.x == 0 && blockIdx.x < 1024) g.xcd_probe[1024 + blockIdx.x] = xcc + 1;
            if ((int)xcc < g.xcd_lo) return;
        }
        W = g.nruns;
    }
    const long total = g.first[g.n];
    const long per = (total + W - 1) / W;
    for (;;) {
        if (PART) {
            __syncthreads();                         // (the previous run's last LDS reads / the previous draw are done)
            if (threadIdx.x == 0) *mbox = atomicAdd(g.run_counter, 1u);
            __syncthreads();
            w = (int)*mbox;
            __syncthreads();                         // (everybody has it: the run's first LDS stores may overwrite the word)
            if (w >= W) return;
        }
        long i0 = (long)w * per, i1 = min(i0 + per, total);
        int pi = 0;
        while (pi + 1 < g.n && g.first[pi + 1] <= i0) ++pi;
        while (i0 < i1) {
            GemmParams p = g.prob[pi];             // by value: the fields live in SGPRs across the k-loop (fetching them from the
                                                   // kernel-argument segment inside the loop measured slower)
            p.sk_part = g.sk_part; p.sk_flag = g.sk_flag; p.sk_err = g.sk_err; p.sk_id = g.sk_id; p.call_err = g.call_err;
            const long pend = min(i1, g.first[pi + 1]);
            long l0 = i0 - g.first[pi];
            const long l1 = pend - g.first[pi];
            while (l0 < l1) {
                const int t = (int)(l0 / p.kt);
                const int it0 = (int)(l0 % p.kt), it1 = (int)min((long)p.kt, it0 + (l1 - l0));
                const bool whole = it0 == 0 && it1 == p.kt;
                SegRole role;
                if (g.sk_part == nullptr) {
                    role = whole ? SegRole{SEG_STORE, 0, 0, 0, false} : seg_atomic(false);
                } else if (it0 > 0) {
                    role = SegRole{SEG_PART, w, 0, 0, false};
                } else {
                    // owner: the slots behind this one whose runs start before the tile's last k-iteration (global iteration space)
                    const long tile_end = g.first[pi] + (l0 - it0) + p.kt;
                    const int c1 = (int)min((long)W, (tile_end + per - 1) / per);
                    role = SegRole{SEG_STORE, 0, w + 1, whole ? w + 1 : c1, false};
                }
                gemm_segment<A_KC, B_KC, MODE>(p, As, Bs, sp, 0, (t / p.gx) * BM, (t % p.gx) * BN, it0 * BK, min(p.K, it1 * BK), role);
                l0 += it1 - it0;
            }
            i0 = pend;
            ++pi;
        }
        if (!PART) return;
    }
}

// zero the C windows of the stream-K tiles (they are accumulated with atomics)
__global__ __launch_bounds__(256) void gemm_zero_tiles_kernel(GemmParams p) {
    const int per_batch = p.gx * p.gy;
    const int tile = p.dp_tiles + blockIdx.x;
    const int bz = tile / per_batch, t = tile % per_batch;
    const int m0 = (t / p.gx) * BM, n0 = (t % p.gx) * BN;
    float* C = p.C + (long)bz * p.sC;
    for (int i = threadIdx.x; i < BM * BN; i += 256) {
        const int m = m0 + i / BN, n = n0 + i % BN;
        if (m < p.M && n < p.N) C[(long)m * p.ldc + n] = 0.f;
    }
}

// ---- arithmetic mode and launch helpers ---------------------------------------------------------------------------------
// 0: v_mfma_f32_32x32x2_f32 ; 1: split-operand bf16 MFMA (fp32-faithful, see split_pair).  OPT_GEMM_ARITH, or the calling
// thread's per-call override (LAS_FLAG_GEMM_F32 -> GemmArithScope)
int gemm_get_arith() { return gemm_arith_effective(); }
void gemm_set_arith(int mode) { opt_set(OPT_GEMM_ARITH, mode ? 1 : 0); }
// schedule knobs (tools/ubench_gemm_sched.py sweeps them inside one process; -1 = built-in default)
enum { TUNE_STREAMK = 0, TUNE_SK_MIN_TILES = 1, TUNE_SPLIT_BELOW = 2, TUNE_SPLIT_TARGET = 3, TUNE_N = 4 };
static const int kTuneOpt[TUNE_N] = {OPT_GEMM_STREAMK, OPT_GEMM_SK_MIN_TILES, OPT_GEMM_SPLIT_BELOW, OPT_GEMM_SPLIT_TARGET};
void gemm_set_tuning(int key, long value) { if (key >= 0 && key < TUNE_N) opt_set(kTuneOpt[key], value); }
static long tune(int key, long dflt) {
    const long v = opt_get(kTuneOpt[key]);
    return v >= 0 ? v : dflt;
}
static long tune_skf_min_run() { return opt_get(OPT_GEMM_SKF_MIN_RUN) >= 0 ? opt_get(OPT_GEMM_SKF_MIN_RUN) : 8; }

template <class Kern>
static int split_kernel_ready(Kern kernel) {       // dynamic LDS beyond 64 KB has to be allowed once per kernel
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, SP_SMEM_BYTES) == hipSuccess;
}
template <bool A_KC, bool B_KC>
static int launch_gemm_ab(const GemmParams& p, int mode, dim3 grid, hipStream_t stream) {
    if (mode == 2) {
        static const int ready = split_kernel_ready(gemm_f32_kernel<A_KC, B_KC, 2>);
        LAS_REQUIRE(ready, "dynamic LDS of the pre-split-operand GEMM");
        hipLaunchKernelGGL((gemm_f32_kernel<A_KC, B_KC, 2>), grid, dim3(GEMM_THREADS), SP_SMEM_BYTES, stream, p);
    } else if (mode == 1) {
        static const int ready = split_kernel_ready(gemm_f32_kernel<A_KC, B_KC, 1>);
        LAS_REQUIRE(ready, "dynamic LDS of the split-operand GEMM");
        hipLaunchKernelGGL((gemm_f32_kernel<A_KC, B_KC, 1>), grid, dim3(GEMM_THREADS), SP_SMEM_BYTES, stream, p);
    } else {
        hipLaunchKernelGGL((gemm_f32_kernel<A_KC, B_KC, 0>), grid, dim3(GEMM_THREADS), 0, stream, p);
    }
    LAS_LAUNCH_CHECK();
    return LAS_OK;
}
static int launch_gemm(const GemmParams& p, bool a_kc, bool b_kc, bool planes, dim3 grid, hipStream_t stream) {
    const int mode = planes ? 2 : (gemm_get_arith() == 1 ? 1 : 0);
    path_note(PATH_GEMM, mode == 2 ? "planes" : (mode == 1 ? "split" : "f32"));
    if (a_kc && b_kc) return launch_gemm_ab<true, true>(p, mode, grid, stream);
    if (a_kc && !b_kc) return launch_gemm_ab<true, false>(p, mode, grid, stream);
    if (!a_kc && b_kc) return launch_gemm_ab<false, true>(p, mode, grid, stream);
    return launch_gemm_ab<false, false>(p, mode, grid, stream);
}
template <bool A_KC, bool B_KC>
static int launch_group_ab(const GemmGroupParams& g, int mode, dim3 grid, hipStream_t stream) {
    if (mode == 2) {
        static const int ready = split_kernel_ready(gemm_group_kernel<A_KC, B_KC, 2>);
        LAS_REQUIRE(ready, "dynamic LDS of the pre-split-operand GEMM");
        hipLaunchKernelGGL((gemm_group_kernel<A_KC, B_KC, 2>), grid, dim3(GEMM_THREADS), SP_SMEM_BYTES, stream, g);
    } else if (mode == 1) {
        static const int ready = split_kernel_ready(gemm_group_kernel<A_KC, B_KC, 1>);
        LAS_REQUIRE(ready, "dynamic LDS of the split-operand GEMM");
        hipLaunchKernelGGL((gemm_group_kernel<A_KC, B_KC, 1>), grid, dim3(GEMM_THREADS), SP_SMEM_BYTES, stream, g);
    } else {
        hipLaunchKernelGGL((gemm_group_kernel<A_KC, B_KC, 0>), grid, dim3(GEMM_THREADS), 0, stream, g);
    }
    LAS_LAUNCH_CHECK();
    return LAS_OK;
}
static int launch_group_part(const GemmGroupParams& g, int mode, dim3 grid, hipStream_t stream) {      // weight-gradient layout, modes 0 / 1
    if (mode == 1) {
        static const int ready = split_kernel_ready(gemm_group_kernel<false, false, 1, true>);
        LAS_REQUIRE(ready, "dynamic LDS of the split-operand GEMM");
        hipLaunchKernelGGL((gemm_group_kernel<false, false, 1, true>), grid, dim3(GEMM_THREADS), SP_SMEM_BYTES, stream, g);
    } else {
        hipLaunchKernelGGL((gemm_group_kernel<false, false, 0, true>), grid, dim3(GEMM_THREADS), 0, stream, g);
    }
    LAS_LAUNCH_CHECK();
    return LAS_OK;
}
static int launch_group(const GemmGroupParams& g, bool a_kc, bool b_kc, bool planes, dim3 grid, hipStream_t stream) {
    const int mode = planes ? 2 : (gemm_get_arith() == 1 ? 1 : 0);
    if (g.nruns > 0) return launch_group_part(g, mode, grid, stream);
    if (a_kc && b_kc) return launch_group_ab<true, true>(g, mode, grid, stream);
    if (a_kc && !b_kc) return launch_group_ab<true, false>(g, mode, grid, stream);
    if (!a_kc && b_kc) return launch_group_ab<false, true>(g, mode, grid, stream);
    return launch_group_ab<false, false>(g, mode, grid, stream);
}

static int gemm_resident_slots() {      // persistent grid: two 256-thread workgroups per CU (34 KB LDS, <= 128 VGPRs each)
    static int slots = -1;
    if (slots < 0) {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 0;
        const int per_cu = (int)opt_get(OPT_GEMM_SLOTS_PER_CU);
        slots = per_cu * cus;
    }
    return slots;
}

// ---- scratch of the stream-K fix-up: one per (device, stream), created on first use -------------------------------------------------
// W parked partial tiles (64 KB each) + W flags + a host-visible error word.  This is the GEMM's own workspace handle (what a BLAS
// handle carries); every other buffer of the library is the caller's.  Kernels of one stream run in order, so one scratch per stream
// is race-free; a stream beyond SK_MAX_SCRATCH (or a first use during stream capture, where hipMalloc is illegal) gets none and
// the GEMM takes the schedules that do not need it.
struct SkScratch { int dev; hipStream_t stream; float* part; unsigned* flag; unsigned* err_host; unsigned* err_dev; int slots; };
constexpr int SK_MAX_SCRATCH = 16;
static std::mutex g_sk_mu;
static SkScratch g_sk[SK_MAX_SCRATCH];
static int g_sk_n = 0;
static std::atomic<unsigned> g_sk_id{1};
static std::atomic<int> g_sk_any{0};      // a scratch exists: until then gemm_sk_check is one relaxed load (no mutex on every GEMM call)

static const SkScratch* sk_scratch(hipStream_t stream, int slots) {
    if (opt_get(OPT_GEMM_SK_FIXUP) <= 0 || slots <= 0) return nullptr;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lk(g_sk_mu);
    for (int i = 0; i < g_sk_n; ++i)
        if (g_sk[i].dev == dev && g_sk[i].stream == stream && g_sk[i].slots >= slots) return &g_sk[i];
    if (g_sk_n >= SK_MAX_SCRATCH) return nullptr;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) return nullptr;
    SkScratch s{dev, stream, nullptr, nullptr, nullptr, nullptr, slots};
    if (hipMalloc(&s.part, sizeof(float) * (size_t)slots * BM * BN) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    if (hipMalloc(&s.flag, sizeof(unsigned) * (size_t)slots) != hipSuccess || hipMemset(s.flag, 0, sizeof(unsigned) * (size_t)slots) != hipSuccess ||
        hipHostMalloc(&s.err_host, sizeof(unsigned), hipHostMallocMapped) != hipSuccess ||
        hipHostGetDevicePointer(reinterpret_cast<void**>(&s.err_dev), s.err_host, 0) != hipSuccess) {
        (void)hipGetLastError();
        if (s.part) (void)hipFree(s.part);
        if (s.flag) (void)hipFree(s.flag);
        if (s.err_host) (void)hipHostFree(s.err_host);
        return nullptr;
    }
    *s.err_host = 0;
    g_sk[g_sk_n] = s;
    g_sk_any.store(1, std::memory_order_release);
    return &g_sk[g_sk_n++];
}
static unsigned sk_next_id() {
    unsigned id = g_sk_id.fetch_add(1, std::memory_order_relaxed);
    if (id == 0) id = g_sk_id.fetch_add(1, std::memory_order_relaxed);
    return id;
}
// A fix-up wait that ran into its spin limit (a contributing workgroup was not resident: another kernel held the CUs) leaves a wrong
// tile behind; the device raises the host-visible word and the next GEMM call (or las_gemm_check) fails loudly.
int gemm_sk_check() {
    if (g_sk_any.load(std::memory_order_acquire) == 0) return LAS_OK;
    std::lock_guard<std::mutex> lk(g_sk_mu);
    bool bad = false;
    for (int i = 0; i < g_sk_n; ++i)
        if (*reinterpret_cast<volatile unsigned*>(g_sk[i].err_host) != 0) { *g_sk[i].err_host = 0; bad = true; }
    return bad ? fail(LAS_ERR_DEVICE, "stream-K fix-up: a contributing workgroup never arrived (GEMM workgroups were not all resident)%s", "") : LAS_OK;
}

constexpr long SKF_MIN_TILES = 64;
bool gemm_sk_fixup_ready(hipStream_t stream, int M, int N) {
    return opt_get(OPT_GEMM_STREAMK) != 0 && (long)cdiv(M, BM) * cdiv(N, BN) >= SKF_MIN_TILES && sk_scratch(stream, gemm_resident_slots()) != nullptr;
}

static bool gemm_aligned(const float* ptr, long ld, long bs) { return ((uintptr_t)ptr % 16 == 0) && (ld % 4 == 0) && (bs % 4 == 0); }

// Pre-split operands index whole 16-byte granules: 8 consecutive elements along the contiguous dimension of each operand
static bool planes_shape_ok(const GemmDesc& d) {
    auto al = [](const void* q) { return q == nullptr || (uintptr_t)q % 16 == 0; };
    return d.K % 8 == 0 && d.lda % 8 == 0 && d.ldb % 8 == 0 && d.sA % 8 == 0 && d.sB % 8 == 0 && al(d.A) && al(d.B) && al(d.A2) && al(d.B2) &&
           (d.a_kc || d.M % 8 == 0) && (d.b_kc || d.N % 8 == 0) && (d.A2 == nullptr || d.K1 % 16 == 0);
}

// fp32 matrix (R x C, row stride ld_src) -> P8x3 granules with row stride ld_dst (elements; multiples of 8): one thread per octet
__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ src, long ld_src, int R, int C8, u32x4* __restrict__ dst, long ldo_dst) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)R * C8) return;
    const int r = (int)(idx / C8), o = (int)(idx % C8);
    const f32x4 v0 = *reinterpret_cast<const f32x4*>(src + (long)r * ld_src + o * 8);
    const f32x4 v1 = *reinterpret_cast<const f32x4*>(src + (long)r * ld_src + o * 8 + 4);
    unsigned a[4], b[4], c[4];
    split_pair(v0[0], v0[1], a[0], b[0], c[0]);
    split_pair(v0[2], v0[3], a[1], b[1], c[1]);
    split_pair(v1[0], v1[1], a[2], b[2], c[2]);
    split_pair(v1[2], v1[3], a[3], b[3], c[3]);
    u32x4* q = dst + ((long)r * ldo_dst + o) * 3;
    q[0] = u32x4{a[0], a[1], a[2], a[3]}; q[1] = u32x4{b[0], b[1], b[2], b[3]}; q[2] = u32x4{c[0], c[1], c[2], c[3]};
}
int split_planes(const float* src, long ld_src, int R, int C, void* dst, long ld_dst, hipStream_t stream) {
    LAS_REQUIRE(src && dst && R > 0 && C > 0, "split_planes arguments");
    LAS_REQUIRE(C % 8 == 0 && ld_src % 4 == 0 && ld_dst % 8 == 0 && ld_dst >= C && (uintptr_t)src % 16 == 0 && (uintptr_t)dst % 16 == 0,
                "split_planes: whole, aligned octets");
    const long n = (long)R * (C / 8);
    hipLaunchKernelGGL(split_planes_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, src, ld_src, R, C / 8,
                       reinterpret_cast<u32x4*>(dst), ld_dst / 8);
    LAS_LAUNCH_CHECK();
    return LAS_OK;
}

// the run counter of the XCD-partitioned group launches: 64 words per device, created on first use; consecutive launches (in-order side stream)
// take consecutive words, each zeroed in front of its launch
static unsigned* group_run_counter(hipStream_t stream) {
    static std::mutex mu;
    static unsigned* buf[16] = {};
    static unsigned next[16] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
    std::lock_guard<std::mutex> lk(mu);
    if (buf[dev] == nullptr) {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(stream, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) return nullptr;
        if (hipMalloc(&buf[dev], sizeof(unsigned) * 64) != hipSuccess) { (void)hipGetLastError(); buf[dev] = nullptr; return nullptr; }
    }
    unsigned* p = buf[dev] + (next[dev]++ & 63u);
    if (hipMemsetAsync(p, 0, sizeof(unsigned), stream) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return p;
}

int gemm_f32_group(const GemmDesc* ds, int n, hipStream_t stream, int xcd_lo, int drawn_runs) {
    const int group_on = (int)opt_get(OPT_GEMM_GROUP);
    const int W = gemm_resident_slots();
    if (xcd_lo < 0 || xcd_lo > 6 || W % 8 != 0) xcd_lo = 0;
    unsigned* run_counter = nullptr;
    if (group_on && xcd_lo == 0 && drawn_runs <= 0) {      // 256-tile form where the group's tiles are big enough
        const int rc = gemm_big_group(ds, n, stream);
        if (rc != LAS_ERR_UNSUPPORTED) return rc;
    }
    bool ok = group_on && W > 0 && n >= 1 && n <= GROUP_MAX;
    // The grouped weight-gradient launches keep the atomic combination by default: their tiles have K = 3 200 .. 12 800 and span 4 - 9
    // workgroup runs, so an owner would fetch up to 512 KB of parked tiles serially at the end of its run, where atomics are fire-and-
    // forget and overlap the next segment (measured: L0 dW group 185 against 164 us, the others +3 .. +7 us).  GEMM_SK_FIXUP=2 selects
    // the fix-up form here too (outputs then need not start from zero; results are run-to-run deterministic).
    const SkScratch* sc = (ok && xcd_lo == 0 && drawn_runs <= 0 && opt_get(OPT_GEMM_SK_FIXUP) >= 2) ? sk_scratch(stream, W) : nullptr;
    for (int i = 0; ok && i < n; ++i) {
        const GemmDesc& d = ds[i];
        ok = d.batch <= 1 && !d.relu && !d.bias0 && !d.bias1 && (d.c_zeroed || d.accumulate || sc != nullptr) && d.a_kc == ds[0].a_kc && d.b_kc == ds[0].b_kc &&
             d.A2 == nullptr && d.M > 0 && d.N > 0 && d.K > 0 && d.planes == ds[0].planes && (!d.planes || planes_shape_ok(d));
    }
    if (!ok) {      // not groupable (or switched off): one launch per problem
        for (int i = 0; i < n; ++i) LAS_TRY(gemm_f32(ds[i], stream));
        return LAS_OK;
    }
    LAS_TRY(gemm_sk_check());
    const bool want_drawn = xcd_lo > 0 || drawn_runs > 0;
    if (want_drawn && (ds[0].a_kc || ds[0].b_kc || ds[0].planes || (run_counter = group_run_counter(stream)) == nullptr)) { xcd_lo = 0; drawn_runs = 0; }
    GemmGroupParams g;
    g.n = n;
    g.first[0] = 0;
    const int xcd_swz = (int)opt_get(OPT_GEMM_XCD_SWZ);
    g.xcd_swz = xcd_swz;
    g.xcd_lo = xcd_lo; g.xcd_probe = xcd_lo > 0 ? xcd_probe_ptr() : nullptr; g.run_counter = run_counter;
    g.nruns = run_counter == nullptr ? 0 : (drawn_runs > 0 ? drawn_runs : (W >> 3) * (8 - xcd_lo));
    for (int i = 0; i < n; ++i) {
        const GemmDesc& d = ds[i];
        GemmParams& p = g.prob[i];
        memset(&p, 0, sizeof(p));
        p.A = d.A; p.B = d.B; p.C = d.C; p.M = d.M; p.N = d.N; p.K = d.K; p.lda = d.lda; p.ldb = d.ldb; p.ldc = d.ldc;
        p.a_vec = d.planes || gemm_aligned(d.A, d.lda, 0); p.b_vec = d.planes || gemm_aligned(d.B, d.ldb, 0);
        p.gx = cdiv(d.N, BN); p.gy = cdiv(d.M, BM); p.kt = std::max(1, cdiv(d.K, BK));
        p.splitk = 1; p.kper = d.K;
        p.accumulate = d.accumulate;           // whole tiles: plain add onto the caller's values instead of a plain store
        g.first[i + 1] = g.first[i] + (long)p.gx * p.gy * p.kt;
    }
    // fix-up schedule: parked partial sums instead of atomics; runs of at least SKF_MIN_RUN k-iterations (small groups use fewer slots)
    g.sk_part = nullptr; g.sk_flag = nullptr; g.sk_err = nullptr; g.sk_id = 0; g.call_err = gemm_call_err_word();
    int Wg = W;
    if (sc != nullptr) {
        const long min_run = std::max<long>(1, tune_skf_min_run());
        Wg = (int)std::min<long>(W, std::max<long>(8, g.first[n] / min_run / 8 * 8));
        g.sk_part = sc->part; g.sk_flag = sc->flag; g.sk_err = sc->err_dev; g.sk_id = sk_next_id();
    }
    return launch_group(g, ds[0].a_kc, ds[0].b_kc, ds[0].planes, dim3(Wg), stream);
}

int gemm_f32(const GemmDesc& d, hipStream_t stream) {
    LAS_REQUIRE(d.M > 0 && d.N > 0 && d.K >= 0, "gemm dims");
    LAS_REQUIRE(d.A && d.B && d.C, "gemm pointers");
    LAS_REQUIRE(d.A2 == nullptr || (d.B2 != nullptr && d.K1 > 0 && d.K1 < d.K && d.K1 % BK == 0 && d.batch <= 1), "second K source");
    LAS_REQUIRE(!d.planes || planes_shape_ok(d), "pre-split operands: K, leading dimensions and the contiguous extents must be multiples of 8, buffers 16-byte aligned");
    {
        const int rc = gemm_big(d, stream);
        if (rc != LAS_ERR_UNSUPPORTED) return rc;
    }
    const bool fastk = d.planes || gemm_get_arith() == 1;      // a k-iteration 2-3x faster than on the fp32 matrix pipe: schedule thresholds follow
    GemmParams p;
    p.A2 = d.A2; p.B2 = d.B2; p.K1 = d.K1;
    p.A = d.A; p.B = d.B; p.C = d.C; p.bias0 = d.bias0; p.bias1 = d.bias1;
    p.M = d.M; p.N = d.N; p.K = d.K; p.lda = d.lda; p.ldb = d.ldb; p.ldc = d.ldc;
    p.sA = d.sA; p.sB = d.sB; p.sC = d.sC; p.sBias0 = d.sBias0; p.sBias1 = d.sBias1;
    const int batch = d.batch > 0 ? d.batch : 1;
    const int gx = cdiv(d.N, BN), gy = cdiv(d.M, BM);
    const long tiles = (long)gx * gy * batch;
    const int kt = std::max(1, cdiv(d.K, BK));
    p.a_vec = d.planes || (gemm_aligned(d.A, d.lda, d.sA) && (d.A2 == nullptr || gemm_aligned(d.A2, d.lda, 0)));
    p.b_vec = d.planes || (gemm_aligned(d.B, d.ldb, d.sB) && (d.B2 == nullptr || gemm_aligned(d.B2, d.ldb, 0)));
    p.gx = gx; p.gy = gy; p.kt = kt;
    p.accumulate = d.accumulate; p.relu = d.relu;
    const int xcd_swz = (int)opt_get(OPT_GEMM_XCD_SWZ);
    p.xcd_swz = xcd_swz;
    p.persistent = 0; p.dp_tiles = 0; p.sk_iters = 0; p.sk_per = 0; p.sk_atomic_whole = 0; p.atomic = 0; p.swz = 0; p.splitk = 1; p.kper = d.K;

    // ---- schedule --------------------------------------------------------------------------------------------------
    const int sk_on = (int)tune(TUNE_STREAMK, 1);
    const int W = gemm_resident_slots();
    const bool may_split = !d.relu && d.K >= 4 * BK;         // a relu epilogue needs the whole sum in one place
    // split-operand arithmetic runs a k-iteration 2-3x faster, so the fixed costs of the stream-K epilogue (zeroing pass, one atomic
    // per output of every partial tile) weigh more: there the persistent schedule is used only when whole tiles fill every slot at
    // least once (plain stores for those, stream-K for the ragged tail), fewer tiles take the classic grid / split-K
    const long min_tiles = tune(TUNE_SK_MIN_TILES, fastk ? W : 0);
    p.sk_part = nullptr; p.sk_flag = nullptr; p.sk_err = nullptr; p.sk_id = 0; p.call_err = gemm_call_err_word();
    LAS_TRY(gemm_sk_check());
    // Stream-K with in-kernel fix-up: the k-iterations of ALL tiles laid end to end and cut into equal runs, one per resident
    // workgroup; a tile that straddles runs is finished by the workgroup holding its first k-iteration, which adds the partial sums the
    // others parked (plain coalesced 64 KB per partial instead of 16 K atomics, no zeroing pass, any epilogue).  Taken whenever the tile
    // count does not fill whole rounds of the resident slots; a run is at least SKF_MIN_RUN k-iterations and a quarter tile.
    {
        const long total = tiles * kt;
        const long skf_min_kt = opt_get(OPT_GEMM_SKF_MIN_KT) >= 0 ? opt_get(OPT_GEMM_SKF_MIN_KT) : 8;
        // a run is at least 8 k-iterations; at most 8 runs share a tile (the owner fetches 64 KB per sharer)
        const long min_run = std::max<long>(tune_skf_min_run(), (kt + 7) / 8);
        const int Wuse = (int)std::min<long>(W, total / std::max<long>(1, min_run) / 8 * 8);
        const bool uneven = (double)tiles / ((double)cdiv(tiles, W) * W) < 0.92;      // share of the resident slots a classic grid keeps busy
        // (few tiles need many sharers per tile to cover the chip, and the owner's serial fetch of their parked tiles then costs more than
        // fire-and-forget atomics — dW_psi, 4 tiles x 200 k-iterations: 47 against 15 us; the logits, 32 tiles x 64: 32 against 18 us inside
        // the training step — so below SKF_MIN_TILES the split-K path stays)
        if (sk_on && d.splitk <= 1 && Wuse >= 16 && kt >= skf_min_kt && uneven && tiles >= SKF_MIN_TILES) {
            if (const SkScratch* sc = sk_scratch(stream, W)) {
                p.persistent = 1; p.dp_tiles = 0; p.sk_iters = total; p.sk_per = (total + Wuse - 1) / Wuse;
                p.sk_part = sc->part; p.sk_flag = sc->flag; p.sk_err = sc->err_dev; p.sk_id = sk_next_id();
                return launch_gemm(p, d.a_kc, d.b_kc, d.planes, dim3((unsigned)cdiv(total, p.sk_per)), stream);
            }
        }
    }
    if (sk_on && d.splitk <= 1 && W > 0 && may_split && kt >= 32 && tiles % W != 0 && tiles * kt >= 4L * W && tiles >= min_tiles) {
        // persistent: whole tiles while they fill every slot, the ragged tail (or everything, when there are fewer tiles than
        // slots) as equal runs of k-iterations
        const long full = (tiles / W) * W;
        // keep the stream-K part at least half a wave of work so that its runs are not dominated by prologue/epilogue
        long dp = full;
        if (dp > 0 && (tiles - dp) * 2 < W) dp -= W;
        p.persistent = 1;
        p.dp_tiles = (int)dp;
        p.sk_iters = (tiles - dp) * kt;
        p.sk_per = (p.sk_iters + W - 1) / W;
        p.sk_atomic_whole = d.accumulate ? 0 : 0;
        const int sk_tiles = (int)(tiles - dp);
        if (!d.accumulate && !d.c_zeroed && sk_tiles > 0) {
            if (dp == 0 && batch == 1 && d.ldc == d.N) {
                LAS_HIP_CHECK(hipMemsetAsync(d.C, 0, sizeof(float) * (size_t)d.M * d.N, stream));
            } else {
                hipLaunchKernelGGL(gemm_zero_tiles_kernel, dim3(sk_tiles), dim3(256), 0, stream, p);
            }
        }
        dim3 grid((unsigned)std::min<long>(W, std::max<long>(dp > 0 ? W : 1, cdiv(p.sk_iters, std::max<long>(1, p.sk_per)))));
        return launch_gemm(p, d.a_kc, d.b_kc, d.planes, grid, stream);
    }

    int splitk = d.splitk > 0 ? d.splitk : 1;
    // (callers pass splitk = 1 where the persistent schedule was the measured best in fp32-MFMA arithmetic; in split-operand
    // arithmetic that schedule is not taken below one tile per slot, so the request falls back to the automatic split)
    const bool can_zero = d.accumulate || d.c_zeroed || d.ldc == d.N || batch == 1;      // split-K partials need a zeroed (or accumulated) C
    if (d.splitk == 0 || (d.splitk == 1 && fastk && may_split && can_zero)) {
        // auto: few output tiles and a long K -> split K so the launch covers the chip (256 CUs)
        const long below = tune(TUNE_SPLIT_BELOW, fastk ? W / 2 : 128);
        if (!d.relu && tiles < below && d.K >= 256) {
            // few output tiles, long K: about two workgroups per CU (they hide each other's barrier stalls) with at least 4 k-tiles each
            const long target = tune(TUNE_SPLIT_TARGET, 512);   // ~2 workgroups per CU: measured best
            splitk = (int)min((long)cdiv(d.K, 4 * BK), max(1L, target / tiles));
        }
    }
    int kper = cdiv(cdiv(d.K, splitk), BK) * BK;
    if (kper == 0) kper = BK;
    splitk = max(1, cdiv(d.K, kper));
    LAS_REQUIRE(!(splitk > 1 && d.relu), "relu epilogue needs splitk==1");
    p.splitk = splitk; p.kper = kper;
    p.atomic = splitk > 1;
    if (p.atomic && !d.accumulate && !d.c_zeroed) {
        // split-K partials are summed with atomics: C must start from zero
        if (d.ldc == d.N) {
            // (batches with gaps between their outputs — interleaved with another GEMM's — are zeroed block by block: the gaps may hold live data)
            if (batch > 1 && d.sC != (long)d.M * d.N)
                LAS_HIP_CHECK(hipMemset2DAsync(d.C, sizeof(float) * (size_t)d.sC, 0, sizeof(float) * (size_t)d.M * d.N, (size_t)batch, stream));
            else
                LAS_HIP_CHECK(hipMemsetAsync(d.C, 0, sizeof(float) * ((size_t)(batch - 1) * d.sC + (size_t)d.M * d.N), stream));
        } else {
            LAS_HIP_CHECK(hipMemset2DAsync(d.C, sizeof(float) * d.ldc, 0, sizeof(float) * d.N, (size_t)d.M, stream));
            LAS_REQUIRE(batch == 1, "split-K with strided C and batch>1");
        }
    }
    p.swz = ((gx * gy) % 8 == 0) && (gx * gy >= 64);
    return launch_gemm(p, d.a_kc, d.b_kc, d.planes, dim3(gx * gy, 1, batch * splitk), stream);
}

}  // namespace las
