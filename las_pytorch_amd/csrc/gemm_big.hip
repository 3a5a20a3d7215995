// 256 x 256 x 16 tiles of the split-operand fp32 GEMM (arithmetic mode 1 of gemm_f32.hip) for gfx950:  C[M,N] (+)= act(A B + bias)
//
// Replaces the same ATen contractions as gemm_f32.hip (reference model/las_model.py:90,174,279 and their autograd, solver/solver.py:95).
//
// Why a second tile size, and what it turned out to be worth.  Ablation builds of the 128 x 128 kernel's main loop on pre-split operands
// (tools/ubench_gemm_planes.py, -DLAS_PLANES_ABL; round 5) showed what it loses: not the operand-split arithmetic (operands split beforehand,
// no VALU at all: 172-194 TF at 4096^3, the same as with it) but the bytes every k-tile moves through the global -> register -> LDS store path
// (no LDS stores: 282 TF; no global loads: 265 TF; MFMAs alone: 296 TF).  Those bytes per MFMA fall with the tile edge, so this kernel runs
// ONE workgroup of four waves per CU, one wave per SIMD with the whole 512-entry register file: a 128 x 128 wave tile's 256 accumulator
// registers live in the AGPR half, three staging sets and one fragment set in the VGPR half; per MFMA it stores and fetches half as much,
// splits half as many values and reads half as many fragments as the 128-tile kernel.
//   * one wave per SIMD has no partner wave to fill MFMA issue gaps: every MFMA group carries an explicit MFMA : VALU : LDS : VMEM
//     interleave (sched_group_barrier), and a fragment plane is refilled during the group AFTER its last use (a refill interleaved with the
//     MFMAs that still read the plane would need a second register set);
//   * LDS buffer offsets are made opaque per step: as known constants they made the compiler hoist ~30 base-address registers (one per
//     64 KB window and access pattern) out of the loop and spill 100-400 of them, with scratch reloads inside the MFMA groups.
// Result (tools/ubench_gemm_big.py): the main loop is faster (4096^3: 223 against 199 TF), but a tile costs ~28 us outside it and these
// problem sizes give 100-400 tiles for 256 CUs: on the GEMMs of the training step the 128-tile kernel stays 10-50 % faster, so automatic
// mode (gemm_big below) takes only problems whose tiles fill whole rounds of the chip with >= 64 k-tiles each.
// LDS: three buffers of bf16 planes, 152 064 bytes (of the CU's 163 840).  Same arithmetic, same LDS images, same conflict-free store /
// read patterns as the 128-tile kernel (see the comments there): whole-K tiles are bit-identical to it.
#include "las_common.h"
#include "las_kernels.h"
#include "options.h"
#include "gemm_common.h"
#include <algorithm>
#include <type_traits>

namespace las {
namespace big {

constexpr int T = 256, BK = 16, THREADS = 256, NLD = 4;
constexpr int KH = T * 4 + 16;                   // dwords per k-half of a K-contiguous operand's plane: [k-half][row][4 dwords]
constexpr int KQ = T * 2;                        // dwords per k-quad of a row-contiguous operand's plane: [k-quad][slot(row)][2 dwords]
constexpr int PLANE = 8 * (T + 8);
static_assert(PLANE >= 2 * KH && PLANE >= 4 * KQ && KH % 4 == 0, "both plane images fit; 16-byte aligned k-halves");
constexpr int OPER = 3 * PLANE, BUF = 2 * OPER, NBUF = 3;
constexpr int SMEM_BYTES = NBUF * BUF * 4;
static_assert(SMEM_BYTES <= 160 * 1024, "one workgroup per CU");
#ifndef LAS_BIG_SGB
#define LAS_BIG_SGB 1       // 1: explicit MFMA : VALU : LDS interleave inside every MFMA group (one wave per SIMD has no partner wave to fill the gaps)
#endif

// slot of a row inside a k-quad of the row-contiguous image: the 128-row pattern of gemm_f32.hip (sp_slot) per half
static __device__ __forceinline__ int slot(int row) {
    const int r = row & 127;
    return (row & 128) + (r & 3) * 32 + (((r >> 2) + 8 * (r & 3)) & 31);
}

// thread -> element map of one operand tile (four 16-byte loads per thread):
//  KC : load i covers row t / 4 + 64 i, k = 4 (t % 4) .. +3
//  !KC: load 2 j + e covers k = 2 kp + e (kp = t / 32), rows 128 j + 4 (t % 32) .. +3; the partner lane 32 up holds the other k-pair of
//       the same quad and rows
// slice sl = 0..3 of the split / store work: KC: load sl;  !KC: (j, h) = (sl / 2, sl % 2): rows h and h + 2 of the four, both k
template <bool KC>
static __device__ __forceinline__ void split_store_slice(unsigned* S, const f32x4 (&reg)[NLD], int sl) {
    const int t = threadIdx.x;
    unsigned a[3], b[3];
    if constexpr (KC) {
        const int row = (t >> 2) + 64 * sl, q = t & 3;
        split_pair(reg[sl][0], reg[sl][1], a[0], a[1], a[2]);
        split_pair(reg[sl][2], reg[sl][3], b[0], b[1], b[2]);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            u32x2 v = {a[pl], b[pl]};
            *reinterpret_cast<u32x2*>(&S[pl * PLANE + (q >> 1) * KH + row * 4 + (q & 1) * 2]) = v;
        }
    } else {
        const int j = sl >> 1, h = sl & 1;
        const int rq = 128 * j + (t & 31) * 4 + h + ((t & 32) ? 2 : 0), kq = t >> 6;
        split_pair(reg[2 * j][h], reg[2 * j + 1][h], a[0], a[1], a[2]);
        split_pair(reg[2 * j][h + 2], reg[2 * j + 1][h + 2], b[0], b[1], b[2]);
        unsigned* dst = S + kq * KQ + slot(rq) * 2;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            auto r = __builtin_amdgcn_permlane32_swap(a[pl], b[pl], false, false);      // {[a.lo | b.lo], [a.hi | b.hi]}
            u32x2 v = {r[0], r[1]};                                                      // (even k-pair, odd k-pair) of this lane's row
            *reinterpret_cast<u32x2*>(dst + pl * PLANE) = v;
        }
    }
}
// the same register image with per-element guards (edge tiles, K ranges that are not multiples of 16, unaligned operands): zero
// outside [0,R) x [k0,kend)
template <bool KC>
static __device__ __forceinline__ void load_guarded(const float* __restrict__ P, long ld, int R, int r0, int k0, int kend, bool vec_ok,
                                                    f32x4 (&reg)[NLD]) {
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if constexpr (KC) {
            const int r = r0 + (t >> 2) + 64 * i, k = k0 + (t & 3) * 4;
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
            const int k = k0 + 2 * (t >> 5) + (i & 1), r = r0 + 128 * (i >> 1) + (t & 31) * 4;
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

struct Frag { u32x4 a[3][4], b[3][4]; };      // [plane][32-row tile]: 8 consecutive k of this lane's row as bf16

// one MFMA, then up to `valu` VALU, one LDS and one vector-memory instruction: the issue slots an MFMA leaves free (32 cycles on its SIMD)
template <int N>
static __device__ __forceinline__ void interleave(int) {
#if LAS_BIG_SGB
#pragma unroll
    for (int n = 0; n < N; ++n) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x080, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    }
#endif
}

// One output tile over the k-range [kbeg, kend): main loop + epilogue.  role: SEG_STORE (bias / accumulate / relu, plain stores) or
// SEG_ATOMIC (one of several contributors: atomicAdd onto a C that holds zero or the value to accumulate onto).
template <bool A_KC, bool B_KC>
__device__ __forceinline__ void segment(const GemmParams& p, unsigned* sp, int bz, int m0, int n0, int kbeg, int kend, const SegRole role) {
    const float* A = p.A + (long)bz * p.sA;
    const float* B = p.B + (long)bz * p.sB;
    float* C = p.C + (long)bz * p.sC;
    const int ntiles = (kend - kbeg + BK - 1) / BK;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = (wave >> 1) * 128, wn = (wave & 1) * 128;
    const int lr = lane & 31, lk = lane >> 5;

    f32x16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const bool fast = p.a_vec && p.b_vec && (m0 + T <= p.M) && (n0 + T <= p.N) && ((kend - kbeg) % BK == 0);
    if (ntiles > 0) {
        // element offset of this thread's load 0; KC: load i is 64 i rows further; row-contiguous: load 2 j + e is e k-rows and 128 j rows further
        const long oA = A_KC ? (long)(m0 + (t >> 2)) * p.lda + (t & 3) * 4 : (long)(2 * (t >> 5)) * p.lda + m0 + (t & 31) * 4;
        const long oB = B_KC ? (long)(n0 + (t >> 2)) * p.ldb + (t & 3) * 4 : (long)(2 * (t >> 5)) * p.ldb + n0 + (t & 31) * 4;
        const long strideA = A_KC ? 1 : p.lda, strideB = B_KC ? 1 : p.ldb;
        // (ONE instance of the main loop: the interior / guarded decision is a uniform branch around the loads, not a second copy of the
        // loop whose 256 accumulator registers would have to be merged behind it)
        auto gload = [&](int k0, f32x4 (&ra_)[NLD], f32x4 (&rb_)[NLD]) {
            const bool second = p.A2 != nullptr && k0 >= p.K1;          // workgroup-uniform
            const float* Ab = second ? p.A2 : A;
            const float* Bb = second ? p.B2 : B;
            const long kk = second ? k0 - p.K1 : k0;
            if (fast) {
                const float* pa = Ab + oA + kk * strideA;
                const float* pb = Bb + oB + kk * strideB;
#pragma unroll
                for (int i = 0; i < NLD; ++i) ra_[i] = *reinterpret_cast<const f32x4*>(pa + (A_KC ? 64 * i * p.lda : (i & 1) * p.lda + 128 * (i >> 1)));
#pragma unroll
                for (int i = 0; i < NLD; ++i) rb_[i] = *reinterpret_cast<const f32x4*>(pb + (B_KC ? 64 * i * p.ldb : (i & 1) * p.ldb + 128 * (i >> 1)));
            } else {
                const int ke = second ? kend - p.K1 : (p.A2 != nullptr ? min(kend, p.K1) : kend);
                load_guarded<A_KC>(Ab, p.lda, p.M, m0, (int)kk, ke, p.a_vec, ra_);
                load_guarded<B_KC>(Bb, p.ldb, p.N, n0, (int)kk, ke, p.b_vec, rb_);
            }
        };
        // fragment offsets (dwords) inside an operand plane, tile 0; tile i is 32 rows further (K-contiguous: + 128 dwords; row-contiguous:
        // slot(row + 32 i) = slot(row) + 8 i within a 128-row half, + 128 slots for the second half)
        const int fa0 = A_KC ? lk * KH + (wm + lr) * 4 : (2 * lk) * KQ + slot(wm + lr) * 2;
        const int fb0 = B_KC ? lk * KH + (wn + lr) * 4 : (2 * lk) * KQ + slot(wn + lr) * 2;
        int fa[4], fb[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            fa[i] = A_KC ? fa0 + i * 128 : (2 * lk) * KQ + slot(wm + i * 32 + lr) * 2;
            fb[i] = B_KC ? fb0 + i * 128 : (2 * lk) * KQ + slot(wn + i * 32 + lr) * 2;
        }
        Frag f;
        // (boff: dword offset of the LDS buffer, an SGPR value the step makes opaque — were it a known constant, the compiler would hoist one
        // base-address VGPR per 64 KB window and access pattern out of the loop and spill them)
        auto rd_a = [&](int boff, int pl) {
            const unsigned* pa = sp + boff + pl * PLANE;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if constexpr (A_KC) {
                    f.a[pl][i] = *reinterpret_cast<const u32x4*>(pa + fa[i]);
                } else {
                    const unsigned* q = pa + fa[i];
                    const unsigned* q2 = q + KQ;
                    asm volatile("" : "+v"(q2));          // (two ds_read_b64: the compiler must not merge them into one ds_read2_b64)
                    const u32x2 lo = *reinterpret_cast<const u32x2*>(q), hi = *reinterpret_cast<const u32x2*>(q2);
                    f.a[pl][i] = u32x4{lo[0], lo[1], hi[0], hi[1]};
                }
            }
        };
        auto rd_b = [&](int boff, int pl) {
            const unsigned* pb = sp + boff + OPER + pl * PLANE;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if constexpr (B_KC) {
                    f.b[pl][i] = *reinterpret_cast<const u32x4*>(pb + fb[i]);
                } else {
                    const unsigned* q = pb + fb[i];
                    const unsigned* q2 = q + KQ;
                    asm volatile("" : "+v"(q2));
                    const u32x2 lo = *reinterpret_cast<const u32x2*>(q), hi = *reinterpret_cast<const u32x2*>(q2);
                    f.b[pl][i] = u32x4{lo[0], lo[1], hi[0], hi[1]};
                }
            }
        };
        auto grp = [&](int pa, int pb) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f.a[pa][i]),
                                                                         __builtin_bit_cast(bf16x8, f.b[pb][j]), acc[i][j], 0, 0, 0);
        };
        {
            // register sets: tile j travels in set j % 3; its loads are issued two steps before its split / store
            f32x4 ra[3][NLD], rb[3][NLD];
            auto sstore_all = [&](int buf, const f32x4 (&ra_)[NLD], const f32x4 (&rb_)[NLD]) {
                unsigned* sa = sp + buf * BUF;
#pragma unroll
                for (int sl = 0; sl < 4; ++sl) { split_store_slice<A_KC>(sa, ra_, sl); split_store_slice<B_KC>(sa + OPER, rb_, sl); }
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
            // step kt with R = kt % 3 (the rotation of gemm_f32.hip's split-operand loop): tile kt is in the fragment registers and LDS buffer
            // R, tile kt+1 in buffer R+1, tile kt+2 in register set R+2 (on its way to buffer R+2), tile kt+3 in flight into set R, tile kt+4
            // is requested into set R+1.  Six MFMA groups of sixteen; a plane of the fragment set is refilled with tile kt+1 as soon as
            // its last group has been issued.
            auto step = [&](auto RC, auto FULLC, int kt) {
                constexpr int R = decltype(RC)::value, R1 = (R + 1) % 3, R2 = (R + 2) % 3;
                constexpr bool FULL = decltype(FULLC)::value;
                const bool nxt = FULL || kt + 1 < ntiles, st = FULL || kt + 2 < ntiles;
                int o1 = R1 * BUF, o2 = R2 * BUF;
                asm volatile("" : "+s"(o1), "+s"(o2));
                unsigned* sa = sp + o2;
                unsigned* sb = sa + OPER;
                if (FULL || kt + 4 < ntiles) gload(kbeg + (kt + 4) * BK, ra[R1], rb[R1]);
                // a plane is refilled during the group AFTER its last use: inside a group the scheduler interleaves LDS reads with the
                // MFMAs, and a read into a plane those MFMAs still use would need a second register set (one wave per SIMD has none to spare)
                grp(1, 1);
                if (st) { split_store_slice<A_KC>(sa, ra[R2], 0); split_store_slice<A_KC>(sa, ra[R2], 1); }
                interleave<16>(0);
                __builtin_amdgcn_sched_barrier(0);
                grp(1, 0);
                if (st) { split_store_slice<A_KC>(sa, ra[R2], 2); split_store_slice<A_KC>(sa, ra[R2], 3); }
                interleave<16>(0);
                __builtin_amdgcn_sched_barrier(0);
                grp(0, 1);
                if (nxt) rd_a(o1, 1);
                if (st) { split_store_slice<B_KC>(sb, rb[R2], 0); split_store_slice<B_KC>(sb, rb[R2], 1); }
                interleave<16>(0);
                __builtin_amdgcn_sched_barrier(0);
                grp(2, 0);
                if (nxt) rd_b(o1, 1);
                if (st) { split_store_slice<B_KC>(sb, rb[R2], 2); split_store_slice<B_KC>(sb, rb[R2], 3); }
                interleave<16>(0);
                __builtin_amdgcn_sched_barrier(0);
                grp(0, 2);
                if (nxt) rd_a(o1, 2);
                interleave<16>(0);
                __builtin_amdgcn_sched_barrier(0);
                grp(0, 0);
                if (nxt) rd_b(o1, 2);
                interleave<16>(0);
                __builtin_amdgcn_sched_barrier(0);
                __syncthreads();
                if (nxt) { rd_a(o1, 0); rd_b(o1, 0); }
                __builtin_amdgcn_sched_barrier(0);
            };
            std::integral_constant<int, 0> c0; std::integral_constant<int, 1> c1; std::integral_constant<int, 2> c2;
            int kt = 0;
            for (; kt + 6 < ntiles; kt += 3) {       // all three steps unguarded
                step(c0, std::true_type{}, kt); step(c1, std::true_type{}, kt + 1); step(c2, std::true_type{}, kt + 2);
            }
            for (; kt < ntiles; kt += 3) {
                step(c0, std::false_type{}, kt);
                if (kt + 1 < ntiles) step(c1, std::false_type{}, kt + 1);
                if (kt + 2 < ntiles) step(c2, std::false_type{}, kt + 2);
            }
        }
        __syncthreads();                             // the segment's last LDS reads precede the next segment's first stores
    }
    // Epilogue.  C/D map of the 32x32 MFMA: col(n) = lane & 31, row(m) = (r & 3) + 8 (r >> 2) + 4 (lane >> 5).
    const bool atomic = role.kind == SEG_ATOMIC;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = n0 + wn + j * 32 + lr;
        if (n >= p.N) continue;
        float bsum = 0.f;
        if (role.add_bias) {
            if (p.bias0) bsum += p.bias0[(long)bz * p.sBias0 + n];
            if (p.bias1) bsum += p.bias1[(long)bz * p.sBias1 + n];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
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

// classic grid: (tiles, 1, batch * splitk); every workgroup owns one tile and one k-slice
template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(THREADS, 1) void gemm_big_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned dyn_[];
    const int bz = blockIdx.z / p.splitk, kz = blockIdx.z % p.splitk;
    // XCD-aware tile order (blocks are dispatched round-robin over the 8 XCDs): each XCD works on a contiguous run of tiles, so the
    // N-tiles of one M-panel share the A panel in that XCD's L2
    int tile = blockIdx.x;
    if (p.swz) tile = (tile & 7) * (gridDim.x >> 3) + (tile >> 3);
    const int kbeg = kz * p.kper;
    segment<A_KC, B_KC>(p, dyn_, bz, (tile / p.gx) * T, (tile % p.gx) * T, kbeg, min(p.K, kbeg + p.kper),
                        p.atomic ? seg_atomic(kz == 0) : seg_store(kz == 0));
}

// several GEMMs of one operand layout in one launch: the k-iterations of all problems laid end to end and cut into W equal runs (stream-K
// across the problems, as gemm_group_kernel); whole tiles are stored (or added) plainly, partial tiles accumulate with atomics onto
// outputs the caller zeroed
template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(THREADS, 1) void gemm_big_group_kernel(GemmGroupParams g) {
    extern __shared__ __attribute__((aligned(16))) unsigned dyn_[];
    const int W = gridDim.x;
    const int w = (W % 8 == 0 && g.xcd_swz) ? (int)(blockIdx.x & 7) * (W >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    const long total = g.first[g.n];
    const long per = (total + W - 1) / W;
    long i0 = (long)w * per, i1 = min(i0 + per, total);
    int pi = 0;
    while (pi + 1 < g.n && g.first[pi + 1] <= i0) ++pi;
    while (i0 < i1) {
        GemmParams p = g.prob[pi];
        const long pend = min(i1, g.first[pi + 1]);
        long l0 = i0 - g.first[pi];
        const long l1 = pend - g.first[pi];
        while (l0 < l1) {
            const int tl = (int)(l0 / p.kt);
            const int it0 = (int)(l0 % p.kt), it1 = (int)min((long)p.kt, it0 + (l1 - l0));
            const bool whole = it0 == 0 && it1 == p.kt;
            segment<A_KC, B_KC>(p, dyn_, 0, (tl / p.gx) * T, (tl % p.gx) * T, it0 * BK, min(p.K, it1 * BK),
                                whole ? SegRole{SEG_STORE, 0, 0, 0, false} : seg_atomic(false));
            l0 += it1 - it0;
        }
        i0 = pend;
        ++pi;
    }
}

template <class Kern>
static int allow_lds(Kern kernel) {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES) == hipSuccess;
}
template <bool A_KC, bool B_KC>
static int launch(const GemmParams& p, dim3 grid, hipStream_t stream) {
    static const int ready = allow_lds(gemm_big_kernel<A_KC, B_KC>);
    LAS_REQUIRE(ready, "dynamic LDS of the 256-tile GEMM");
    hipLaunchKernelGGL((gemm_big_kernel<A_KC, B_KC>), grid, dim3(THREADS), SMEM_BYTES, stream, p);
    LAS_LAUNCH_CHECK();
    return LAS_OK;
}
template <bool A_KC, bool B_KC>
static int launch_group(const GemmGroupParams& g, dim3 grid, hipStream_t stream) {
    static const int ready = allow_lds(gemm_big_group_kernel<A_KC, B_KC>);
    LAS_REQUIRE(ready, "dynamic LDS of the 256-tile GEMM");
    hipLaunchKernelGGL((gemm_big_group_kernel<A_KC, B_KC>), grid, dim3(THREADS), SMEM_BYTES, stream, g);
    LAS_LAUNCH_CHECK();
    return LAS_OK;
}

static int device_cus() {
    static int cus = -1;
    if (cus < 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 0;
        cus = n;
    }
    return cus;
}
static bool aligned(const float* ptr, long ld, long bs) { return ((uintptr_t)ptr % 16 == 0) && (ld % 4 == 0) && (bs % 4 == 0); }

// share of the padded tile grid that is real output (a 256-tile on N = 160 multiplies 37 % padding)
static double fill(int M, int N) { return (double)M * N / ((double)cdiv(M, T) * T * (double)cdiv(N, T) * T); }

}  // namespace big

// The 256-tile kernel takes a GEMM when its tiles are mostly real output, K is long enough to amortise the 256 KB epilogue of a tile, and a
// k-split exists that fills the chip's one-workgroup-per-CU slots; LAS_ERR_UNSUPPORTED hands the problem back to the 128-tile kernel
// (no error string: it is the normal way out).  GEMM_BIG = 0 switches it off (A/B).
int gemm_big(const GemmDesc& d, hipStream_t stream) {
    using namespace big;
    const long mode = opt_get(OPT_GEMM_BIG);
    if (mode == 0 || d.planes || gemm_get_arith() != 1) return LAS_ERR_UNSUPPORTED;
    const int cus = device_cus();
    if (cus <= 0 || d.M <= 0 || d.N <= 0 || d.K < 8 * BK) return LAS_ERR_UNSUPPORTED;
    if (fill(d.M, d.N) < (mode >= 2 ? 0.30 : 0.85)) return LAS_ERR_UNSUPPORTED;
    const int batch = d.batch > 0 ? d.batch : 1;
    const int gx = cdiv(d.N, T), gy = cdiv(d.M, T);
    const long tiles = (long)gx * gy * batch;
    const int kt = cdiv(d.K, BK);
    // k-split: the candidate with the best slot utilisation, at least 8 k-tiles per slice; a relu epilogue needs the whole sum in one place
    const bool can_zero = d.accumulate || d.c_zeroed || d.ldc == d.N || batch == 1;
    // Measured (tools/ubench_gemm_big.py, profiles/r05_gemm_big.txt): a 256-tile costs ~28 us of prologue + epilogue (256 KB of output per
    // workgroup, nothing else resident on the CU to hide it) plus 2.4 us per k-tile, so it beats two co-resident 128-tile workgroups only
    // where whole rounds of tiles fill the chip and K is long: automatic mode takes >= 90 % slot utilisation with >= 64 k-tiles per slice
    // (4096^3: 223 against 199 TF; 8192 x 1024 x 1024 x 2: 195 against 180 TF; the training step's own shapes — 100 to 400 such tiles, K <=
    // 2048 — do not qualify and stay on the 128-tile kernel, which is 10-50 % faster there).  Mode 2 (A/B, tests): whenever legal.
    int best = 0; double best_u = 0.0;
    const int smax = (d.relu || !can_zero || d.A2 != nullptr) ? 1 : 8;
    for (int s = 1; s <= smax; ++s) {
        if (d.splitk > 1 && s != d.splitk) continue;
        if (s > 1 && kt / s < (mode >= 2 ? 8 : 64)) break;
        const long wgs = tiles * s;
        const double u = (double)wgs / ((double)cdiv(wgs, cus) * cus) - 0.05 * (s - 1);      // every extra slice re-reads and re-writes C once
        if (u > best_u + 1e-9) { best_u = u; best = s; }
    }
    if (best == 0) return LAS_ERR_UNSUPPORTED;
    if (mode < 2 && (best_u < 0.90 || kt / best < 64)) return LAS_ERR_UNSUPPORTED;
    int splitk = best;
    int kper = cdiv(cdiv(d.K, splitk), BK) * BK;
    splitk = std::max(1, cdiv(d.K, kper));
    GemmParams p;
    memset(&p, 0, sizeof(p));
    p.A = d.A; p.B = d.B; p.C = d.C; p.bias0 = d.bias0; p.bias1 = d.bias1; p.A2 = d.A2; p.B2 = d.B2; p.K1 = d.K1;
    p.M = d.M; p.N = d.N; p.K = d.K; p.lda = d.lda; p.ldb = d.ldb; p.ldc = d.ldc;
    p.sA = d.sA; p.sB = d.sB; p.sC = d.sC; p.sBias0 = d.sBias0; p.sBias1 = d.sBias1;
    p.a_vec = aligned(d.A, d.lda, d.sA) && (d.A2 == nullptr || aligned(d.A2, d.lda, 0));
    p.b_vec = aligned(d.B, d.ldb, d.sB) && (d.B2 == nullptr || aligned(d.B2, d.ldb, 0));
    p.gx = gx; p.gy = gy; p.kt = kt; p.accumulate = d.accumulate; p.relu = d.relu;
    p.splitk = splitk; p.kper = kper; p.atomic = splitk > 1;
    if (p.atomic && !d.accumulate && !d.c_zeroed) {      // split-K partials are summed with atomics: C must start from zero
        if (d.ldc == d.N) {
            // (batches with gaps between their outputs — interleaved with another GEMM's — are zeroed block by block: the gaps may hold live data)
            if (batch > 1 && d.sC != (long)d.M * d.N)
                LAS_HIP_CHECK(hipMemset2DAsync(d.C, sizeof(float) * (size_t)d.sC, 0, sizeof(float) * (size_t)d.M * d.N, (size_t)batch, stream));
            else
                LAS_HIP_CHECK(hipMemsetAsync(d.C, 0, sizeof(float) * ((size_t)(batch - 1) * d.sC + (size_t)d.M * d.N), stream));
        } else {
            LAS_HIP_CHECK(hipMemset2DAsync(d.C, sizeof(float) * d.ldc, 0, sizeof(float) * d.N, (size_t)d.M, stream));
        }
    }
    p.swz = ((gx * gy) % 8 == 0) && (gx * gy >= 64);
    path_note(PATH_GEMM, "split256");
    const dim3 grid(gx * gy, 1, batch * splitk);
    if (d.a_kc && d.b_kc) return launch<true, true>(p, grid, stream);
    if (d.a_kc && !d.b_kc) return launch<true, false>(p, grid, stream);
    if (!d.a_kc && d.b_kc) return launch<false, true>(p, grid, stream);
    return launch<false, false>(p, grid, stream);
}

// grouped form (the weight-gradient contractions of a backward pass): same conditions as gemm_f32_group plus well-filled 256-tiles
int gemm_big_group(const GemmDesc* ds, int n, hipStream_t stream) {
    using namespace big;
    // (measured slower than the 128-tile grouped launch on every weight-gradient group of the training step — L1 group 264 against 191 us —:
    // a run of k-iterations ends in a 256 KB atomic epilogue; only mode 2 takes it)
    if (opt_get(OPT_GEMM_BIG) < 2 || gemm_get_arith() != 1 || n < 1 || n > GROUP_MAX) return LAS_ERR_UNSUPPORTED;
    const int cus = device_cus();
    if (cus <= 0) return LAS_ERR_UNSUPPORTED;
    double real = 0.0, padded = 0.0;
    for (int i = 0; i < n; ++i) {
        const GemmDesc& d = ds[i];
        if (d.planes || d.batch > 1 || d.relu || d.bias0 || d.bias1 || !(d.c_zeroed || d.accumulate) || d.a_kc != ds[0].a_kc || d.b_kc != ds[0].b_kc ||
            d.A2 != nullptr || d.M <= 0 || d.N <= 0 || d.K < 8 * BK)
            return LAS_ERR_UNSUPPORTED;
        real += (double)d.M * d.N * d.K;
        padded += (double)cdiv(d.M, T) * T * (double)cdiv(d.N, T) * T * d.K;
    }
    if (real / padded < 0.50) return LAS_ERR_UNSUPPORTED;      // (flop-weighted: a vocabulary-sized member costs little inside a group of big ones)
    GemmGroupParams g;
    g.n = n; g.first[0] = 0; g.xcd_swz = (int)opt_get(OPT_GEMM_XCD_SWZ);
    g.sk_part = nullptr; g.sk_flag = nullptr; g.sk_err = nullptr; g.sk_id = 0; g.call_err = nullptr;
    for (int i = 0; i < n; ++i) {
        const GemmDesc& d = ds[i];
        GemmParams& p = g.prob[i];
        memset(&p, 0, sizeof(p));
        p.A = d.A; p.B = d.B; p.C = d.C; p.M = d.M; p.N = d.N; p.K = d.K; p.lda = d.lda; p.ldb = d.ldb; p.ldc = d.ldc;
        p.a_vec = aligned(d.A, d.lda, 0); p.b_vec = aligned(d.B, d.ldb, 0);
        p.gx = cdiv(d.N, T); p.gy = cdiv(d.M, T); p.kt = std::max(1, cdiv(d.K, BK));
        p.splitk = 1; p.kper = d.K; p.accumulate = d.accumulate;
        g.first[i + 1] = g.first[i] + (long)p.gx * p.gy * p.kt;
    }
    if (g.first[n] < 8L * cus) return LAS_ERR_UNSUPPORTED;      // fewer than 8 k-iterations per workgroup: the 128-tile schedule is finer
    path_note(PATH_GEMM, "split256");
    if (ds[0].a_kc && ds[0].b_kc) return launch_group<true, true>(g, dim3(cus), stream);
    if (ds[0].a_kc && !ds[0].b_kc) return launch_group<true, false>(g, dim3(cus), stream);
    if (!ds[0].a_kc && ds[0].b_kc) return launch_group<false, true>(g, dim3(cus), stream);
    return launch_group<false, false>(g, dim3(cus), stream);
}

}  // namespace las
