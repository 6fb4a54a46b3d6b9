// split_common.hpp - device helpers shared by the two translation units of the split arithmetic (split_arith.hip: the contractions per hyperedge;
// split_node.hip: the node-level contractions and linear maps): wave roles and their issue priority, the three-bf16 and two-fp16 operand splits, the
// transposed-read fragment helpers, row tiles by node type.  Everything here is inline device code in an anonymous namespace.
#pragma once
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "ablate.hpp"
#include "common.hpp"
#include "split.hpp"

namespace {

typedef short v8s __attribute__((ext_vector_type(8)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));

constexpr int kSplitTE = 32;            // hyperedges per tile
constexpr int kSplitRanges = 128;       // contiguous tile ranges at d = 128 (x 2 column halves = 256 workgroups, one per CU)
constexpr int kSplitThreads = 512;

// Issue priority of the two wave roles (s_setprio, 0 .. 3; A/B: tools/ab_variant.sh NAME -DIHG_SERVICE_PRIO=n -DIHG_MATRIX_PRIO=m).  The
// service waves are the second-dispatched half of the workgroup - the arbitration loser at equal priority (oldest first).
#ifndef IHG_SERVICE_PRIO
#define IHG_SERVICE_PRIO 3           // measured at C3 (same box, us): forward 1,874 -> 1,766, weight gradients 1,454 -> 1,342, member gradients 1,870 -> 1,834;
#endif                               // priority 1: 1,787 / 1,366 / 1,827; matrix waves at 1 instead: no change (profiles/r3/ab_priority.txt)
#ifndef IHG_MATRIX_PRIO
#define IHG_MATRIX_PRIO 0
#endif
// Left to the scheduler, an LDS fragment read that the source issues a step ahead is sunk to just in front of its first MFMA (the weight
// planes hold 192 registers, and shortening live ranges wins): every step then sits out the LDS latency.  A scheduling barrier on both sides
// of a step's MFMA group keeps the reads of step s + 1 in front of the MFMAs of step s.  -DIHG_NO_PIN: the scheduler's order (A/B).
#ifdef IHG_NO_PIN
#define IHG_PIN_ORDER()
#else
#define IHG_PIN_ORDER() __builtin_amdgcn_sched_barrier(0)
#endif

__device__ __forceinline__ void role_priority(bool service) {
    if (service) {
        if (IHG_SERVICE_PRIO) __builtin_amdgcn_s_setprio(IHG_SERVICE_PRIO);
    } else {
        if (IHG_MATRIX_PRIO) __builtin_amdgcn_s_setprio(IHG_MATRIX_PRIO);
    }
}

__device__ __forceinline__ unsigned pack_hi(float a, float b) {          // {top half of b, top half of a}
    return __builtin_amdgcn_perm(__float_as_uint(b), __float_as_uint(a), 0x07060302u);
}
// row r of a table whose rows are ld floats apart (ld < 2^31: the *_ok() predicates): ONE v_mad_u64_u32 where the 64 x 64-bit product of
// an int64 leading dimension costs three quarter-rate multiplies per row - service-wave cycles the matrix pipe waits for
__device__ __forceinline__ const float* row_at(const float* base, int32_t r, uint32_t ld) {
    return base + static_cast<uint64_t>(static_cast<uint32_t>(r)) * static_cast<uint64_t>(ld);
}
__device__ __forceinline__ float top16(float x) { return __uint_as_float(__float_as_uint(x) & 0xffff0000u); }

// Two fp32 values -> one dword of each of the three bf16 planes (low half: xa's term, high half: xb's): mask, subtract, mask, subtract, and
// three v_perm_b32 that pack the top halves - 11 vector instructions per pair.  (Tried: the remainder x - top16(x) as ONE
// v_dot2c_f32_bf16 of the packed plane with the selector {-1, 0} accumulated onto x, 7 instructions per pair and bit-identical planes
// - tools/dot2_probe.hip - but the dot instruction is not a full-rate one: forward +2 %, weight gradients +8 %.)
__device__ __forceinline__ void split_pair(float xa, float xb, unsigned (&w)[3]) {
    const float ra = xa - top16(xa), rb = xb - top16(xb);
    const float la = ra - top16(ra), lb = rb - top16(rb);
    w[0] = pack_hi(xa, xb);
    w[1] = pack_hi(ra, rb);
    w[2] = pack_hi(la, lb);
}

// eight consecutive k of one row / column -> the three bf16 planes of that MFMA fragment
struct Planes {
    v4u p[3];
};
__device__ __forceinline__ Planes split8(v4f x0, v4f x1) {
    Planes out;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const v4f x = half == 0 ? x0 : x1;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            unsigned w[3];
            split_pair(x[2 * i], x[2 * i + 1], w);
#pragma unroll
            for (int p = 0; p < 3; ++p) out.p[p][2 * half + i] = w[p];
        }
    }
    return out;
}

// ------------------------------------------------------------------------------------------------
// fp32 through TWO fp16 terms (node-level contraction, d = 128).  An fp32 value, first multiplied by a power of two that brings its row's (or its weight column's)
// largest magnitude to [2^13, 2^14), is hi + lo with hi = fp16(x) (11 significand bits, round to nearest even) and lo = fp16(x - hi) (the difference is exact in fp32;
// lo carries the next 11 bits wherever |x| >= 2^-3, i.e. within 2^-17 of the row's largest entry - below that it is an fp16 subnormal with an ABSOLUTE error of 2^-25,
// 2^-38 of the row's largest entry).  A product a b is then three partial products - hi lo + lo hi + hi hi, each exact in fp32, accumulated by
// v_mfma_f32_16x16x32_f16 - instead of six: half the matrix-pipe time, and the split costs 4 vector instructions per element instead of 5.5.  What is left out
// (lo lo, and the rounding of lo) is bounded by 3 x 2^-22 |a b|; rounding to nearest, not truncation, so there is no one-sided bias (numpy emulation of both
// schemes against float64 on normal, wide-range (2e-4 .. 3e3), one-huge-many-tiny and low-16-bits-set operands: 0.8 - 1.8e-7 per-row against 1.0 - 2.6e-7 for the
// three-bf16 scheme).  The scales are powers of two: applying and removing them is exact.  fp16's narrow exponent is what the scaling is for: scaled magnitudes
// stay below 2^14, partial products below 2^28, sums over 1,024 of them below 2^38.
// ------------------------------------------------------------------------------------------------
typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef _Float16 v2h __attribute__((ext_vector_type(2)));

// 2^(13 - floor(log2 m)) and its inverse for a magnitude m >= 0 (zero, denormals and magnitudes beyond 2^100 either way: exponent clamped - such rows are all-zero
// or all-huge relative to anything they meet)
__device__ __forceinline__ float scale_up_for(float m, float& inverse) {
    int e = static_cast<int>((__float_as_uint(m) >> 23) & 0xffu);         // biased exponent of the largest magnitude
    e = e < 27 ? 27 : (e > 227 ? 227 : e);
    inverse = __uint_as_float(static_cast<unsigned>(e - 13) << 23);       // 2^(e - 127 - 13)
    return __uint_as_float(static_cast<unsigned>(267 - e) << 23);         // 2^(13 - (e - 127))
}
// two scaled fp32 values -> one dword of each plane (low half: xa's term)
__device__ __forceinline__ void split_pair_h2(float xa, float xb, unsigned& hi, unsigned& lo) {
    const v2h h = v2h{static_cast<_Float16>(xa), static_cast<_Float16>(xb)};
    const v2h l = v2h{static_cast<_Float16>(xa - static_cast<float>(h[0])), static_cast<_Float16>(xb - static_cast<float>(h[1]))};
    hi = __builtin_bit_cast(unsigned, h);
    lo = __builtin_bit_cast(unsigned, l);
}
// The largest magnitude of a row piece and of the row, for scale_up_for().  fmaxf() costs a canonicalising v_max per operand under IEEE mode (a third of the
// instructions of the maximum); v_max3_f32 with |.| modifiers takes two values per instruction.  m >= 0 throughout, so across lanes the maximum is taken on the bit
// patterns (unsigned order = float order) with DPP moves inside the row's lanes - no LDS round trip (ds_bpermute, three or four in series, each waited for in order).
__device__ __forceinline__ float abs_max3(float a, float b, float m) {
    float r;
    asm("v_max3_f32 %0, |%1|, |%2|, %3" : "=v"(r) : "v"(a), "v"(b), "v"(m));
    return r;
}
__device__ __forceinline__ float abs_max_of(v4f a, v4f b, float m) { return abs_max3(b[2], b[3], abs_max3(b[0], b[1], abs_max3(a[2], a[3], abs_max3(a[0], a[1], m)))); }
template <int CTRL>
__device__ __forceinline__ unsigned dpp_max_u32(unsigned u) {
    const unsigned other = static_cast<unsigned>(__builtin_amdgcn_update_dpp(0, static_cast<int>(u), CTRL, 0xf, 0xf, true));
    return u > other ? u : other;
}
// maximum over the 8 (16) consecutive lanes that hold one row: quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror (, row_mirror)
template <int LANES>
__device__ __forceinline__ float row_lanes_max(float m) {
    static_assert(LANES == 8 || LANES == 16, "a row's threads are 8 or 16 consecutive lanes");
    unsigned u = __float_as_uint(m);
    u = dpp_max_u32<0xB1>(u);
    u = dpp_max_u32<0x4E>(u);
    u = dpp_max_u32<0x141>(u);
    if (LANES == 16) u = dpp_max_u32<0x140>(u);
    return __uint_as_float(u);
}

// partial products in accumulation order (A plane, B plane): smallest first; planes: 0 = hi, 1 = lo
__device__ constexpr int kTermA2[3] = {0, 1, 0};
__device__ constexpr int kTermB2[3] = {1, 0, 0};

// partial products in accumulation order (A plane, B plane): smallest first
__device__ constexpr int kTermA[6] = {0, 2, 1, 0, 1, 0};
__device__ constexpr int kTermB[6] = {2, 0, 1, 1, 0, 0};

__device__ __forceinline__ int tr_swizzle(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }

__device__ __forceinline__ v8s read_tr_fragment(const unsigned char* lo, const unsigned char* hi) {
    typedef short v4s __attribute__((ext_vector_type(4)));
    const v4s a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)lo);
    const v4s b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)hi);
    return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}

// (both translation units: the member gradients scale their weight columns with it)
// scale of every output column of the node-level linear maps (= weight row c, or column c with transpose): wsc[type][c] = scale_up_for(max_k |w(c, k)|), winv its inverse.
// One wave per (type, c).
__global__ __launch_bounds__(kBlockThreads) void dense_weight_scales_kernel(const float* __restrict__ w, int64_t ld_w, int64_t type_stride, int n_types, int d, int transpose,
                                                                            float* __restrict__ wsc, float* __restrict__ winv) {
    const int lane = threadIdx.x & 63;
    const int64_t unit = global_wave_id();
    if (unit >= static_cast<int64_t>(n_types) * d) return;
    const int type = static_cast<int>(unit) / d, c = static_cast<int>(unit) % d;
    const float* wt = w + type * type_stride;
    float m = 0.f;
    for (int k = lane; k < d; k += kWave) m = fmaxf(m, fabsf(transpose == 0 ? wt[static_cast<int64_t>(c) * ld_w + k] : wt[static_cast<int64_t>(k) * ld_w + c]));
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    float inv;
    const float sc = scale_up_for(m, inv);
    if (lane == 0) {
        wsc[unit] = sc;
        winv[unit] = inv;
    }
}

// leading dimensions the kernels' 32-bit row arithmetic takes (row_at): 16-byte rows, below 2^31 floats
inline bool ld_ok(int64_t ld) { return ld > 0 && ld % 4 == 0 && ld < (int64_t{1} << 31); }

}  // namespace
