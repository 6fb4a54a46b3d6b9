// common.hpp - shared pieces of libihgnn_hip (gfx950 only, wave = 64).  Everything here has internal linkage per translation unit
// except the thread-local error buffer, which lives in host.cpp.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "ihgnn_hip.h"

extern thread_local char ihg_error_buffer[512];

namespace {

inline int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(ihg_error_buffer, sizeof(ihg_error_buffer), fmt, ap);
    va_end(ap);
    return code;
}

inline int check_launch(const char* what) {
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) return fail(IHG_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(err));
    return IHG_OK;
}

constexpr int kWave = 64;
constexpr int kBlockThreads = 256;
constexpr int kWavesPerBlock = kBlockThreads / kWave;
constexpr int kMaxBlocks = 256 * 8;   // 256 CUs x 8 resident 256-thread blocks: grid-stride beyond that

inline int grid_for_waves(int64_t waves) {
    int64_t blocks = (waves + kWavesPerBlock - 1) / kWavesPerBlock;
    if (blocks < 1) blocks = 1;
    if (blocks > kMaxBlocks) blocks = kMaxBlocks;
    return static_cast<int>(blocks);
}

// ------------------------------------------------------------------------------------------------
// Row fragments: VEC = 4 -> one float4 (16 B) per lane per row, VEC = 1 -> one float.
// ------------------------------------------------------------------------------------------------
template <int VEC> struct Frag;
template <> struct Frag<4> {
    float4 v;
    __device__ static Frag zero() { return {make_float4(0.f, 0.f, 0.f, 0.f)}; }
    __device__ static Frag load(const float* p) { return {*reinterpret_cast<const float4*>(p)}; }
    // a row that is read once and not again soon: non-temporal
    __device__ static Frag load_stream(const float* p) {
        typedef float vec4 __attribute__((ext_vector_type(4)));
        const vec4 t = __builtin_nontemporal_load(reinterpret_cast<const vec4*>(p));
        return {make_float4(t[0], t[1], t[2], t[3])};
    }
    __device__ void store(float* p) const { *reinterpret_cast<float4*>(p) = v; }
    // write-once output that nobody re-reads soon: non-temporal, so the stream does not push the gathered table out of the caches
    __device__ void store_stream(float* p) const {
        typedef float vec4 __attribute__((ext_vector_type(4)));
        __builtin_nontemporal_store(vec4{v.x, v.y, v.z, v.w}, reinterpret_cast<vec4*>(p));
    }
    __device__ void add_scaled(const Frag& o, float s) { v.x += s * o.v.x; v.y += s * o.v.y; v.z += s * o.v.z; v.w += s * o.v.w; }
    __device__ void add(const Frag& o) { v.x += o.v.x; v.y += o.v.y; v.z += o.v.z; v.w += o.v.w; }
    __device__ void mul(float s) { v.x *= s; v.y *= s; v.z *= s; v.w *= s; }
    __device__ void div(float s) { v.x /= s; v.y /= s; v.z /= s; v.w /= s; }
};
template <> struct Frag<1> {
    float v;
    __device__ static Frag zero() { return {0.f}; }
    __device__ static Frag load(const float* p) { return {*p}; }
    __device__ static Frag load_stream(const float* p) { return {__builtin_nontemporal_load(p)}; }
    __device__ void store(float* p) const { *p = v; }
    __device__ void store_stream(float* p) const { __builtin_nontemporal_store(v, p); }
    __device__ void add_scaled(const Frag& o, float s) { v += s * o.v; }
    __device__ void add(const Frag& o) { v += o.v; }
    __device__ void mul(float s) { v *= s; }
    __device__ void div(float s) { v /= s; }
};

__device__ __forceinline__ int64_t global_wave_id() {
    return static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + (threadIdx.x >> 6);
}
__device__ __forceinline__ int64_t global_wave_count() { return static_cast<int64_t>(gridDim.x) * kWavesPerBlock; }

typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));

// 16-byte store of a value that is written once and not re-read by this kernel (streams of hyperedge rows): non-temporal
__device__ __forceinline__ void store_stream4(float* p, v4f v) { __builtin_nontemporal_store(v, reinterpret_cast<v4f*>(p)); }

constexpr int kRowPad = 4;       // floats; breaks the power-of-two row stride for ds_read_b128

// row of accumulator register r in lane `lane` of a 32x32 MFMA result (column = lane & 31)
__device__ __forceinline__ int acc_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// Sum `n_slabs` slabs of `total` floats each at element `idx`: the workgroup's 4 waves take the slabs round-robin with 8
// loads in flight per lane, then combine through LDS in wave order (fixed tree => bitwise reproducible).
// Must be called by all 256 threads of a block with idx = blockIdx.x * 64 + (threadIdx.x & 63); returns the sum to wave 0.
__device__ __forceinline__ float slab_sum(const float* __restrict__ slabs, int n_slabs, int64_t total, int64_t idx, bool live) {
    __shared__ float part[kWavesPerBlock][kWave];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float a[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) a[u] = 0.f;
    if (live) {
        int sl = wave;
        for (; sl + 7 * kWavesPerBlock < n_slabs; sl += 8 * kWavesPerBlock) {
#pragma unroll
            for (int u = 0; u < 8; ++u) a[u] += slabs[static_cast<int64_t>(sl + u * kWavesPerBlock) * total + idx];
        }
        for (; sl < n_slabs; sl += kWavesPerBlock) a[0] += slabs[static_cast<int64_t>(sl) * total + idx];
    }
    part[wave][lane] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    __syncthreads();
    const float sum = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
    __syncthreads();
    return sum;
}


}  // namespace

// Rows of a node-level matrix that is not ONE contiguous [N, ld] block: the rows of every node type (users, queries, items) start at an address of
// their own - the input features X0 are rows 1.. of the user table, the bag means of the queries and rows 1.. of the item table (Models/RawGnn.py:112,
// Models/EmbeddingLayers.py:70-79), and assembling them costs four [N, d]-sized copies per training step.  p[t] + v * ld addresses GLOBAL node row v of
// type t: p[t] = (first row of type t) - type_begin[t] * ld, a virtual base.  A contiguous matrix is three equal pointers.
struct TypedRows {
    const float* p[3];
};
struct TypedRowsOut {
    float* p[3];
};

namespace {
// p[t] by compares, not by indexing: a per-lane index into a by-value kernel argument makes the compiler fetch the pointer from the argument segment with a vector load
// (and wait for every outstanding request in front of it)
__device__ __forceinline__ const float* typed_base(const TypedRows& r, int t) { return t == 0 ? r.p[0] : (t == 1 ? r.p[1] : r.p[2]); }
__device__ __forceinline__ float* typed_base(const TypedRowsOut& r, int t) { return t == 0 ? r.p[0] : (t == 1 ? r.p[1] : r.p[2]); }
inline TypedRows typed_rows(const float* base) { return TypedRows{{base, base, base}}; }
inline TypedRowsOut typed_rows_out(float* base) { return TypedRowsOut{{base, base, base}}; }
inline TypedRows typed_rows(const float* const* first_rows, const int64_t* type_begin, int64_t ld) {
    TypedRows t;
    for (int k = 0; k < 3; ++k) t.p[k] = reinterpret_cast<const float*>(reinterpret_cast<uintptr_t>(first_rows[k]) - static_cast<uintptr_t>(type_begin[k] * ld * 4));
    return t;
}
inline TypedRowsOut typed_rows_out(float* const* first_rows, const int64_t* type_begin, int64_t ld) {
    TypedRowsOut t;
    for (int k = 0; k < 3; ++k) t.p[k] = reinterpret_cast<float*>(reinterpret_cast<uintptr_t>(first_rows[k]) - static_cast<uintptr_t>(type_begin[k] * ld * 4));
    return t;
}

// p[0 .. n) = 0 (small fills that would otherwise be torch launches inside a training step: padding rows of gradients, bias gradients, row masks)
__global__ __launch_bounds__(kBlockThreads) void zero_floats_kernel(float* __restrict__ p, int64_t n) {
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlockThreads + threadIdx.x; i < n; i += static_cast<int64_t>(gridDim.x) * kBlockThreads) p[i] = 0.f;
}
inline void launch_zero_floats(float* p, int64_t n, hipStream_t s) {
    if (n <= 0) return;
    const int grid = static_cast<int>(std::min<int64_t>((n + kBlockThreads - 1) / kBlockThreads, kMaxBlocks));
    hipLaunchKernelGGL(zero_floats_kernel, dim3(grid), dim3(kBlockThreads), 0, s, p, n);
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
inline bool mfma_dim(int dim) { return dim == 32 || dim == 64 || dim == 128 || dim == 256; }

}  // namespace
