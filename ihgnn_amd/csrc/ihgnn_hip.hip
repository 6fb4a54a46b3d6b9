// libihgnn_hip: hand-written gfx950 (MI355X / CDNA4) kernels for the IHGNN hypergraph message-passing path,
// behind the C ABI of include/ihgnn_hip.h.  Wave = 64 lanes everywhere; no CUDA-compat paths.
//
// Layout idea shared by the two HBM-bound kernels (K5 node->hyperedge, K7 hyperedge->node):
//   a feature row of `dim` floats is owned by a GROUP of G = dim/4 lanes (16 B per lane, so one group
//   instruction moves one whole row and one wave instruction moves 64/G rows = 1 KiB), every lane keeps its
//   own 4 columns in registers for the whole reduction (no cross-lane adds), and the only cross-lane traffic
//   is the index stream: indices are fetched once per wave with a single coalesced load and handed to their
//   group with wavefront shuffles (ds_bpermute), so the dependent row gathers of several hyperedges are in
//   flight together.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "ihgnn_hip.h"

namespace {

thread_local char g_error[512] = "";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
    return code;
}

int check_launch(const char* what) {
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
    __device__ void store(float* p) const { *reinterpret_cast<float4*>(p) = v; }
    __device__ void add_scaled(const Frag& o, float s) { v.x += s * o.v.x; v.y += s * o.v.y; v.z += s * o.v.z; v.w += s * o.v.w; }
    __device__ void add(const Frag& o) { v.x += o.v.x; v.y += o.v.y; v.z += o.v.z; v.w += o.v.w; }
    __device__ void mul(float s) { v.x *= s; v.y *= s; v.z *= s; v.w *= s; }
    __device__ void div(float s) { v.x /= s; v.y /= s; v.z /= s; v.w /= s; }
};
template <> struct Frag<1> {
    float v;
    __device__ static Frag zero() { return {0.f}; }
    __device__ static Frag load(const float* p) { return {*p}; }
    __device__ void store(float* p) const { *p = v; }
    __device__ void add_scaled(const Frag& o, float s) { v += s * o.v; }
    __device__ void add(const Frag& o) { v += o.v; }
    __device__ void mul(float s) { v *= s; }
    __device__ void div(float s) { v /= s; }
};

__device__ __forceinline__ int64_t global_wave_id() {
    return static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + (threadIdx.x >> 6);
}
__device__ __forceinline__ int64_t global_wave_count() { return static_cast<int64_t>(gridDim.x) * kWavesPerBlock; }

// ================================================================================================
// K5  node -> hyperedge gather-sum
//   G lanes own one hyperedge row; a wave works on EPW = (64/G)*U consecutive hyperedges per iteration:
//   one coalesced load brings their 3*EPW member ids (<= 64 ints), shuffles hand each group its ids, then
//   3*U independent row gathers per lane are issued before the first add.  Group g takes hyperedges
//   e0 + g + (64/G)*t so that each store instruction of the wave writes (64/G) consecutive rows = 1 KiB.
// ================================================================================================
template <int VEC, int G, int U>
__global__ __launch_bounds__(kBlockThreads) void edge_gather_sum_kernel(
    const float* __restrict__ src, int64_t ld_src, const int32_t* __restrict__ i3,
    const float* __restrict__ node_scale, const float* __restrict__ bias, float alpha,
    float* __restrict__ out, int64_t ld_out, int64_t n_edges, int dim_vec) {
    constexpr int GPW = kWave / G;
    constexpr int EPW = GPW * U;
    static_assert(EPW * 3 <= kWave, "member ids of one wave iteration must fit one coalesced load");
    const int lane = threadIdx.x & (kWave - 1);
    const int lig = lane & (G - 1);
    const int grp = lane / G;
    const int64_t n_ids = n_edges * 3;

    for (int64_t e0 = global_wave_id() * EPW; e0 < n_edges; e0 += global_wave_count() * EPW) {
        const int64_t pos = e0 * 3 + lane;
        const bool have = lane < EPW * 3 && pos < n_ids;
        const int my_id = have ? i3[pos] : 0;
        const float my_scale = (node_scale != nullptr && have) ? node_scale[my_id] : 1.f;

        int ids[U][3];
        float sc[U][3];
#pragma unroll
        for (int t = 0; t < U; ++t) {
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                const int from = (grp + GPW * t) * 3 + m;
                ids[t][m] = __shfl(my_id, from);
                sc[t][m] = __shfl(my_scale, from);
            }
        }
        for (int c = lig; c < dim_vec; c += G) {
            Frag<VEC> rows[U][3];
#pragma unroll
            for (int t = 0; t < U; ++t) {
                const bool live = e0 + grp + GPW * t < n_edges;
#pragma unroll
                for (int m = 0; m < 3; ++m)
                    rows[t][m] = live ? Frag<VEC>::load(src + static_cast<int64_t>(ids[t][m]) * ld_src + c * VEC)
                                      : Frag<VEC>::zero();
            }
            Frag<VEC> b = bias != nullptr ? Frag<VEC>::load(bias + c * VEC) : Frag<VEC>::zero();
#pragma unroll
            for (int t = 0; t < U; ++t) {
                const int64_t e = e0 + grp + GPW * t;
                if (e >= n_edges) continue;
                Frag<VEC> acc = Frag<VEC>::zero();
                acc.add_scaled(rows[t][0], sc[t][0]);      // (u + q) + i, the order of a row-major SpMM row
                acc.add_scaled(rows[t][1], sc[t][1]);
                acc.add_scaled(rows[t][2], sc[t][2]);
                acc.mul(alpha);
                acc.add(b);
                acc.store(out + e * ld_out + c * VEC);
            }
        }
    }
}

// ================================================================================================
// K7  hyperedge -> node segment-sum (also: EmbeddingBag mean forward/backward, scatter-add backward)
//   G lanes own one output row and walk its id list in chunks of G ids: one coalesced id load per chunk,
//   shuffles broadcast each id inside the group, UNR row gathers in flight per lane, adds in list order.
//   The chunk loop is made wave-uniform with a cross-group max so the shuffles always run converged.
// ================================================================================================
template <int VEC, int G>
__device__ __forceinline__ Frag<VEC> accumulate_list(const float* __restrict__ src, int64_t ld_src,
                                                     const int32_t* __restrict__ ids, const float* __restrict__ src_scale,
                                                     int begin, int len, int wave_max_len, int lane, int col) {
    constexpr int UNR = G < 8 ? G : 8;
    const int lig = lane & (G - 1);
    const int group_base = lane & ~(G - 1);
    Frag<VEC> acc = Frag<VEC>::zero();
    for (int base = 0; base < wave_max_len; base += G) {
        const bool have = base + lig < len;
        const int my_id = have ? ids[begin + base + lig] : -1;
        const float my_w = (src_scale != nullptr && have) ? src_scale[my_id] : 1.f;
#pragma unroll 1
        for (int j = 0; j < G; j += UNR) {
            if (base + j >= wave_max_len) break;      // wave-uniform: nothing left in any group
            int id[UNR];
            float w[UNR];
            Frag<VEC> row[UNR];
#pragma unroll
            for (int k = 0; k < UNR; ++k) {
                id[k] = __shfl(my_id, group_base + j + k);
                w[k] = __shfl(my_w, group_base + j + k);
            }
#pragma unroll
            for (int k = 0; k < UNR; ++k)
                row[k] = (id[k] >= 0 && col >= 0) ? Frag<VEC>::load(src + static_cast<int64_t>(id[k]) * ld_src + col * VEC)
                                                  : Frag<VEC>::zero();
#pragma unroll
            for (int k = 0; k < UNR; ++k) acc.add_scaled(row[k], w[k]);
        }
    }
    return acc;
}

template <int G>
__device__ __forceinline__ int wave_max_over_groups(int v) {
#pragma unroll
    for (int o = kWave / 2; o >= G; o >>= 1) {
        const int other = __shfl_xor(v, o);
        v = other > v ? other : v;
    }
    return v;
}

template <int VEC>
__device__ __forceinline__ void apply_out_scale(Frag<VEC>& acc, const float* out_scale, int mode, int64_t row) {
    if (mode == IHG_SCALE_MULTIPLY) {
        acc.mul(out_scale[row]);
    } else if (mode == IHG_SCALE_DIVIDE) {
        const float s = out_scale[row];
        if (s != 0.f) acc.div(s);
    }
}

template <int VEC, int G>
__global__ __launch_bounds__(kBlockThreads) void node_segment_sum_kernel(
    const float* __restrict__ src, int64_t ld_src, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ ids,
    const float* __restrict__ src_scale, const float* __restrict__ out_scale, int mode,
    float* __restrict__ out, int64_t ld_out, int64_t n_rows, int dim_vec, int heavy_threshold) {
    constexpr int GPW = kWave / G;
    const int lane = threadIdx.x & (kWave - 1);
    const int lig = lane & (G - 1);
    const int grp = lane / G;
    for (int64_t r0 = global_wave_id() * GPW; r0 < n_rows; r0 += global_wave_count() * GPW) {
        const int64_t r = r0 + grp;
        int begin = 0, len = 0;
        bool heavy = false;
        if (r < n_rows) {
            begin = rowptr[r];
            len = rowptr[r + 1] - begin;
            if (heavy_threshold > 0 && len > heavy_threshold) { heavy = true; len = 0; }
        }
        const int wave_len = wave_max_over_groups<G>(len);
        const int col_iters = (dim_vec + G - 1) / G;
        for (int ci = 0; ci < col_iters; ++ci) {
            const int c = ci * G + lig;
            const int col = c < dim_vec ? c : -1;
            Frag<VEC> acc = accumulate_list<VEC, G>(src, ld_src, ids, src_scale, begin, len, wave_len, lane, col);
            if (r < n_rows && !heavy && col >= 0) {
                apply_out_scale<VEC>(acc, out_scale, mode, r);
                acc.store(out + r * ld_out + col * VEC);
            }
        }
    }
}

// Split rows: one group per segment -> partials[s,:]; then one group per heavy row adds its partials in order.
template <int VEC, int G>
__global__ __launch_bounds__(kBlockThreads) void heavy_partial_kernel(
    const float* __restrict__ src, int64_t ld_src, const int32_t* __restrict__ ids, const float* __restrict__ src_scale,
    const int32_t* __restrict__ seg_begin, const int32_t* __restrict__ seg_end, int64_t n_segments,
    float* __restrict__ partials, int dim, int dim_vec) {
    constexpr int GPW = kWave / G;
    const int lane = threadIdx.x & (kWave - 1);
    const int lig = lane & (G - 1);
    const int grp = lane / G;
    for (int64_t s0 = global_wave_id() * GPW; s0 < n_segments; s0 += global_wave_count() * GPW) {
        const int64_t s = s0 + grp;
        int begin = 0, len = 0;
        if (s < n_segments) { begin = seg_begin[s]; len = seg_end[s] - begin; }
        const int wave_len = wave_max_over_groups<G>(len);
        const int col_iters = (dim_vec + G - 1) / G;
        for (int ci = 0; ci < col_iters; ++ci) {
            const int c = ci * G + lig;
            const int col = c < dim_vec ? c : -1;
            Frag<VEC> acc = accumulate_list<VEC, G>(src, ld_src, ids, src_scale, begin, len, wave_len, lane, col);
            if (s < n_segments && col >= 0) acc.store(partials + s * dim + col * VEC);
        }
    }
}

template <int VEC, int G>
__global__ __launch_bounds__(kBlockThreads) void heavy_finish_kernel(
    const float* __restrict__ partials, const int32_t* __restrict__ heavy_rows, const int32_t* __restrict__ heavy_segptr,
    int64_t n_heavy, const float* __restrict__ out_scale, int mode, float* __restrict__ out, int64_t ld_out, int dim, int dim_vec) {
    constexpr int GPW = kWave / G;
    const int lane = threadIdx.x & (kWave - 1);
    const int lig = lane & (G - 1);
    const int grp = lane / G;
    for (int64_t h0 = global_wave_id() * GPW; h0 < n_heavy; h0 += global_wave_count() * GPW) {
        const int64_t h = h0 + grp;
        if (h >= n_heavy) continue;
        const int64_t row = heavy_rows[h];
        const int s_begin = heavy_segptr[h], s_end = heavy_segptr[h + 1];
        for (int c = lig; c < dim_vec; c += G) {
            Frag<VEC> acc = Frag<VEC>::zero();
            for (int s = s_begin; s < s_end; ++s) acc.add(Frag<VEC>::load(partials + static_cast<int64_t>(s) * dim + c * VEC));
            apply_out_scale<VEC>(acc, out_scale, mode, row);
            acc.store(out + row * ld_out + c * VEC);
        }
    }
}

// ================================================================================================
// Interactive step, generic form (any dim / stride).  One thread per output element; correct for every
// shape, used when the MFMA-tiled kernels' shape constraints do not hold.
// ================================================================================================
__global__ __launch_bounds__(kBlockThreads) void interact_fwd_generic_kernel(
    const float* __restrict__ h, int64_t ld_h, const float* __restrict__ p, int64_t ld_p, const int32_t* __restrict__ i3,
    const float* __restrict__ w, int64_t ld_w, int order, float* __restrict__ out, int64_t ld_out, int64_t n_edges, int dim) {
    const int64_t total = n_edges * dim;
    for (int64_t idx = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; idx < total;
         idx += static_cast<int64_t>(gridDim.x) * blockDim.x) {
        const int64_t e = idx / dim;
        const int j = static_cast<int>(idx - e * dim);
        const int64_t u = i3[e * 3], q = i3[e * 3 + 1], i = i3[e * 3 + 2];
        float acc = 0.f;
        if (p != nullptr) acc = (p[u * ld_p + j] + p[q * ld_p + j]) + p[i * ld_p + j];
        const float* wj = w + j * ld_w + 3 * static_cast<int64_t>(dim);
        const float* hu = h + u * ld_h;
        const float* hq = h + q * ld_h;
        const float* hi = h + i * ld_h;
        for (int c = 0; c < dim; ++c) {
            const float a = hu[c], b = hq[c], d = hi[c];
            const float uq = a * b;
            acc += wj[c] * uq;
            acc += wj[dim + c] * (b * d);
            acc += wj[2 * dim + c] * (d * a);
            if (order == 3) acc += wj[3 * dim + c] * (uq * d);
        }
        out[e * ld_out + j] = acc;
    }
}

// g[e, s, c]: gradient w.r.t. member s's transformed feature through the product terms.
__global__ __launch_bounds__(kBlockThreads) void interact_bwd_members_generic_kernel(
    const float* __restrict__ h, int64_t ld_h, const int32_t* __restrict__ i3, const float* __restrict__ w, int64_t ld_w,
    int order, const float* __restrict__ dout, int64_t ld_dout, float* __restrict__ g, int64_t n_edges, int dim) {
    const int64_t total = n_edges * dim;
    for (int64_t idx = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; idx < total;
         idx += static_cast<int64_t>(gridDim.x) * blockDim.x) {
        const int64_t e = idx / dim;
        const int c = static_cast<int>(idx - e * dim);
        const int64_t u = i3[e * 3], q = i3[e * 3 + 1], i = i3[e * 3 + 2];
        const float* de = dout + e * ld_dout;
        float z_uq = 0.f, z_qi = 0.f, z_iu = 0.f, z_uqi = 0.f;
        for (int j = 0; j < dim; ++j) {
            const float* wj = w + j * ld_w + 3 * static_cast<int64_t>(dim) + c;
            const float d = de[j];
            z_uq += d * wj[0];
            z_qi += d * wj[dim];
            z_iu += d * wj[2 * dim];
            if (order == 3) z_uqi += d * wj[3 * dim];
        }
        const float a = h[u * ld_h + c], b = h[q * ld_h + c], d = h[i * ld_h + c];
        float* ge = g + e * 3 * dim + c;
        ge[0] = z_uq * b + z_iu * d + z_uqi * (b * d);
        ge[dim] = z_uq * a + z_qi * d + z_uqi * (a * d);
        ge[2 * dim] = z_qi * b + z_iu * a + z_uqi * (a * b);
    }
}

// dW[j, (3+blk)*dim + c] = sum_e dout[e, j] * z_blk[e, c]; one thread per weight element, edges in order.
__global__ __launch_bounds__(kBlockThreads) void interact_bwd_weight_generic_kernel(
    const float* __restrict__ h, int64_t ld_h, const int32_t* __restrict__ i3, int order,
    const float* __restrict__ dout, int64_t ld_dout, float* __restrict__ dw, int64_t ld_dw, int64_t n_edges, int dim) {
    const int blocks = order == 3 ? 4 : 3;
    const int64_t total = static_cast<int64_t>(dim) * blocks * dim;
    for (int64_t idx = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; idx < total;
         idx += static_cast<int64_t>(gridDim.x) * blockDim.x) {
        const int c = static_cast<int>(idx % dim);
        const int blk = static_cast<int>((idx / dim) % blocks);
        const int j = static_cast<int>(idx / (static_cast<int64_t>(dim) * blocks));
        float acc = 0.f;
        for (int64_t e = 0; e < n_edges; ++e) {
            const int64_t u = i3[e * 3], q = i3[e * 3 + 1], i = i3[e * 3 + 2];
            const float a = h[u * ld_h + c], b = h[q * ld_h + c], d = h[i * ld_h + c];
            float z;
            if (blk == 0) z = a * b;
            else if (blk == 1) z = b * d;
            else if (blk == 2) z = d * a;
            else z = (a * b) * d;
            acc += dout[e * ld_dout + j] * z;
        }
        dw[j * ld_dw + (3 + blk) * static_cast<int64_t>(dim) + c] = acc;
    }
}

// ------------------------------------------------------------------------------------------------
// Dispatch helpers
// ------------------------------------------------------------------------------------------------
inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// Smallest power of two >= n, clamped to [4, 64].
inline int group_lanes(int n) {
    int g = 4;
    while (g < n && g < kWave) g <<= 1;
    return g;
}

template <int VEC>
int launch_edge_gather_sum(const float* src, int64_t ld_src, const int32_t* i3, const float* node_scale, const float* bias,
                           float alpha, float* out, int64_t ld_out, int64_t n_edges, int dim, hipStream_t stream) {
    const int dim_vec = dim / VEC;
    const int g = group_lanes(dim_vec);
#define IHG_LAUNCH_K5(G, U)                                                                                         \
    {                                                                                                               \
        constexpr int EPW = (kWave / G) * U;                                                                        \
        const int grid = grid_for_waves((n_edges + EPW - 1) / EPW);                                                 \
        hipLaunchKernelGGL((edge_gather_sum_kernel<VEC, G, U>), dim3(grid), dim3(kBlockThreads), 0, stream, src,    \
                           ld_src, i3, node_scale, bias, alpha, out, ld_out, n_edges, dim_vec);                     \
    }
    switch (g) {
        case 4: IHG_LAUNCH_K5(4, 1) break;
        case 8: IHG_LAUNCH_K5(8, 2) break;
        case 16: IHG_LAUNCH_K5(16, 4) break;
        case 32: IHG_LAUNCH_K5(32, 4) break;
        default: IHG_LAUNCH_K5(64, 4) break;
    }
#undef IHG_LAUNCH_K5
    return check_launch("ihg_edge_gather_sum");
}

template <int VEC, int G>
void launch_segment_sum_g(const float* src, int64_t ld_src, const int32_t* rowptr, const int32_t* ids, const float* src_scale,
                          const float* out_scale, int mode, float* out, int64_t ld_out, int64_t n_rows, int dim_vec,
                          int heavy_threshold, hipStream_t stream) {
    constexpr int GPW = kWave / G;
    const int grid = grid_for_waves((n_rows + GPW - 1) / GPW);
    hipLaunchKernelGGL((node_segment_sum_kernel<VEC, G>), dim3(grid), dim3(kBlockThreads), 0, stream, src, ld_src, rowptr,
                       ids, src_scale, out_scale, mode, out, ld_out, n_rows, dim_vec, heavy_threshold);
}

template <int VEC>
int launch_segment_sum(const float* src, int64_t ld_src, const int32_t* rowptr, const int32_t* ids, const float* src_scale,
                       const float* out_scale, int mode, float* out, int64_t ld_out, int64_t n_rows, int dim,
                       int heavy_threshold, hipStream_t stream) {
    const int dim_vec = dim / VEC;
    switch (group_lanes(dim_vec)) {
        case 4: launch_segment_sum_g<VEC, 4>(src, ld_src, rowptr, ids, src_scale, out_scale, mode, out, ld_out, n_rows, dim_vec, heavy_threshold, stream); break;
        case 8: launch_segment_sum_g<VEC, 8>(src, ld_src, rowptr, ids, src_scale, out_scale, mode, out, ld_out, n_rows, dim_vec, heavy_threshold, stream); break;
        case 16: launch_segment_sum_g<VEC, 16>(src, ld_src, rowptr, ids, src_scale, out_scale, mode, out, ld_out, n_rows, dim_vec, heavy_threshold, stream); break;
        case 32: launch_segment_sum_g<VEC, 32>(src, ld_src, rowptr, ids, src_scale, out_scale, mode, out, ld_out, n_rows, dim_vec, heavy_threshold, stream); break;
        default: launch_segment_sum_g<VEC, 64>(src, ld_src, rowptr, ids, src_scale, out_scale, mode, out, ld_out, n_rows, dim_vec, heavy_threshold, stream); break;
    }
    return check_launch("ihg_node_segment_sum");
}

template <int VEC, int G>
void launch_heavy_g(const float* src, int64_t ld_src, const int32_t* ids, const float* src_scale, const float* out_scale, int mode,
                    const int32_t* seg_begin, const int32_t* seg_end, int64_t n_segments, const int32_t* heavy_rows,
                    const int32_t* heavy_segptr, int64_t n_heavy, float* partials, float* out, int64_t ld_out, int dim,
                    int dim_vec, hipStream_t stream) {
    constexpr int GPW = kWave / G;
    hipLaunchKernelGGL((heavy_partial_kernel<VEC, G>), dim3(grid_for_waves((n_segments + GPW - 1) / GPW)), dim3(kBlockThreads), 0,
                       stream, src, ld_src, ids, src_scale, seg_begin, seg_end, n_segments, partials, dim, dim_vec);
    hipLaunchKernelGGL((heavy_finish_kernel<VEC, G>), dim3(grid_for_waves((n_heavy + GPW - 1) / GPW)), dim3(kBlockThreads), 0,
                       stream, partials, heavy_rows, heavy_segptr, n_heavy, out_scale, mode, out, ld_out, dim, dim_vec);
}

template <int VEC>
int launch_heavy(const float* src, int64_t ld_src, const int32_t* ids, const float* src_scale, const float* out_scale, int mode,
                 const int32_t* seg_begin, const int32_t* seg_end, int64_t n_segments, const int32_t* heavy_rows,
                 const int32_t* heavy_segptr, int64_t n_heavy, float* partials, float* out, int64_t ld_out, int dim,
                 hipStream_t stream) {
    const int dim_vec = dim / VEC;
#define IHG_HEAVY(G) launch_heavy_g<VEC, G>(src, ld_src, ids, src_scale, out_scale, mode, seg_begin, seg_end, n_segments, heavy_rows, heavy_segptr, n_heavy, partials, out, ld_out, dim, dim_vec, stream)
    switch (group_lanes(dim_vec)) {
        case 4: IHG_HEAVY(4); break;
        case 8: IHG_HEAVY(8); break;
        case 16: IHG_HEAVY(16); break;
        case 32: IHG_HEAVY(32); break;
        default: IHG_HEAVY(64); break;
    }
#undef IHG_HEAVY
    return check_launch("ihg_node_segment_sum_heavy");
}

inline bool scale_mode_ok(int mode, const float* scale) {
    if (mode == IHG_SCALE_NONE) return true;
    return (mode == IHG_SCALE_MULTIPLY || mode == IHG_SCALE_DIVIDE) && scale != nullptr;
}

}  // namespace

// =================================================================================================
// C ABI
// =================================================================================================
extern "C" {

int32_t ihg_abi_version(void) { return 1; }

const char* ihg_last_error_string(void) { return g_error; }

int ihg_build_csr(const int64_t* triples, int64_t n_edges, int64_t n_users, int64_t n_queries, int64_t n_items,
                  int32_t* i3, int32_t* rowptr, int32_t* edge_ids, float* degree) {
    if (n_edges < 0 || n_users < 0 || n_queries < 0 || n_items < 0) return fail(IHG_ERR_INVALID, "ihg_build_csr: negative size");
    const int64_t n_nodes = n_users + n_queries + n_items;
    if (n_nodes >= INT32_MAX || n_edges * 3 >= INT32_MAX) return fail(IHG_ERR_INVALID, "ihg_build_csr: graph exceeds int32 indexing");
    if ((n_edges > 0 && (triples == nullptr || i3 == nullptr || edge_ids == nullptr)) || rowptr == nullptr || degree == nullptr)
        return fail(IHG_ERR_INVALID, "ihg_build_csr: null buffer");
    const int64_t offset[3] = {0, n_users, n_users + n_queries};
    const int64_t limit[3] = {n_users, n_queries, n_items};
    std::memset(rowptr, 0, sizeof(int32_t) * static_cast<size_t>(n_nodes + 1));
    for (int64_t e = 0; e < n_edges; ++e) {
        for (int m = 0; m < 3; ++m) {
            const int64_t local = triples[e * 3 + m];
            if (local < 0 || local >= limit[m])
                return fail(IHG_ERR_INVALID, "ihg_build_csr: hyperedge %lld member %d id %lld out of range [0,%lld)",
                            static_cast<long long>(e), m, static_cast<long long>(local), static_cast<long long>(limit[m]));
            const int32_t node = static_cast<int32_t>(local + offset[m]);
            i3[e * 3 + m] = node;
            ++rowptr[node + 1];
        }
    }
    for (int64_t v = 0; v < n_nodes; ++v) {
        const int32_t d = rowptr[v + 1];
        degree[v] = d == 0 ? 1e-8f : static_cast<float>(d);
        rowptr[v + 1] = rowptr[v] + d;
    }
    std::vector<int32_t> cursor(rowptr, rowptr + n_nodes);
    for (int64_t e = 0; e < n_edges; ++e)           // ascending e => ascending hyperedge ids inside every node
        for (int m = 0; m < 3; ++m) edge_ids[cursor[i3[e * 3 + m]]++] = static_cast<int32_t>(e);
    return IHG_OK;
}

int ihg_transpose_csr(const int32_t* ptr, const int32_t* ids, int64_t n_rows, int64_t n_cols, int32_t* t_ptr, int32_t* t_rows) {
    if (n_rows < 0 || n_cols < 0 || ptr == nullptr || t_ptr == nullptr) return fail(IHG_ERR_INVALID, "ihg_transpose_csr: bad argument");
    const int64_t nnz = ptr[n_rows];
    if (nnz > 0 && (ids == nullptr || t_rows == nullptr)) return fail(IHG_ERR_INVALID, "ihg_transpose_csr: null buffer");
    std::memset(t_ptr, 0, sizeof(int32_t) * static_cast<size_t>(n_cols + 1));
    for (int64_t k = 0; k < nnz; ++k) {
        if (ids[k] < 0 || ids[k] >= n_cols) return fail(IHG_ERR_INVALID, "ihg_transpose_csr: id %d out of range", ids[k]);
        ++t_ptr[ids[k] + 1];
    }
    for (int64_t c = 0; c < n_cols; ++c) t_ptr[c + 1] += t_ptr[c];
    std::vector<int32_t> cursor(t_ptr, t_ptr + n_cols);
    for (int64_t r = 0; r < n_rows; ++r)
        for (int32_t k = ptr[r]; k < ptr[r + 1]; ++k) t_rows[cursor[ids[k]]++] = static_cast<int32_t>(r);
    return IHG_OK;
}

int ihg_edge_gather_sum(const float* src, int64_t ld_src, const int32_t* i3, const float* node_scale, const float* bias,
                        float alpha, float* out, int64_t ld_out, int64_t n_edges, int32_t dim, ihg_stream_t stream) {
    if (n_edges < 0 || dim <= 0 || ld_src < dim || ld_out < dim) return fail(IHG_ERR_INVALID, "ihg_edge_gather_sum: bad size (E=%lld dim=%d ld_src=%lld ld_out=%lld)", (long long)n_edges, dim, (long long)ld_src, (long long)ld_out);
    if (n_edges == 0) return IHG_OK;
    if (src == nullptr || i3 == nullptr || out == nullptr) return fail(IHG_ERR_INVALID, "ihg_edge_gather_sum: null pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool wide = dim % 4 == 0 && ld_src % 4 == 0 && ld_out % 4 == 0 && aligned16(src) && aligned16(out) && (bias == nullptr || aligned16(bias));
    return wide ? launch_edge_gather_sum<4>(src, ld_src, i3, node_scale, bias, alpha, out, ld_out, n_edges, dim, s)
                : launch_edge_gather_sum<1>(src, ld_src, i3, node_scale, bias, alpha, out, ld_out, n_edges, dim, s);
}

int ihg_node_segment_sum(const float* src, int64_t ld_src, const int32_t* rowptr, const int32_t* ids, const float* src_scale,
                         const float* out_scale, int32_t out_scale_mode, float* out, int64_t ld_out, int64_t n_rows, int32_t dim,
                         int32_t heavy_threshold, ihg_stream_t stream) {
    if (n_rows < 0 || dim <= 0 || ld_src < dim || ld_out < dim) return fail(IHG_ERR_INVALID, "ihg_node_segment_sum: bad size");
    if (!scale_mode_ok(out_scale_mode, out_scale)) return fail(IHG_ERR_INVALID, "ihg_node_segment_sum: bad out_scale_mode %d", out_scale_mode);
    if (n_rows == 0) return IHG_OK;
    if (src == nullptr || rowptr == nullptr || ids == nullptr || out == nullptr) return fail(IHG_ERR_INVALID, "ihg_node_segment_sum: null pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool wide = dim % 4 == 0 && ld_src % 4 == 0 && ld_out % 4 == 0 && aligned16(src) && aligned16(out);
    return wide ? launch_segment_sum<4>(src, ld_src, rowptr, ids, src_scale, out_scale, out_scale_mode, out, ld_out, n_rows, dim, heavy_threshold, s)
                : launch_segment_sum<1>(src, ld_src, rowptr, ids, src_scale, out_scale, out_scale_mode, out, ld_out, n_rows, dim, heavy_threshold, s);
}

int ihg_node_segment_sum_heavy(const float* src, int64_t ld_src, const int32_t* ids, const float* src_scale, const float* out_scale,
                               int32_t out_scale_mode, const int32_t* seg_begin, const int32_t* seg_end, int64_t n_segments,
                               const int32_t* heavy_rows, const int32_t* heavy_segptr, int64_t n_heavy, float* partials,
                               float* out, int64_t ld_out, int32_t dim, ihg_stream_t stream) {
    if (n_segments < 0 || n_heavy < 0 || dim <= 0 || ld_src < dim || ld_out < dim) return fail(IHG_ERR_INVALID, "ihg_node_segment_sum_heavy: bad size");
    if (!scale_mode_ok(out_scale_mode, out_scale)) return fail(IHG_ERR_INVALID, "ihg_node_segment_sum_heavy: bad out_scale_mode %d", out_scale_mode);
    if (n_heavy == 0) return IHG_OK;
    if (src == nullptr || ids == nullptr || seg_begin == nullptr || seg_end == nullptr || heavy_rows == nullptr || heavy_segptr == nullptr ||
        partials == nullptr || out == nullptr)
        return fail(IHG_ERR_INVALID, "ihg_node_segment_sum_heavy: null pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool wide = dim % 4 == 0 && ld_src % 4 == 0 && ld_out % 4 == 0 && aligned16(src) && aligned16(out) && aligned16(partials);
    return wide ? launch_heavy<4>(src, ld_src, ids, src_scale, out_scale, out_scale_mode, seg_begin, seg_end, n_segments, heavy_rows, heavy_segptr, n_heavy, partials, out, ld_out, dim, s)
                : launch_heavy<1>(src, ld_src, ids, src_scale, out_scale, out_scale_mode, seg_begin, seg_end, n_segments, heavy_rows, heavy_segptr, n_heavy, partials, out, ld_out, dim, s);
}

int ihg_bag_mean_fwd(const float* table, int64_t ld_table, const int32_t* bag_ptr, const int32_t* words, const float* bag_len,
                     float* out, int64_t ld_out, int64_t n_bags, int32_t dim, ihg_stream_t stream) {
    if (bag_len == nullptr && n_bags > 0) return fail(IHG_ERR_INVALID, "ihg_bag_mean_fwd: null bag_len");
    return ihg_node_segment_sum(table, ld_table, bag_ptr, words, nullptr, bag_len, IHG_SCALE_DIVIDE, out, ld_out, n_bags, dim, 0, stream);
}

int ihg_bag_mean_bwd(const float* dout, int64_t ld_dout, const int32_t* word_ptr, const int32_t* word_bags, const float* inv_len,
                     float* dtable, int64_t ld_dtable, int64_t n_table_rows, int32_t dim, ihg_stream_t stream) {
    if (inv_len == nullptr && n_table_rows > 0) return fail(IHG_ERR_INVALID, "ihg_bag_mean_bwd: null inv_len");
    return ihg_node_segment_sum(dout, ld_dout, word_ptr, word_bags, inv_len, nullptr, IHG_SCALE_NONE, dtable, ld_dtable, n_table_rows, dim, 0, stream);
}

int ihg_interact_fwd(const float* h, int64_t ld_h, const float* p, int64_t ld_p, const int32_t* i3, const float* w, int64_t ld_w,
                     int32_t order, float* out, int64_t ld_out, int64_t n_edges, int32_t dim, ihg_stream_t stream) {
    if (order != 2 && order != 3) return fail(IHG_ERR_INVALID, "ihg_interact_fwd: order must be 2 or 3, got %d", order);
    const int k = order == 3 ? 7 : 6;
    if (n_edges < 0 || dim <= 0 || ld_h < dim || ld_out < dim || ld_w < static_cast<int64_t>(k) * dim || (p != nullptr && ld_p < dim))
        return fail(IHG_ERR_INVALID, "ihg_interact_fwd: bad size");
    if (n_edges == 0) return IHG_OK;
    if (h == nullptr || i3 == nullptr || w == nullptr || out == nullptr) return fail(IHG_ERR_INVALID, "ihg_interact_fwd: null pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int64_t total = n_edges * dim;
    const int grid = static_cast<int>(std::min<int64_t>((total + kBlockThreads - 1) / kBlockThreads, kMaxBlocks * 4));
    hipLaunchKernelGGL(interact_fwd_generic_kernel, dim3(grid), dim3(kBlockThreads), 0, s, h, ld_h, p, ld_p, i3, w, ld_w, order, out, ld_out, n_edges, dim);
    return check_launch("ihg_interact_fwd");
}

int64_t ihg_interact_bwd_workspace_bytes(int64_t n_edges, int32_t dim, int32_t order) {
    (void)n_edges; (void)dim; (void)order;
    return 0;
}

int ihg_interact_bwd(const float* h, int64_t ld_h, const int32_t* i3, const float* w, int64_t ld_w, int32_t order,
                     const float* dout, int64_t ld_dout, float* g, float* dw, int64_t ld_dw, void* workspace,
                     int64_t workspace_bytes, int64_t n_edges, int32_t dim, ihg_stream_t stream) {
    if (order != 2 && order != 3) return fail(IHG_ERR_INVALID, "ihg_interact_bwd: order must be 2 or 3, got %d", order);
    const int k = order == 3 ? 7 : 6;
    if (n_edges < 0 || dim <= 0 || ld_h < dim || ld_dout < dim || ld_w < static_cast<int64_t>(k) * dim || ld_dw < static_cast<int64_t>(k) * dim)
        return fail(IHG_ERR_INVALID, "ihg_interact_bwd: bad size");
    if (workspace_bytes < ihg_interact_bwd_workspace_bytes(n_edges, dim, order)) return fail(IHG_ERR_WORKSPACE, "ihg_interact_bwd: workspace too small");
    (void)workspace;
    if (h == nullptr || i3 == nullptr || w == nullptr || dout == nullptr || g == nullptr || dw == nullptr)
        return fail(IHG_ERR_INVALID, "ihg_interact_bwd: null pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (n_edges > 0) {
        const int64_t total = n_edges * dim;
        const int grid = static_cast<int>(std::min<int64_t>((total + kBlockThreads - 1) / kBlockThreads, kMaxBlocks * 4));
        hipLaunchKernelGGL(interact_bwd_members_generic_kernel, dim3(grid), dim3(kBlockThreads), 0, s, h, ld_h, i3, w, ld_w, order, dout, ld_dout, g, n_edges, dim);
    }
    const int64_t wtotal = static_cast<int64_t>(dim) * (order == 3 ? 4 : 3) * dim;
    const int wgrid = static_cast<int>(std::min<int64_t>((wtotal + kBlockThreads - 1) / kBlockThreads, kMaxBlocks * 4));
    hipLaunchKernelGGL(interact_bwd_weight_generic_kernel, dim3(wgrid), dim3(kBlockThreads), 0, s, h, ld_h, i3, order, dout, ld_dout, dw, ld_dw, n_edges, dim);
    return check_launch("ihg_interact_bwd");
}

}  // extern "C"
