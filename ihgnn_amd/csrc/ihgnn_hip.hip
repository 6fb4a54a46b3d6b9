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

#include <algorithm>
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
                                                     const float* __restrict__ entry_scale,
                                                     int begin, int len, int wave_max_len, int lane, int col) {
    constexpr int UNR = G < 8 ? G : 8;
    const int lig = lane & (G - 1);
    const int group_base = lane & ~(G - 1);
    Frag<VEC> acc = Frag<VEC>::zero();
    for (int base = 0; base < wave_max_len; base += G) {
        const bool have = base + lig < len;
        const int my_id = have ? ids[begin + base + lig] : -1;
        float my_w = (src_scale != nullptr && have) ? src_scale[my_id] : 1.f;
        if (entry_scale != nullptr && have) my_w *= entry_scale[begin + base + lig];
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

// Work list of one launch: first the fixed-length segments of the split (heavy) rows, then the light rows in `row_order`
// (decreasing length).  Unit u < n_segments writes partials[u]; unit u >= n_segments writes its output row.
template <int VEC, int G>
__global__ __launch_bounds__(kBlockThreads) void node_segment_sum_kernel(
    const float* __restrict__ src, int64_t ld_src, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ ids,
    const int32_t* __restrict__ row_order, const float* __restrict__ src_scale, const float* __restrict__ entry_scale,
    const float* __restrict__ out_scale, int mode,
    float* __restrict__ out, int64_t ld_out, int64_t n_rows, int dim, int dim_vec, int heavy_threshold,
    const int32_t* __restrict__ seg_begin, const int32_t* __restrict__ seg_end, int64_t n_segments, float* __restrict__ partials,
    const float* __restrict__ self_weight) {
    constexpr int GPW = kWave / G;
    const int lane = threadIdx.x & (kWave - 1);
    const int lig = lane & (G - 1);
    const int grp = lane / G;
    const int64_t n_units = n_segments + n_rows;
    for (int64_t u0 = global_wave_id() * GPW; u0 < n_units; u0 += global_wave_count() * GPW) {
        const int64_t u = u0 + grp;
        int begin = 0, len = 0;
        float* dst = nullptr;
        int64_t scale_row = -1;
        if (u < n_segments) {
            begin = seg_begin[u];
            len = seg_end[u] - begin;
            dst = partials + u * dim;
        } else if (u < n_units) {
            int64_t r = u - n_segments;
            if (row_order != nullptr) r = row_order[r];
            begin = rowptr[r];
            len = rowptr[r + 1] - begin;
            if (heavy_threshold > 0 && len > heavy_threshold) {
                len = 0;                                    // finished from the partials
            } else {
                dst = out + r * ld_out;
                scale_row = r;
            }
        }
        const int wave_len = wave_max_over_groups<G>(len);
        const int col_iters = (dim_vec + G - 1) / G;
        for (int ci = 0; ci < col_iters; ++ci) {
            const int c = ci * G + lig;
            const int col = c < dim_vec ? c : -1;
            Frag<VEC> acc = accumulate_list<VEC, G>(src, ld_src, ids, src_scale, entry_scale, begin, len, wave_len, lane, col);
            if (dst != nullptr && col >= 0) {
                if (scale_row >= 0) {
                    if (self_weight != nullptr)             // square operators: the row's own source row, weighted
                        acc.add_scaled(Frag<VEC>::load(src + scale_row * ld_src + col * VEC),
                                       self_weight[scale_row] * (src_scale != nullptr ? src_scale[scale_row] : 1.f));
                    apply_out_scale<VEC>(acc, out_scale, mode, scale_row);
                }
                acc.store(dst + col * VEC);
            }
        }
    }
}

// One workgroup per heavy row: its 256/G lane groups take the row's partials round-robin (4 loads in flight each), the
// per-group sums are combined through LDS in group order - a fixed summation tree, so the result is bitwise reproducible.
template <int VEC, int G>
__global__ __launch_bounds__(kBlockThreads) void heavy_finish_kernel(
    const float* __restrict__ partials, const int32_t* __restrict__ heavy_rows, const int32_t* __restrict__ heavy_segptr,
    int64_t n_heavy, const float* __restrict__ out_scale, int mode, float* __restrict__ out, int64_t ld_out, int dim, int dim_vec,
    const float* __restrict__ src, int64_t ld_src, const float* __restrict__ src_scale, const float* __restrict__ self_weight) {
    constexpr int GROUPS = kBlockThreads / G;
    __shared__ __attribute__((aligned(16))) float red[GROUPS][G * VEC];
    const int lig = threadIdx.x & (G - 1);
    const int grp = threadIdx.x / G;
    for (int64_t h = blockIdx.x; h < n_heavy; h += gridDim.x) {
        const int64_t row = heavy_rows[h];
        const int s_begin = heavy_segptr[h], s_end = heavy_segptr[h + 1];
        for (int c0 = 0; c0 < dim_vec; c0 += G) {
            const int c = c0 + lig;
            Frag<VEC> acc = Frag<VEC>::zero();
            if (c < dim_vec) {
                int sgm = s_begin + grp;
                for (; sgm + 3 * GROUPS < s_end; sgm += 4 * GROUPS) {
                    const Frag<VEC> a0 = Frag<VEC>::load(partials + static_cast<int64_t>(sgm) * dim + c * VEC);
                    const Frag<VEC> a1 = Frag<VEC>::load(partials + static_cast<int64_t>(sgm + GROUPS) * dim + c * VEC);
                    const Frag<VEC> a2 = Frag<VEC>::load(partials + static_cast<int64_t>(sgm + 2 * GROUPS) * dim + c * VEC);
                    const Frag<VEC> a3 = Frag<VEC>::load(partials + static_cast<int64_t>(sgm + 3 * GROUPS) * dim + c * VEC);
                    acc.add(a0); acc.add(a1); acc.add(a2); acc.add(a3);
                }
                for (; sgm < s_end; sgm += GROUPS) acc.add(Frag<VEC>::load(partials + static_cast<int64_t>(sgm) * dim + c * VEC));
                acc.store(&red[grp][lig * VEC]);
            }
            __syncthreads();
            if (grp == 0 && c < dim_vec) {
                Frag<VEC> total = Frag<VEC>::load(&red[0][lig * VEC]);
                const int used = s_end - s_begin < GROUPS ? s_end - s_begin : GROUPS;
                for (int g2 = 1; g2 < used; ++g2) total.add(Frag<VEC>::load(&red[g2][lig * VEC]));
                if (self_weight != nullptr)
                    total.add_scaled(Frag<VEC>::load(src + row * ld_src + c * VEC), self_weight[row] * (src_scale != nullptr ? src_scale[row] : 1.f));
                apply_out_scale<VEC>(total, out_scale, mode, row);
                total.store(out + row * ld_out + c * VEC);
            }
            __syncthreads();
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


// ================================================================================================
// Interactive step on the matrix cores (exact fp32: v_mfma_f32_32x32x2_f32, bit-for-bit an fmaf chain).
//
//   fwd      C[e][j]        = sum_b sum_c z_b[e][c] * W[j][(3+b)d + c]          (+ hoisted first-order rows)
//   members  dz_b[e][c]     = sum_j dout[e][j] * W[j][(3+b)d + c]    -> g[e, slot, c] by the product rule
//   weights  dW[j][(3+b)d+c]= sum_e dout[e][j] * z_b[e][c]
//
// A 32x32x2 MFMA takes ONE float per lane per operand: lane l gives A[i = l&31][k = l>>5] and B[k = l>>5][j = l&31].
// The contraction index may be visited in any order as long as A and B agree, so in fwd / members lane-half h of
// MFMA step s (s = 0..3) is given k = 8t + 4h + s: each lane then needs 4 CONSECUTIVE k per operand, i.e. one
// ds_read_b128 (A side, gathered rows staged in LDS with a 16-B row pad -> conflict-free) and one 16-B global load
// (B side) per 4 MFMAs.  The weights are re-packed once per call into that fragment order (pack_weights_kernel,
// <= 1 MB) so that every B load of a wave is one contiguous, fully coalesced 1 KiB from L2.
// z_b is never stored: it is formed in registers from the three staged member rows right before the MFMAs.
// ================================================================================================
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int kRowPad = 4;       // floats; breaks the power-of-two row stride for ds_read_b128

// wp_fwd[jt][b][t][lane][4] = W[32jt + (lane&31)][(3+b)d + 8t + 4(lane>>5) + s]        (k runs along c)
// wp_bwd[ct][b][t][lane][4] = W[8t + 4(lane>>5) + s][(3+b)d + 32ct + (lane&31)]        (k runs along j)
__global__ __launch_bounds__(kBlockThreads) void pack_weights_kernel(const float* __restrict__ w, int64_t ld_w, int d, int nblk,
                                                                     float* __restrict__ wp_fwd, float* __restrict__ wp_bwd) {
    const int t_count = d / 8;
    const int total = (d / 32) * nblk * t_count * kWave;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int lane = idx & (kWave - 1);
        const int t = (idx >> 6) % t_count;
        const int b = ((idx >> 6) / t_count) % nblk;
        const int xt = (idx >> 6) / (t_count * nblk);
        const int r = lane & 31, half = lane >> 5;
        if (wp_fwd != nullptr) {
            const float* src = w + static_cast<int64_t>(32 * xt + r) * ld_w + static_cast<int64_t>(3 + b) * d + 8 * t + 4 * half;
            *reinterpret_cast<float4*>(wp_fwd + static_cast<int64_t>(idx) * 4) = make_float4(src[0], src[1], src[2], src[3]);
        }
        if (wp_bwd != nullptr) {
            const float* src = w + static_cast<int64_t>(8 * t + 4 * half) * ld_w + static_cast<int64_t>(3 + b) * d + 32 * xt + r;
            *reinterpret_cast<float4*>(wp_bwd + static_cast<int64_t>(idx) * 4) = make_float4(src[0], src[ld_w], src[2 * ld_w], src[3 * ld_w]);
        }
    }
}

template <int D> struct TileShape {
    static constexpr int KC = D < 64 ? D : 64;             // staged column chunk of the member rows
    static constexpr int ET = D == 32 ? 4 : 2;             // 32-edge tiles per workgroup tile
    static constexpr int TE = ET * 32;
    static constexpr int NJ = D == 32 ? 1 : D / 64;        // 32-wide output column tiles per wave
    static constexpr int STRIDE = KC + kRowPad;
};

__device__ __forceinline__ int acc_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

template <int D, int NBLK>
__global__ __launch_bounds__(kBlockThreads) void interact_fwd_mfma_kernel(
    const float* __restrict__ h, int64_t ld_h, const float* __restrict__ p, int64_t ld_p, const int32_t* __restrict__ i3,
    const float* __restrict__ wp, float* __restrict__ out, int64_t ld_out, int64_t n_edges) {
    using S = TileShape<D>;
    __shared__ __attribute__((aligned(16))) float tile[3][S::TE][S::STRIDE];
    __shared__ int ids[S::TE][3];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int et = D == 32 ? wave : (wave & 1);
    const int jt0 = D == 32 ? 0 : (wave >> 1);
    const int row = et * 32 + (lane & 31), half = lane >> 5;
    const int64_t n_tiles = (n_edges + S::TE - 1) / S::TE;
    const v4f* wp4 = reinterpret_cast<const v4f*>(wp);

    for (int64_t tile_id = blockIdx.x; tile_id < n_tiles; tile_id += gridDim.x) {
        const int64_t e_base = tile_id * S::TE;
        __syncthreads();                                      // previous tile's epilogue is done with ids[]
        for (int k = tid; k < S::TE * 3; k += kBlockThreads) {
            const int64_t pos = e_base * 3 + k;
            (&ids[0][0])[k] = pos < n_edges * 3 ? i3[pos] : 0;
        }
        v16f acc[S::NJ];
#pragma unroll
        for (int x = 0; x < S::NJ; ++x)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[x][r] = 0.f;

        for (int kc = 0; kc < D / S::KC; ++kc) {
            __syncthreads();                                  // ids visible; previous chunk's reads finished
            constexpr int V4_PER_ROW = S::KC / 4;
            constexpr int LOADS = 3 * S::TE * V4_PER_ROW / kBlockThreads;
            float4 stage[LOADS];
#pragma unroll
            for (int x = 0; x < LOADS; ++x) {
                const int idx = tid + kBlockThreads * x;
                const int c4 = idx % V4_PER_ROW, r = (idx / V4_PER_ROW) % S::TE, m = idx / (V4_PER_ROW * S::TE);
                stage[x] = *reinterpret_cast<const float4*>(h + static_cast<int64_t>(ids[r][m]) * ld_h + kc * S::KC + c4 * 4);
            }
#pragma unroll
            for (int x = 0; x < LOADS; ++x) {
                const int idx = tid + kBlockThreads * x;
                const int c4 = idx % V4_PER_ROW, r = (idx / V4_PER_ROW) % S::TE, m = idx / (V4_PER_ROW * S::TE);
                *reinterpret_cast<float4*>(&tile[m][r][c4 * 4]) = stage[x];
            }
            __syncthreads();
#pragma unroll 2
            for (int t = 0; t < S::KC / 8; ++t) {
                const int col = 8 * t + 4 * half;
                const v4f au = *reinterpret_cast<const v4f*>(&tile[0][row][col]);
                const v4f aq = *reinterpret_cast<const v4f*>(&tile[1][row][col]);
                const v4f ai = *reinterpret_cast<const v4f*>(&tile[2][row][col]);
                v4f z[4];
                z[0] = au * aq;
                z[1] = aq * ai;
                z[2] = ai * au;
                z[3] = z[0] * ai;
                const int tg = kc * (S::KC / 8) + t;
#pragma unroll
                for (int x = 0; x < S::NJ; ++x) {
                    const int jt = jt0 + 2 * x;
                    v4f bf[NBLK];
#pragma unroll
                    for (int b = 0; b < NBLK; ++b) bf[b] = wp4[(static_cast<int64_t>(jt * NBLK + b) * (D / 8) + tg) * kWave + lane];
#pragma unroll
                    for (int b = 0; b < NBLK; ++b)
#pragma unroll
                        for (int s = 0; s < 4; ++s) acc[x] = __builtin_amdgcn_mfma_f32_32x32x2f32(z[b][s], bf[b][s], acc[x], 0, 0, 0);
                }
            }
        }
        // epilogue: add the hoisted first-order rows and store
#pragma unroll
        for (int x = 0; x < S::NJ; ++x) {
            const int j = (jt0 + 2 * x) * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int er = et * 32 + acc_row(r, lane);
                const int64_t e = e_base + er;
                if (e < n_edges) {
                    const float first = (p[static_cast<int64_t>(ids[er][0]) * ld_p + j] + p[static_cast<int64_t>(ids[er][1]) * ld_p + j]) +
                                        p[static_cast<int64_t>(ids[er][2]) * ld_p + j];
                    out[e * ld_out + j] = acc[x][r] + first;
                }
            }
        }
    }
}

// members: one workgroup tile = TE consecutive hyperedges; the dout rows are streamed (not gathered) into LDS at full
// width, each wave then runs its (edge tile, column tile) jobs one after the other with NBLK accumulators.
template <int D, int NBLK>
__global__ __launch_bounds__(kBlockThreads) void interact_bwd_members_mfma_kernel(
    const float* __restrict__ h, int64_t ld_h, const int32_t* __restrict__ i3, const float* __restrict__ wq,
    const float* __restrict__ dout, int64_t ld_dout, float* __restrict__ g, int64_t n_edges) {
    constexpr int ET = D == 32 ? 4 : 2, TE = ET * 32, STRIDE = D + kRowPad, JOBS = ET * (D / 32);
    __shared__ __attribute__((aligned(16))) float dtile[TE][STRIDE];
    __shared__ int ids[TE][3];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5;
    const int64_t n_tiles = (n_edges + TE - 1) / TE;
    const v4f* wq4 = reinterpret_cast<const v4f*>(wq);

    for (int64_t tile_id = blockIdx.x; tile_id < n_tiles; tile_id += gridDim.x) {
        const int64_t e_base = tile_id * TE;
        __syncthreads();
        for (int k = tid; k < TE * 3; k += kBlockThreads) {
            const int64_t pos = e_base * 3 + k;
            (&ids[0][0])[k] = pos < n_edges * 3 ? i3[pos] : 0;
        }
        constexpr int V4_PER_ROW = D / 4;
        for (int idx = tid; idx < TE * V4_PER_ROW; idx += kBlockThreads) {
            const int c4 = idx % V4_PER_ROW, r = idx / V4_PER_ROW;
            const int64_t e = e_base + r;
            const float4 v = e < n_edges ? *reinterpret_cast<const float4*>(dout + e * ld_dout + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4*>(&dtile[r][c4 * 4]) = v;
        }
        __syncthreads();
        for (int job = wave; job < JOBS; job += kWavesPerBlock) {
            const int et = job % ET, ct = job / ET;
            const int row = et * 32 + (lane & 31);
            v16f acc[NBLK];
#pragma unroll
            for (int b = 0; b < NBLK; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
#pragma unroll 2
            for (int t = 0; t < D / 8; ++t) {
                const v4f a = *reinterpret_cast<const v4f*>(&dtile[row][8 * t + 4 * half]);
                v4f bf[NBLK];
#pragma unroll
                for (int b = 0; b < NBLK; ++b) bf[b] = wq4[(static_cast<int64_t>(ct * NBLK + b) * (D / 8) + t) * kWave + lane];
#pragma unroll
                for (int b = 0; b < NBLK; ++b)
#pragma unroll
                    for (int s = 0; s < 4; ++s) acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], bf[b][s], acc[b], 0, 0, 0);
            }
            const int c = ct * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int er = et * 32 + acc_row(r, lane);
                const int64_t e = e_base + er;
                if (e < n_edges) {
                    const float a = h[static_cast<int64_t>(ids[er][0]) * ld_h + c];
                    const float b = h[static_cast<int64_t>(ids[er][1]) * ld_h + c];
                    const float dd = h[static_cast<int64_t>(ids[er][2]) * ld_h + c];
                    const float z_uq = acc[0][r], z_qi = acc[1][r], z_iu = acc[2][r];
                    const float z_uqi = NBLK == 4 ? acc[NBLK - 1][r] : 0.f;
                    float* ge = g + e * 3 * D + c;
                    ge[0] = z_uq * b + z_iu * dd + z_uqi * (b * dd);
                    ge[D] = z_uq * a + z_qi * dd + z_uqi * (a * dd);
                    ge[2 * D] = z_qi * b + z_iu * a + z_uqi * (a * b);
                }
            }
        }
    }
}


// ------------------------------------------------------------------------------------------------
// Pipelined forms of the forward and member-gradient kernels for D <= 64 (a whole member row is one staged chunk).
// Per workgroup iteration: [barrier] registers -> LDS (tile n) [barrier]; issue the row gathers of tile n+1 into
// registers; multiply tile n out of LDS.  The gathers (ids -> rows: two dependent trips to memory) complete behind 128
// MFMAs per wave instead of in front of them; the epilogue operands (hoisted first-order sums / member rows) ride the
// same prefetch and are read back from LDS, so the only global traffic inside the compute phase is the packed-weight
// stream (double-buffered one k-step ahead) and the result stores.
// ------------------------------------------------------------------------------------------------
template <int D, int NBLK>
__global__ __launch_bounds__(kBlockThreads, 2) void interact_fwd_pipe_kernel(
    const float* __restrict__ h, int64_t ld_h, const float* __restrict__ p, int64_t ld_p, const int32_t* __restrict__ i3,
    const float* __restrict__ wp, float* __restrict__ out, int64_t ld_out, int64_t n_edges) {
    static_assert(D == 32 || D == 64, "pipelined form stages whole rows");
    using S = TileShape<D>;
    constexpr int V4 = D / 4, LOADS = 3 * S::TE * V4 / kBlockThreads, PL = LOADS / 3, T_STEPS = D / 8;
    __shared__ __attribute__((aligned(16))) float tile[3][S::TE][S::STRIDE];
    __shared__ __attribute__((aligned(16))) float psum[S::TE][D];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int et = D == 32 ? wave : (wave & 1);
    const int jt = D == 32 ? 0 : (wave >> 1);
    const int row = et * 32 + (lane & 31), half = lane >> 5;
    const int64_t n_tiles = (n_edges + S::TE - 1) / S::TE;
    const v4f* wfrag = reinterpret_cast<const v4f*>(wp) + static_cast<int64_t>(jt) * NBLK * T_STEPS * kWave + lane;

    v4f hreg[LOADS], preg[LOADS];
    int64_t cur = -1, nxt = blockIdx.x;
    while (true) {
        if (cur >= 0) {
            __syncthreads();
#pragma unroll
            for (int x = 0; x < LOADS; ++x) {
                const int idx = tid + kBlockThreads * x;
                *reinterpret_cast<v4f*>(&tile[idx / (V4 * S::TE)][(idx / V4) % S::TE][(idx % V4) * 4]) = hreg[x];
            }
#pragma unroll
            for (int x = 0; x < PL; ++x) {
                const int idx = tid + kBlockThreads * x;
                *reinterpret_cast<v4f*>(&psum[idx / V4][(idx % V4) * 4]) = (preg[x] + preg[x + PL]) + preg[x + 2 * PL];
            }
            __syncthreads();
        }
        const bool have_next = nxt < n_tiles;
        if (have_next) {
            const int64_t e_base = nxt * S::TE;
#pragma unroll
            for (int x = 0; x < LOADS; ++x) {
                const int idx = tid + kBlockThreads * x;
                const int c4 = idx % V4, r = (idx / V4) % S::TE, m = idx / (V4 * S::TE);
                const int64_t e = e_base + r;
                const int64_t node = e < n_edges ? i3[e * 3 + m] : 0;
                hreg[x] = *reinterpret_cast<const v4f*>(h + node * ld_h + c4 * 4);
                preg[x] = *reinterpret_cast<const v4f*>(p + node * ld_p + c4 * 4);
            }
        }
        if (cur >= 0) {
            v16f acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            v4f bcur[NBLK], bnext[NBLK];
#pragma unroll
            for (int b = 0; b < NBLK; ++b) bcur[b] = wfrag[(b * T_STEPS) * kWave];
#pragma unroll
            for (int t = 0; t < T_STEPS; ++t) {
                if (t + 1 < T_STEPS) {
#pragma unroll
                    for (int b = 0; b < NBLK; ++b) bnext[b] = wfrag[(b * T_STEPS + t + 1) * kWave];
                }
                const int col = 8 * t + 4 * half;
                const v4f au = *reinterpret_cast<const v4f*>(&tile[0][row][col]);
                const v4f aq = *reinterpret_cast<const v4f*>(&tile[1][row][col]);
                const v4f ai = *reinterpret_cast<const v4f*>(&tile[2][row][col]);
                v4f z[4];
                z[0] = au * aq;
                z[1] = aq * ai;
                z[2] = ai * au;
                z[3] = z[0] * ai;
#pragma unroll
                for (int b = 0; b < NBLK; ++b)
#pragma unroll
                    for (int s2 = 0; s2 < 4; ++s2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(z[b][s2], bcur[b][s2], acc, 0, 0, 0);
#pragma unroll
                for (int b = 0; b < NBLK; ++b) bcur[b] = bnext[b];
            }
            const int j = jt * 32 + (lane & 31);
            const int64_t e_base = cur * S::TE;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int er = et * 32 + acc_row(r, lane);
                const int64_t e = e_base + er;
                if (e < n_edges) out[e * ld_out + j] = acc[r] + psum[er][j];
            }
        }
        if (!have_next) break;
        cur = nxt;
        nxt += gridDim.x;
    }
}

template <int D, int NBLK>
__global__ __launch_bounds__(kBlockThreads, 2) void interact_bwd_members_pipe_kernel(
    const float* __restrict__ h, int64_t ld_h, const int32_t* __restrict__ i3, const float* __restrict__ wq,
    const float* __restrict__ dout, int64_t ld_dout, float* __restrict__ g, int64_t n_edges) {
    static_assert(D == 32 || D == 64, "pipelined form stages whole rows");
    constexpr int ET = D == 32 ? 4 : 2, TE = ET * 32, STRIDE = D + kRowPad, V4 = D / 4, T_STEPS = D / 8;
    constexpr int DL = TE * V4 / kBlockThreads, HL = 3 * DL;
    __shared__ __attribute__((aligned(16))) float dtile[TE][STRIDE];
    __shared__ __attribute__((aligned(16))) float htile[3][TE][D];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5;
    const int et = wave % ET, ct = wave / ET;             // ET * (D/32) == 4 jobs: one per wave
    const int row = et * 32 + (lane & 31);
    const int64_t n_tiles = (n_edges + TE - 1) / TE;
    const v4f* wfrag = reinterpret_cast<const v4f*>(wq) + static_cast<int64_t>(ct) * NBLK * T_STEPS * kWave + lane;

    v4f dreg[DL], hreg[HL];
    int64_t cur = -1, nxt = blockIdx.x;
    while (true) {
        if (cur >= 0) {
            __syncthreads();
#pragma unroll
            for (int x = 0; x < DL; ++x) {
                const int idx = tid + kBlockThreads * x;
                *reinterpret_cast<v4f*>(&dtile[idx / V4][(idx % V4) * 4]) = dreg[x];
            }
#pragma unroll
            for (int x = 0; x < HL; ++x) {
                const int idx = tid + kBlockThreads * x;
                *reinterpret_cast<v4f*>(&htile[idx / (V4 * TE)][(idx / V4) % TE][(idx % V4) * 4]) = hreg[x];
            }
            __syncthreads();
        }
        const bool have_next = nxt < n_tiles;
        if (have_next) {
            const int64_t e_base = nxt * TE;
#pragma unroll
            for (int x = 0; x < DL; ++x) {
                const int idx = tid + kBlockThreads * x;
                const int64_t e = e_base + idx / V4;
                dreg[x] = e < n_edges ? *reinterpret_cast<const v4f*>(dout + e * ld_dout + (idx % V4) * 4) : v4f{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int x = 0; x < HL; ++x) {
                const int idx = tid + kBlockThreads * x;
                const int c4 = idx % V4, r = (idx / V4) % TE, m = idx / (V4 * TE);
                const int64_t e = e_base + r;
                const int64_t node = e < n_edges ? i3[e * 3 + m] : 0;
                hreg[x] = *reinterpret_cast<const v4f*>(h + node * ld_h + c4 * 4);
            }
        }
        if (cur >= 0) {
            v16f acc[NBLK];
#pragma unroll
            for (int b = 0; b < NBLK; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
            v4f bcur[NBLK], bnext[NBLK];
#pragma unroll
            for (int b = 0; b < NBLK; ++b) bcur[b] = wfrag[(b * T_STEPS) * kWave];
#pragma unroll
            for (int t = 0; t < T_STEPS; ++t) {
                if (t + 1 < T_STEPS) {
#pragma unroll
                    for (int b = 0; b < NBLK; ++b) bnext[b] = wfrag[(b * T_STEPS + t + 1) * kWave];
                }
                const v4f a = *reinterpret_cast<const v4f*>(&dtile[row][8 * t + 4 * half]);
#pragma unroll
                for (int b = 0; b < NBLK; ++b)
#pragma unroll
                    for (int s2 = 0; s2 < 4; ++s2) acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s2], bcur[b][s2], acc[b], 0, 0, 0);
#pragma unroll
                for (int b = 0; b < NBLK; ++b) bcur[b] = bnext[b];
            }
            const int c = ct * 32 + (lane & 31);
            const int64_t e_base = cur * TE;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int er = et * 32 + acc_row(r, lane);
                const int64_t e = e_base + er;
                if (e < n_edges) {
                    const float a = htile[0][er][c], b = htile[1][er][c], dd = htile[2][er][c];
                    const float z_uq = acc[0][r], z_qi = acc[1][r], z_iu = acc[2][r];
                    const float z_uqi = NBLK == 4 ? acc[NBLK - 1][r] : 0.f;
                    float* ge = g + e * 3 * D + c;
                    ge[0] = z_uq * b + z_iu * dd + z_uqi * (b * dd);
                    ge[D] = z_uq * a + z_qi * dd + z_uqi * (a * dd);
                    ge[2 * D] = z_qi * b + z_iu * a + z_uqi * (a * b);
                }
            }
        }
        if (!have_next) break;
        cur = nxt;
        nxt += gridDim.x;
    }
}


// ------------------------------------------------------------------------------------------------
// Wave-specialised forms for D <= 64: one 512-thread workgroup per CU, persistent over hyperedge tiles.
//   waves 4-7 (one per SIMD)  LOADERS:   gather the member rows (and the epilogue operands) of tile n+2 into registers,
//                                         drop tile n+1 into the other half of a double-buffered LDS image;
//   waves 0-3 (one per SIMD)  CONSUMERS: multiply tile n out of LDS on the matrix cores; the weight fragments they need
//                                         (128 VGPRs at D = 64) are loaded ONCE per kernel and stay in registers.
// One workgroup barrier per tile.  vmcnt retires loads in issue order, so a wave that both prefetches rows and streams
// weight fragments stalls its MFMAs behind its own prefetch; splitting the roles gives each role its own counter and
// leaves the consumers with no loads at all in steady state - their stream is LDS reads, MFMAs and result stores.
// ------------------------------------------------------------------------------------------------
constexpr int kWsThreads = 512;

template <int D, int NBLK>
__global__ __launch_bounds__(kWsThreads, 2) void interact_fwd_ws_kernel(
    const float* __restrict__ h, int64_t ld_h, const float* __restrict__ p, int64_t ld_p, const int32_t* __restrict__ i3,
    const float* __restrict__ wp, float* __restrict__ out, int64_t ld_out, int64_t n_edges) {
    static_assert(D == 32 || D == 64, "wave-specialised form stages whole rows");
    using S = TileShape<D>;
    constexpr int V4 = D / 4, LOADS = 3 * S::TE * V4 / kBlockThreads, PL = LOADS / 3, T_STEPS = D / 8;
    struct Buffer {
        float tile[3][S::TE][S::STRIDE];
        float psum[S::TE][D];
    };
    __shared__ __attribute__((aligned(16))) Buffer buf[2];
    const int64_t n_tiles = (n_edges + S::TE - 1) / S::TE;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;

    if (wave >= 4) {
        // ---------------- loaders ----------------
        // Schedule per trip (tile T is deposited while the consumers multiply the previous one):
        //   deposit rows(T)  ->  issue rows(T+g) with the ids fetched a trip ago  ->  issue ids(T+2g)  ->  barrier.
        // vmcnt retires in issue order, so every wait in a trip is for loads that were issued a whole trip earlier.
        const int tid = threadIdx.x - kBlockThreads;
        const int64_t g = gridDim.x;
        v4f hr[LOADS], pr[LOADS];
        int node[LOADS], node_next[LOADS];
        auto load_ids = [&](int64_t tile_id, int (&dst)[LOADS]) {
            const int64_t e_base = tile_id * S::TE;
#pragma unroll
            for (int x = 0; x < LOADS; ++x) {
                const int idx = tid + kBlockThreads * x;
                const int64_t e = e_base + (idx / V4) % S::TE;
                dst[x] = e < n_edges ? i3[e * 3 + idx / (V4 * S::TE)] : 0;
            }
        };
        auto issue_rows = [&](const int (&src)[LOADS]) {
#pragma unroll
            for (int x = 0; x < LOADS; ++x) {
                const int idx = tid + kBlockThreads * x;
                hr[x] = *reinterpret_cast<const v4f*>(h + static_cast<int64_t>(src[x]) * ld_h + (idx % V4) * 4);
                pr[x] = *reinterpret_cast<const v4f*>(p + static_cast<int64_t>(src[x]) * ld_p + (idx % V4) * 4);
            }
        };
        int64_t t = blockIdx.x;
        if (t < n_tiles) {
            load_ids(t, node);
            issue_rows(node);
        }
        if (t + g < n_tiles) load_ids(t + g, node_next);
        int which = 0;
        while (t < n_tiles) {
            Buffer& b = buf[which];
#pragma unroll
            for (int x = 0; x < LOADS; ++x) {
                const int idx = tid + kBlockThreads * x;
                *reinterpret_cast<v4f*>(&b.tile[idx / (V4 * S::TE)][(idx / V4) % S::TE][(idx % V4) * 4]) = hr[x];
            }
#pragma unroll
            for (int x = 0; x < PL; ++x) {
                const int idx = tid + kBlockThreads * x;
                *reinterpret_cast<v4f*>(&b.psum[idx / V4][(idx % V4) * 4]) = (pr[x] + pr[x + PL]) + pr[x + 2 * PL];
            }
            if (t + g < n_tiles) issue_rows(node_next);
            if (t + 2 * g < n_tiles) load_ids(t + 2 * g, node_next);
            __syncthreads();
            t += g;
            which ^= 1;
        }
        return;
    }
    // ---------------- consumers ----------------
    const int et = D == 32 ? wave : (wave & 1);
    const int jt = D == 32 ? 0 : (wave >> 1);
    const int row = et * 32 + (lane & 31), half = lane >> 5;
    const v4f* wfrag = reinterpret_cast<const v4f*>(wp) + static_cast<int64_t>(jt) * NBLK * T_STEPS * kWave + lane;
    v4f wreg[NBLK][T_STEPS];
#pragma unroll
    for (int b = 0; b < NBLK; ++b)
#pragma unroll
        for (int ts = 0; ts < T_STEPS; ++ts) wreg[b][ts] = wfrag[(b * T_STEPS + ts) * kWave];
    const int j = jt * 32 + (lane & 31);
    int which = 0;
    for (int64_t t = blockIdx.x; t < n_tiles; t += gridDim.x, which ^= 1) {
        __syncthreads();
        const Buffer& b = buf[which];
        v16f acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        // A operands one k-step ahead of the MFMAs that use them: the LDS round trip hides behind 16 MFMAs
        v4f au = *reinterpret_cast<const v4f*>(&b.tile[0][row][4 * half]);
        v4f aq = *reinterpret_cast<const v4f*>(&b.tile[1][row][4 * half]);
        v4f ai = *reinterpret_cast<const v4f*>(&b.tile[2][row][4 * half]);
#pragma unroll
        for (int ts = 0; ts < T_STEPS; ++ts) {
            v4f z[4];
            z[0] = au * aq;
            z[1] = aq * ai;
            z[2] = ai * au;
            z[3] = z[0] * ai;
            if (ts + 1 < T_STEPS) {
                const int col = 8 * (ts + 1) + 4 * half;
                au = *reinterpret_cast<const v4f*>(&b.tile[0][row][col]);
                aq = *reinterpret_cast<const v4f*>(&b.tile[1][row][col]);
                ai = *reinterpret_cast<const v4f*>(&b.tile[2][row][col]);
            }
#pragma unroll
            for (int bk = 0; bk < NBLK; ++bk)
#pragma unroll
                for (int s2 = 0; s2 < 4; ++s2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(z[bk][s2], wreg[bk][ts][s2], acc, 0, 0, 0);
        }
        // epilogue: all 16 first-order sums are read from LDS in one batch, then added and stored
        const int64_t e_base = t * S::TE;
        float first[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) first[r] = b.psum[et * 32 + acc_row(r, lane)][j];
        float* orow = out + (e_base + et * 32) * ld_out + j;
        if (e_base + S::TE <= n_edges) {
#pragma unroll
            for (int r = 0; r < 16; ++r) orow[static_cast<int64_t>(acc_row(r, lane)) * ld_out] = acc[r] + first[r];
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (e_base + et * 32 + acc_row(r, lane) < n_edges) orow[static_cast<int64_t>(acc_row(r, lane)) * ld_out] = acc[r] + first[r];
        }
    }
}

template <int D, int NBLK>
__global__ __launch_bounds__(kWsThreads, 2) void interact_bwd_members_ws_kernel(
    const float* __restrict__ h, int64_t ld_h, const int32_t* __restrict__ i3, const float* __restrict__ wq,
    const float* __restrict__ dout, int64_t ld_dout, float* __restrict__ g, int64_t n_edges) {
    static_assert(D == 32 || D == 64, "wave-specialised form stages whole rows");
    constexpr int ET = D == 32 ? 4 : 2, TE = ET * 32, STRIDE = D + kRowPad, V4 = D / 4, T_STEPS = D / 8;
    constexpr int DL = TE * V4 / kBlockThreads, HL = 3 * DL;
    struct Buffer {
        float dtile[TE][STRIDE];
        float htile[3][TE][D];
    };
    __shared__ __attribute__((aligned(16))) Buffer buf[2];
    const int64_t n_tiles = (n_edges + TE - 1) / TE;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;

    if (wave >= 4) {
        // loaders: deposit rows(T) -> issue rows(T+g) with ids fetched a trip ago -> issue ids(T+2g) -> barrier
        const int tid = threadIdx.x - kBlockThreads;
        const int64_t g = gridDim.x;
        v4f dr[DL], hr[HL];
        int node[HL], node_next[HL];
        auto load_ids = [&](int64_t tile_id, int (&dst)[HL]) {
            const int64_t e_base = tile_id * TE;
#pragma unroll
            for (int x = 0; x < HL; ++x) {
                const int idx = tid + kBlockThreads * x;
                const int64_t e = e_base + (idx / V4) % TE;
                dst[x] = e < n_edges ? i3[e * 3 + idx / (V4 * TE)] : 0;
            }
        };
        auto issue_rows = [&](int64_t tile_id, const int (&src)[HL]) {
            const int64_t e_base = tile_id * TE;
#pragma unroll
            for (int x = 0; x < DL; ++x) {
                const int idx = tid + kBlockThreads * x;
                const int64_t e = e_base + idx / V4;
                dr[x] = e < n_edges ? *reinterpret_cast<const v4f*>(dout + e * ld_dout + (idx % V4) * 4) : v4f{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int x = 0; x < HL; ++x) {
                const int idx = tid + kBlockThreads * x;
                hr[x] = *reinterpret_cast<const v4f*>(h + static_cast<int64_t>(src[x]) * ld_h + (idx % V4) * 4);
            }
        };
        int64_t t = blockIdx.x;
        if (t < n_tiles) {
            load_ids(t, node);
            issue_rows(t, node);
        }
        if (t + g < n_tiles) load_ids(t + g, node_next);
        int which = 0;
        while (t < n_tiles) {
            Buffer& b = buf[which];
#pragma unroll
            for (int x = 0; x < DL; ++x) {
                const int idx = tid + kBlockThreads * x;
                *reinterpret_cast<v4f*>(&b.dtile[idx / V4][(idx % V4) * 4]) = dr[x];
            }
#pragma unroll
            for (int x = 0; x < HL; ++x) {
                const int idx = tid + kBlockThreads * x;
                *reinterpret_cast<v4f*>(&b.htile[idx / (V4 * TE)][(idx / V4) % TE][(idx % V4) * 4]) = hr[x];
            }
            if (t + g < n_tiles) issue_rows(t + g, node_next);
            if (t + 2 * g < n_tiles) load_ids(t + 2 * g, node_next);
            __syncthreads();
            t += g;
            which ^= 1;
        }
        return;
    }
    const int et = wave % ET, ct = wave / ET;
    const int row = et * 32 + (lane & 31), half = lane >> 5;
    const v4f* wfrag = reinterpret_cast<const v4f*>(wq) + static_cast<int64_t>(ct) * NBLK * T_STEPS * kWave + lane;
    v4f wreg[NBLK][T_STEPS];
#pragma unroll
    for (int b = 0; b < NBLK; ++b)
#pragma unroll
        for (int ts = 0; ts < T_STEPS; ++ts) wreg[b][ts] = wfrag[(b * T_STEPS + ts) * kWave];
    const int c = ct * 32 + (lane & 31);
    int which = 0;
    for (int64_t t = blockIdx.x; t < n_tiles; t += gridDim.x, which ^= 1) {
        __syncthreads();
        const Buffer& b = buf[which];
        v16f acc[NBLK];
#pragma unroll
        for (int bk = 0; bk < NBLK; ++bk)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[bk][r] = 0.f;
        v4f a = *reinterpret_cast<const v4f*>(&b.dtile[row][4 * half]);
#pragma unroll
        for (int ts = 0; ts < T_STEPS; ++ts) {
            const v4f a_now = a;
            if (ts + 1 < T_STEPS) a = *reinterpret_cast<const v4f*>(&b.dtile[row][8 * (ts + 1) + 4 * half]);
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2)
#pragma unroll
                for (int bk = 0; bk < NBLK; ++bk) acc[bk] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_now[s2], wreg[bk][ts][s2], acc[bk], 0, 0, 0);
        }
        // epilogue in batches of four rows: 12 LDS reads in flight, then the product rule and 12 stores
        const int64_t e_base = t * TE;
        const bool full = e_base + TE <= n_edges;
        float* gbase = g + (e_base + et * 32) * 3 * D + c;
#pragma unroll
        for (int r0 = 0; r0 < 16; r0 += 4) {
            float hu[4], hq[4], hi[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int er = et * 32 + acc_row(r0 + k, lane);
                hu[k] = b.htile[0][er][c];
                hq[k] = b.htile[1][er][c];
                hi[k] = b.htile[2][er][c];
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int r = r0 + k;
                const float z_uq = acc[0][r], z_qi = acc[1][r], z_iu = acc[2][r];
                const float z_uqi = NBLK == 4 ? acc[NBLK - 1][r] : 0.f;
                const float gu = z_uq * hq[k] + z_iu * hi[k] + z_uqi * (hq[k] * hi[k]);
                const float gq = z_uq * hu[k] + z_qi * hi[k] + z_uqi * (hu[k] * hi[k]);
                const float gi = z_qi * hq[k] + z_iu * hu[k] + z_uqi * (hu[k] * hq[k]);
                if (full || e_base + et * 32 + acc_row(r, lane) < n_edges) {
                    float* ge = gbase + static_cast<int64_t>(acc_row(r, lane)) * 3 * D;
                    ge[0] = gu;
                    ge[D] = gq;
                    ge[2 * D] = gi;
                }
            }
        }
    }
}


// ------------------------------------------------------------------------------------------------
// Wave-specialised member-gradient kernel for D = 128.  The weight fragments no longer fit in registers (512 VGPRs), so the
// consumers stream them from L2 (packed, two k-steps ahead) while the loaders stream the dout rows into a double-buffered LDS
// image; the member values of the product rule are requested by the consumers right behind the first fragments of a job, so
// they arrive under the MFMAs.  (A chunked wave-specialised FORWARD for D = 128 was built and measured equal to the plain
// MFMA tiling - 3.45 vs 3.41 ms at E = 2.2 M - so the forward keeps the plain kernel at this width.)
// ------------------------------------------------------------------------------------------------
template <int D, int NBLK>
__global__ __launch_bounds__(kWsThreads, 2) void interact_bwd_members_wsbig_kernel(
    const float* __restrict__ h, int64_t ld_h, const int32_t* __restrict__ i3, const float* __restrict__ wq,
    const float* __restrict__ dout, int64_t ld_dout, float* __restrict__ g_out, int64_t n_edges) {
    static_assert(D == 128, "chunked wave-specialised form");
    constexpr int TE = 64, STRIDE = D + kRowPad, V4 = D / 4, DL = TE * V4 / kBlockThreads, T_STEPS = D / 8, JOBS = 2 * (D / 32);
    __shared__ __attribute__((aligned(16))) float dtile[2][TE][STRIDE];
    __shared__ int ids[2][TE][3];
    const int64_t n_tiles = (n_edges + TE - 1) / TE;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t g = gridDim.x;

    if (wave >= 4) {
        // loaders: the dout rows are a plain stream; one tile of lead in registers
        const int tid = threadIdx.x - kBlockThreads;
        v4f dr[DL];
        int my_id = 0;
        auto issue = [&](int64_t tile_id) {
            const int64_t e_base = tile_id * TE;
#pragma unroll
            for (int x = 0; x < DL; ++x) {
                const int idx = tid + kBlockThreads * x;
                const int64_t e = e_base + idx / V4;
                dr[x] = e < n_edges ? *reinterpret_cast<const v4f*>(dout + e * ld_dout + (idx % V4) * 4) : v4f{0.f, 0.f, 0.f, 0.f};
            }
            const int64_t pos = e_base * 3 + tid;
            my_id = (tid < TE * 3 && pos < n_edges * 3) ? i3[pos] : 0;
        };
        int64_t t = blockIdx.x;
        if (t < n_tiles) issue(t);
        int which = 0;
        while (t < n_tiles) {
#pragma unroll
            for (int x = 0; x < DL; ++x) {
                const int idx = tid + kBlockThreads * x;
                *reinterpret_cast<v4f*>(&dtile[which][idx / V4][(idx % V4) * 4]) = dr[x];
            }
            if (tid < TE * 3) (&ids[which][0][0])[tid] = my_id;
            if (t + g < n_tiles) issue(t + g);
            __syncthreads();
            t += g;
            which ^= 1;
        }
        return;
    }
    const int half = lane >> 5;
    const v4f* wq4 = reinterpret_cast<const v4f*>(wq) + lane;
    int which = 0;
    for (int64_t t = blockIdx.x; t < n_tiles; t += g, which ^= 1) {
        __syncthreads();
        const int64_t e_base = t * TE;
        const bool full = e_base + TE <= n_edges;
        for (int job = wave; job < JOBS; job += 4) {
            const int et = job & 1, ct = job >> 1;
            const int row = et * 32 + (lane & 31);
            const int c = ct * 32 + (lane & 31);
            v16f acc[NBLK];
#pragma unroll
            for (int bk = 0; bk < NBLK; ++bk)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[bk][r] = 0.f;
            // weight fragments two k-steps ahead; the member values of the epilogue are requested right behind the first two
            // fragment sets, so they arrive under the MFMAs instead of in front of the stores
            v4f b0[NBLK], b1[NBLK], b2[NBLK];
#pragma unroll
            for (int bk = 0; bk < NBLK; ++bk) {
                b0[bk] = wq4[(static_cast<int64_t>(ct * NBLK + bk) * T_STEPS + 0) * kWave];
                b1[bk] = wq4[(static_cast<int64_t>(ct * NBLK + bk) * T_STEPS + 1) * kWave];
            }
            float hu[16], hq[16], hi[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int er = et * 32 + acc_row(r, lane);
                hu[r] = h[static_cast<int64_t>(ids[which][er][0]) * ld_h + c];
                hq[r] = h[static_cast<int64_t>(ids[which][er][1]) * ld_h + c];
                hi[r] = h[static_cast<int64_t>(ids[which][er][2]) * ld_h + c];
            }
#pragma unroll
            for (int ts = 0; ts < T_STEPS; ++ts) {
                if (ts + 2 < T_STEPS) {
#pragma unroll
                    for (int bk = 0; bk < NBLK; ++bk) b2[bk] = wq4[(static_cast<int64_t>(ct * NBLK + bk) * T_STEPS + ts + 2) * kWave];
                }
                const v4f a = *reinterpret_cast<const v4f*>(&dtile[which][row][8 * ts + 4 * half]);
#pragma unroll
                for (int s2 = 0; s2 < 4; ++s2)
#pragma unroll
                    for (int bk = 0; bk < NBLK; ++bk) acc[bk] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s2], b0[bk][s2], acc[bk], 0, 0, 0);
#pragma unroll
                for (int bk = 0; bk < NBLK; ++bk) {
                    b0[bk] = b1[bk];
                    b1[bk] = b2[bk];
                }
            }
            float* gbase = g_out + (e_base + et * 32) * 3 * D + c;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float z_uq = acc[0][r], z_qi = acc[1][r], z_iu = acc[2][r];
                const float z_uqi = NBLK == 4 ? acc[NBLK - 1][r] : 0.f;
                if (full || e_base + et * 32 + acc_row(r, lane) < n_edges) {
                    float* ge = gbase + static_cast<int64_t>(acc_row(r, lane)) * 3 * D;
                    ge[0] = z_uq * hq[r] + z_iu * hi[r] + z_uqi * (hq[r] * hi[r]);
                    ge[D] = z_uq * hu[r] + z_qi * hi[r] + z_uqi * (hu[r] * hi[r]);
                    ge[2 * D] = z_qi * hq[r] + z_iu * hu[r] + z_uqi * (hu[r] * hq[r]);
                }
            }
        }
    }
}

// dW, same roles: workgroup (x, y) owns the 64 x 64 x NBLK sub-block y = (js, cs) of the d x NBLK*d gradient (d a multiple of 64)
// for the hyperedge tiles x, x + gridDim.x, ...; its loaders fetch the matching 64-column slices of dout and of the member rows.
template <int NBLK>
__global__ __launch_bounds__(kWsThreads, 2) void interact_bwd_weight_ws_kernel(
    const float* __restrict__ h, int64_t ld_h, const int32_t* __restrict__ i3, const float* __restrict__ dout, int64_t ld_dout,
    float* __restrict__ slabs, int64_t n_edges, int d) {
    constexpr int D = 64, TE = 64, V4 = D / 4, DL = TE * V4 / kBlockThreads, HL = 3 * DL;
    struct Buffer {
        float dtile[TE][D];
        float mtile[3][TE][D];
    };
    __shared__ __attribute__((aligned(16))) Buffer buf[2];
    const int64_t n_tiles = (n_edges + TE - 1) / TE;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int subs = d / D;
    const int js = blockIdx.y / subs, cs = blockIdx.y % subs;

    if (wave >= 4) {
        // loaders: deposit rows(T) -> issue rows(T+g) with ids fetched a trip ago -> issue ids(T+2g) -> barrier
        const int tid = threadIdx.x - kBlockThreads;
        const int64_t g = gridDim.x;
        v4f dr[DL], hr[HL];
        int node[HL], node_next[HL];
        auto load_ids = [&](int64_t tile_id, int (&dst)[HL]) {
            const int64_t e_base = tile_id * TE;
#pragma unroll
            for (int x = 0; x < HL; ++x) {
                const int idx = tid + kBlockThreads * x;
                const int64_t e = e_base + (idx / V4) % TE;
                dst[x] = e < n_edges ? i3[e * 3 + idx / (V4 * TE)] : 0;
            }
        };
        auto issue_rows = [&](int64_t tile_id, const int (&src)[HL]) {
            const int64_t e_base = tile_id * TE;
#pragma unroll
            for (int x = 0; x < DL; ++x) {
                const int idx = tid + kBlockThreads * x;
                const int64_t e = e_base + idx / V4;
                dr[x] = e < n_edges ? *reinterpret_cast<const v4f*>(dout + e * ld_dout + js * D + (idx % V4) * 4) : v4f{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int x = 0; x < HL; ++x) {
                const int idx = tid + kBlockThreads * x;
                hr[x] = *reinterpret_cast<const v4f*>(h + static_cast<int64_t>(src[x]) * ld_h + cs * D + (idx % V4) * 4);
            }
        };
        int64_t t = blockIdx.x;
        if (t < n_tiles) {
            load_ids(t, node);
            issue_rows(t, node);
        }
        if (t + g < n_tiles) load_ids(t + g, node_next);
        int which = 0;
        while (t < n_tiles) {
            Buffer& b = buf[which];
#pragma unroll
            for (int x = 0; x < DL; ++x) {
                const int idx = tid + kBlockThreads * x;
                *reinterpret_cast<v4f*>(&b.dtile[idx / V4][(idx % V4) * 4]) = dr[x];
            }
#pragma unroll
            for (int x = 0; x < HL; ++x) {
                const int idx = tid + kBlockThreads * x;
                *reinterpret_cast<v4f*>(&b.mtile[idx / (V4 * TE)][(idx / V4) % TE][(idx % V4) * 4]) = hr[x];
            }
            if (t + g < n_tiles) issue_rows(t + g, node_next);
            if (t + 2 * g < n_tiles) load_ids(t + 2 * g, node_next);
            __syncthreads();
            t += g;
            which ^= 1;
        }
        return;
    }
    const int half = lane >> 5, l31 = lane & 31;
    const int jt = wave & 1, ct = wave >> 1;
    v16f acc[NBLK];
#pragma unroll
    for (int bk = 0; bk < NBLK; ++bk)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[bk][r] = 0.f;
    int which = 0;
    for (int64_t t = blockIdx.x; t < n_tiles; t += gridDim.x, which ^= 1) {
        __syncthreads();
        const Buffer& b = buf[which];
#pragma unroll 8
        for (int kk = 0; kk < TE / 2; ++kk) {
            const int e = 2 * kk + half;
            const float a = b.dtile[e][jt * 32 + l31];
            const float hu = b.mtile[0][e][ct * 32 + l31], hq = b.mtile[1][e][ct * 32 + l31], hi = b.mtile[2][e][ct * 32 + l31];
            float z[4];
            z[0] = hu * hq;
            z[1] = hq * hi;
            z[2] = hi * hu;
            z[3] = z[0] * hi;
#pragma unroll
            for (int bk = 0; bk < NBLK; ++bk) acc[bk] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, z[bk], acc[bk], 0, 0, 0);
        }
    }
    float* slab = slabs + static_cast<int64_t>(blockIdx.x) * d * NBLK * d;          // slab x: a full [d][NBLK*d] matrix
#pragma unroll
    for (int bk = 0; bk < NBLK; ++bk)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            slab[static_cast<int64_t>(js * D + jt * 32 + acc_row(r, lane)) * NBLK * d + bk * d + cs * D + ct * 32 + l31] = acc[bk][r];
}

// weights: workgroup (x, y) owns the SW x (NBLK*SW) sub-block y = (js, cs) of dW for the hyperedge tiles x, x + gridDim.x, ...
// and keeps it in MFMA accumulators for the whole sweep (contraction index = hyperedge, 2 per MFMA); it ends by writing
// its partial sub-block into slab x, and slab_reduce_kernel adds the slabs in a fixed order (bitwise reproducible).
// Software pipeline: the rows of tile n+1 are fetched into registers while tile n is multiplied out of LDS, so the
// gather latency (ids -> rows, two dependent trips) hides behind 32 MFMA steps.
template <int SW, int NBLK>
__global__ __launch_bounds__(kBlockThreads, 2) void interact_bwd_weight_mfma_kernel(
    const float* __restrict__ h, int64_t ld_h, const int32_t* __restrict__ i3, const float* __restrict__ dout, int64_t ld_dout,
    float* __restrict__ slabs, int64_t n_edges, int d) {
    constexpr int TE = 64;
    constexpr int WT = SW / 32;                       // 32-wide tiles per side of the sub-block (2 for SW = 64, 1 for 32)
    constexpr int TILES = WT * WT * NBLK;             // accumulator tiles of the sub-block
    constexpr int PER_WAVE = (TILES + kWavesPerBlock - 1) / kWavesPerBlock;
    constexpr int V4_PER_ROW = SW / 4;
    constexpr int D_LOADS = TE * V4_PER_ROW / kBlockThreads;          // dout tile float4s per thread
    constexpr int M_LOADS = 3 * TE * V4_PER_ROW / kBlockThreads;      // member-row float4s per thread
    __shared__ __attribute__((aligned(16))) float dtile[TE][SW];
    __shared__ __attribute__((aligned(16))) float mtile[3][TE][SW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int subs = d / SW;
    const int js = blockIdx.y / subs, cs = blockIdx.y % subs;
    const int64_t n_tiles = (n_edges + TE - 1) / TE;
    // wave -> (jt, ct) pair, all NBLK blocks (SW = 64: 4 pairs, one per wave); SW = 32: one pair, block b = wave
    const int jt = WT == 2 ? (wave & 1) : 0, ct = WT == 2 ? (wave >> 1) : 0;
    v16f acc[PER_WAVE];
#pragma unroll
    for (int x = 0; x < PER_WAVE; ++x)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[x][r] = 0.f;

    v4f dreg[D_LOADS], mreg[M_LOADS];
    int64_t cur = -1, nxt = blockIdx.x;
    while (true) {
        if (cur >= 0) {
            __syncthreads();                              // everyone is done reading the previous tile
#pragma unroll
            for (int x = 0; x < D_LOADS; ++x) {
                const int idx = tid + kBlockThreads * x;
                *reinterpret_cast<v4f*>(&dtile[idx / V4_PER_ROW][(idx % V4_PER_ROW) * 4]) = dreg[x];
            }
#pragma unroll
            for (int x = 0; x < M_LOADS; ++x) {
                const int idx = tid + kBlockThreads * x;
                *reinterpret_cast<v4f*>(&mtile[idx / (V4_PER_ROW * TE)][(idx / V4_PER_ROW) % TE][(idx % V4_PER_ROW) * 4]) = mreg[x];
            }
            __syncthreads();
        }
        const bool have_next = nxt < n_tiles;
        if (have_next) {                                  // the one fetch site: rows of tile `nxt` -> registers
            const int64_t e_base = nxt * TE;
#pragma unroll
            for (int x = 0; x < D_LOADS; ++x) {
                const int idx = tid + kBlockThreads * x;
                const int c4 = idx % V4_PER_ROW, r = idx / V4_PER_ROW;
                const int64_t e = e_base + r;
                dreg[x] = e < n_edges ? *reinterpret_cast<const v4f*>(dout + e * ld_dout + js * SW + c4 * 4) : v4f{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int x = 0; x < M_LOADS; ++x) {
                const int idx = tid + kBlockThreads * x;
                const int c4 = idx % V4_PER_ROW, r = (idx / V4_PER_ROW) % TE, m = idx / (V4_PER_ROW * TE);
                const int64_t e = e_base + r;
                const int64_t node = e < n_edges ? i3[e * 3 + m] : 0;
                mreg[x] = *reinterpret_cast<const v4f*>(h + node * ld_h + cs * SW + c4 * 4);
            }
        }
        if (cur >= 0) {
#pragma unroll 4
            for (int kk = 0; kk < TE / 2; ++kk) {
                const int e = 2 * kk + half;
                const float a = dtile[e][jt * 32 + l31];
                const float hu = mtile[0][e][ct * 32 + l31], hq = mtile[1][e][ct * 32 + l31], hi = mtile[2][e][ct * 32 + l31];
                float z[4];
                z[0] = hu * hq;
                z[1] = hq * hi;
                z[2] = hi * hu;
                z[3] = z[0] * hi;
                if (WT == 2) {
#pragma unroll
                    for (int b = 0; b < NBLK; ++b) acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, z[b], acc[b], 0, 0, 0);
                } else {
                    const float zb = wave == 0 ? z[0] : wave == 1 ? z[1] : wave == 2 ? z[2] : z[3];
                    if (wave < NBLK) acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, zb, acc[0], 0, 0, 0);
                }
            }
        }
        if (!have_next) break;
        cur = nxt;
        nxt += gridDim.x;
    }
    // slab[x] is a full [d][NBLK*d] matrix; element (j, b*d + c)
    float* slab = slabs + static_cast<int64_t>(blockIdx.x) * d * NBLK * d;
#pragma unroll
    for (int x = 0; x < PER_WAVE; ++x) {
        const int b = WT == 2 ? x : wave;
        if (b >= NBLK) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = js * SW + jt * 32 + acc_row(r, lane);
            const int c = cs * SW + ct * 32 + l31;
            slab[static_cast<int64_t>(j) * NBLK * d + b * d + c] = acc[x][r];
        }
    }
}

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

__global__ __launch_bounds__(kBlockThreads) void slab_reduce_kernel(const float* __restrict__ slabs, int n_slabs, int d, int nblk,
                                                                    float* __restrict__ dw, int64_t ld_dw) {
    const int width = nblk * d;
    const int64_t total = static_cast<int64_t>(d) * width;
    for (int64_t base = static_cast<int64_t>(blockIdx.x) * kWave; base < total; base += static_cast<int64_t>(gridDim.x) * kWave) {
        const int64_t idx = base + (threadIdx.x & 63);
        const float acc = slab_sum(slabs, n_slabs, total, idx, idx < total);
        if ((threadIdx.x >> 6) == 0 && idx < total) {
            const int j = static_cast<int>(idx / width), col = static_cast<int>(idx - static_cast<int64_t>(j) * width);
            dw[static_cast<int64_t>(j) * ld_dw + 3 * static_cast<int64_t>(d) + col] = acc;
        }
    }
}

// ================================================================================================
// Node-level dense transforms (K4 and the hoisted first-order blocks): out[v] = x[v] * W_type(v)^T (+ bias).
// Nodes are typed by contiguous id ranges (users | queries | items); a layer either uses one weight for every row
// (feature_transform) or one d x d block per type (the u / q / i blocks of aggregation.weight).  All three kernels
// are HBM-bound streams over [N, d]; the matrix cores only keep the arithmetic out of the way:
//   row_gemm_kernel        out[v][n]   = sum_k in[v][k] * B_t[k][n] (+ bias[n])     fwd (B = W^T) and input-grad (B = W)
//   dense_weight_grad      dW_t[c][j]  = sum_{v in t} dout[v][c] * x[v][j],  db[c] = sum_v dout[v][c]
// ================================================================================================
struct TypePlan {
    int64_t begin[4];        // row ranges of the three node types: [begin[t], begin[t+1])
    int tile_prefix[4];      // cumulative workgroup tiles per type
};

// pk[type][xt][t][lane][4]:  transpose == 0:  W_t[32xt + r][8t + 4h + s]     (B[k][n] = W[n][k],  out = in * W^T)
//                            transpose == 1:  W_t[8t + 4h + s][32xt + r]     (B[k][n] = W[k][n],  out = in * W)
// with W_t[a][b] = w[a * ld_w + t * type_stride + b], r = lane & 31, h = lane >> 5.
__global__ __launch_bounds__(kBlockThreads) void pack_dense_kernel(const float* __restrict__ w, int64_t ld_w, int64_t type_stride,
                                                                   int n_types, int d, int transpose, float* __restrict__ pk) {
    const int t_count = d / 8;
    const int per_type = (d / 32) * t_count * kWave;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < per_type * n_types; idx += gridDim.x * blockDim.x) {
        const int type = idx / per_type, rem = idx - type * per_type;
        const int lane = rem & (kWave - 1), t = (rem >> 6) % t_count, xt = (rem >> 6) / t_count;
        const int r = lane & 31, half = lane >> 5;
        const float* wt = w + type * type_stride;
        float4 v;
        if (transpose == 0) {
            const float* src = wt + static_cast<int64_t>(32 * xt + r) * ld_w + 8 * t + 4 * half;
            v = make_float4(src[0], src[1], src[2], src[3]);
        } else {
            const float* src = wt + static_cast<int64_t>(8 * t + 4 * half) * ld_w + 32 * xt + r;
            v = make_float4(src[0], src[ld_w], src[2 * ld_w], src[3 * ld_w]);
        }
        *reinterpret_cast<float4*>(pk + static_cast<int64_t>(idx) * 4) = v;
    }
}

// Register-prefetch pipeline: the rows of tile n+1 are fetched while tile n is multiplied.  vmcnt retires in issue order, so
// the (tiny, cache-resident) weight fragments of the current tile are pulled into registers BEFORE the prefetch is issued;
// D = 256 would need 512 registers for that and keeps the plain fetch-then-multiply order.
template <int D>
__global__ __launch_bounds__(kBlockThreads) void row_gemm_kernel(const float* __restrict__ in, int64_t ld_in, const float* __restrict__ pk,
                                                                 int64_t pk_type_stride, const float* __restrict__ bias, int bias_mask,
                                                                 TypePlan plan, float* __restrict__ out, int64_t ld_out) {
    constexpr int ET = D == 32 ? 4 : 2, TE = ET * 32, STRIDE = D + kRowPad, JOBS = ET * (D / 32);
    constexpr int V4_PER_ROW = D / 4, LOADS = TE * V4_PER_ROW / kBlockThreads, T_STEPS = D / 8;
    constexpr int JOBS_PER_WAVE = JOBS / kWavesPerBlock;
    constexpr bool HOLD_B = D <= 128;
    __shared__ __attribute__((aligned(16))) float xt[TE][STRIDE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5;
    const int total_tiles = plan.tile_prefix[3];
    auto tile_type = [&](int tile_id) { return tile_id >= plan.tile_prefix[2] ? 2 : (tile_id >= plan.tile_prefix[1] ? 1 : 0); };

    v4f xreg[LOADS];
    v4f breg[HOLD_B ? JOBS_PER_WAVE : 1][HOLD_B ? T_STEPS : 1];
    int cur = -1, nxt = blockIdx.x;
    while (true) {
        if (cur >= 0) {
            __syncthreads();
#pragma unroll
            for (int k = 0; k < LOADS; ++k) {
                const int idx = tid + kBlockThreads * k;
                *reinterpret_cast<v4f*>(&xt[idx / V4_PER_ROW][(idx % V4_PER_ROW) * 4]) = xreg[k];
            }
            __syncthreads();
            if (HOLD_B) {
                const v4f* pk4 = reinterpret_cast<const v4f*>(pk + tile_type(cur) * pk_type_stride) + lane;
#pragma unroll
                for (int jw = 0; jw < JOBS_PER_WAVE; ++jw) {
                    const int ct = (wave + jw * kWavesPerBlock) / ET;
#pragma unroll
                    for (int t = 0; t < T_STEPS; ++t) breg[jw][t] = pk4[(static_cast<int64_t>(ct) * T_STEPS + t) * kWave];
                }
            }
        }
        const bool have_next = nxt < total_tiles;
        if (have_next && (HOLD_B || cur < 0)) {
            const int type = tile_type(nxt);
            const int64_t r_base = plan.begin[type] + static_cast<int64_t>(nxt - plan.tile_prefix[type]) * TE;
            const int64_t r_end = plan.begin[type + 1];
#pragma unroll
            for (int k = 0; k < LOADS; ++k) {
                const int idx = tid + kBlockThreads * k;
                const int64_t v = r_base + idx / V4_PER_ROW;
                xreg[k] = v < r_end ? *reinterpret_cast<const v4f*>(in + v * ld_in + (idx % V4_PER_ROW) * 4) : v4f{0.f, 0.f, 0.f, 0.f};
            }
        }
        if (cur >= 0) {
            const int type = tile_type(cur);
            const int64_t r_base = plan.begin[type] + static_cast<int64_t>(cur - plan.tile_prefix[type]) * TE;
            const int64_t r_end = plan.begin[type + 1];
            const v4f* pk4 = reinterpret_cast<const v4f*>(pk + type * pk_type_stride) + lane;
            const bool with_bias = bias != nullptr && ((bias_mask >> type) & 1);
#pragma unroll
            for (int jw = 0; jw < JOBS_PER_WAVE; ++jw) {
                const int job = wave + jw * kWavesPerBlock;
                const int et = job % ET, ct = job / ET;
                const int row = et * 32 + (lane & 31);
                v16f acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
                for (int t = 0; t < T_STEPS; ++t) {
                    const v4f a = *reinterpret_cast<const v4f*>(&xt[row][8 * t + 4 * half]);
                    const v4f bf = HOLD_B ? breg[HOLD_B ? jw : 0][HOLD_B ? t : 0] : pk4[(static_cast<int64_t>(ct) * T_STEPS + t) * kWave];
#pragma unroll
                    for (int s2 = 0; s2 < 4; ++s2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s2], bf[s2], acc, 0, 0, 0);
                }
                const int c = ct * 32 + (lane & 31);
                const float bv = with_bias ? bias[c] : 0.f;
                float* orow = out + (r_base + et * 32) * ld_out + c;
                if (r_base + TE <= r_end) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) orow[static_cast<int64_t>(acc_row(r, lane)) * ld_out] = acc[r] + bv;
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (r_base + et * 32 + acc_row(r, lane) < r_end) orow[static_cast<int64_t>(acc_row(r, lane)) * ld_out] = acc[r] + bv;
                }
            }
        }
        if (!HOLD_B && cur >= 0 && have_next) {           // plain order for D = 256: fetch the next tile after the multiply
            const int type = tile_type(nxt);
            const int64_t r_base = plan.begin[type] + static_cast<int64_t>(nxt - plan.tile_prefix[type]) * TE;
            const int64_t r_end = plan.begin[type + 1];
#pragma unroll
            for (int k = 0; k < LOADS; ++k) {
                const int idx = tid + kBlockThreads * k;
                const int64_t v = r_base + idx / V4_PER_ROW;
                xreg[k] = v < r_end ? *reinterpret_cast<const v4f*>(in + v * ld_in + (idx % V4_PER_ROW) * 4) : v4f{0.f, 0.f, 0.f, 0.f};
            }
        }
        if (!have_next) break;
        cur = nxt;
        nxt += gridDim.x;
    }
}

// grid = (slabs, (d/SW)^2 sub-blocks, weight types).  Slab layout: [type][slab][d][d] then bias part [type][slab][d].
// Same register-prefetch pipeline as the interactive weight-gradient kernel.
template <int SW>
__global__ __launch_bounds__(kBlockThreads) void dense_weight_grad_kernel(const float* __restrict__ dout, int64_t ld_dout,
                                                                          const float* __restrict__ x, int64_t ld_x, TypePlan plan,
                                                                          int single_weight, float* __restrict__ slabs,
                                                                          float* __restrict__ bias_slabs, int d) {
    constexpr int TE = 64, WT = SW / 32, V4_PER_ROW = SW / 4, LOADS = TE * V4_PER_ROW / kBlockThreads;
    __shared__ __attribute__((aligned(16))) float dtile[TE][SW];
    __shared__ __attribute__((aligned(16))) float xtile[TE][SW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int subs = d / SW;
    const int js = blockIdx.y / subs, cs = blockIdx.y % subs;
    const int type = blockIdx.z;
    const int64_t r_begin = single_weight ? plan.begin[0] : plan.begin[type];
    const int64_t r_end = single_weight ? plan.begin[3] : plan.begin[type + 1];
    const int64_t n_tiles = (r_end - r_begin + TE - 1) / TE;
    const int jt = WT == 2 ? (wave & 1) : 0, ct = WT == 2 ? (wave >> 1) : 0;
    const bool active = WT == 2 || wave == 0;
    v16f acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float colsum = 0.f;

    float4 dreg[LOADS], xreg[LOADS];
    int64_t cur = -1, nxt = blockIdx.x;
    while (true) {
        if (cur >= 0) {
            __syncthreads();
#pragma unroll
            for (int k = 0; k < LOADS; ++k) {
                const int idx = tid + kBlockThreads * k;
                *reinterpret_cast<float4*>(&dtile[idx / V4_PER_ROW][(idx % V4_PER_ROW) * 4]) = dreg[k];
                *reinterpret_cast<float4*>(&xtile[idx / V4_PER_ROW][(idx % V4_PER_ROW) * 4]) = xreg[k];
            }
            __syncthreads();
        }
        const bool have_next = nxt < n_tiles;
        if (have_next) {
            const int64_t r_base = r_begin + nxt * TE;
#pragma unroll
            for (int k = 0; k < LOADS; ++k) {
                const int idx = tid + kBlockThreads * k;
                const int c4 = idx % V4_PER_ROW, r = idx / V4_PER_ROW;
                const int64_t v = r_base + r;
                const bool live = v < r_end;
                dreg[k] = live ? *reinterpret_cast<const float4*>(dout + v * ld_dout + js * SW + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
                xreg[k] = live ? *reinterpret_cast<const float4*>(x + v * ld_x + cs * SW + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        if (cur >= 0) {
            if (active) {
#pragma unroll 4
                for (int kk = 0; kk < TE / 2; ++kk) {
                    const int e = 2 * kk + half;
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(dtile[e][jt * 32 + l31], xtile[e][ct * 32 + l31], acc, 0, 0, 0);
                }
            }
            if (cs == 0 && tid < SW) {
                float part = 0.f;
#pragma unroll 8
                for (int r = 0; r < TE; ++r) part += dtile[r][tid];
                colsum += part;
            }
        }
        if (!have_next) break;
        cur = nxt;
        nxt += gridDim.x;
    }
    const int n_slabs = gridDim.x;
    float* slab = slabs + (static_cast<int64_t>(type) * n_slabs + blockIdx.x) * d * d;
    if (active) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
            slab[static_cast<int64_t>(js * SW + jt * 32 + acc_row(r, lane)) * d + cs * SW + ct * 32 + l31] = acc[r];
    }
    if (cs == 0 && tid < SW) bias_slabs[(static_cast<int64_t>(type) * n_slabs + blockIdx.x) * d + js * SW + tid] = colsum;
}

__global__ __launch_bounds__(kBlockThreads) void dense_slab_reduce_kernel(const float* __restrict__ slabs, const float* __restrict__ bias_slabs,
                                                                          int n_slabs, int n_types, int d, float* __restrict__ dw, int64_t ld_dw,
                                                                          int64_t dw_type_stride, float* __restrict__ dbias, int bias_mask) {
    const int64_t per_type = static_cast<int64_t>(d) * d;
    const int64_t w_items = per_type * n_types;
    const int64_t total = w_items + d;
    for (int64_t base = static_cast<int64_t>(blockIdx.x) * kWave; base < total; base += static_cast<int64_t>(gridDim.x) * kWave) {
        const int64_t idx = base + (threadIdx.x & 63);
        const bool first_wave = (threadIdx.x >> 6) == 0;
        if (base < w_items) {                               // w_items is a multiple of 64: a block never straddles the two parts
            const int type = static_cast<int>(idx / per_type);
            const int64_t rem = idx - type * per_type;
            const float acc = slab_sum(slabs + static_cast<int64_t>(type) * n_slabs * per_type, n_slabs, per_type, rem, idx < w_items);
            if (first_wave && idx < w_items) {
                const int c = static_cast<int>(rem / d), j = static_cast<int>(rem - static_cast<int64_t>(c) * d);
                dw[static_cast<int64_t>(c) * ld_dw + type * dw_type_stride + j] = acc;
            }
        } else {
            const int c = static_cast<int>(idx - w_items);
            float acc = 0.f;
            for (int type = 0; type < n_types; ++type) {
                const bool use = n_types == 1 || ((bias_mask >> type) & 1);
                const float part = slab_sum(bias_slabs + static_cast<int64_t>(type) * n_slabs * d, n_slabs, d, c, use && c < d);
                acc += part;
            }
            if (first_wave && c < d && dbias != nullptr) dbias[c] = acc;
        }
    }
}

inline TypePlan make_plan(const int64_t* type_begin, int tile_rows) {
    TypePlan plan;
    int acc = 0;
    for (int t = 0; t < 4; ++t) plan.begin[t] = type_begin[t];
    for (int t = 0; t < 3; ++t) {
        plan.tile_prefix[t] = acc;
        acc += static_cast<int>((type_begin[t + 1] - type_begin[t] + tile_rows - 1) / tile_rows);
    }
    plan.tile_prefix[3] = acc;
    return plan;
}
constexpr int kDenseSlabs = 256;

int launch_row_gemm(int dim, const float* in, int64_t ld_in, const float* w, int64_t ld_w, int64_t w_type_stride, int transpose,
                    const float* bias, int bias_mask, const int64_t* type_begin, float* out, int64_t ld_out, float* pk, hipStream_t s) {
    const int n_types = w_type_stride == 0 ? 1 : 3;
    const int pack_items = n_types * (dim / 32) * (dim / 8) * kWave;
    hipLaunchKernelGGL(pack_dense_kernel, dim3((pack_items + kBlockThreads - 1) / kBlockThreads), dim3(kBlockThreads), 0, s, w, ld_w, w_type_stride,
                       n_types, dim, transpose, pk);
    const int64_t pk_type_stride = n_types == 1 ? 0 : static_cast<int64_t>(dim) * dim;
    const TypePlan plan = make_plan(type_begin, dim == 32 ? 128 : 64);
    if (plan.tile_prefix[3] == 0) return IHG_OK;
    const int grid = std::min(plan.tile_prefix[3], 256 * 4);
#define IHG_RG(D) hipLaunchKernelGGL((row_gemm_kernel<D>), dim3(grid), dim3(kBlockThreads), 0, s, in, ld_in, pk, pk_type_stride, bias, bias_mask, plan, out, ld_out)
    switch (dim) {
        case 32: IHG_RG(32); break;
        case 64: IHG_RG(64); break;
        case 128: IHG_RG(128); break;
        default: IHG_RG(256); break;
    }
#undef IHG_RG
    return IHG_OK;
}

inline bool mfma_dim(int dim) { return dim == 32 || dim == 64 || dim == 128 || dim == 256; }
inline int64_t packed_weight_floats(int dim, int order) { return static_cast<int64_t>(order == 3 ? 4 : 3) * dim * dim; }
inline int weight_slabs(int dim) {
    const int subs = dim >= 64 ? (dim / 64) * (dim / 64) : 1;
    int n = 512 / subs;
    return n < 8 ? 8 : n;
}
constexpr int kFwdGrid = 256 * 3;
constexpr int kPipeGrid = 256;          // wave-specialised kernels: one 512-thread workgroup per CU


// ================================================================================================
// Batch tail (HEM scoring head over the concatenated layer outputs, Models/PredictionLayers.py:21-44 after
// Models/RawGnn.py:122-131): one wave per batch row, lanes over the columns of every layer output; no [N, D] concat.
//   fwd  score[r] = sum_l sum_c X_l[item_r][c] * (lam * X_l[query_r][c] + (1 - lam) * X_l[user_r][c]) + bias[item_r]
//   bwd  rowgrad[0B + r] = ds (1 - lam) X[item_r]   (w.r.t. the user row),   rowgrad[1B + r] = ds lam X[item_r]   (query row),
//        rowgrad[2B + r] = ds (lam X[query_r] + (1 - lam) X[user_r])   (item row); [3B, L1*d] dense and conflict-free -
//        the caller adds duplicate rows with one deterministic scatter.
// ================================================================================================
struct LayerPtrs {
    const float* x[8];
};

__global__ __launch_bounds__(kBlockThreads) void hem_score_fwd_kernel(LayerPtrs layers, int n_layers, int64_t ld, int dim,
                                                                      const int64_t* __restrict__ rows, const int64_t* __restrict__ items,
                                                                      const float* __restrict__ bias, float lam, float* __restrict__ scores,
                                                                      int64_t batch) {
    const int lane = threadIdx.x & 63;
    for (int64_t r = global_wave_id(); r < batch; r += global_wave_count()) {
        const int64_t u = rows[r], q = rows[batch + r], it = rows[2 * batch + r];
        float acc = 0.f;
        for (int l = 0; l < n_layers; ++l) {
            const float* x = layers.x[l];
            for (int c = lane; c < dim; c += kWave) {
                const float m = lam * x[q * ld + c] + (1.f - lam) * x[u * ld + c];
                acc += x[it * ld + c] * m;
            }
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o);
        if (lane == 0) scores[r] = acc + bias[items[r]];
    }
}

__global__ __launch_bounds__(kBlockThreads) void hem_score_bwd_kernel(LayerPtrs layers, int n_layers, int64_t ld, int dim,
                                                                      const int64_t* __restrict__ rows, const float* __restrict__ dscores,
                                                                      float grad_scale, float lam, float* __restrict__ rowgrad, int64_t width,
                                                                      int64_t batch) {
    const int lane = threadIdx.x & 63;
    for (int64_t r = global_wave_id(); r < batch; r += global_wave_count()) {
        const int64_t u = rows[r], q = rows[batch + r], it = rows[2 * batch + r];
        const float ds = dscores[r] * grad_scale;
        if (lane == 0 && width > static_cast<int64_t>(n_layers) * dim) {       // optional extra column: d bias, carried by the item row
            const int64_t col = static_cast<int64_t>(n_layers) * dim;
            rowgrad[r * width + col] = 0.f;
            rowgrad[(batch + r) * width + col] = 0.f;
            rowgrad[(2 * batch + r) * width + col] = ds;
        }
        for (int l = 0; l < n_layers; ++l) {
            const float* x = layers.x[l];
            for (int c = lane; c < dim; c += kWave) {
                const float xu = x[u * ld + c], xq = x[q * ld + c], xi = x[it * ld + c];
                const int64_t col = static_cast<int64_t>(l) * dim + c;
                rowgrad[r * width + col] = ds * (1.f - lam) * xi;
                rowgrad[(batch + r) * width + col] = ds * lam * xi;
                rowgrad[(2 * batch + r) * width + col] = ds * (lam * xq + (1.f - lam) * xu);
            }
        }
    }
}



constexpr int kScatterThreads = 1024;

// Mean binary cross-entropy with logits over a batch and its gradient, one workgroup, fixed reduction tree
// (nn.BCEWithLogitsLoss(), Main.py:191): loss = mean(max(s,0) - s*y + log1p(exp(-|s|))), dscores = (sigmoid(s) - y) / n.
__global__ __launch_bounds__(kScatterThreads) void bce_with_logits_kernel(const float* __restrict__ scores, const float* __restrict__ labels, int n,
                                                                          float* __restrict__ loss, float* __restrict__ dscores) {
    __shared__ float part[kScatterThreads];
    float acc = 0.f;
    const float inv_n = 1.f / static_cast<float>(n);
    for (int k = threadIdx.x; k < n; k += kScatterThreads) {
        const float sc = scores[k], y = labels[k];
        const float e = expf(-fabsf(sc));
        acc += fmaxf(sc, 0.f) - sc * y + log1pf(e);
        const float sig = sc >= 0.f ? 1.f / (1.f + e) : e / (1.f + e);
        dscores[k] = (sig - y) * inv_n;
    }
    part[threadIdx.x] = acc;
    __syncthreads();
    for (int off = kScatterThreads / 2; off > 0; off >>= 1) {
        if (static_cast<int>(threadIdx.x) < off) part[threadIdx.x] += part[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) *loss = part[0] * inv_n;
}

// Deterministic scatter-add of a small batch of rows into a large dense matrix: dense[rows[k], :] += rowgrad[k, :], duplicates
// summed in batch order, no atomics, no sort.  One wave per batch position k: it scans rows[0..k) for an earlier occurrence of
// its destination (ballot over 64 ids at a time; the id array is a few KB and cache-resident) and retires if there is one;
// otherwise it is the leader of that destination, walks rows[k..n) in order and adds every matching batch row, then writes the
// destination row once.  O(n^2 / 64) wave-steps in total - microseconds for the few thousand rows of a training batch; replaces
// index_put_(accumulate=True) (bounds checks, device radix sort, scatter kernel).
constexpr int kScatterMax = 16384;

// Destination addressing: column c of batch row k goes to dense[(c / block_width) * block_stride + row * ld_dense + c % block_width]
// (block_width = width, block_stride = 0 is a plain matrix; block_width = d, block_stride = N*d lands layer l's columns in its own
// contiguous [N, d] matrix); an optional last column goes to tail[row - tail_row_offset] (the bias gradient).
__global__ __launch_bounds__(kBlockThreads) void batch_scatter_kernel(const float* __restrict__ rowgrad, int64_t ld_rowgrad, int width,
                                                                      const int64_t* __restrict__ rows, int n, float* __restrict__ dense,
                                                                      int64_t ld_dense, int block_width, int64_t block_stride,
                                                                      float* __restrict__ tail, int64_t tail_row_offset, int64_t tail_rows) {
    __shared__ int32_t key[kScatterMax];                    // every workgroup keeps the whole id list in LDS (<= 64 KiB)
    for (int k = threadIdx.x; k < n; k += kBlockThreads) key[k] = static_cast<int32_t>(rows[k]);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    for (int64_t k = global_wave_id(); k < n; k += global_wave_count()) {
        const int32_t mine = key[k];
        bool follower = false;
        for (int base = 0; base < k; base += kWave) {
            const int j = base + lane;
            if (__ballot(j < k && key[j] == mine) != 0ull) { follower = true; break; }
        }
        if (follower) continue;
        for (int c0 = 0; c0 < width; c0 += 4 * kWave) {
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
            for (int base = static_cast<int>(k) & ~(kWave - 1); base < n; base += kWave) {
                const int j = base + lane;
                unsigned long long mask = __ballot(j >= k && j < n && key[j] == mine);
                while (mask != 0ull) {
                    // up to 16 members per trip: all their loads are issued before the first add (a hot destination - a
                    // popular query - can own hundreds of batch rows); absent slots add an exact 0
                    constexpr int MEMBERS = 16;
                    float v[MEMBERS][4];
#pragma unroll
                    for (int u = 0; u < MEMBERS; ++u) {
                        const bool have = mask != 0ull;
                        const int bit = have ? __ffsll(static_cast<long long>(mask)) - 1 : 0;
                        if (have) mask &= mask - 1;
                        const float* src = rowgrad + static_cast<int64_t>(base + bit) * ld_rowgrad;
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int c = c0 + q * kWave + lane;
                            v[u][q] = (have && c < width) ? src[c] : 0.f;
                        }
                    }
#pragma unroll
                    for (int u = 0; u < MEMBERS; ++u)
#pragma unroll
                        for (int q = 0; q < 4; ++q) acc[q] += v[u][q];
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c = c0 + q * kWave + lane;
                if (c >= width) continue;
                if (tail != nullptr && c == width - 1) {
                    const int64_t tr = static_cast<int64_t>(mine) - tail_row_offset;
                    if (tr >= 0 && tr < tail_rows) tail[tr] += acc[q];
                } else {
                    dense[(c / block_width) * block_stride + static_cast<int64_t>(mine) * ld_dense + c % block_width] += acc[q];
                }
            }
        }
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

struct HeavyPlan {
    const int32_t* seg_begin;
    const int32_t* seg_end;
    int64_t n_segments;
    const int32_t* heavy_rows;
    const int32_t* heavy_segptr;
    int64_t n_heavy;
    float* partials;
};

template <int VEC, int G>
void launch_segment_sum_g(const float* src, int64_t ld_src, const int32_t* rowptr, const int32_t* ids, const int32_t* row_order,
                          const float* src_scale, const float* entry_scale, const float* out_scale, int mode, float* out, int64_t ld_out, int64_t n_rows, int dim,
                          int heavy_threshold, const HeavyPlan& hp, const float* self_weight, hipStream_t stream) {
    constexpr int GPW = kWave / G;
    const int dim_vec = dim / VEC;
    const int grid = grid_for_waves((n_rows + hp.n_segments + GPW - 1) / GPW);
    hipLaunchKernelGGL((node_segment_sum_kernel<VEC, G>), dim3(grid), dim3(kBlockThreads), 0, stream, src, ld_src, rowptr, ids, row_order,
                       src_scale, entry_scale, out_scale, mode, out, ld_out, n_rows, dim, dim_vec, heavy_threshold, hp.seg_begin, hp.seg_end,
                       hp.n_segments, hp.partials, self_weight);
    if (hp.n_heavy > 0)
        hipLaunchKernelGGL((heavy_finish_kernel<VEC, G>), dim3(static_cast<int>(std::min<int64_t>(hp.n_heavy, kMaxBlocks * 4))),
                           dim3(kBlockThreads), 0, stream, hp.partials, hp.heavy_rows, hp.heavy_segptr, hp.n_heavy, out_scale, mode, out,
                           ld_out, dim, dim_vec, src, ld_src, src_scale, self_weight);
}

template <int VEC>
int launch_segment_sum(const float* src, int64_t ld_src, const int32_t* rowptr, const int32_t* ids, const int32_t* row_order,
                       const float* src_scale, const float* entry_scale, const float* out_scale, int mode, float* out, int64_t ld_out, int64_t n_rows, int dim,
                       int heavy_threshold, const HeavyPlan& hp, const float* self_weight, hipStream_t stream) {
#define IHG_K7(G) launch_segment_sum_g<VEC, G>(src, ld_src, rowptr, ids, row_order, src_scale, entry_scale, out_scale, mode, out, ld_out, n_rows, dim, heavy_threshold, hp, self_weight, stream)
    switch (group_lanes(dim / VEC)) {
        case 4: IHG_K7(4); break;
        case 8: IHG_K7(8); break;
        case 16: IHG_K7(16); break;
        case 32: IHG_K7(32); break;
        default: IHG_K7(64); break;
    }
#undef IHG_K7
    return check_launch("ihg_node_segment_sum");
}

inline bool scale_mode_ok(int mode, const float* scale) {
    if (mode == IHG_SCALE_NONE) return true;
    return (mode == IHG_SCALE_MULTIPLY || mode == IHG_SCALE_DIVIDE) && scale != nullptr;
}

template <int NBLK>
void launch_interact_fwd_mfma(int dim, const float* h, int64_t ld_h, const float* p, int64_t ld_p, const int32_t* i3, const float* wp,
                              float* out, int64_t ld_out, int64_t n_edges, hipStream_t s) {
#define IHG_FWD(D)                                                                                                          \
    {                                                                                                                       \
        const int64_t tiles = (n_edges + TileShape<D>::TE - 1) / TileShape<D>::TE;                                          \
        const int grid = static_cast<int>(std::min<int64_t>(tiles, kFwdGrid));                                              \
        hipLaunchKernelGGL((interact_fwd_mfma_kernel<D, NBLK>), dim3(grid), dim3(kBlockThreads), 0, s, h, ld_h, p, ld_p, i3, wp, out, ld_out, n_edges); \
    }
#define IHG_FWD_PIPE(D)                                                                                                     \
    {                                                                                                                       \
        const int64_t tiles = (n_edges + TileShape<D>::TE - 1) / TileShape<D>::TE;                                          \
        const int grid = static_cast<int>(std::min<int64_t>(tiles, kPipeGrid));                                             \
        hipLaunchKernelGGL((interact_fwd_ws_kernel<D, NBLK>), dim3(grid), dim3(kWsThreads), 0, s, h, ld_h, p, ld_p, i3, wp, out, ld_out, n_edges); \
    }
    switch (dim) {
        case 32: IHG_FWD_PIPE(32) break;
        case 64: IHG_FWD_PIPE(64) break;
        case 128: IHG_FWD(128) break;
        default: IHG_FWD(256) break;
    }
#undef IHG_FWD
#undef IHG_FWD_PIPE
}

template <int NBLK>
void launch_interact_bwd_mfma(int dim, const float* h, int64_t ld_h, const int32_t* i3, const float* wq, const float* dout, int64_t ld_dout,
                              float* g, float* slabs, float* dw, int64_t ld_dw, int64_t n_edges, hipStream_t s) {
#define IHG_MEM(D)                                                                                                          \
    {                                                                                                                       \
        constexpr int TE = D == 32 ? 128 : 64;                                                                              \
        const int grid = static_cast<int>(std::min<int64_t>((n_edges + TE - 1) / TE, kFwdGrid));                            \
        hipLaunchKernelGGL((interact_bwd_members_mfma_kernel<D, NBLK>), dim3(grid), dim3(kBlockThreads), 0, s, h, ld_h, i3, wq, dout, ld_dout, g, n_edges); \
    }
#define IHG_MEM_PIPE(D)                                                                                                     \
    {                                                                                                                       \
        constexpr int TE = D == 32 ? 128 : 64;                                                                              \
        const int grid = static_cast<int>(std::min<int64_t>((n_edges + TE - 1) / TE, kPipeGrid));                           \
        hipLaunchKernelGGL((interact_bwd_members_ws_kernel<D, NBLK>), dim3(grid), dim3(kWsThreads), 0, s, h, ld_h, i3, wq, dout, ld_dout, g, n_edges); \
    }
    switch (dim) {
        case 32: IHG_MEM_PIPE(32) break;
        case 64: IHG_MEM_PIPE(64) break;
        case 128: {
            const int grid = static_cast<int>(std::min<int64_t>((n_edges + 63) / 64, kPipeGrid));
            hipLaunchKernelGGL((interact_bwd_members_wsbig_kernel<128, NBLK>), dim3(grid), dim3(kWsThreads), 0, s, h, ld_h, i3, wq, dout, ld_dout, g, n_edges);
        } break;
        default: IHG_MEM(256) break;
    }
#undef IHG_MEM
#undef IHG_MEM_PIPE
    const int subs_ws = (dim / 64) * (dim / 64);
    const int n_slabs = static_cast<int>(std::min<int64_t>(dim >= 64 ? std::max(kPipeGrid / subs_ws, 8) : weight_slabs(dim), (n_edges + 63) / 64));
    if (dim >= 64) {
        hipLaunchKernelGGL((interact_bwd_weight_ws_kernel<NBLK>), dim3(n_slabs, subs_ws), dim3(kWsThreads), 0, s, h, ld_h, i3, dout, ld_dout, slabs, n_edges, dim);
    } else {
        hipLaunchKernelGGL((interact_bwd_weight_mfma_kernel<32, NBLK>), dim3(n_slabs, 1), dim3(kBlockThreads), 0, s, h, ld_h, i3, dout, ld_dout, slabs, n_edges, dim);
    }
    const int total = dim * NBLK * dim;
    hipLaunchKernelGGL(slab_reduce_kernel, dim3((total + kWave - 1) / kWave), dim3(kBlockThreads), 0, s, slabs, n_slabs, dim, NBLK, dw, ld_dw);
}

// Parses the space-separated integers of one CSV field into `out`; returns false on a malformed token.
bool parse_int_list(const char* p, const char* end, std::vector<int64_t>& out) {
    out.clear();
    while (p < end) {
        while (p < end && (*p == ' ' || *p == '\t' || *p == '\r')) ++p;
        if (p >= end) break;
        bool neg = false;
        if (*p == '-' || *p == '+') { neg = *p == '-'; ++p; }
        if (p >= end || *p < '0' || *p > '9') return false;
        int64_t v = 0;
        while (p < end && *p >= '0' && *p <= '9') v = v * 10 + (*p++ - '0');
        out.push_back(neg ? -v : v);
    }
    return true;
}

}  // namespace

// =================================================================================================
// C ABI
// =================================================================================================
extern "C" {

int32_t ihg_abi_version(void) { return 12; }

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


int ihg_build_pair_csr(const int64_t* triples, int64_t n_edges, int64_t n_users, int64_t n_queries, int64_t n_items,
                       int32_t completeness, int32_t self_loops, int32_t* rowptr, int32_t* cols, float* vals, float* degree,
                       int64_t capacity, int64_t* nnz_out) {
    if (n_edges < 0 || n_users < 0 || n_queries < 0 || n_items < 0 || completeness < 0 || completeness > 3)
        return fail(IHG_ERR_INVALID, "ihg_build_pair_csr: bad argument");
    const int64_t n_nodes = n_users + n_queries + n_items;
    if (rowptr == nullptr || degree == nullptr || nnz_out == nullptr || (n_edges > 0 && (triples == nullptr || cols == nullptr || vals == nullptr)))
        return fail(IHG_ERR_INVALID, "ihg_build_pair_csr: null buffer");
    if (n_nodes >= INT32_MAX) return fail(IHG_ERR_INVALID, "ihg_build_pair_csr: graph exceeds int32 indexing");
    // member pairs joined by one interaction (Helpers/Graph.py:40-63): uqi = all three pairs, otherwise a single pair
    static const int kPairs[4][3][2] = {{{0, 1}, {1, 2}, {2, 0}}, {{0, 1}, {0, 1}, {0, 1}}, {{0, 2}, {0, 2}, {0, 2}}, {{1, 2}, {1, 2}, {1, 2}}};
    const int n_pairs = completeness == 0 ? 3 : 1;
    const int64_t offset[3] = {0, n_users, n_users + n_queries};
    const int64_t limit[3] = {n_users, n_queries, n_items};
    std::vector<uint64_t> keys;
    keys.reserve(static_cast<size_t>(n_edges) * n_pairs * 2 + (self_loops ? n_nodes : 0));
    std::vector<float> deg(static_cast<size_t>(n_nodes), self_loops ? 1.f : 0.f);
    if (self_loops)
        for (int64_t v = 0; v < n_nodes; ++v) keys.push_back((static_cast<uint64_t>(v) << 32) | static_cast<uint64_t>(v));
    for (int64_t e = 0; e < n_edges; ++e) {
        int64_t node[3];
        for (int m = 0; m < 3; ++m) {
            const int64_t local = triples[e * 3 + m];
            if (local < 0 || local >= limit[m]) return fail(IHG_ERR_INVALID, "ihg_build_pair_csr: interaction %lld member %d out of range", (long long)e, m);
            node[m] = local + offset[m];
        }
        for (int k = 0; k < n_pairs; ++k) {
            const uint64_t a = static_cast<uint64_t>(node[kPairs[completeness][k][0]]), b = static_cast<uint64_t>(node[kPairs[completeness][k][1]]);
            keys.push_back((a << 32) | b);
            keys.push_back((b << 32) | a);
            deg[a] += 1.f;                                   // Graph.py:48,54,60,66: +2 per member for uqi, +1 per pair member otherwise
            deg[b] += 1.f;
        }
    }
    std::sort(keys.begin(), keys.end());
    std::memset(rowptr, 0, sizeof(int32_t) * static_cast<size_t>(n_nodes + 1));
    int64_t nnz = 0;
    for (size_t k = 0; k < keys.size();) {
        size_t j = k;
        while (j < keys.size() && keys[j] == keys[k]) ++j;   // duplicates are summed by coalesce() (Graph.py:73-79)
        if (nnz >= capacity) return fail(IHG_ERR_WORKSPACE, "ihg_build_pair_csr: capacity %lld too small", (long long)capacity);
        cols[nnz] = static_cast<int32_t>(keys[k] & 0xffffffffu);
        vals[nnz] = static_cast<float>(j - k);
        ++rowptr[(keys[k] >> 32) + 1];
        ++nnz;
        k = j;
    }
    for (int64_t v = 0; v < n_nodes; ++v) {
        rowptr[v + 1] += rowptr[v];
        degree[v] = (!self_loops && deg[v] == 0.f) ? 1e-8f : deg[v];
    }
    *nnz_out = nnz;
    return IHG_OK;
}


// ------------------------------------------------------------------------------------------------
// HOST: search-log CSV ingestion.
// ------------------------------------------------------------------------------------------------

int ihg_parse_search_logs(const char* path, int64_t* n_logs, int64_t* n_pos, int64_t* n_neg, int64_t* pos, int64_t pos_capacity,
                          int64_t* neg, int64_t neg_capacity) {
    if (path == nullptr || n_logs == nullptr || n_pos == nullptr || n_neg == nullptr) return fail(IHG_ERR_INVALID, "ihg_parse_search_logs: null argument");
    FILE* f = std::fopen(path, "rb");
    if (f == nullptr) return fail(IHG_ERR_INVALID, "ihg_parse_search_logs: cannot open %s", path);
    std::fseek(f, 0, SEEK_END);
    const long size = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    std::vector<char> text(static_cast<size_t>(size > 0 ? size : 0) + 1);
    const size_t got = size > 0 ? std::fread(text.data(), 1, static_cast<size_t>(size), f) : 0;
    std::fclose(f);
    text[got] = '\n';
    const char* p = text.data();
    const char* const end = p + got + 1;
    while (p < end && *p != '\n') ++p;                      // header line (SearchLogCollection.py:28)
    ++p;
    int64_t logs = 0, pcount = 0, ncount = 0, line_no = 1;
    std::vector<int64_t> items, flags, scalar;
    while (p < end) {
        const char* eol = p;
        while (eol < end && *eol != '\n') ++eol;
        ++line_no;
        const char* q = p;
        bool blank = true;
        for (const char* c = p; c < eol; ++c)
            if (*c != ' ' && *c != '\r' && *c != '\t') { blank = false; break; }
        if (!blank) {
            const char* field[9];
            int n_fields = 0;
            field[n_fields++] = q;
            for (const char* c = q; c < eol && n_fields < 9; ++c)
                if (*c == ',') field[n_fields++] = c + 1;
            int commas = 0;
            for (const char* c = q; c < eol; ++c) commas += *c == ',';
            if (commas != 7) return fail(IHG_ERR_INVALID, "ihg_parse_search_logs: %s line %lld has %d columns, expected 8", path, (long long)line_no, commas + 1);
            field[8] = eol + 1;
            auto fend = [&](int k) { return field[k + 1] - 1; };
            if (!parse_int_list(field[0], fend(0), scalar) || scalar.size() != 1) return fail(IHG_ERR_INVALID, "ihg_parse_search_logs: %s line %lld: bad user id", path, (long long)line_no);
            const int64_t user = scalar[0];
            if (!parse_int_list(field[1], fend(1), scalar) || scalar.size() != 1) return fail(IHG_ERR_INVALID, "ihg_parse_search_logs: %s line %lld: bad query id", path, (long long)line_no);
            const int64_t query = scalar[0];
            if (!parse_int_list(field[3], fend(3), items) || !parse_int_list(field[6], fend(6), flags))
                return fail(IHG_ERR_INVALID, "ihg_parse_search_logs: %s line %lld: bad item / interaction list", path, (long long)line_no);
            const size_t n = items.size() < flags.size() ? items.size() : flags.size();
            for (size_t k = 0; k < n; ++k) {
                if (flags[k] > 0) {
                    if (pos != nullptr) {
                        if (pcount >= pos_capacity) return fail(IHG_ERR_WORKSPACE, "ihg_parse_search_logs: positive buffer too small");
                        pos[pcount * 3] = user; pos[pcount * 3 + 1] = query; pos[pcount * 3 + 2] = items[k];
                    }
                    ++pcount;
                } else {
                    if (neg != nullptr) {
                        if (ncount >= neg_capacity) return fail(IHG_ERR_WORKSPACE, "ihg_parse_search_logs: negative buffer too small");
                        neg[ncount * 3] = user; neg[ncount * 3 + 1] = query; neg[ncount * 3 + 2] = items[k];
                    }
                    ++ncount;
                }
            }
            ++logs;
        }
        p = eol + 1;
    }
    *n_logs = logs;
    *n_pos = pcount;
    *n_neg = ncount;
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

int ihg_node_segment_sum(const float* src, int64_t ld_src, const int32_t* rowptr, const int32_t* ids, const int32_t* row_order,
                         const float* src_scale, const float* entry_scale, const float* out_scale, int32_t out_scale_mode, float* out, int64_t ld_out,
                         int64_t n_rows, int32_t dim, int32_t heavy_threshold, const int32_t* seg_begin, const int32_t* seg_end,
                         int64_t n_segments, const int32_t* heavy_rows, const int32_t* heavy_segptr, int64_t n_heavy, float* partials,
                         const float* self_weight, ihg_stream_t stream) {
    if (n_rows < 0 || dim <= 0 || ld_src < dim || ld_out < dim || n_segments < 0 || n_heavy < 0) return fail(IHG_ERR_INVALID, "ihg_node_segment_sum: bad size");
    if (!scale_mode_ok(out_scale_mode, out_scale)) return fail(IHG_ERR_INVALID, "ihg_node_segment_sum: bad out_scale_mode %d", out_scale_mode);
    if (n_rows == 0) return IHG_OK;
    if (src == nullptr || rowptr == nullptr || ids == nullptr || out == nullptr) return fail(IHG_ERR_INVALID, "ihg_node_segment_sum: null pointer");
    if (n_heavy > 0 && (heavy_threshold <= 0 || seg_begin == nullptr || seg_end == nullptr || heavy_rows == nullptr || heavy_segptr == nullptr || partials == nullptr))
        return fail(IHG_ERR_INVALID, "ihg_node_segment_sum: incomplete split-row plan");
    if (n_heavy == 0) {
        n_segments = 0;
        heavy_threshold = 0;
    }
    const HeavyPlan hp{seg_begin, seg_end, n_segments, heavy_rows, heavy_segptr, n_heavy, partials};
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool wide = dim % 4 == 0 && ld_src % 4 == 0 && ld_out % 4 == 0 && aligned16(src) && aligned16(out) && (n_heavy == 0 || aligned16(partials));
    return wide ? launch_segment_sum<4>(src, ld_src, rowptr, ids, row_order, src_scale, entry_scale, out_scale, out_scale_mode, out, ld_out, n_rows, dim, heavy_threshold, hp, self_weight, s)
                : launch_segment_sum<1>(src, ld_src, rowptr, ids, row_order, src_scale, entry_scale, out_scale, out_scale_mode, out, ld_out, n_rows, dim, heavy_threshold, hp, self_weight, s);
}

int ihg_bag_mean_fwd(const float* table, int64_t ld_table, const int32_t* bag_ptr, const int32_t* words, const float* bag_len,
                     float* out, int64_t ld_out, int64_t n_bags, int32_t dim, ihg_stream_t stream) {
    if (bag_len == nullptr && n_bags > 0) return fail(IHG_ERR_INVALID, "ihg_bag_mean_fwd: null bag_len");
    return ihg_node_segment_sum(table, ld_table, bag_ptr, words, nullptr, nullptr, nullptr, bag_len, IHG_SCALE_DIVIDE, out, ld_out, n_bags, dim, 0, nullptr, nullptr, 0, nullptr, nullptr, 0, nullptr, nullptr, stream);
}

int ihg_bag_mean_bwd(const float* dout, int64_t ld_dout, const int32_t* word_ptr, const int32_t* word_bags, const float* inv_len,
                     float* dtable, int64_t ld_dtable, int64_t n_table_rows, int32_t dim, ihg_stream_t stream) {
    if (inv_len == nullptr && n_table_rows > 0) return fail(IHG_ERR_INVALID, "ihg_bag_mean_bwd: null inv_len");
    return ihg_node_segment_sum(dout, ld_dout, word_ptr, word_bags, nullptr, inv_len, nullptr, nullptr, IHG_SCALE_NONE, dtable, ld_dtable, n_table_rows, dim, 0, nullptr, nullptr, 0, nullptr, nullptr, 0, nullptr, nullptr, stream);
}

int64_t ihg_interact_fwd_workspace_bytes(int64_t n_edges, int32_t dim, int32_t order) {
    (void)n_edges;
    if (!mfma_dim(dim) || (order != 2 && order != 3)) return 0;
    return packed_weight_floats(dim, order) * static_cast<int64_t>(sizeof(float));
}

int ihg_interact_fwd(const float* h, int64_t ld_h, const float* p, int64_t ld_p, const int32_t* i3, const float* w, int64_t ld_w,
                     int32_t order, float* out, int64_t ld_out, void* workspace, int64_t workspace_bytes, int64_t n_edges,
                     int32_t dim, ihg_stream_t stream) {
    if (order != 2 && order != 3) return fail(IHG_ERR_INVALID, "ihg_interact_fwd: order must be 2 or 3, got %d", order);
    const int k = order == 3 ? 7 : 6;
    if (n_edges < 0 || dim <= 0 || ld_h < dim || ld_out < dim || ld_w < static_cast<int64_t>(k) * dim || (p != nullptr && ld_p < dim))
        return fail(IHG_ERR_INVALID, "ihg_interact_fwd: bad size");
    if (n_edges == 0) return IHG_OK;
    if (h == nullptr || i3 == nullptr || w == nullptr || out == nullptr) return fail(IHG_ERR_INVALID, "ihg_interact_fwd: null pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool tiled = mfma_dim(dim) && p != nullptr && ld_h % 4 == 0 && ld_w % 4 == 0 && aligned16(h) && aligned16(w) && workspace != nullptr &&
                       aligned16(workspace);
    if (tiled) {
        if (workspace_bytes < ihg_interact_fwd_workspace_bytes(n_edges, dim, order)) return fail(IHG_ERR_WORKSPACE, "ihg_interact_fwd: workspace too small");
        float* wp = static_cast<float*>(workspace);
        const int nblk = order == 3 ? 4 : 3;
        const int pack_items = (dim / 32) * nblk * (dim / 8) * kWave;
        hipLaunchKernelGGL(pack_weights_kernel, dim3((pack_items + kBlockThreads - 1) / kBlockThreads), dim3(kBlockThreads), 0, s, w, ld_w, dim, nblk, wp,
                           static_cast<float*>(nullptr));
        if (nblk == 4) launch_interact_fwd_mfma<4>(dim, h, ld_h, p, ld_p, i3, wp, out, ld_out, n_edges, s);
        else launch_interact_fwd_mfma<3>(dim, h, ld_h, p, ld_p, i3, wp, out, ld_out, n_edges, s);
        return check_launch("ihg_interact_fwd");
    }
    const int64_t total = n_edges * dim;
    const int grid = static_cast<int>(std::min<int64_t>((total + kBlockThreads - 1) / kBlockThreads, kMaxBlocks * 4));
    hipLaunchKernelGGL(interact_fwd_generic_kernel, dim3(grid), dim3(kBlockThreads), 0, s, h, ld_h, p, ld_p, i3, w, ld_w, order, out, ld_out, n_edges, dim);
    return check_launch("ihg_interact_fwd");
}

int64_t ihg_interact_bwd_workspace_bytes(int64_t n_edges, int32_t dim, int32_t order) {
    (void)n_edges;
    if (!mfma_dim(dim) || (order != 2 && order != 3)) return 0;
    const int64_t w_floats = packed_weight_floats(dim, order);
    return (w_floats + static_cast<int64_t>(weight_slabs(dim)) * w_floats) * static_cast<int64_t>(sizeof(float));
}

int ihg_interact_bwd(const float* h, int64_t ld_h, const int32_t* i3, const float* w, int64_t ld_w, int32_t order,
                     const float* dout, int64_t ld_dout, float* g, float* dw, int64_t ld_dw, void* workspace,
                     int64_t workspace_bytes, int64_t n_edges, int32_t dim, ihg_stream_t stream) {
    if (order != 2 && order != 3) return fail(IHG_ERR_INVALID, "ihg_interact_bwd: order must be 2 or 3, got %d", order);
    const int k = order == 3 ? 7 : 6;
    if (n_edges < 0 || dim <= 0 || ld_h < dim || ld_dout < dim || ld_w < static_cast<int64_t>(k) * dim || ld_dw < static_cast<int64_t>(k) * dim)
        return fail(IHG_ERR_INVALID, "ihg_interact_bwd: bad size");
    if (h == nullptr || i3 == nullptr || w == nullptr || dout == nullptr || g == nullptr || dw == nullptr)
        return fail(IHG_ERR_INVALID, "ihg_interact_bwd: null pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool tiled = mfma_dim(dim) && n_edges > 0 && ld_h % 4 == 0 && ld_w % 4 == 0 && ld_dout % 4 == 0 && aligned16(h) && aligned16(w) &&
                       aligned16(dout) && workspace != nullptr && aligned16(workspace);
    if (tiled) {
        if (workspace_bytes < ihg_interact_bwd_workspace_bytes(n_edges, dim, order)) return fail(IHG_ERR_WORKSPACE, "ihg_interact_bwd: workspace too small");
        const int nblk = order == 3 ? 4 : 3;
        float* wq = static_cast<float*>(workspace);
        float* slabs = wq + packed_weight_floats(dim, order);
        const int pack_items = (dim / 32) * nblk * (dim / 8) * kWave;
        hipLaunchKernelGGL(pack_weights_kernel, dim3((pack_items + kBlockThreads - 1) / kBlockThreads), dim3(kBlockThreads), 0, s, w, ld_w, dim, nblk,
                           static_cast<float*>(nullptr), wq);
        if (nblk == 4) launch_interact_bwd_mfma<4>(dim, h, ld_h, i3, wq, dout, ld_dout, g, slabs, dw, ld_dw, n_edges, s);
        else launch_interact_bwd_mfma<3>(dim, h, ld_h, i3, wq, dout, ld_dout, g, slabs, dw, ld_dw, n_edges, s);
        return check_launch("ihg_interact_bwd");
    }
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


int64_t ihg_node_linear_workspace_bytes(int32_t dim) {
    if (!mfma_dim(dim)) return -1;
    const int64_t packed = 3LL * dim * dim;
    const int64_t slabs = 3LL * kDenseSlabs * (static_cast<int64_t>(dim) * dim + dim);
    return (packed + slabs) * static_cast<int64_t>(sizeof(float));
}

static int node_linear_common_check(const char* what, int32_t dim, int64_t ld_a, int64_t ld_b, int64_t ld_w, const int64_t* type_begin,
                                    const void* workspace, int64_t workspace_bytes) {
    if (!mfma_dim(dim)) return fail(IHG_ERR_INVALID, "%s: dim %d is not one of 32/64/128/256", what, dim);
    if (type_begin == nullptr || workspace == nullptr) return fail(IHG_ERR_INVALID, "%s: null pointer", what);
    if (ld_a < dim || ld_b < dim || ld_w < dim || ld_a % 4 || ld_b % 4) return fail(IHG_ERR_INVALID, "%s: bad leading dimension", what);
    if (!(type_begin[0] <= type_begin[1] && type_begin[1] <= type_begin[2] && type_begin[2] <= type_begin[3])) return fail(IHG_ERR_INVALID, "%s: type ranges not ascending", what);
    if (workspace_bytes < ihg_node_linear_workspace_bytes(dim)) return fail(IHG_ERR_WORKSPACE, "%s: workspace too small", what);
    if (!aligned16(workspace)) return fail(IHG_ERR_INVALID, "%s: workspace not 16-byte aligned", what);
    return IHG_OK;
}

int ihg_node_linear_fwd(const float* x, int64_t ld_x, const float* w, int64_t ld_w, int64_t w_type_stride, const float* bias,
                        int32_t bias_type_mask, const int64_t* type_begin, float* out, int64_t ld_out, void* workspace,
                        int64_t workspace_bytes, int32_t dim, ihg_stream_t stream) {
    if (int rc = node_linear_common_check("ihg_node_linear_fwd", dim, ld_x, ld_out, ld_w, type_begin, workspace, workspace_bytes)) return rc;
    if (type_begin[3] == type_begin[0]) return IHG_OK;
    if (x == nullptr || w == nullptr || out == nullptr || !aligned16(x)) return fail(IHG_ERR_INVALID, "ihg_node_linear_fwd: null or unaligned pointer");
    launch_row_gemm(dim, x, ld_x, w, ld_w, w_type_stride, 0, bias, bias_type_mask, type_begin, out, ld_out, static_cast<float*>(workspace),
                    static_cast<hipStream_t>(stream));
    return check_launch("ihg_node_linear_fwd");
}

int ihg_node_linear_bwd_input(const float* dout, int64_t ld_dout, const float* w, int64_t ld_w, int64_t w_type_stride,
                              const int64_t* type_begin, float* dx, int64_t ld_dx, void* workspace, int64_t workspace_bytes,
                              int32_t dim, ihg_stream_t stream) {
    if (int rc = node_linear_common_check("ihg_node_linear_bwd_input", dim, ld_dout, ld_dx, ld_w, type_begin, workspace, workspace_bytes)) return rc;
    if (type_begin[3] == type_begin[0]) return IHG_OK;
    if (dout == nullptr || w == nullptr || dx == nullptr || !aligned16(dout)) return fail(IHG_ERR_INVALID, "ihg_node_linear_bwd_input: null or unaligned pointer");
    launch_row_gemm(dim, dout, ld_dout, w, ld_w, w_type_stride, 1, nullptr, 0, type_begin, dx, ld_dx, static_cast<float*>(workspace),
                    static_cast<hipStream_t>(stream));
    return check_launch("ihg_node_linear_bwd_input");
}

int ihg_node_linear_bwd_weight(const float* dout, int64_t ld_dout, const float* x, int64_t ld_x, const int64_t* type_begin,
                               float* dw, int64_t ld_dw, int64_t dw_type_stride, float* dbias, int32_t bias_type_mask,
                               void* workspace, int64_t workspace_bytes, int32_t dim, ihg_stream_t stream) {
    if (int rc = node_linear_common_check("ihg_node_linear_bwd_weight", dim, ld_dout, ld_x, ld_dw, type_begin, workspace, workspace_bytes)) return rc;
    if (dout == nullptr || x == nullptr || dw == nullptr || !aligned16(dout) || !aligned16(x)) return fail(IHG_ERR_INVALID, "ihg_node_linear_bwd_weight: null or unaligned pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int n_types = dw_type_stride == 0 ? 1 : 3;
    float* slabs = static_cast<float*>(workspace) + 3LL * dim * dim;
    float* bias_slabs = slabs + 3LL * kDenseSlabs * dim * dim;
    const TypePlan plan = make_plan(type_begin, 64);
    if (dim == 32) {
        hipLaunchKernelGGL((dense_weight_grad_kernel<32>), dim3(kDenseSlabs, 1, n_types), dim3(kBlockThreads), 0, s, dout, ld_dout, x, ld_x, plan,
                           n_types == 1 ? 1 : 0, slabs, bias_slabs, dim);
    } else {
        const int subs = (dim / 64) * (dim / 64);
        hipLaunchKernelGGL((dense_weight_grad_kernel<64>), dim3(kDenseSlabs, subs, n_types), dim3(kBlockThreads), 0, s, dout, ld_dout, x, ld_x, plan,
                           n_types == 1 ? 1 : 0, slabs, bias_slabs, dim);
    }
    const int total = dim * dim * n_types + dim;
    hipLaunchKernelGGL(dense_slab_reduce_kernel, dim3((total + kWave - 1) / kWave), dim3(kBlockThreads), 0, s, slabs, bias_slabs,
                       kDenseSlabs, n_types, dim, dw, ld_dw, dw_type_stride, dbias, bias_type_mask);
    return check_launch("ihg_node_linear_bwd_weight");
}


static int hem_common_check(const char* what, const float* const* layers, int32_t n_layers, int64_t ld, int32_t dim, const int64_t* rows, int64_t batch) {
    if (n_layers < 1 || n_layers > 8) return fail(IHG_ERR_INVALID, "%s: 1..8 layer outputs supported, got %d", what, n_layers);
    if (layers == nullptr || rows == nullptr || dim <= 0 || ld < dim || batch < 0) return fail(IHG_ERR_INVALID, "%s: bad argument", what);
    for (int l = 0; l < n_layers; ++l)
        if (layers[l] == nullptr) return fail(IHG_ERR_INVALID, "%s: null layer pointer", what);
    return IHG_OK;
}

int ihg_hem_score_fwd(const float* const* layers, int32_t n_layers, int64_t ld, int32_t dim, const int64_t* rows, const int64_t* items,
                      const float* bias, float lambda_muq, float* scores, int64_t batch, ihg_stream_t stream) {
    if (int rc = hem_common_check("ihg_hem_score_fwd", layers, n_layers, ld, dim, rows, batch)) return rc;
    if (batch == 0) return IHG_OK;
    if (items == nullptr || bias == nullptr || scores == nullptr) return fail(IHG_ERR_INVALID, "ihg_hem_score_fwd: null pointer");
    LayerPtrs lp{};
    for (int l = 0; l < n_layers; ++l) lp.x[l] = layers[l];
    hipLaunchKernelGGL(hem_score_fwd_kernel, dim3(grid_for_waves(batch)), dim3(kBlockThreads), 0, static_cast<hipStream_t>(stream), lp, n_layers, ld,
                       dim, rows, items, bias, lambda_muq, scores, batch);
    return check_launch("ihg_hem_score_fwd");
}

int ihg_hem_score_bwd(const float* const* layers, int32_t n_layers, int64_t ld, int32_t dim, const int64_t* rows, const float* dscores,
                      float grad_scale, float lambda_muq, float* rowgrad, int64_t ld_rowgrad, int64_t batch, ihg_stream_t stream) {
    if (int rc = hem_common_check("ihg_hem_score_bwd", layers, n_layers, ld, dim, rows, batch)) return rc;
    if (batch == 0) return IHG_OK;
    if (dscores == nullptr || rowgrad == nullptr || ld_rowgrad < static_cast<int64_t>(n_layers) * dim) return fail(IHG_ERR_INVALID, "ihg_hem_score_bwd: null pointer or short row stride");
    LayerPtrs lp{};
    for (int l = 0; l < n_layers; ++l) lp.x[l] = layers[l];
    hipLaunchKernelGGL(hem_score_bwd_kernel, dim3(grid_for_waves(batch)), dim3(kBlockThreads), 0, static_cast<hipStream_t>(stream), lp, n_layers, ld,
                       dim, rows, dscores, grad_scale, lambda_muq, rowgrad, ld_rowgrad, batch);
    return check_launch("ihg_hem_score_bwd");
}

int ihg_bce_with_logits(const float* scores, const float* labels, int64_t n, float* loss, float* dscores, ihg_stream_t stream) {
    if (n <= 0 || n > (1 << 24) || scores == nullptr || labels == nullptr || loss == nullptr || dscores == nullptr)
        return fail(IHG_ERR_INVALID, "ihg_bce_with_logits: bad argument");
    hipLaunchKernelGGL(bce_with_logits_kernel, dim3(1), dim3(kScatterThreads), 0, static_cast<hipStream_t>(stream), scores, labels, static_cast<int>(n), loss, dscores);
    return check_launch("ihg_bce_with_logits");
}


int64_t ihg_batch_scatter_workspace_bytes(int64_t n_rows) {
    return (n_rows < 0 || n_rows > kScatterMax) ? -1 : 0;
}

int ihg_batch_scatter_add(const float* rowgrad, int64_t ld_rowgrad, int32_t width, const int64_t* rows, int64_t n_rows, float* dense,
                          int64_t ld_dense, int32_t block_width, int64_t block_stride, float* tail, int64_t tail_row_offset,
                          int64_t tail_rows, ihg_stream_t stream) {
    if (n_rows < 0 || n_rows > kScatterMax) return fail(IHG_ERR_INVALID, "ihg_batch_scatter_add: 0..%d rows supported, got %lld", kScatterMax, (long long)n_rows);
    if (width <= 0 || ld_rowgrad < width || block_width <= 0 || ld_dense < block_width) return fail(IHG_ERR_INVALID, "ihg_batch_scatter_add: bad width / stride");
    if (n_rows == 0) return IHG_OK;
    if (rowgrad == nullptr || rows == nullptr || dense == nullptr) return fail(IHG_ERR_INVALID, "ihg_batch_scatter_add: null pointer");
    hipLaunchKernelGGL(batch_scatter_kernel, dim3(grid_for_waves(n_rows)), dim3(kBlockThreads), 0, static_cast<hipStream_t>(stream), rowgrad, ld_rowgrad,
                       width, rows, static_cast<int>(n_rows), dense, ld_dense, block_width, block_stride, tail, tail_row_offset, tail_rows);
    return check_launch("ihg_batch_scatter_add");
}

}  // extern "C"
