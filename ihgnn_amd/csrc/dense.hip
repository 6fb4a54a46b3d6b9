// dense.hip - node-level dense transforms (feature_transform and the hoisted first-order blocks): typed row GEMM, its input
// gradient and its weight / bias gradient.
#include "common.hpp"
#include "narrow.hpp"
#include "split.hpp"

namespace {

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
                                                                 int64_t bias_type_stride, TypePlan plan, float* __restrict__ out, int64_t ld_out) {
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
                const float bv = with_bias ? bias[type * bias_type_stride + c] : 0.f;
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
// FUSE_DX (d == SW == 64): the dout tile is in LDS anyway, so the input gradient dx = dout * W_type of the same rows is
// computed here too (weight fragments in registers for the whole kernel) and the separate row-GEMM launch over dout goes away.
template <int SW, bool FUSE_DX>
__global__ __launch_bounds__(kBlockThreads) void dense_weight_grad_kernel(const float* __restrict__ dout, int64_t ld_dout,
                                                                          const float* __restrict__ x, int64_t ld_x, TypePlan plan,
                                                                          int single_weight, float* __restrict__ slabs,
                                                                          float* __restrict__ bias_slabs, int d, const float* __restrict__ w,
                                                                          int64_t ld_w, int64_t w_type_stride, float* __restrict__ dx, int64_t ld_dx,
                                                                          int dx_accumulate) {
    static_assert(!FUSE_DX || SW == 64, "the fused input gradient covers whole 64-wide rows");
    constexpr int TE = 64, WT = SW / 32, V4_PER_ROW = SW / 4, LOADS = TE * V4_PER_ROW / kBlockThreads;
    constexpr int DSTRIDE = FUSE_DX ? SW + kRowPad : SW;      // padded rows for the ds_read_b128 A-operand reads of the dx product
    __shared__ __attribute__((aligned(16))) float dtile[TE][DSTRIDE];
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
    // dx job of this wave: rows et * 32 ..., columns cx * 32 ...; B[k][n] = W_type[k][n], k-order 8 t + 4 half + s as in row_gemm_kernel
    const int et = wave & 1, cx = wave >> 1;
    v4f breg[FUSE_DX ? SW / 8 : 1];
    if (FUSE_DX) {
        const float* wt = w + type * w_type_stride + cx * 32 + l31;
#pragma unroll
        for (int t = 0; t < SW / 8; ++t)
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2) breg[t][s2] = wt[static_cast<int64_t>(8 * t + 4 * half + s2) * ld_w];
    }

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
            if (FUSE_DX) {
                v16f gx;
#pragma unroll
                for (int r = 0; r < 16; ++r) gx[r] = 0.f;
#pragma unroll
                for (int t = 0; t < SW / 8; ++t) {
                    const v4f av = *reinterpret_cast<const v4f*>(&dtile[et * 32 + l31][8 * t + 4 * half]);
#pragma unroll
                    for (int s2 = 0; s2 < 4; ++s2) gx = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s2], breg[FUSE_DX ? t : 0][s2], gx, 0, 0, 0);
                }
                const int64_t r_base = r_begin + cur * TE;
                float* orow = dx + (r_base + et * 32) * ld_dx + cx * 32 + l31;
                if (dx_accumulate) {                        // dx already holds another contribution to the same gradient
                    float old[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        old[r] = r_base + et * 32 + acc_row(r, lane) < r_end ? orow[static_cast<int64_t>(acc_row(r, lane)) * ld_dx] : 0.f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) gx[r] += old[r];
                }
                if (r_base + TE <= r_end) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) orow[static_cast<int64_t>(acc_row(r, lane)) * ld_dx] = gx[r];
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (r_base + et * 32 + acc_row(r, lane) < r_end) orow[static_cast<int64_t>(acc_row(r, lane)) * ld_dx] = gx[r];
                }
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
                                                                          int64_t dw_type_stride, float* __restrict__ dbias, int bias_mask,
                                                                          int64_t dbias_type_stride) {
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
                if (dbias_type_stride != 0 && first_wave && c < d && dbias != nullptr) dbias[type * dbias_type_stride + c] = use ? part : 0.f;
            }
            if (dbias_type_stride == 0 && first_wave && c < d && dbias != nullptr) dbias[c] = acc;
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
// ------------------------------------------------------------------------------------------------
// Row GEMM in the strip style (D = 128), cf. interact.hip: eight waves, wave w owns the 16 output columns 16 w .. with the whole
// contraction index - its weight fragments (8 float4 = 32 VGPRs per node type) stay in registers, the input rows come in by
// LDS-DMA (tiles of 64 rows, double-buffered, swizzled for the ds_read_b128 A-operand reads), v_mfma_f32_16x16x4_f32 with four
// independent accumulator tiles per wave, results leave through an LDS image as whole 16-byte-per-lane rows.  One barrier per
// tile; the stores of tile k - 1 and the fill of tile k + 1 are issued inside the MFMA phase of tile k, half a phase apart on
// the two waves of a SIMD.
// pk_strip[type][strip][g][lane][4]:  transpose == 0:  W_t[16 strip + (lane & 15)][16 g + 4 (lane >> 4) + s]     (out = in * W^T)
//                                     transpose == 1:  W_t[16 g + 4 (lane >> 4) + s][16 strip + (lane & 15)]     (out = in * W)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlockThreads) void pack_dense_strip_kernel(const float* __restrict__ w, int64_t ld_w, int64_t type_stride,
                                                                         int n_types, int d, int transpose, float* __restrict__ pk) {
    const int kg = d / 16;
    const int per_type = (d / 16) * kg * kWave;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < per_type * n_types; idx += gridDim.x * blockDim.x) {
        const int type = idx / per_type, rem = idx - type * per_type;
        const int lane = rem & (kWave - 1), g = (rem >> 6) % kg, strip = (rem >> 6) / kg;
        const int c = lane & 15, kq = lane >> 4;
        const float* wt = w + type * type_stride;
        float4 v;
        if (transpose == 0) {
            const float* src = wt + static_cast<int64_t>(16 * strip + c) * ld_w + 16 * g + 4 * kq;
            v = make_float4(src[0], src[1], src[2], src[3]);
        } else {
            const float* src = wt + static_cast<int64_t>(16 * g + 4 * kq) * ld_w + 16 * strip + c;
            v = make_float4(src[0], src[ld_w], src[2 * ld_w], src[3 * ld_w]);
        }
        *reinterpret_cast<float4*>(pk + static_cast<int64_t>(idx) * 4) = v;
    }
}

__device__ __forceinline__ void dense_lds_dma16(const float* src, float* lds_piece) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)lds_piece, 16, 0, 0);
}

template <int D, int TE>
__global__ __launch_bounds__(512, 4) void row_gemm_strip_kernel(const float* __restrict__ in, int64_t ld_in, const float* __restrict__ pk,
                                                             int64_t pk_type_stride, const float* __restrict__ bias, int bias_mask,
                                                             int64_t bias_type_stride, TypePlan plan, float* __restrict__ out, int64_t ld_out) {
    static_assert(D == 128, "eight 16-column strips");
    constexpr int RT = TE / 16, KG = D / 16, OSTRIDE = D + 4, PIECES = TE * D * 4 / 1024 / 8;     // DMA pieces per wave per tile (TE = 32: 2)
    __shared__ __attribute__((aligned(16))) float xt[2][TE][D];
    __shared__ __attribute__((aligned(16))) float ot[2][TE][OSTRIDE];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int total_tiles = plan.tile_prefix[3];
    const int grid = gridDim.x;
    const int n_my = static_cast<int>(blockIdx.x) < total_tiles ? (total_tiles - static_cast<int>(blockIdx.x) + grid - 1) / grid : 0;
    if (n_my == 0) return;
    auto tile_type = [&](int tile_id) { return tile_id >= plan.tile_prefix[2] ? 2 : (tile_id >= plan.tile_prefix[1] ? 1 : 0); };
    auto tile_rows = [&](int tile_id, int64_t& r_base, int64_t& r_end) {
        const int type = tile_type(tile_id);
        r_base = plan.begin[type] + static_cast<int64_t>(tile_id - plan.tile_prefix[type]) * TE;
        r_end = plan.begin[type + 1];
    };
    int tl = tid;
    auto issue_dma = [&](int k) {
        int64_t r_base, r_end;
        tile_rows(static_cast<int>(blockIdx.x) + k * grid, r_base, r_end);
        const int prow = (tl >> 5) & 1, pchunk = tl & 31;
#pragma unroll
        for (int kk = 0; kk < PIECES; ++kk) {
            const int r0 = 2 * (wave * PIECES + kk);
            int64_t v = r_base + r0 + prow;
            v = v < r_end ? v : r_end - 1;                               // rows past the type's end re-read its last row (never stored)
            dense_lds_dma16(in + v * ld_in + ((pchunk ^ ((r0 + prow) & 15)) << 2), &xt[k & 1][r0][0]);
        }
    };
    auto store_out = [&](int k) {
        int64_t r_base, r_end;
        const int tile_id = static_cast<int>(blockIdx.x) + k * grid;
        tile_rows(tile_id, r_base, r_end);
        const int type = tile_type(tile_id);
        const bool with_bias = bias != nullptr && ((bias_mask >> type) & 1);
        const int c4 = (tl & 31) * 4;
        v4f bv = v4f{0.f, 0.f, 0.f, 0.f};
        if (with_bias) bv = *reinterpret_cast<const v4f*>(bias + type * bias_type_stride + c4);
#pragma unroll
        for (int x = 0; x < TE / 16; ++x) {
            const int row = (tl >> 5) + 16 * x;                          // 512 threads = 16 rows x 32 vectors per pass
            const v4f v = *reinterpret_cast<const v4f*>(&ot[k & 1][row][c4]) + bv;
            if (r_base + row < r_end) *reinterpret_cast<v4f*>(out + (r_base + row) * ld_out + c4) = v;
        }
    };
    v4f wreg[KG];
    int cur_type = -1;
    issue_dma(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const bool late = wave >= 4;
    for (int k = 0; k < n_my; ++k) {
        const int tile_id = static_cast<int>(blockIdx.x) + k * grid;
        const int type = tile_type(tile_id);
        if (type != cur_type) {                                           // at most three times per kernel: tiles are ordered by type
            const v4f* pk4 = reinterpret_cast<const v4f*>(pk + type * pk_type_stride) + static_cast<int64_t>(wave) * KG * kWave + (tid & 63);
#pragma unroll
            for (int g = 0; g < KG; ++g) wreg[g] = pk4[g * kWave];
            cur_type = type;
        }
        asm volatile("" : "+v"(tl));
        const int arow = tl & 15, kq = (tl >> 4) & 3;
        int lane_off = arow * (D * 4) + ((kq ^ arow) << 4);
        asm volatile("" : "+v"(lane_off));
        const char* xbase = reinterpret_cast<const char*>(&xt[k & 1][0][0]);
        v4f acc[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc[rt] = v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int g = 0; g < KG; ++g) {
            if (g == 0 || g == KG / 2) {
                if (late == (g != 0)) {
                    if (k + 1 < n_my) issue_dma(k + 1);                     // that buffer's last reader was tile k - 1
                    if (k > 0) store_out(k - 1);
                }
            }
            v4f a[RT];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) a[rt] = *reinterpret_cast<const v4f*>(xbase + (lane_off ^ (g << 6)) + rt * 16 * (D * 4));
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2)
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rt][s2], wreg[g][s2], acc[rt], 0, 0, 0);
        }
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) ot[k & 1][rt * 16 + 4 * kq + r][16 * wave + arow] = acc[rt][r];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    asm volatile("" : "+v"(tl));
    store_out(n_my - 1);
}

constexpr int kDenseSlabs = 256;

// in_typed / out_typed: the rows of every node type start at an address of their own (TypedRows; `in` / `out` are then ignored) - on the bf16-split kernels only
int launch_row_gemm(int dim, const float* in, int64_t ld_in, const float* w, int64_t ld_w, int64_t w_type_stride, int transpose,
                    const float* bias, int bias_mask, int64_t bias_type_stride, const int64_t* type_begin, float* out, int64_t ld_out, float* pk,
                    hipStream_t s, const TypedRows* in_typed = nullptr, const TypedRowsOut* out_typed = nullptr, int accumulate = 0) {
    const int n_types = w_type_stride == 0 ? 1 : 3;
    const bool out_ok = out_typed != nullptr ? (aligned16(out_typed->p[0]) && aligned16(out_typed->p[1]) && aligned16(out_typed->p[2])) : aligned16(out);
    const bool in_ok = in_typed != nullptr ? (aligned16(in_typed->p[0]) && aligned16(in_typed->p[1]) && aligned16(in_typed->p[2])) : aligned16(in);
    if (narrow_linear_ok(dim, ld_in, ld_out) && in_ok && out_ok && (bias == nullptr || (aligned16(bias) && bias_type_stride % 4 == 0))) {      // d = 32 / 64: narrow.hip (fp32 MFMA, any arithmetic mode)
        launch_row_gemm_narrow(dim, in_typed != nullptr ? *in_typed : typed_rows(in), ld_in, w, ld_w, w_type_stride, transpose, bias, bias_mask, bias_type_stride, type_begin,
                               out_typed != nullptr ? *out_typed : typed_rows_out(out), ld_out, accumulate, pk, s);
        return IHG_OK;
    }
    if (out_ok && split_row_gemm_ok(dim, nullptr, ld_out, bias, bias_type_stride)) {   // the bf16 planes sit behind the slabs (see ihg_node_linear_workspace_bytes)
        void* planes = pk + 3LL * dim * dim + 3LL * kDenseSlabs * (static_cast<int64_t>(dim) * dim + dim);
        launch_row_gemm_split(dim, in_typed != nullptr ? *in_typed : typed_rows(in), ld_in, w, ld_w, w_type_stride, transpose, bias, bias_mask, bias_type_stride, type_begin,
                              out_typed != nullptr ? *out_typed : typed_rows_out(out), ld_out, planes, s, accumulate);
        return IHG_OK;
    }
    if (accumulate) return fail(IHG_ERR_INVALID, "node-level linear map accumulating into its output: needs the bf16-split kernels (dim 128 / 256, aligned rows)");
    if (in_typed != nullptr || out_typed != nullptr) return fail(IHG_ERR_INVALID, "node-level linear map over typed rows: needs the bf16-split kernels (dim 128 / 256, aligned rows)");
    if (dim == 128 && aligned16(out) && ld_out % 4 == 0 && (bias == nullptr || (aligned16(bias) && bias_type_stride % 4 == 0))) {
        const int items = n_types * (dim / 16) * (dim / 16) * kWave;
        hipLaunchKernelGGL(pack_dense_strip_kernel, dim3((items + kBlockThreads - 1) / kBlockThreads), dim3(kBlockThreads), 0, s, w, ld_w, w_type_stride,
                           n_types, dim, transpose, pk);
        // tiles of 32 rows, two workgroups per CU (66 KB of LDS each): a pure stream over [N, d] wants bytes in flight more than big tiles
        const TypePlan plan = make_plan(type_begin, 32);
        if (plan.tile_prefix[3] == 0) return IHG_OK;
        hipLaunchKernelGGL((row_gemm_strip_kernel<128, 32>), dim3(std::min(plan.tile_prefix[3], 512)), dim3(512), 0, s, in, ld_in, pk,
                           n_types == 1 ? int64_t{0} : static_cast<int64_t>(dim) * dim, bias, bias_mask, bias_type_stride, plan, out, ld_out);
        return IHG_OK;
    }
    const int pack_items = n_types * (dim / 32) * (dim / 8) * kWave;
    hipLaunchKernelGGL(pack_dense_kernel, dim3((pack_items + kBlockThreads - 1) / kBlockThreads), dim3(kBlockThreads), 0, s, w, ld_w, w_type_stride,
                       n_types, dim, transpose, pk);
    const int64_t pk_type_stride = n_types == 1 ? 0 : static_cast<int64_t>(dim) * dim;
    const TypePlan plan = make_plan(type_begin, dim == 32 ? 128 : 64);
    if (plan.tile_prefix[3] == 0) return IHG_OK;
    const int grid = std::min(plan.tile_prefix[3], 256 * 4);
#define IHG_RG(D) hipLaunchKernelGGL((row_gemm_kernel<D>), dim3(grid), dim3(kBlockThreads), 0, s, in, ld_in, pk, pk_type_stride, bias, bias_mask, bias_type_stride, plan, out, ld_out)
    switch (dim) {
        case 32: IHG_RG(32); break;
        case 64: IHG_RG(64); break;
        case 128: IHG_RG(128); break;
        default: IHG_RG(256); break;
    }
#undef IHG_RG
    return IHG_OK;
}

// ------------------------------------------------------------------------------------------------
// Composition of the two linear maps of a first-order layer (tiny: 3 d^3 multiply-adds; one thread per output element,
// sums in index order).  A = aggregation.weight [d, 3d] (blocks A_u | A_q | A_i), c its bias, W / b = feature_transform.
//   fwd:  We[i][t d + j] = sum_k A[i][t d + k] W[k][j]        be[t][i] = sum_k A[i][t d + k] b[k] + (t == 0 ? c[i] : 0)
//   bwd:  dA[i][t d + k] = sum_j dWe[i][t d + j] W[k][j] + dbe[t][i] b[k]      dW[k][j] = sum_t sum_i A[i][t d + k] dWe[i][t d + j]
//         db[k] = sum_t sum_i A[i][t d + k] dbe[t][i]                           dc[i] = dbe[0][i]
// ------------------------------------------------------------------------------------------------
// sum_k x[k * sx] * y[k * sy], k < n: eight loads of each operand in flight, four partial sums combined in a fixed order
__device__ __forceinline__ float strided_dot(const float* __restrict__ x, int64_t sx, const float* __restrict__ y, int64_t sy, int n) {
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    int k = 0;
    for (; k + 8 <= n; k += 8) {
        float xv[8], yv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            xv[u] = x[(k + u) * sx];
            yv[u] = y[(k + u) * sy];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc[u & 3] += xv[u] * yv[u];
    }
    for (; k < n; ++k) acc[k & 3] += x[k * sx] * y[k * sy];
    return (acc[0] + acc[1]) + (acc[2] + acc[3]);
}

__global__ __launch_bounds__(kBlockThreads) void compose_fwd_kernel(const float* __restrict__ a, int64_t ld_a, const float* __restrict__ c,
                                                                    const float* __restrict__ w, int64_t ld_w, const float* __restrict__ b,
                                                                    float* __restrict__ we, int64_t ld_we, float* __restrict__ be, int d) {
    const int64_t n_w = 3LL * d * d, total = n_w + 3LL * d;
    for (int64_t idx = static_cast<int64_t>(blockIdx.x) * kBlockThreads + threadIdx.x; idx < total; idx += static_cast<int64_t>(gridDim.x) * kBlockThreads) {
        if (idx < n_w) {                                    // lanes run over j: W rows coalesced, the A element is a broadcast
            const int i = static_cast<int>(idx / (3 * d)), col = static_cast<int>(idx % (3 * d)), t = col / d, j = col % d;
            we[i * ld_we + col] = strided_dot(a + i * ld_a + t * d, 1, w + j, ld_w, d);
        } else {
            const int r = static_cast<int>(idx - n_w), t = r / d, i = r % d;
            be[r] = strided_dot(a + i * ld_a + t * d, 1, b, 1, d) + (t == 0 && c != nullptr ? c[i] : 0.f);
        }
    }
}

__global__ __launch_bounds__(kBlockThreads) void compose_bwd_kernel(const float* __restrict__ a, int64_t ld_a, const float* __restrict__ w, int64_t ld_w,
                                                                    const float* __restrict__ b, const float* __restrict__ dwe, int64_t ld_dwe,
                                                                    const float* __restrict__ dbe, float* __restrict__ da, int64_t ld_da,
                                                                    float* __restrict__ dc, float* __restrict__ dw, int64_t ld_dw,
                                                                    float* __restrict__ db, int d) {
    // part 1, one WAVE per element of dA (both operands are contiguous in the summation index j: lanes run over j)
    const int lane = threadIdx.x & 63;
    const int64_t n_a = 3LL * d * d;
    for (int64_t o = global_wave_id(); o < n_a; o += global_wave_count()) {
        const int i = static_cast<int>(o / (3 * d)), col = static_cast<int>(o % (3 * d)), t = col / d, k = col % d;
        const float* grow = dwe + i * ld_dwe + t * d;
        const float* wrow = w + k * ld_w;
        float acc = 0.f;
        for (int j = lane; j < d; j += kWave) acc += grow[j] * wrow[j];
        for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o);      // butterfly: every lane ends with the same, order-fixed sum
        if (lane == 0) da[i * ld_da + col] = acc + dbe[t * d + i] * b[k];
    }
    // part 2, one THREAD per element of dW / db / dc (lanes run over the output column: coalesced rows, broadcast scalars)
    const int64_t n_w = static_cast<int64_t>(d) * d, total = n_w + 2LL * d;
    for (int64_t idx = static_cast<int64_t>(blockIdx.x) * kBlockThreads + threadIdx.x; idx < total; idx += static_cast<int64_t>(gridDim.x) * kBlockThreads) {
        if (idx < n_w) {
            const int k = static_cast<int>(idx / d), j = static_cast<int>(idx % d);
            float acc = 0.f;
            for (int t = 0; t < 3; ++t) acc += strided_dot(a + t * d + k, ld_a, dwe + t * d + j, ld_dwe, d);
            dw[k * ld_dw + j] = acc;
        } else if (idx < n_w + d) {
            const int k = static_cast<int>(idx - n_w);
            float acc = 0.f;
            for (int t = 0; t < 3; ++t) acc += strided_dot(a + t * d + k, ld_a, dbe + t * d, 1, d);
            db[k] = acc;
        } else {
            const int i = static_cast<int>(idx - n_w - d);
            if (dc != nullptr) dc[i] = dbe[i];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Any-width forms of the three node-level products (d not in {32, 64, 128, 256}, or rows that are not 16-byte aligned): one thread
// per output element, sums in index order.  Correct for every shape; the tiled kernels above are the fast path.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int type_of_row(const TypePlan& plan, int64_t v) { return v >= plan.begin[2] ? 2 : (v >= plan.begin[1] ? 1 : 0); }

// transpose == 0: out[v][n] = sum_k in[v][k] W_t[n][k] (+ bias);   transpose == 1: out[v][k] = sum_n in[v][n] W_t[n][k]
__global__ __launch_bounds__(kBlockThreads) void row_gemm_generic_kernel(const float* __restrict__ in, int64_t ld_in, const float* __restrict__ w, int64_t ld_w,
                                                                         int64_t w_type_stride, int transpose, const float* __restrict__ bias, int bias_mask,
                                                                         int64_t bias_type_stride, TypePlan plan, float* __restrict__ out, int64_t ld_out, int d) {
    const int64_t rows = plan.begin[3] - plan.begin[0];
    const int64_t total = rows * d;
    for (int64_t idx = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; idx < total; idx += static_cast<int64_t>(gridDim.x) * blockDim.x) {
        const int64_t v = plan.begin[0] + idx / d;
        const int n = static_cast<int>(idx % d);
        const int type = type_of_row(plan, v);
        const float* wt = w + type * w_type_stride;
        const float* row = in + v * ld_in;
        float acc = 0.f;
        if (transpose == 0) {
            for (int k = 0; k < d; ++k) acc += row[k] * wt[static_cast<int64_t>(n) * ld_w + k];
            if (bias != nullptr && ((bias_mask >> type) & 1)) acc += bias[type * bias_type_stride + n];
        } else {
            for (int k = 0; k < d; ++k) acc += row[k] * wt[static_cast<int64_t>(k) * ld_w + n];
        }
        out[v * ld_out + n] = acc;
    }
}

// dW_t[c][j] = sum_{v in t} dout[v][c] x[v][j] and the bias gradient for ANY width: grid = (row slabs, 16 x 16 output tiles, weight
// types); a workgroup walks its slab of the type's rows in chunks of 32 staged in LDS, thread (c, j) of the tile keeps one sum (rows in
// index order); per-slab partials in dense_weight_grad_kernel's slab layout ([type][slab][d][d], bias part [type][slab][d]), added up in
// slab order by dense_generic_reduce_kernel - deterministic, and N / slabs serial steps per thread instead of N.
constexpr int kGenericSlabs = 64;
constexpr int kGenericChunk = 32;

__global__ __launch_bounds__(kBlockThreads) void dense_weight_grad_generic_kernel(const float* __restrict__ dout, int64_t ld_dout, const float* __restrict__ x,
                                                                                  int64_t ld_x, TypePlan plan, int single_weight, int d,
                                                                                  float* __restrict__ slabs, float* __restrict__ bias_slabs) {
    __shared__ float dt[kGenericChunk][17], xt[kGenericChunk][17];
    const int tiles = (d + 15) / 16;
    const int tc = blockIdx.y / tiles, tj = blockIdx.y % tiles, type = blockIdx.z;
    const int n_slabs = gridDim.x;
    const int64_t r_begin = single_weight ? plan.begin[0] : plan.begin[type], r_end = single_weight ? plan.begin[3] : plan.begin[type + 1];
    const int64_t per = (r_end - r_begin + n_slabs - 1) / n_slabs;
    const int64_t v0 = r_begin + blockIdx.x * per, v1 = v0 + per < r_end ? v0 + per : r_end;
    const int tid = threadIdx.x, c = tid >> 4, j = tid & 15;
    float acc = 0.f, colsum = 0.f;
    for (int64_t base = v0; base < v1; base += kGenericChunk) {
        for (int idx = tid; idx < kGenericChunk * 16; idx += kBlockThreads) {
            const int r = idx >> 4, k = idx & 15;
            const int64_t v = base + r;
            const bool live = v < v1;
            dt[r][k] = live && 16 * tc + k < d ? dout[v * ld_dout + 16 * tc + k] : 0.f;
            xt[r][k] = live && 16 * tj + k < d ? x[v * ld_x + 16 * tj + k] : 0.f;
        }
        __syncthreads();
#pragma unroll 8
        for (int r = 0; r < kGenericChunk; ++r) acc += dt[r][c] * xt[r][j];
        if (tj == 0 && j == 0) {
#pragma unroll 8
            for (int r = 0; r < kGenericChunk; ++r) colsum += dt[r][c];
        }
        __syncthreads();
    }
    const int64_t slab = static_cast<int64_t>(type) * n_slabs + blockIdx.x;
    if (16 * tc + c < d && 16 * tj + j < d) slabs[(slab * d + 16 * tc + c) * d + 16 * tj + j] = acc;
    if (tj == 0 && j == 0 && 16 * tc + c < d) bias_slabs[slab * d + 16 * tc + c] = colsum;
}

__global__ __launch_bounds__(kBlockThreads) void dense_generic_reduce_kernel(const float* __restrict__ slabs, const float* __restrict__ bias_slabs, int n_slabs,
                                                                             int n_types, int d, float* __restrict__ dw, int64_t ld_dw, int64_t dw_type_stride,
                                                                             float* __restrict__ dbias, int bias_mask, int64_t dbias_type_stride) {
    const int64_t per_type = static_cast<int64_t>(d) * d;
    const int64_t total = per_type * n_types + d;
    for (int64_t idx = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; idx < total; idx += static_cast<int64_t>(gridDim.x) * blockDim.x) {
        if (idx < per_type * n_types) {
            const int type = static_cast<int>(idx / per_type);
            const int64_t rem = idx % per_type;
            const float* src = slabs + static_cast<int64_t>(type) * n_slabs * per_type + rem;
            float acc = 0.f;
            for (int sl = 0; sl < n_slabs; ++sl) acc += src[sl * per_type];
            dw[(rem / d) * ld_dw + type * dw_type_stride + rem % d] = acc;
        } else if (dbias != nullptr) {
            const int c = static_cast<int>(idx - per_type * n_types);
            float all = 0.f;
            for (int type = 0; type < n_types; ++type) {
                const bool use = n_types == 1 || ((bias_mask >> type) & 1);
                float part = 0.f;
                if (use)
                    for (int sl = 0; sl < n_slabs; ++sl) part += bias_slabs[(static_cast<int64_t>(type) * n_slabs + sl) * d + c];
                all += part;
                if (dbias_type_stride != 0) dbias[type * dbias_type_stride + c] = part;
            }
            if (dbias_type_stride == 0) dbias[c] = all;
        }
    }
}

inline bool tiled_node_linear(int dim, int64_t ld_a, int64_t ld_b, const void* a, const void* workspace) {
    return mfma_dim(dim) && ld_a % 4 == 0 && ld_b % 4 == 0 && aligned16(a) && workspace != nullptr && aligned16(workspace);
}

inline void launch_row_gemm_generic(int dim, const float* in, int64_t ld_in, const float* w, int64_t ld_w, int64_t w_type_stride, int transpose,
                                    const float* bias, int bias_mask, int64_t bias_type_stride, const int64_t* type_begin, float* out, int64_t ld_out,
                                    hipStream_t s) {
    const TypePlan plan = make_plan(type_begin, 64);
    const int64_t total = (type_begin[3] - type_begin[0]) * dim;
    const int grid = static_cast<int>(std::min<int64_t>((total + kBlockThreads - 1) / kBlockThreads, kMaxBlocks * 4));
    hipLaunchKernelGGL(row_gemm_generic_kernel, dim3(grid), dim3(kBlockThreads), 0, s, in, ld_in, w, ld_w, w_type_stride, transpose, bias, bias_mask,
                       bias_type_stride, plan, out, ld_out, dim);
}

}  // namespace

extern "C" {

int64_t ihg_node_linear_workspace_bytes(int32_t dim) {
    if (dim <= 0) return -1;
    if (!mfma_dim(dim)) return 3LL * kGenericSlabs * (static_cast<int64_t>(dim) * dim + dim) * static_cast<int64_t>(sizeof(float));      // any-width kernels: row-slab partials only
    const int64_t packed = 3LL * dim * dim;
    const int64_t slabs = 3LL * kDenseSlabs * (static_cast<int64_t>(dim) * dim + dim);
    return (packed + slabs + split_dense_plane_floats(dim)) * static_cast<int64_t>(sizeof(float));
}

static int node_linear_common_check(const char* what, int32_t dim, int64_t ld_a, int64_t ld_b, int64_t ld_w, const int64_t* type_begin,
                                    const void* workspace, int64_t workspace_bytes) {
    if (dim <= 0) return fail(IHG_ERR_INVALID, "%s: dim %d", what, dim);
    if (type_begin == nullptr) return fail(IHG_ERR_INVALID, "%s: null pointer", what);
    if (ld_a < dim || ld_b < dim || ld_w < dim) return fail(IHG_ERR_INVALID, "%s: bad leading dimension", what);
    if (!mfma_dim(dim) || ld_a % 4 || ld_b % 4 || workspace == nullptr) return IHG_OK;        // the any-width kernels take it from here
    if (!(type_begin[0] <= type_begin[1] && type_begin[1] <= type_begin[2] && type_begin[2] <= type_begin[3])) return fail(IHG_ERR_INVALID, "%s: type ranges not ascending", what);
    if (workspace_bytes < ihg_node_linear_workspace_bytes(dim)) return fail(IHG_ERR_WORKSPACE, "%s: workspace too small", what);
    if (!aligned16(workspace)) return fail(IHG_ERR_INVALID, "%s: workspace not 16-byte aligned", what);
    return IHG_OK;
}

int ihg_node_linear_fwd(const float* x, int64_t ld_x, const float* w, int64_t ld_w, int64_t w_type_stride, const float* bias,
                        int32_t bias_type_mask, int64_t bias_type_stride, const int64_t* type_begin, float* out, int64_t ld_out, void* workspace,
                        int64_t workspace_bytes, int32_t dim, ihg_stream_t stream) {
    if (int rc = node_linear_common_check("ihg_node_linear_fwd", dim, ld_x, ld_out, ld_w, type_begin, workspace, workspace_bytes)) return rc;
    if (type_begin[3] == type_begin[0]) return IHG_OK;
    if (x == nullptr || w == nullptr || out == nullptr) return fail(IHG_ERR_INVALID, "ihg_node_linear_fwd: null pointer");
    if (tiled_node_linear(dim, ld_x, ld_out, x, workspace))
        launch_row_gemm(dim, x, ld_x, w, ld_w, w_type_stride, 0, bias, bias_type_mask, bias_type_stride, type_begin, out, ld_out,
                        static_cast<float*>(workspace), static_cast<hipStream_t>(stream));
    else
        launch_row_gemm_generic(dim, x, ld_x, w, ld_w, w_type_stride, 0, bias, bias_type_mask, bias_type_stride, type_begin, out, ld_out,
                                static_cast<hipStream_t>(stream));
    return check_launch("ihg_node_linear_fwd");
}

int ihg_node_linear_bwd_input(const float* dout, int64_t ld_dout, const float* w, int64_t ld_w, int64_t w_type_stride,
                              const int64_t* type_begin, float* dx, int64_t ld_dx, void* workspace, int64_t workspace_bytes,
                              int32_t dim, ihg_stream_t stream) {
    if (int rc = node_linear_common_check("ihg_node_linear_bwd_input", dim, ld_dout, ld_dx, ld_w, type_begin, workspace, workspace_bytes)) return rc;
    if (type_begin[3] == type_begin[0]) return IHG_OK;
    if (dout == nullptr || w == nullptr || dx == nullptr) return fail(IHG_ERR_INVALID, "ihg_node_linear_bwd_input: null pointer");
    if (tiled_node_linear(dim, ld_dout, ld_dx, dout, workspace))
        launch_row_gemm(dim, dout, ld_dout, w, ld_w, w_type_stride, 1, nullptr, 0, 0, type_begin, dx, ld_dx, static_cast<float*>(workspace),
                        static_cast<hipStream_t>(stream));
    else
        launch_row_gemm_generic(dim, dout, ld_dout, w, ld_w, w_type_stride, 1, nullptr, 0, 0, type_begin, dx, ld_dx, static_cast<hipStream_t>(stream));
    return check_launch("ihg_node_linear_bwd_input");
}

int32_t ihg_node_linear_bwd_accumulates(int32_t dim, int64_t ld_dout, int64_t ld_x, int64_t ld_dx) {
    // the weight-gradient kernels that form dx in the same pass and can add it onto what dx already holds: d = 64 (fp32 MFMA), d = 128 (bf16-split)
    if (ld_dout % 4 || ld_x % 4 || ld_dx % 4) return 0;
    return dim == 64 || dim == kNarrowDim || ((dim == 128 || dim == 256) && split_arith_enabled()) ? 1 : 0;       // (d = 256: the input gradient is a row-GEMM launch of its own, which adds onto dx)
}

int ihg_node_linear_bwd_weight(const float* dout, int64_t ld_dout, const float* x, int64_t ld_x, const int64_t* type_begin,
                               float* dw, int64_t ld_dw, int64_t dw_type_stride, float* dbias, int32_t bias_type_mask, int64_t dbias_type_stride,
                               const float* w, int64_t ld_w, float* dx, int64_t ld_dx, int32_t dx_accumulate,
                               void* workspace, int64_t workspace_bytes, int32_t dim, ihg_stream_t stream) {
    if (int rc = node_linear_common_check("ihg_node_linear_bwd_weight", dim, ld_dout, ld_x, ld_dw, type_begin, workspace, workspace_bytes)) return rc;
    if (dout == nullptr || x == nullptr || dw == nullptr) return fail(IHG_ERR_INVALID, "ihg_node_linear_bwd_weight: null pointer");
    if (dx != nullptr && (w == nullptr || ld_w < dim || ld_dx < dim)) return fail(IHG_ERR_INVALID, "ihg_node_linear_bwd_weight: dx needs w and row strides >= dim");
    if (dx_accumulate && (dx == nullptr || !ihg_node_linear_bwd_accumulates(dim, ld_dout, ld_x, ld_dx)))
        return fail(IHG_ERR_INVALID, "ihg_node_linear_bwd_weight: dx_accumulate needs dx and a fused input-gradient kernel (ihg_node_linear_bwd_accumulates)");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int n_types = dw_type_stride == 0 ? 1 : 3;
    if (!tiled_node_linear(dim, ld_dout, ld_x, dout, workspace) || !aligned16(x) || (dx != nullptr && ld_dx % 4)) {
        // the any-width kernels OVERWRITE dx: a caller that asked for dx += (the member gradients are already in it) on rows this branch takes - unaligned dout / x
        // at a width ihg_node_linear_bwd_accumulates said yes to - is refused, not silently given a dx without its first addend
        if (dx_accumulate) return fail(IHG_ERR_INVALID, "ihg_node_linear_bwd_weight: dx_accumulate needs 16-byte aligned dout and x rows (the any-width kernels overwrite dx)");
        if (dx != nullptr) launch_row_gemm_generic(dim, dout, ld_dout, w, ld_w, dw_type_stride, 1, nullptr, 0, 0, type_begin, dx, ld_dx, s);
        // row-slab partials at the front of the workspace (every width's workspace holds at least kGenericSlabs of them), then a fixed-order sum
        const int64_t need = 3LL * kGenericSlabs * (static_cast<int64_t>(dim) * dim + dim) * static_cast<int64_t>(sizeof(float));
        if (workspace == nullptr || workspace_bytes < need) return fail(IHG_ERR_WORKSPACE, "ihg_node_linear_bwd_weight: workspace too small (any-width path)");
        float* gslabs = static_cast<float*>(workspace);
        float* gbias = gslabs + 3LL * kGenericSlabs * dim * dim;
        const int tiles = (dim + 15) / 16;
        hipLaunchKernelGGL(dense_weight_grad_generic_kernel, dim3(kGenericSlabs, tiles * tiles, n_types), dim3(kBlockThreads), 0, s, dout, ld_dout, x, ld_x,
                           make_plan(type_begin, 64), n_types == 1 ? 1 : 0, dim, gslabs, gbias);
        const int64_t total = static_cast<int64_t>(dim) * dim * n_types + dim;
        hipLaunchKernelGGL(dense_generic_reduce_kernel, dim3(static_cast<int>(std::min<int64_t>((total + kBlockThreads - 1) / kBlockThreads, kMaxBlocks))),
                           dim3(kBlockThreads), 0, s, gslabs, gbias, kGenericSlabs, n_types, dim, dw, ld_dw, dw_type_stride, dbias, bias_type_mask,
                           n_types == 1 ? int64_t{0} : dbias_type_stride);
        return check_launch("ihg_node_linear_bwd_weight");
    }
    float* slabs = static_cast<float*>(workspace) + 3LL * dim * dim;
    float* bias_slabs = slabs + 3LL * kDenseSlabs * dim * dim;
    const TypePlan plan = make_plan(type_begin, 64);
    // weights of type t are the column block t of w exactly as for dw: w_type_stride == dw_type_stride
    int n_slabs = kDenseSlabs;
    if (narrow_linear_ok(dim, ld_dout, ld_x) && aligned16(dout) && (dx == nullptr || (aligned16(dx) && aligned16(w) && ld_w % 4 == 0))) {
        // d = 32 / 64: weight, bias and input gradient in one pass over (dout, x) (narrow.hip)
        const TypedRowsOut dx_rows = typed_rows_out(dx);
        n_slabs = launch_dense_weight_narrow(dim, dout, ld_dout, typed_rows(x), ld_x, type_begin, n_types, slabs, bias_slabs, w, ld_w, dw_type_stride, dx != nullptr ? &dx_rows : nullptr, ld_dx,
                                             dx_accumulate, static_cast<float*>(workspace), s);
    } else if (dx_accumulate && dim == kNarrowDim) {
        return fail(IHG_ERR_INVALID, "ihg_node_linear_bwd_weight: dx_accumulate at dim 32 needs 16-byte aligned rows");
    } else if (split_dense_weight_ok(dim, dout, ld_dout, x, ld_x)) {             // bf16-split contraction (d = 128, 256); the input gradient stays a row-GEMM launch
        const bool fused_dx = dx != nullptr && dim == 128 && aligned16(dx) && ld_dx % 4 == 0;
        if (dx_accumulate && !fused_dx && !(dim == 256 && aligned16(dx) && ld_dx % 4 == 0))
            return fail(IHG_ERR_INVALID, "ihg_node_linear_bwd_weight: dx_accumulate needs 16-byte aligned dx rows");
        if (dx != nullptr && !fused_dx) {
            if (int rc = launch_row_gemm(dim, dout, ld_dout, w, ld_w, dw_type_stride, 1, nullptr, 0, 0, type_begin, dx, ld_dx, static_cast<float*>(workspace), s, nullptr, nullptr, dx_accumulate)) return rc;
        }
        void* planes = static_cast<float*>(workspace) + 3LL * dim * dim + 3LL * kDenseSlabs * (static_cast<int64_t>(dim) * dim + dim);
        const TypedRowsOut dx_rows = typed_rows_out(dx);
        n_slabs = launch_dense_weight_split(dim, dout, ld_dout, typed_rows(x), ld_x, type_begin, n_types, slabs, bias_slabs, w, ld_w, dw_type_stride, fused_dx ? &dx_rows : nullptr, ld_dx,
                                            planes, s, fused_dx ? dx_accumulate : 0);
    } else if (dim == 64 && dx != nullptr) {
        hipLaunchKernelGGL((dense_weight_grad_kernel<64, true>), dim3(kDenseSlabs, 1, n_types), dim3(kBlockThreads), 0, s, dout, ld_dout, x, ld_x, plan,
                           n_types == 1 ? 1 : 0, slabs, bias_slabs, dim, w, ld_w, dw_type_stride, dx, ld_dx, dx_accumulate);
    } else {
        if (dx != nullptr) {                               // other widths: the row-GEMM pass over dout stays a launch of its own
            launch_row_gemm(dim, dout, ld_dout, w, ld_w, dw_type_stride, 1, nullptr, 0, 0, type_begin, dx, ld_dx, static_cast<float*>(workspace), s);
        }
        if (dim == 32) {
            hipLaunchKernelGGL((dense_weight_grad_kernel<32, false>), dim3(kDenseSlabs, 1, n_types), dim3(kBlockThreads), 0, s, dout, ld_dout, x, ld_x, plan,
                               n_types == 1 ? 1 : 0, slabs, bias_slabs, dim, static_cast<const float*>(nullptr), int64_t{0}, int64_t{0},
                               static_cast<float*>(nullptr), int64_t{0}, 0);
        } else {
            const int subs = (dim / 64) * (dim / 64);
            hipLaunchKernelGGL((dense_weight_grad_kernel<64, false>), dim3(kDenseSlabs, subs, n_types), dim3(kBlockThreads), 0, s, dout, ld_dout, x, ld_x, plan,
                               n_types == 1 ? 1 : 0, slabs, bias_slabs, dim, static_cast<const float*>(nullptr), int64_t{0}, int64_t{0},
                               static_cast<float*>(nullptr), int64_t{0}, 0);
        }
    }
    const int total = dim * dim * n_types + dim;
    hipLaunchKernelGGL(dense_slab_reduce_kernel, dim3((total + kWave - 1) / kWave), dim3(kBlockThreads), 0, s, slabs, bias_slabs,
                       n_slabs, n_types, dim, dw, ld_dw, dw_type_stride, dbias, bias_type_mask, n_types == 1 ? int64_t{0} : dbias_type_stride);
    return check_launch("ihg_node_linear_bwd_weight");
}

int32_t ihg_node_linear_typed_supported(int32_t dim, int64_t ld_x, int64_t ld_out) {
    return ((split_arith_enabled() && (dim == 128 || dim == 256)) || dim == kNarrowDim || dim == 64) && ld_x >= dim && ld_out >= dim && ld_x % 4 == 0 && ld_out % 4 == 0 ? 1 : 0;
}

int ihg_node_linear_fwd_typed(const float* const* x_rows, int64_t ld_x, const float* w, int64_t ld_w, int64_t w_type_stride, const float* bias,
                              int32_t bias_type_mask, int64_t bias_type_stride, const int64_t* type_begin, float* out, int64_t ld_out, void* workspace,
                              int64_t workspace_bytes, int32_t dim, ihg_stream_t stream) {
    if (int rc = node_linear_common_check("ihg_node_linear_fwd_typed", dim, ld_x, ld_out, ld_w, type_begin, workspace, workspace_bytes)) return rc;
    if (type_begin[3] == type_begin[0]) return IHG_OK;
    if (x_rows == nullptr || w == nullptr || out == nullptr) return fail(IHG_ERR_INVALID, "ihg_node_linear_fwd_typed: null pointer");
    for (int t = 0; t < 3; ++t)
        if (type_begin[t + 1] > type_begin[t] && (x_rows[t] == nullptr || !aligned16(x_rows[t]))) return fail(IHG_ERR_INVALID, "ihg_node_linear_fwd_typed: rows of type %d null or not 16-byte aligned", t);
    if (!ihg_node_linear_typed_supported(dim, ld_x, ld_out) || workspace == nullptr || !aligned16(workspace))
        return fail(IHG_ERR_INVALID, "ihg_node_linear_fwd_typed: not available for this shape (ihg_node_linear_typed_supported)");
    const TypedRows in = typed_rows(x_rows, type_begin, ld_x);
    if (int rc = launch_row_gemm(dim, nullptr, ld_x, w, ld_w, w_type_stride, 0, bias, bias_type_mask, bias_type_stride, type_begin, out, ld_out, static_cast<float*>(workspace),
                                 static_cast<hipStream_t>(stream), &in, nullptr))
        return rc;
    return check_launch("ihg_node_linear_fwd_typed");
}

int ihg_node_linear_bwd_weight_typed(const float* dout, int64_t ld_dout, const float* const* x_rows, int64_t ld_x, const int64_t* type_begin,
                                     float* dw, int64_t ld_dw, int64_t dw_type_stride, float* dbias, int32_t bias_type_mask, int64_t dbias_type_stride,
                                     const float* w, int64_t ld_w, float* const* dx_rows, int64_t ld_dx, int32_t zero_row_before_mask,
                                     void* workspace, int64_t workspace_bytes, int32_t dim, ihg_stream_t stream) {
    if (int rc = node_linear_common_check("ihg_node_linear_bwd_weight_typed", dim, ld_dout, ld_x, ld_dw, type_begin, workspace, workspace_bytes)) return rc;
    if (dout == nullptr || x_rows == nullptr || dw == nullptr) return fail(IHG_ERR_INVALID, "ihg_node_linear_bwd_weight_typed: null pointer");
    if (dx_rows != nullptr && (w == nullptr || ld_w < dim || ld_dx < dim)) return fail(IHG_ERR_INVALID, "ihg_node_linear_bwd_weight_typed: dx needs w and row strides >= dim");
    for (int t = 0; t < 3; ++t) {
        if (type_begin[t + 1] == type_begin[t]) continue;
        if (x_rows[t] == nullptr || !aligned16(x_rows[t]) || (dx_rows != nullptr && (dx_rows[t] == nullptr || !aligned16(dx_rows[t]))))
            return fail(IHG_ERR_INVALID, "ihg_node_linear_bwd_weight_typed: rows of type %d null or not 16-byte aligned", t);
    }
    const bool narrow = narrow_linear_ok(dim, ld_dout, ld_x) && aligned16(dout) && (dx_rows == nullptr || (aligned16(w) && ld_w % 4 == 0 && ld_dx % 4 == 0));
    if (!ihg_node_linear_typed_supported(dim, ld_x, dx_rows != nullptr ? ld_dx : ld_x) || (!narrow && !split_dense_weight_ok(dim, dout, ld_dout, x_rows[0], ld_x)) || workspace == nullptr ||
        !aligned16(workspace))
        return fail(IHG_ERR_INVALID, "ihg_node_linear_bwd_weight_typed: not available for this shape (ihg_node_linear_typed_supported)");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int n_types = dw_type_stride == 0 ? 1 : 3;
    float* slabs = static_cast<float*>(workspace) + 3LL * dim * dim;
    float* bias_slabs = slabs + 3LL * kDenseSlabs * dim * dim;
    void* planes = static_cast<float*>(workspace) + 3LL * dim * dim + 3LL * kDenseSlabs * (static_cast<int64_t>(dim) * dim + dim);
    const TypedRows xin = typed_rows(x_rows, type_begin, ld_x);
    TypedRowsOut dxo = typed_rows_out(nullptr);
    if (dx_rows != nullptr) {
        dxo = typed_rows_out(dx_rows, type_begin, ld_dx);
        // gradient tables with a padding row in front of a type's rows (row 0 of the [U + 1, d] / [I + 1, d] embedding tables, Models/EmbeddingLayers.py:33-35): zeroed here
        for (int t = 0; t < 3; ++t)
            if ((zero_row_before_mask >> t) & 1) launch_zero_floats(dx_rows[t] - ld_dx, dim, s);
    }
    const bool fused_dx = dx_rows != nullptr && dim == 128;
    int n_slabs = 0;
    if (narrow) {
        n_slabs = launch_dense_weight_narrow(dim, dout, ld_dout, xin, ld_x, type_begin, n_types, slabs, bias_slabs, w, ld_w, dw_type_stride, dx_rows != nullptr ? &dxo : nullptr, ld_dx, 0, static_cast<float*>(workspace), s);
    } else {
        if (dx_rows != nullptr && !fused_dx) {
            if (int rc = launch_row_gemm(dim, dout, ld_dout, w, ld_w, dw_type_stride, 1, nullptr, 0, 0, type_begin, nullptr, ld_dx, static_cast<float*>(workspace), s, nullptr, &dxo)) return rc;
        }
        n_slabs = launch_dense_weight_split(dim, dout, ld_dout, xin, ld_x, type_begin, n_types, slabs, bias_slabs, w, ld_w, dw_type_stride, fused_dx ? &dxo : nullptr, ld_dx, planes, s, 0);
    }
    const int total = dim * dim * n_types + dim;
    hipLaunchKernelGGL(dense_slab_reduce_kernel, dim3((total + kWave - 1) / kWave), dim3(kBlockThreads), 0, s, slabs, bias_slabs,
                       n_slabs, n_types, dim, dw, ld_dw, dw_type_stride, dbias, bias_type_mask, n_types == 1 ? int64_t{0} : dbias_type_stride);
    return check_launch("ihg_node_linear_bwd_weight_typed");
}

int ihg_compose_first_order_fwd(const float* a, int64_t ld_a, const float* c, const float* w, int64_t ld_w, const float* b, float* w_eff,
                                int64_t ld_w_eff, float* b_eff, int32_t dim, ihg_stream_t stream) {
    if (dim <= 0 || dim > 1024 || ld_a < 3LL * dim || ld_w < dim || ld_w_eff < 3LL * dim) return fail(IHG_ERR_INVALID, "ihg_compose_first_order_fwd: bad size");
    if (a == nullptr || w == nullptr || b == nullptr || w_eff == nullptr || b_eff == nullptr) return fail(IHG_ERR_INVALID, "ihg_compose_first_order_fwd: null pointer");
    const int64_t total = 3LL * dim * dim + 3LL * dim;
    hipLaunchKernelGGL(compose_fwd_kernel, dim3(static_cast<int>((total + kBlockThreads - 1) / kBlockThreads)), dim3(kBlockThreads), 0,
                       static_cast<hipStream_t>(stream), a, ld_a, c, w, ld_w, b, w_eff, ld_w_eff, b_eff, dim);
    return check_launch("ihg_compose_first_order_fwd");
}

int ihg_compose_first_order_bwd(const float* a, int64_t ld_a, const float* w, int64_t ld_w, const float* b, const float* dw_eff, int64_t ld_dw_eff,
                                const float* db_eff, float* da, int64_t ld_da, float* dc, float* dw, int64_t ld_dw, float* db, int32_t dim,
                                ihg_stream_t stream) {
    if (dim <= 0 || dim > 1024 || ld_a < 3LL * dim || ld_w < dim || ld_dw_eff < 3LL * dim || ld_da < 3LL * dim || ld_dw < dim)
        return fail(IHG_ERR_INVALID, "ihg_compose_first_order_bwd: bad size");
    if (a == nullptr || w == nullptr || b == nullptr || dw_eff == nullptr || db_eff == nullptr || da == nullptr || dw == nullptr || db == nullptr)
        return fail(IHG_ERR_INVALID, "ihg_compose_first_order_bwd: null pointer");
    const int64_t waves = 3LL * dim * dim;                 // one wave per dA element; the thread-per-element part needs far fewer blocks
    hipLaunchKernelGGL(compose_bwd_kernel, dim3(static_cast<int>(std::min<int64_t>((waves + kWavesPerBlock - 1) / kWavesPerBlock, 256 * 16))), dim3(kBlockThreads), 0,
                       static_cast<hipStream_t>(stream), a, ld_a, w, ld_w, b, dw_eff, ld_dw_eff, db_eff, da, ld_da, dc, dw, ld_dw, db, dim);
    return check_launch("ihg_compose_first_order_bwd");
}
}  // extern "C"
