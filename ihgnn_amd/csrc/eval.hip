// eval.hip - evaluation scoring: HEM scores of many (user, query) pairs against EVERY item with a running top-k per
// pair, on the matrix cores, without materialising the [pairs, items] score matrix.
//
// Reference: per search log, RawGnn.forward(u * ones(I), q * ones(I), None) (Models/RawGnn.py:124-137) broadcasts one user row and
// one query row to [I, D], HemPredictionLayer scores them (Models/PredictionLayers.py:35-43) and Metrics.calculate_on_all_items
// (Helpers/Metrics.py:60-61) sorts all I scores and keeps ten.  Here a workgroup owns a block of 32 pairs - their mixed rows
// m = lam * F[q] + (1 - lam) * F[u] sit in LDS for the whole kernel - and its four waves stream disjoint runs of item rows as the
// A operand of v_mfma_f32_32x32x2_f32 (exact fp32): the 32 x 32 result tile has the PAIR on the lane (column) and 16 items in the
// lane's accumulator registers, so every lane keeps the running top-k of "its" pair over the items it has seen in registers, with
// no cross-lane traffic in the loop.  The partial lists (2 lane halves x 4 waves x item slices per pair) are merged by a second,
// tiny kernel.  Order: higher score first, equal scores by lower item index (= a stable descending sort; the reference's
// torch.sort is unstable on ties).
#include "common.hpp"

#include <cfloat>
#include <climits>

namespace {

constexpr int kTopMax = 10;                 // list length kept per lane (Metrics.py:60: top 10)
constexpr int kPairsPerBlock = 32;
constexpr int kEvalWaves = 4;

__device__ __forceinline__ bool ranks_before(float s, int i, float v, int j) { return s > v || (s == v && i < j); }

struct TopList {
    float val[kTopMax];
    int idx[kTopMax];
    __device__ void init() {
#pragma unroll
        for (int p = 0; p < kTopMax; ++p) {
            val[p] = -FLT_MAX;
            idx[p] = INT_MAX;
        }
    }
    // sorted insert (best first); the caller has checked that (s, i) ranks before the last slot
    __device__ void insert(float s, int i) {
        val[kTopMax - 1] = s;
        idx[kTopMax - 1] = i;
#pragma unroll
        for (int p = kTopMax - 1; p > 0; --p) {
            const bool up = ranks_before(val[p], idx[p], val[p - 1], idx[p - 1]);
            const float v0 = val[p - 1], v1 = val[p];
            const int i0 = idx[p - 1], i1 = idx[p];
            val[p - 1] = up ? v1 : v0;
            val[p] = up ? v0 : v1;
            idx[p - 1] = up ? i1 : i0;
            idx[p] = up ? i0 : i1;
        }
    }
};

// grid (pair blocks, item slices); 256 threads.  partial[(pair * n_lists + list) * kTopMax + p]
__global__ __launch_bounds__(kBlockThreads, 2) void score_topk_kernel(
    const float* __restrict__ feat, int64_t ld, int dim, int64_t item_row0, int64_t n_items, const float* __restrict__ bias,
    const int64_t* __restrict__ users, const int64_t* __restrict__ queries, int64_t query_row0, float lam, int64_t n_pairs,
    float* __restrict__ part_val, int32_t* __restrict__ part_idx) {
    extern __shared__ __attribute__((aligned(16))) float mixed[];           // [32][dim8 + kRowPad]
    const int dim8 = (dim + 7) & ~7;
    const int stride = dim8 + kRowPad;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t pair0 = static_cast<int64_t>(blockIdx.x) * kPairsPerBlock;
    // mixed rows of this block's pairs (PredictionLayers.py:35), zero-padded to a multiple of 8 columns; pairs past the end repeat the last one
    for (int idx = tid; idx < kPairsPerBlock * (dim8 / 4); idx += kBlockThreads) {
        const int r = idx / (dim8 / 4), c4 = idx % (dim8 / 4);
        int64_t pr = pair0 + r;
        pr = pr < n_pairs ? pr : n_pairs - 1;
        float4 m = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c4 * 4 < dim) {
            const float4 q = *reinterpret_cast<const float4*>(feat + (queries[pr] + query_row0) * ld + c4 * 4);
            const float4 u = *reinterpret_cast<const float4*>(feat + users[pr] * ld + c4 * 4);
            m = make_float4(lam * q.x + (1 - lam) * u.x, lam * q.y + (1 - lam) * u.y, lam * q.z + (1 - lam) * u.z, lam * q.w + (1 - lam) * u.w);
        }
        *reinterpret_cast<float4*>(&mixed[r * stride + c4 * 4]) = m;
    }
    __syncthreads();

    // this wave's run of 32-item tiles inside this block's item slice
    const int64_t n_tiles = (n_items + 31) / 32;
    const int64_t n_runs = static_cast<int64_t>(gridDim.y) * kEvalWaves;
    const int64_t run = static_cast<int64_t>(blockIdx.y) * kEvalWaves + wave;
    const int64_t tile_begin = n_tiles * run / n_runs, tile_end = n_tiles * (run + 1) / n_runs;
    const int r31 = lane & 31, half = lane >> 5;
    const float* mrow = mixed + r31 * stride + 4 * half;
    const int t_steps = dim8 / 8;
    TopList top;
    top.init();

    for (int64_t tile = tile_begin; tile < tile_end; ++tile) {
        int64_t item = tile * 32 + r31;
        item = item < n_items ? item : n_items - 1;                          // rows past the end re-read the last item; masked below
        const float* arow = feat + (item_row0 + item) * ld + 4 * half;
        v16f acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        // item rows straight from L2 / Infinity Cache into registers, four k-groups ahead of the MFMAs that use them
        constexpr int PF = 4;
        v4f a[PF];
#pragma unroll
        for (int x = 0; x < PF; ++x) a[x] = 8 * x + 4 * half < dim ? *reinterpret_cast<const v4f*>(arow + 8 * x) : v4f{0.f, 0.f, 0.f, 0.f};
        for (int t0 = 0; t0 < t_steps; t0 += PF) {
#pragma unroll
            for (int x = 0; x < PF; ++x) {
                const int t = t0 + x;
                if (t < t_steps) {
                    const v4f av = a[x];
                    if (8 * (t + PF) + 4 * half < dim) a[x] = *reinterpret_cast<const v4f*>(arow + 8 * (t + PF));      // dim % 8 == 4: no read past the row
                    const v4f bv = *reinterpret_cast<const v4f*>(mrow + 8 * t);
#pragma unroll
                    for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], bv[s], acc, 0, 0, 0);
                }
            }
        }
        // this lane: pair r31, items tile * 32 + acc_row(r, lane)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            const int64_t i0 = tile * 32 + 8 * r4 + 4 * half;
            float bv[4];
#pragma unroll
            for (int x = 0; x < 4; ++x) bv[x] = i0 + x < n_items ? bias[i0 + x] : 0.f;
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                const float s = acc[4 * r4 + x] + bv[x];
                const int i = static_cast<int>(i0 + x);
                if (i0 + x < n_items && ranks_before(s, i, top.val[kTopMax - 1], top.idx[kTopMax - 1])) top.insert(s, i);
            }
        }
    }
    const int64_t pair = pair0 + r31;
    if (pair < n_pairs) {
        const int64_t n_lists = n_runs * 2;
        const int64_t base = (pair * n_lists + run * 2 + half) * kTopMax;
#pragma unroll
        for (int p = 0; p < kTopMax; ++p) {
            part_val[base + p] = top.val[p];
            part_idx[base + p] = top.idx[p];
        }
    }
}

// one wave per pair: the best k of its n_lists * kTopMax candidates, best first
__global__ __launch_bounds__(kBlockThreads) void merge_topk_kernel(const float* __restrict__ part_val, const int32_t* __restrict__ part_idx,
                                                                   int64_t n_pairs, int n_cand, int k, float* __restrict__ out_val,
                                                                   int32_t* __restrict__ out_idx) {
    const int lane = threadIdx.x & 63;
    for (int64_t pair = global_wave_id(); pair < n_pairs; pair += global_wave_count()) {
    const float* pv = part_val + pair * n_cand;
    const int32_t* pi = part_idx + pair * n_cand;
    float last_v = FLT_MAX;
    int last_i = -1;
    for (int round = 0; round < k; ++round) {
        // best candidate that ranks strictly after the previous winner (candidates are distinct items, so "after" = not yet taken)
        float bv = -FLT_MAX;
        int bi = INT_MAX;
        for (int c = lane; c < n_cand; c += kWave) {
            const float v = pv[c];
            const int i = pi[c];
            if (ranks_before(last_v, last_i, v, i) && ranks_before(v, i, bv, bi)) {
                bv = v;
                bi = i;
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float ov = __shfl_xor(bv, off);
            const int oi = __shfl_xor(bi, off);
            if (ranks_before(ov, oi, bv, bi)) {
                bv = ov;
                bi = oi;
            }
        }
        if (lane == 0) {
            out_val[pair * k + round] = bv;
            out_idx[pair * k + round] = bi == INT_MAX ? -1 : bi;
        }
        last_v = bv;
        last_i = bi;
    }
    }
}

inline int eval_slices(int64_t n_pairs, int64_t n_items) {
    const int64_t blocks = (n_pairs + kPairsPerBlock - 1) / kPairsPerBlock;
    const int64_t tiles = (n_items + 31) / 32;
    int64_t slices = (1024 + blocks - 1) / blocks;                          // ~4 workgroups per CU in all
    const int64_t most = std::max<int64_t>(1, tiles / (kEvalWaves * 4));    // at least 4 tiles per wave
    slices = std::max<int64_t>(1, std::min<int64_t>({slices, most, 64}));
    return static_cast<int>(slices);
}

}  // namespace

extern "C" {

int64_t ihg_score_topk_workspace_bytes(int64_t n_pairs, int64_t n_items) {
    if (n_pairs <= 0 || n_items <= 0) return 0;
    const int64_t n_lists = static_cast<int64_t>(eval_slices(n_pairs, n_items)) * kEvalWaves * 2;
    return n_pairs * n_lists * kTopMax * 8;
}

int ihg_score_topk(const float* features, int64_t ld, int32_t dim, int64_t query_row0, int64_t item_row0, int64_t n_items, const float* item_bias,
                   const int64_t* users, const int64_t* queries, float lambda_muq, int64_t n_pairs, int32_t k, float* top_scores,
                   int32_t* top_items, void* workspace, int64_t workspace_bytes, ihg_stream_t stream) {
    if (n_pairs < 0 || n_items <= 0 || dim <= 0 || k <= 0 || k > kTopMax || ld < dim) return fail(IHG_ERR_INVALID, "ihg_score_topk: bad size (k <= %d)", kTopMax);
    if (n_pairs == 0) return IHG_OK;
    if (features == nullptr || item_bias == nullptr || users == nullptr || queries == nullptr || top_scores == nullptr || top_items == nullptr)
        return fail(IHG_ERR_INVALID, "ihg_score_topk: null pointer");
    if (dim % 4 != 0 || ld % 4 != 0 || !aligned16(features)) return fail(IHG_ERR_INVALID, "ihg_score_topk: rows must be 16-byte aligned with dim %% 4 == 0");
    if (n_items > INT_MAX - 64) return fail(IHG_ERR_INVALID, "ihg_score_topk: item ids are int32");
    const int dim8 = (dim + 7) & ~7;
    const size_t lds = static_cast<size_t>(kPairsPerBlock) * (dim8 + kRowPad) * sizeof(float);
    if (lds > 160 * 1024) return fail(IHG_ERR_INVALID, "ihg_score_topk: feature width %d does not fit the LDS pair block", dim);
    if (workspace == nullptr || !aligned16(workspace) || workspace_bytes < ihg_score_topk_workspace_bytes(n_pairs, n_items))
        return fail(IHG_ERR_WORKSPACE, "ihg_score_topk: workspace too small");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int slices = eval_slices(n_pairs, n_items);
    const int64_t n_lists = static_cast<int64_t>(slices) * kEvalWaves * 2;
    float* part_val = static_cast<float*>(workspace);
    int32_t* part_idx = reinterpret_cast<int32_t*>(part_val + n_pairs * n_lists * kTopMax);
    static bool attr_set[64] = {};                           // per device ordinal: the opt-in to > 64 KB of LDS belongs to the device's code object
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess) (void)hipGetLastError();
    if (device < 0 || device >= 64 || !attr_set[device]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(score_topk_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            (void)hipGetLastError();
        if (device >= 0 && device < 64) attr_set[device] = true;
    }
    const int64_t blocks = (n_pairs + kPairsPerBlock - 1) / kPairsPerBlock;
    hipLaunchKernelGGL(score_topk_kernel, dim3(static_cast<unsigned>(blocks), slices), dim3(kBlockThreads), lds, s, features, ld, dim, item_row0, n_items,
                       item_bias, users, queries, query_row0, lambda_muq, n_pairs, part_val, part_idx);
    hipLaunchKernelGGL(merge_topk_kernel, dim3(grid_for_waves(n_pairs)), dim3(kBlockThreads), 0, s, part_val, part_idx, n_pairs,
                       static_cast<int>(n_lists * kTopMax), k, top_scores, top_items);
    return check_launch("ihg_score_topk");
}

}  // extern "C"
