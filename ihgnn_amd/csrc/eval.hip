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

// ------------------------------------------------------------------------------------------------
// Arithmetic: fp32 through TWO fp16 terms (split_arith.hip has the scheme and its error analysis): every item row and every mixed row is multiplied by the power of
// two that brings its largest magnitude to [2^13, 2^14), x = hi + lo with hi = fp16(x), lo = fp16(x - hi), and a product is hi lo + lo hi + hi hi on
// v_mfma_f32_32x32x16_f16, fp32 accumulation; the score is the accumulator times the two inverse scales (exact: powers of two) plus the bias.  Unlike the training
// kernels nothing is split inside the hot loop: the item rows are split ONCE per call into MFMA-fragment order (score_prepare_kernel: 1 KB per (tile, k-step, plane), a
// fully coalesced wave load), the mixed rows of a pair block once into LDS - the loop is loads, LDS reads and MFMAs.  A workgroup owns PB x 32 pairs (PB = 2 where their
// planes fit LDS: widths up to 624) and its eight waves stream disjoint runs of item tiles: every item row is fetched once per 32 PB pairs.
// ------------------------------------------------------------------------------------------------
typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef _Float16 v2h __attribute__((ext_vector_type(2)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));

constexpr int kEvalThreads = 512;

__device__ __forceinline__ float eval_scale_up_for(float m, float& inverse) {      // 2^(13 - floor(log2 m)) and its inverse (exponent clamped; split_arith.hip scale_up_for)
    int e = static_cast<int>((__float_as_uint(m) >> 23) & 0xffu);
    e = e < 27 ? 27 : (e > 227 ? 227 : e);
    inverse = __uint_as_float(static_cast<unsigned>(e - 13) << 23);
    return __uint_as_float(static_cast<unsigned>(267 - e) << 23);
}
__device__ __forceinline__ void eval_split8(const float (&x)[8], v4u& hi, v4u& lo) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const v2h h = v2h{static_cast<_Float16>(x[2 * i]), static_cast<_Float16>(x[2 * i + 1])};
        const v2h l = v2h{static_cast<_Float16>(x[2 * i] - static_cast<float>(h[0])), static_cast<_Float16>(x[2 * i + 1] - static_cast<float>(h[1]))};
        hi[i] = __builtin_bit_cast(unsigned, h);
        lo[i] = __builtin_bit_cast(unsigned, l);
    }
}

// Item rows -> frag[tile][ks][plane][lane][8 x fp16] (tile = 32 items, ks = 16 columns; lane = 32 (column half of the k-step) + item % 32: the A fragment of
// v_mfma_f32_32x32x16_f16), aux[item] = {inverse scale, bias}.  One wave per item row; columns past `dim` are zero.
__global__ __launch_bounds__(kBlockThreads) void score_prepare_kernel(const float* __restrict__ feat, int64_t ld, int dim, int ksteps, int64_t item_row0, int64_t n_items,
                                                                      const float* __restrict__ bias, v4u* __restrict__ frag, float2* __restrict__ aux) {
    const int lane = threadIdx.x & 63;
    for (int64_t item = global_wave_id(); item < n_items; item += global_wave_count()) {
        const float* row = feat + (item_row0 + item) * ld;
        float m = 0.f;
        for (int c = lane; c < dim; c += kWave) m = fmaxf(m, fabsf(row[c]));
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
        float inv;
        const float sc = eval_scale_up_for(m, inv);
        if (lane == 0) aux[item] = make_float2(inv, bias[item]);
        const int64_t tile = item >> 5;
        for (int chunk = lane; chunk < 2 * ksteps; chunk += kWave) {      // eight consecutive columns
            float x[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) x[i] = 8 * chunk + i < dim ? row[8 * chunk + i] * sc : 0.f;
            v4u hi, lo;
            eval_split8(x, hi, lo);
            const int64_t base = ((tile * ksteps + (chunk >> 1)) * 2) * kWave + 32 * (chunk & 1) + (item & 31);
            frag[base] = hi;
            frag[base + kWave] = lo;
        }
    }
}

// grid (pair blocks of 32 PB pairs, item slices); 512 threads.  partial[(pair * n_lists + list) * kTopMax + p]
template <int PB>
__global__ __launch_bounds__(kEvalThreads) void score_topk_kernel(
    const float* __restrict__ feat, int64_t ld, int dim, int ksteps, const v4u* __restrict__ frag, const float2* __restrict__ aux, int64_t n_items,
    const int64_t* __restrict__ users, const int64_t* __restrict__ queries, int64_t query_row0, float lam, int64_t n_pairs,
    float* __restrict__ part_val, int32_t* __restrict__ part_idx) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];    // mixed planes: [32 PB][hi: 2 dp bytes | lo: 2 dp bytes | 16 bytes of padding], then pinv[32 PB]
    const int dp = 16 * ksteps;
    const int row_bytes = 4 * dp + 16;                                      // (the padding spreads the pairs' equal chunks over the banks)
    float* pinv = reinterpret_cast<float*>(smem + static_cast<size_t>(32 * PB) * row_bytes);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t pair0 = static_cast<int64_t>(blockIdx.x) * (32 * PB);
    // mixed rows m = lam F[q] + (1 - lam) F[u] (PredictionLayers.py:35) of this block's pairs, one wave per row: largest magnitude, scale, two fp16 planes;
    // pairs past the end repeat the last one
    for (int r = wave; r < 32 * PB; r += kEvalThreads / kWave) {
        int64_t pr = pair0 + r;
        pr = pr < n_pairs ? pr : n_pairs - 1;
        const float* qrow = feat + (queries[pr] + query_row0) * ld;
        const float* urow = feat + users[pr] * ld;
        float m = 0.f;
        for (int c = lane; c < dim; c += kWave) m = fmaxf(m, fabsf(lam * qrow[c] + (1.f - lam) * urow[c]));
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
        float inv;
        const float sc = eval_scale_up_for(m, inv);
        if (lane == 0) pinv[r] = inv;
        for (int chunk = lane; chunk < 2 * ksteps; chunk += kWave) {
            float x[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int c = 8 * chunk + i;
                x[i] = c < dim ? (lam * qrow[c] + (1.f - lam) * urow[c]) * sc : 0.f;
            }
            v4u hi, lo;
            eval_split8(x, hi, lo);
            *reinterpret_cast<v4u*>(smem + static_cast<size_t>(r) * row_bytes + 16 * chunk) = hi;
            *reinterpret_cast<v4u*>(smem + static_cast<size_t>(r) * row_bytes + 2 * dp + 16 * chunk) = lo;
        }
    }
    __syncthreads();

    // this wave's run of 32-item tiles inside this block's item slice
    constexpr int WAVES = kEvalThreads / kWave;
    const int64_t n_tiles = (n_items + 31) / 32;
    const int64_t n_runs = static_cast<int64_t>(gridDim.y) * WAVES;
    const int64_t run = static_cast<int64_t>(blockIdx.y) * WAVES + wave;
    const int64_t tile_begin = n_tiles * run / n_runs, tile_end = n_tiles * (run + 1) / n_runs;
    const int n31 = lane & 31, half = lane >> 5;
    float my_pinv[PB];
    const unsigned char* brow[PB];
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) {
        my_pinv[pb] = pinv[32 * pb + n31];
        brow[pb] = smem + static_cast<size_t>(32 * pb + n31) * row_bytes + 16 * half;
    }
    TopList top[PB];
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) top[pb].init();

    constexpr int PF = 4;                                                   // k-steps of item fragments in flight
    for (int64_t tile = tile_begin; tile < tile_end; ++tile) {
        const v4u* af = frag + (tile * ksteps * 2) * kWave + lane;
        v16f acc[PB];
#pragma unroll
        for (int pb = 0; pb < PB; ++pb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[pb][r] = 0.f;
        v4u a[PF][2];
#pragma unroll
        for (int x = 0; x < PF; ++x) {
            const int ks = x < ksteps ? x : ksteps - 1;
            a[x][0] = af[(2 * ks) * kWave];
            a[x][1] = af[(2 * ks + 1) * kWave];
        }
        for (int k0 = 0; k0 < ksteps; k0 += PF) {
#pragma unroll
            for (int x = 0; x < PF; ++x) {
                const int ks = k0 + x;
                if (ks < ksteps) {
                    const v8h ahi = __builtin_bit_cast(v8h, a[x][0]), alo = __builtin_bit_cast(v8h, a[x][1]);
                    const int nk = ks + PF < ksteps ? ks + PF : ksteps - 1;     // (past the end: the last k-step again, read and dropped)
                    a[x][0] = af[(2 * nk) * kWave];
                    a[x][1] = af[(2 * nk + 1) * kWave];
#pragma unroll
                    for (int pb = 0; pb < PB; ++pb) {
                        const v8h bhi = *reinterpret_cast<const v8h*>(brow[pb] + 32 * ks), blo = *reinterpret_cast<const v8h*>(brow[pb] + 2 * dp + 32 * ks);
                        acc[pb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi, blo, acc[pb], 0, 0, 0);
                        acc[pb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo, bhi, acc[pb], 0, 0, 0);
                        acc[pb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi, bhi, acc[pb], 0, 0, 0);
                    }
                }
            }
        }
        // this lane: pairs 32 pb + n31, items tile * 32 + acc_row(r, lane)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            const int64_t i0 = tile * 32 + 8 * r4 + 4 * half;
            float2 ax[4];
#pragma unroll
            for (int x = 0; x < 4; ++x) ax[x] = i0 + x < n_items ? aux[i0 + x] : make_float2(0.f, 0.f);
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                const int i = static_cast<int>(i0 + x);
#pragma unroll
                for (int pb = 0; pb < PB; ++pb) {
                    const float s = acc[pb][4 * r4 + x] * (ax[x].x * my_pinv[pb]) + ax[x].y;
                    if (i0 + x < n_items && ranks_before(s, i, top[pb].val[kTopMax - 1], top[pb].idx[kTopMax - 1])) top[pb].insert(s, i);
                }
            }
        }
    }
    const int64_t n_lists = n_runs * 2;
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) {
        const int64_t pair = pair0 + 32 * pb + n31;
        if (pair < n_pairs) {
            const int64_t base = (pair * n_lists + run * 2 + half) * kTopMax;
#pragma unroll
            for (int p = 0; p < kTopMax; ++p) {
                part_val[base + p] = top[pb].val[p];
                part_idx[base + p] = top[pb].idx[p];
            }
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

constexpr int kEvalWavesPerBlock = kEvalThreads / kWave;

inline int eval_ksteps(int dim) { return (dim + 15) / 16; }
inline size_t eval_lds_bytes(int dim, int pb) { return static_cast<size_t>(32 * pb) * (4 * 16 * eval_ksteps(dim) + 16) + static_cast<size_t>(32 * pb) * sizeof(float); }
inline int eval_pair_tiles(int dim) { return eval_lds_bytes(dim, 2) <= 160 * 1024 ? 2 : 1; }      // pair tiles of 32 per workgroup

inline int eval_slices(int64_t n_pairs, int64_t n_items, int dim) {
    const int64_t blocks = (n_pairs + 32 * eval_pair_tiles(dim) - 1) / (32 * eval_pair_tiles(dim));
    const int64_t tiles = (n_items + 31) / 32;
    int64_t slices = (512 + blocks - 1) / blocks;                           // ~2 workgroups per CU in all (one resident per CU: LDS)
    const int64_t most = std::max<int64_t>(1, tiles / (kEvalWavesPerBlock * 4));    // at least 4 tiles per wave
    slices = std::max<int64_t>(1, std::min<int64_t>({slices, most, 64}));
    return static_cast<int>(slices);
}

inline int64_t eval_frag_bytes(int64_t n_items, int dim) { return ((n_items + 31) / 32) * eval_ksteps(dim) * 2 * kWave * 16; }
inline int64_t eval_aux_bytes(int64_t n_items) { return ((n_items + 31) / 32) * 32 * 8; }

}  // namespace

extern "C" {

// the widest feature row whose 32-pair block (two fp16 planes + the row pad + the pair scales) fits the 160 KB of LDS: 1264 (79 k-steps of 16)
int32_t ihg_score_topk_max_dim(void) {
    int dim = 16;
    while (eval_lds_bytes(dim + 16, 1) <= 160 * 1024) dim += 16;
    return dim;
}

int64_t ihg_score_topk_workspace_bytes(int64_t n_pairs, int64_t n_items, int32_t dim) {
    if (n_pairs <= 0 || n_items <= 0 || dim <= 0) return 0;
    const int64_t n_lists = static_cast<int64_t>(eval_slices(n_pairs, n_items, dim)) * kEvalWavesPerBlock * 2;
    return eval_frag_bytes(n_items, dim) + eval_aux_bytes(n_items) + n_pairs * n_lists * kTopMax * 8;
}

int ihg_score_topk(const float* features, int64_t ld, int32_t dim, int64_t query_row0, int64_t item_row0, int64_t n_items, const float* item_bias,
                   const int64_t* users, const int64_t* queries, float lambda_muq, int64_t n_pairs, int32_t k, float* top_scores,
                   int32_t* top_items, void* workspace, int64_t workspace_bytes, ihg_stream_t stream) {
    if (n_pairs < 0 || n_items <= 0 || dim <= 0 || k <= 0 || k > kTopMax || ld < dim) return fail(IHG_ERR_INVALID, "ihg_score_topk: bad size (k <= %d)", kTopMax);
    if (n_pairs == 0) return IHG_OK;
    if (features == nullptr || item_bias == nullptr || users == nullptr || queries == nullptr || top_scores == nullptr || top_items == nullptr)
        return fail(IHG_ERR_INVALID, "ihg_score_topk: null pointer");
    if (n_items > INT_MAX - 64) return fail(IHG_ERR_INVALID, "ihg_score_topk: item ids are int32");
    if (eval_lds_bytes(dim, 1) > 160 * 1024) return fail(IHG_ERR_INVALID, "ihg_score_topk: feature width %d does not fit the LDS pair block (widest: %d)", dim, ihg_score_topk_max_dim());
    if (workspace == nullptr || !aligned16(workspace) || workspace_bytes < ihg_score_topk_workspace_bytes(n_pairs, n_items, dim))
        return fail(IHG_ERR_WORKSPACE, "ihg_score_topk: workspace too small");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int ksteps = eval_ksteps(dim), pb = eval_pair_tiles(dim);
    const int slices = eval_slices(n_pairs, n_items, dim);
    const int64_t n_lists = static_cast<int64_t>(slices) * kEvalWavesPerBlock * 2;
    v4u* frag = static_cast<v4u*>(workspace);
    float2* aux = reinterpret_cast<float2*>(static_cast<unsigned char*>(workspace) + eval_frag_bytes(n_items, dim));
    float* part_val = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(aux) + eval_aux_bytes(n_items));
    int32_t* part_idx = reinterpret_cast<int32_t*>(part_val + n_pairs * n_lists * kTopMax);
    hipLaunchKernelGGL(score_prepare_kernel, dim3(grid_for_waves(n_items)), dim3(kBlockThreads), 0, s, features, ld, dim, ksteps, item_row0, n_items, item_bias, frag, aux);
    static bool attr_set[64] = {};                           // per device ordinal: the opt-in to > 64 KB of LDS belongs to the device's code object
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess) (void)hipGetLastError();
    if (device < 0 || device >= 64 || !attr_set[device]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(score_topk_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) (void)hipGetLastError();
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(score_topk_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) (void)hipGetLastError();
        if (device >= 0 && device < 64) attr_set[device] = true;
    }
    const int64_t blocks = (n_pairs + 32 * pb - 1) / (32 * pb);
    const size_t lds = eval_lds_bytes(dim, pb);
    if (pb == 2)
        hipLaunchKernelGGL(score_topk_kernel<2>, dim3(static_cast<unsigned>(blocks), slices), dim3(kEvalThreads), lds, s, features, ld, dim, ksteps, frag, aux, n_items, users, queries,
                           query_row0, lambda_muq, n_pairs, part_val, part_idx);
    else
        hipLaunchKernelGGL(score_topk_kernel<1>, dim3(static_cast<unsigned>(blocks), slices), dim3(kEvalThreads), lds, s, features, ld, dim, ksteps, frag, aux, n_items, users, queries,
                           query_row0, lambda_muq, n_pairs, part_val, part_idx);
    hipLaunchKernelGGL(merge_topk_kernel, dim3(grid_for_waves(n_pairs)), dim3(kBlockThreads), 0, s, part_val, part_idx, n_pairs,
                       static_cast<int>(n_lists * kTopMax), k, top_scores, top_items);
    return check_launch("ihg_score_topk");
}

}  // extern "C"
