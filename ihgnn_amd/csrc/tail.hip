// tail.hip - the batch tail: HEM scores straight from the layer outputs, BCE-with-logits, per-row gradients and the
// deterministic sort-free scatter of duplicate rows.
#include "common.hpp"

namespace {

// ================================================================================================
// Batch tail (HEM scoring head over the concatenated layer outputs, Models/PredictionLayers.py:21-44 after
// Models/RawGnn.py:122-131): one wave per batch row, lanes over the columns of every layer output; no [N, D] concat.
//   fwd  score[r] = sum_l sum_c X_l[item_r][c] * (lam * X_l[query_r][c] + (1 - lam) * X_l[user_r][c]) + bias[item_r]
//   bwd  rowgrad[0B + r] = ds (1 - lam) X[item_r]   (w.r.t. the user row),   rowgrad[1B + r] = ds lam X[item_r]   (query row),
//        rowgrad[2B + r] = ds (lam X[query_r] + (1 - lam) X[user_r])   (item row); [3B, L1*d] dense and conflict-free -
//        the caller adds duplicate rows with one deterministic scatter.
// ================================================================================================
// per layer: the matrix (virtual base, common.hpp TypedRows) its user / query / item rows are read from - three equal pointers for an ordinary [N, d] layer
// output; layer 0 may be the embedding tables themselves (rows 1.. of the user table, the queries' bag means, rows 1.. of the item table): a batch's user,
// query and item rows are typed by construction, so the head needs no type lookup
struct LayerPtrs {
    const float* x[8][3];
    int64_t ld[8];
};

// rows_upper (optional): the rows of the layers ABOVE layer 0 where they are numbered differently from layer 0's - a layout that leaves the isolated nodes out of its own
// numbering (IncidenceLayout.node_map) while layer 0 is read from the embedding tables by public id; a negative entry is an isolated node: its rows above layer 0 are zero
__global__ __launch_bounds__(kBlockThreads) void hem_score_fwd_kernel(LayerPtrs layers, int n_layers, int dim,
                                                                      const int64_t* __restrict__ rows, const int64_t* __restrict__ items,
                                                                      const float* __restrict__ bias, float lam, float* __restrict__ scores,
                                                                      int64_t batch, const int64_t* __restrict__ rows_upper = nullptr) {
    const int lane = threadIdx.x & 63;
    for (int64_t r = global_wave_id(); r < batch; r += global_wave_count()) {
        float acc = 0.f;
        for (int l = 0; l < n_layers; ++l) {
            const int64_t* rr = (l > 0 && rows_upper != nullptr) ? rows_upper : rows;
            const int64_t u = rr[r], q = rr[batch + r], it = rr[2 * batch + r];
            if (it < 0 || (u < 0 && q < 0)) continue;       // an isolated item (or both other rows zero): the layer adds nothing
            const float* xu = layers.x[l][0] + (u < 0 ? 0 : u) * layers.ld[l];
            const float* xq = layers.x[l][1] + (q < 0 ? 0 : q) * layers.ld[l];
            const float* xi = layers.x[l][2] + it * layers.ld[l];
            const float wu = u < 0 ? 0.f : 1.f - lam, wq = q < 0 ? 0.f : lam;
            for (int c = lane; c < dim; c += kWave) {
                const float m = wq * xq[c] + wu * xu[c];
                acc += xi[c] * m;
            }
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o);
        if (lane == 0) scores[r] = acc + bias[items[r]];
    }
}

__global__ __launch_bounds__(kBlockThreads) void hem_score_bwd_kernel(LayerPtrs layers, int n_layers, int dim,
                                                                      const int64_t* __restrict__ rows, const float* __restrict__ dscores,
                                                                      float grad_scale, float lam, float* __restrict__ rowgrad, int64_t width,
                                                                      int64_t batch, const float* __restrict__ grad_scale_device = nullptr,
                                                                      const int64_t* __restrict__ rows_upper = nullptr) {
    const int lane = threadIdx.x & 63;
    if (grad_scale_device != nullptr) grad_scale *= *grad_scale_device;      // the upstream gradient of the loss, still on the device (no host read, no extra launch)
    for (int64_t r = global_wave_id(); r < batch; r += global_wave_count()) {
        const float ds = dscores[r] * grad_scale;
        if (lane == 0 && width > static_cast<int64_t>(n_layers) * dim) {       // optional extra column: d bias, carried by the item row
            const int64_t col = static_cast<int64_t>(n_layers) * dim;
            rowgrad[r * width + col] = 0.f;
            rowgrad[(batch + r) * width + col] = 0.f;
            rowgrad[(2 * batch + r) * width + col] = ds;
        }
        for (int l = 0; l < n_layers; ++l) {
            const int64_t* rr = (l > 0 && rows_upper != nullptr) ? rows_upper : rows;
            const int64_t u = rr[r], q = rr[batch + r], it = rr[2 * batch + r];
            // (an isolated node's rows above layer 0 are zero constants: the row gradients written for them are never added anywhere - ihg_batch_rows_add / _put skip negative rows)
            const float* pu = layers.x[l][0] + (u < 0 ? 0 : u) * layers.ld[l];
            const float* pq = layers.x[l][1] + (q < 0 ? 0 : q) * layers.ld[l];
            const float* pi = layers.x[l][2] + (it < 0 ? 0 : it) * layers.ld[l];
            for (int c = lane; c < dim; c += kWave) {
                const float xu = u < 0 ? 0.f : pu[c], xq = q < 0 ? 0.f : pq[c], xi = it < 0 ? 0.f : pi[c];
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
constexpr int kScatterMax = 16384;                      // ids of one launch in 64 KiB of LDS: any single-GPU batch (3 x 1,100 rows) ...
constexpr int kScatterMaxWide = 32768;                  // ... and the wide instance (128 of gfx950's 160 KiB, one workgroup per CU) for the UNION of the ranks' batches in the
                                                        // data-parallel cotangent exchange (8 ranks x 3,300 rows = 26,400; ihgnn_amd/distributed.py: CotangentSync)

// Destination addressing: column c of batch row k goes to dense[(c / block_width) * block_stride + row * ld_dense + c % block_width]
// (block_width = width, block_stride = 0 is a plain matrix; block_width = d, block_stride = N*d lands layer l's columns in its own
// contiguous [N, d] matrix); an optional last column goes to tail[row - tail_row_offset] (the bias gradient).
// COMBINE: nothing is scattered; the sum over a destination's batch rows replaces the FIRST of them in rowgrad itself (a row
// is read by the wave of its destination's first occurrence only, which is also its only writer) and leader[k] says whether
// batch row k is such a first occurrence.  batch_rows_add_kernel then adds leader rows wherever they are needed.
template <bool COMBINE, int CAP = kScatterMax>
__global__ __launch_bounds__(kBlockThreads) void batch_scatter_kernel(float* __restrict__ rowgrad, int64_t ld_rowgrad, int width,
                                                                      const int64_t* __restrict__ rows, int n, float* __restrict__ dense,
                                                                      int64_t ld_dense, int block_width, int64_t block_stride,
                                                                      float* __restrict__ tail, int64_t tail_row_offset, int64_t tail_rows,
                                                                      int32_t* __restrict__ leader, int block_rows) {
    // block_rows: rows of different consecutive blocks of this many batch rows never share a destination (the user / query / item
    // thirds of a batch address disjoint node ranges), so a wave only scans its own block; n = one block is the general case.
    // A WORKGROUP belongs to one block (blockIdx.y) and keeps only that block's ids in LDS (CAP >= block_rows): a third of the fill and of the LDS of the whole list - the
    // union of eight ranks' batches (3 x 8,800 rows) runs on the 64 KiB instance, two workgroups per CU.  A NEGATIVE id is a row that takes no part (its gradient was
    // already summed into another row - a rank's own duplicates, combined before the exchange): never a leader, never matched.
    __shared__ int32_t key[CAP];
    const int lo = static_cast<int>(blockIdx.y) * block_rows;
    const int hi = lo + block_rows < n ? lo + block_rows : n;
    for (int k = lo + threadIdx.x; k < hi; k += kBlockThreads) key[k - lo] = static_cast<int32_t>(rows[k]);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = static_cast<int>(blockIdx.x) * kWavesPerBlock + (threadIdx.x >> 6);
    for (int64_t k = lo + wave; k < hi; k += static_cast<int64_t>(gridDim.x) * kWavesPerBlock) {
        const int32_t mine = key[k - lo];
        if (mine < 0) {
            if (COMBINE && lane == 0) leader[k] = 0;
            continue;
        }
        bool follower = false;
        for (int base = lo; base < k; base += kWave) {
            const int j = base + lane;
            if (__ballot(j < k && key[j - lo] == mine) != 0ull) { follower = true; break; }
        }
        if (COMBINE && lane == 0) leader[k] = follower ? 0 : 1;
        if (follower) continue;
        for (int c0 = 0; c0 < width; c0 += 4 * kWave) {
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
            for (int base = static_cast<int>(k) & ~(kWave - 1); base < hi; base += kWave) {
                const int j = base + lane;
                unsigned long long mask = __ballot(j >= k && j < hi && key[j - lo] == mine);
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
                if (COMBINE) {
                    rowgrad[k * ld_rowgrad + c] = acc[q];
                } else if (tail != nullptr && c == width - 1) {
                    const int64_t tr = static_cast<int64_t>(mine) - tail_row_offset;
                    if (tr >= 0 && tr < tail_rows) tail[tr] += acc[q];
                } else {
                    dense[(c / block_width) * block_stride + static_cast<int64_t>(mine) * ld_dense + c % block_width] += acc[q];
                }
            }
        }
    }
}

// dense[rows[k], 0:width] += src[k, 0:width] for the leader rows of a combined batch (no two leaders share a destination);
// with `tail`, the single column src[k, 0] goes to tail[rows[k] - tail_row_offset] instead.
// dense: virtual bases per node type (TypedRowsOut; a plain matrix: three equal pointers), begin1 / begin2 = first query / item row; ASSIGN: = instead of +=
template <bool ASSIGN>
__global__ __launch_bounds__(kBlockThreads) void batch_rows_add_kernel(const float* __restrict__ src, int64_t ld_src, int width,
                                                                       const int64_t* __restrict__ rows, const int32_t* __restrict__ leader, int n,
                                                                       TypedRowsOut dense, int64_t begin1, int64_t begin2, int64_t ld_dense, float* __restrict__ tail,
                                                                       int64_t tail_row_offset, int64_t tail_rows) {
    const int lane = threadIdx.x & 63;
    for (int64_t k = global_wave_id(); k < n; k += global_wave_count()) {
        if (leader[k] == 0) continue;
        const int64_t row = rows[k];
        if (row < 0) continue;                               // (an isolated node under a compact layout: its rows above layer 0 are constants)
        if (tail != nullptr) {
            const int64_t tr = row - tail_row_offset;
            if (lane == 0 && tr >= 0 && tr < tail_rows) tail[tr] += src[k * ld_src];
            continue;
        }
        float* dst = typed_base(dense, row >= begin2 ? 2 : (row >= begin1 ? 1 : 0)) + row * ld_dense;
        for (int c = lane; c < width; c += kWave) dst[c] = ASSIGN ? src[k * ld_src + c] : dst[c] + src[k * ld_src + c];
    }
}

// mask[rows[k]] = value for the k < n listed rows (the row mask of the last layer's sparse cotangent: set before its pull, cleared after)
__global__ __launch_bounds__(kBlockThreads) void mark_rows_kernel(const int64_t* __restrict__ rows64, const int32_t* __restrict__ rows32, int64_t n, uint8_t* __restrict__ mask,
                                                                  uint8_t value) {
    for (int64_t k = static_cast<int64_t>(blockIdx.x) * kBlockThreads + threadIdx.x; k < n; k += static_cast<int64_t>(gridDim.x) * kBlockThreads)
        mask[rows64 != nullptr ? rows64[k] : static_cast<int64_t>(rows32[k])] = value;
}

// rows[0 .. 3 b) = users | queries + query_row0 | items + item_row0 (the global node rows of a batch, Models/RawGnn.py:128-131: torch.cat of three index vectors + two adds)
__global__ __launch_bounds__(kBlockThreads) void batch_node_rows_kernel(const int64_t* __restrict__ users, const int64_t* __restrict__ queries, const int64_t* __restrict__ items,
                                                                        int64_t b, int64_t query_row0, int64_t item_row0, int64_t* __restrict__ rows, int32_t* __restrict__ rows32) {
    for (int64_t k = static_cast<int64_t>(blockIdx.x) * kBlockThreads + threadIdx.x; k < 3 * b; k += static_cast<int64_t>(gridDim.x) * kBlockThreads) {
        const int64_t r = k < b ? users[k] : (k < 2 * b ? queries[k - b] + query_row0 : items[k - 2 * b] + item_row0);
        rows[k] = r;
        if (rows32 != nullptr) rows32[k] = static_cast<int32_t>(r);
    }
}

// base[rows[k], 0 .. dim) = 0 for the k < n listed rows (one wave per row)
__global__ __launch_bounds__(kBlockThreads) void zero_rows_kernel(float* __restrict__ base, int64_t ld, int dim, const int64_t* __restrict__ rows, int64_t n) {
    const int lane = threadIdx.x & 63;
    for (int64_t k = global_wave_id(); k < n; k += global_wave_count())
        for (int c = lane; c < dim; c += kWave) base[rows[k] * ld + c] = 0.f;
}

// ------------------------------------------------------------------------------------------------
// Adam over a list of tensors in one launch (pointer table passed by value): a pure stream, 7 passes of 4 bytes per parameter.
// ------------------------------------------------------------------------------------------------
constexpr int kAdamMaxTensors = 24;
constexpr int kAdamChunk = 4096;                            // elements per workgroup trip

struct AdamTable {
    float* param[kAdamMaxTensors];
    const float* grad[kAdamMaxTensors];
    float* exp_avg[kAdamMaxTensors];
    float* exp_avg_sq[kAdamMaxTensors];
    int64_t chunk_begin[kAdamMaxTensors + 1];               // prefix of ceil(n / kAdamChunk)
    int64_t count[kAdamMaxTensors];
    int n_tensors;
};

__device__ __forceinline__ void adam_update(float& p, float g, float& m, float& v, float beta1, float beta2, float eps, float weight_decay,
                                            float step_size, float bias2_sqrt) {
    g = weight_decay != 0.f ? g + weight_decay * p : g;
    m = m + (g - m) * (1.f - beta1);                        // torch's lerp form
    v = beta2 * v + (1.f - beta2) * g * g;
    const float denom = sqrtf(v) / bias2_sqrt + eps;
    p = p - step_size * (m / denom);
}

// `scalars` != nullptr: {step_size, bias2_sqrt} are read from device memory at launch time (a captured launch replays with the step's own values)
__global__ __launch_bounds__(kBlockThreads) void adam_kernel(AdamTable tab, float beta1, float beta2, float eps, float weight_decay, float step_size,
                                                             float bias2_sqrt, const float* __restrict__ scalars) {
    if (scalars != nullptr) {
        step_size = scalars[0];
        bias2_sqrt = scalars[1];
    }
    const int64_t total_chunks = tab.chunk_begin[tab.n_tensors];
    for (int64_t chunk = blockIdx.x; chunk < total_chunks; chunk += gridDim.x) {
        int t = 0;
        while (chunk >= tab.chunk_begin[t + 1]) ++t;
        const int64_t base = (chunk - tab.chunk_begin[t]) * kAdamChunk;
        const int64_t n = tab.count[t];
        float* __restrict__ p = tab.param[t];
        const float* __restrict__ g = tab.grad[t];
        float* __restrict__ m = tab.exp_avg[t];
        float* __restrict__ v = tab.exp_avg_sq[t];
        const bool vec = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(v)) & 15) == 0;
        if (vec && base + kAdamChunk <= n) {
#pragma unroll
            for (int k = 0; k < kAdamChunk / (4 * kBlockThreads); ++k) {
                const int64_t i = base + (threadIdx.x + k * kBlockThreads) * 4;
                v4f pv = *reinterpret_cast<const v4f*>(p + i), mv = *reinterpret_cast<const v4f*>(m + i), vv = *reinterpret_cast<const v4f*>(v + i);
                const v4f gv = *reinterpret_cast<const v4f*>(g + i);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    float pc = pv[c], mc = mv[c], vc = vv[c];
                    adam_update(pc, gv[c], mc, vc, beta1, beta2, eps, weight_decay, step_size, bias2_sqrt);
                    pv[c] = pc; mv[c] = mc; vv[c] = vc;
                }
                *reinterpret_cast<v4f*>(p + i) = pv;
                *reinterpret_cast<v4f*>(m + i) = mv;
                *reinterpret_cast<v4f*>(v + i) = vv;
            }
        } else {
            for (int64_t i = base + threadIdx.x; i < base + kAdamChunk && i < n; i += kBlockThreads)
                adam_update(p[i], g[i], m[i], v[i], beta1, beta2, eps, weight_decay, step_size, bias2_sqrt);
        }
    }
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// Negative sampling on the device: `random.sample(range(n_items), k)` per positive (Dataset.py:107-109) - k DISTINCT
// items, uniform over the catalogue, the positive itself not excluded.  Counter-based generator: draw (row, slot, attempt) is a
// pure function of (seed, batch counter), so a run is reproducible whatever the launch geometry; a repeated item inside a
// sample is rejected and redrawn (k <= 16, so the check is a register scan).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t mix64(uint64_t x) {                  // splitmix64 finaliser
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

constexpr int kSampleMax = 16;

__global__ __launch_bounds__(kBlockThreads) void sample_negatives_kernel(uint64_t seed, uint64_t counter, int64_t n_rows, int64_t n_items, int k,
                                                                         int64_t* __restrict__ out) {
    for (int64_t row = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; row < n_rows; row += static_cast<int64_t>(gridDim.x) * blockDim.x) {
        int64_t picked[kSampleMax];
        const uint64_t key = mix64(seed ^ mix64(counter)) ^ mix64(static_cast<uint64_t>(row) * 0xD1B54A32D192ED03ull);
        uint64_t attempt = 0;
#pragma unroll 1
        for (int slot = 0; slot < k; ++slot) {
            int64_t item;
            bool fresh;
            do {
                const uint64_t r = mix64(key + (attempt++) * 0x2545F4914F6CDD1Dull);
                item = static_cast<int64_t>(__umul64hi(r, static_cast<uint64_t>(n_items)));      // uniform on [0, n_items) up to 2^-64 n_items
                fresh = true;
                for (int j = 0; j < slot; ++j) fresh = fresh && picked[j] != item;
            } while (!fresh);
            picked[slot] = item;
            out[row * k + slot] = item;
        }
    }
}

extern "C" {

int ihg_sample_negatives(uint64_t seed, uint64_t counter, int64_t n_rows, int64_t n_items, int32_t k, int64_t* out, ihg_stream_t stream) {
    if (n_rows < 0 || n_items <= 0 || k <= 0 || k > kSampleMax || k > n_items)
        return fail(IHG_ERR_INVALID, "ihg_sample_negatives: need 1 <= k <= min(%d, n_items)", kSampleMax);
    if (n_rows == 0) return IHG_OK;
    if (out == nullptr) return fail(IHG_ERR_INVALID, "ihg_sample_negatives: null pointer");
    const int grid = static_cast<int>(std::min<int64_t>((n_rows + kBlockThreads - 1) / kBlockThreads, kMaxBlocks));
    hipLaunchKernelGGL(sample_negatives_kernel, dim3(grid), dim3(kBlockThreads), 0, static_cast<hipStream_t>(stream), seed, counter, n_rows, n_items, k, out);
    return check_launch("ihg_sample_negatives");
}


// layers[l] for l >= 1 (and for l == 0 without layer0_rows) are plain [N, ld] matrices; with layer0_rows the three node types of layer 0 start at their own addresses
static LayerPtrs layer_ptrs(const float* const* layers, int32_t n_layers, int64_t ld, const float* const* layer0_rows, int64_t ld0, const int64_t* type_begin) {
    LayerPtrs lp{};
    for (int l = 0; l < n_layers; ++l) {
        for (int t = 0; t < 3; ++t) lp.x[l][t] = layers[l];
        lp.ld[l] = ld;
    }
    if (layer0_rows != nullptr) {
        const TypedRows t0 = typed_rows(layer0_rows, type_begin, ld0);
        for (int t = 0; t < 3; ++t) lp.x[0][t] = t0.p[t];
        lp.ld[0] = ld0;
    }
    return lp;
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
    const LayerPtrs lp = layer_ptrs(layers, n_layers, ld, nullptr, 0, nullptr);
    hipLaunchKernelGGL(hem_score_fwd_kernel, dim3(grid_for_waves(batch)), dim3(kBlockThreads), 0, static_cast<hipStream_t>(stream), lp, n_layers,
                       dim, rows, items, bias, lambda_muq, scores, batch);
    return check_launch("ihg_hem_score_fwd");
}

int ihg_hem_score_bwd(const float* const* layers, int32_t n_layers, int64_t ld, int32_t dim, const int64_t* rows, const float* dscores,
                      float grad_scale, float lambda_muq, float* rowgrad, int64_t ld_rowgrad, int64_t batch, ihg_stream_t stream) {
    if (int rc = hem_common_check("ihg_hem_score_bwd", layers, n_layers, ld, dim, rows, batch)) return rc;
    if (batch == 0) return IHG_OK;
    if (dscores == nullptr || rowgrad == nullptr || ld_rowgrad < static_cast<int64_t>(n_layers) * dim) return fail(IHG_ERR_INVALID, "ihg_hem_score_bwd: null pointer or short row stride");
    const LayerPtrs lp = layer_ptrs(layers, n_layers, ld, nullptr, 0, nullptr);
    hipLaunchKernelGGL(hem_score_bwd_kernel, dim3(grid_for_waves(batch)), dim3(kBlockThreads), 0, static_cast<hipStream_t>(stream), lp, n_layers,
                       dim, rows, dscores, grad_scale, lambda_muq, rowgrad, ld_rowgrad, batch);
    return check_launch("ihg_hem_score_bwd");
}

int ihg_hem_score_fwd_typed0(const float* const* layers, int32_t n_layers, int64_t ld, int32_t dim, const float* const* layer0_rows, int64_t ld0,
                             const int64_t* type_begin, const int64_t* rows, const int64_t* rows_upper, const int64_t* items, const float* bias, float lambda_muq,
                             float* scores, int64_t batch, ihg_stream_t stream) {
    if (layer0_rows != nullptr && (type_begin == nullptr || ld0 < dim)) return fail(IHG_ERR_INVALID, "ihg_hem_score_fwd_typed0: type ranges / row stride of layer 0 missing");
    if (n_layers < 1 || n_layers > 8 || ld < dim || dim <= 0 || batch < 0 || layers == nullptr) return fail(IHG_ERR_INVALID, "ihg_hem_score_fwd_typed0: bad size");
    if (batch == 0) return IHG_OK;
    if (rows == nullptr || items == nullptr || bias == nullptr || scores == nullptr) return fail(IHG_ERR_INVALID, "ihg_hem_score_fwd_typed0: null pointer");
    const LayerPtrs lp = layer_ptrs(layers, n_layers, ld, layer0_rows, ld0, type_begin);
    hipLaunchKernelGGL(hem_score_fwd_kernel, dim3(grid_for_waves(batch)), dim3(kBlockThreads), 0, static_cast<hipStream_t>(stream), lp, n_layers,
                       dim, rows, items, bias, lambda_muq, scores, batch, rows_upper);
    return check_launch("ihg_hem_score_fwd_typed0");
}

int ihg_hem_score_bwd_typed0(const float* const* layers, int32_t n_layers, int64_t ld, int32_t dim, const float* const* layer0_rows, int64_t ld0,
                             const int64_t* type_begin, const int64_t* rows, const int64_t* rows_upper, const float* dscores, const float* grad_scale_device,
                             float grad_scale, float lambda_muq, float* rowgrad, int64_t ld_rowgrad, int64_t batch, ihg_stream_t stream) {
    if (layer0_rows != nullptr && (type_begin == nullptr || ld0 < dim)) return fail(IHG_ERR_INVALID, "ihg_hem_score_bwd_typed0: type ranges / row stride of layer 0 missing");
    if (n_layers < 1 || n_layers > 8 || ld < dim || dim <= 0 || batch < 0 || layers == nullptr) return fail(IHG_ERR_INVALID, "ihg_hem_score_bwd_typed0: bad size");
    if (batch == 0) return IHG_OK;
    if (rows == nullptr || dscores == nullptr || rowgrad == nullptr || ld_rowgrad < static_cast<int64_t>(n_layers) * dim)
        return fail(IHG_ERR_INVALID, "ihg_hem_score_bwd_typed0: null pointer or short row stride");
    const LayerPtrs lp = layer_ptrs(layers, n_layers, ld, layer0_rows, ld0, type_begin);
    hipLaunchKernelGGL(hem_score_bwd_kernel, dim3(grid_for_waves(batch)), dim3(kBlockThreads), 0, static_cast<hipStream_t>(stream), lp, n_layers,
                       dim, rows, dscores, grad_scale, lambda_muq, rowgrad, ld_rowgrad, batch, grad_scale_device, rows_upper);
    return check_launch("ihg_hem_score_bwd_typed0");
}

int ihg_bce_with_logits(const float* scores, const float* labels, int64_t n, float* loss, float* dscores, ihg_stream_t stream) {
    if (n <= 0 || n > (1 << 24) || scores == nullptr || labels == nullptr || loss == nullptr || dscores == nullptr)
        return fail(IHG_ERR_INVALID, "ihg_bce_with_logits: bad argument");
    hipLaunchKernelGGL(bce_with_logits_kernel, dim3(1), dim3(kScatterThreads), 0, static_cast<hipStream_t>(stream), scores, labels, static_cast<int>(n), loss, dscores);
    return check_launch("ihg_bce_with_logits");
}


int64_t ihg_batch_scatter_workspace_bytes(int64_t n_rows) {
    return (n_rows < 0 || n_rows > kScatterMaxWide) ? -1 : 0;
}

int32_t ihg_batch_scatter_max_rows(void) { return kScatterMaxWide; }

int ihg_batch_scatter_add(const float* rowgrad, int64_t ld_rowgrad, int32_t width, const int64_t* rows, int64_t n_rows, float* dense,
                          int64_t ld_dense, int32_t block_width, int64_t block_stride, float* tail, int64_t tail_row_offset,
                          int64_t tail_rows, ihg_stream_t stream) {
    if (n_rows < 0 || n_rows > kScatterMaxWide) return fail(IHG_ERR_INVALID, "ihg_batch_scatter_add: 0..%d rows supported, got %lld", kScatterMaxWide, (long long)n_rows);
    if (width <= 0 || ld_rowgrad < width || block_width <= 0 || ld_dense < block_width) return fail(IHG_ERR_INVALID, "ihg_batch_scatter_add: bad width / stride");
    if (n_rows == 0) return IHG_OK;
    if (rowgrad == nullptr || rows == nullptr || dense == nullptr) return fail(IHG_ERR_INVALID, "ihg_batch_scatter_add: null pointer");
    const dim3 grid(grid_for_waves(n_rows), 1);              // one block: any two rows may share a destination
    if (n_rows > kScatterMax)
        hipLaunchKernelGGL((batch_scatter_kernel<false, kScatterMaxWide>), grid, dim3(kBlockThreads), 0, static_cast<hipStream_t>(stream),
                           const_cast<float*>(rowgrad), ld_rowgrad, width, rows, static_cast<int>(n_rows), dense, ld_dense, block_width, block_stride, tail,
                           tail_row_offset, tail_rows, static_cast<int32_t*>(nullptr), static_cast<int>(n_rows));
    else
        hipLaunchKernelGGL((batch_scatter_kernel<false>), grid, dim3(kBlockThreads), 0, static_cast<hipStream_t>(stream),
                           const_cast<float*>(rowgrad), ld_rowgrad, width, rows, static_cast<int>(n_rows), dense, ld_dense, block_width, block_stride, tail,
                           tail_row_offset, tail_rows, static_cast<int32_t*>(nullptr), static_cast<int>(n_rows));
    return check_launch("ihg_batch_scatter_add");
}

int ihg_batch_combine(float* rowgrad, int64_t ld_rowgrad, int32_t width, const int64_t* rows, int64_t n_rows, int64_t disjoint_block_rows,
                      int32_t* leader, ihg_stream_t stream) {
    if (n_rows < 0 || n_rows > kScatterMaxWide) return fail(IHG_ERR_INVALID, "ihg_batch_combine: 0..%d rows supported, got %lld", kScatterMaxWide, (long long)n_rows);
    if (width <= 0 || ld_rowgrad < width) return fail(IHG_ERR_INVALID, "ihg_batch_combine: bad width / stride");
    if (n_rows == 0) return IHG_OK;
    if (rowgrad == nullptr || rows == nullptr || leader == nullptr) return fail(IHG_ERR_INVALID, "ihg_batch_combine: null pointer");
    if (disjoint_block_rows <= 0 || disjoint_block_rows > n_rows) disjoint_block_rows = n_rows;
    const int n_blocks = static_cast<int>((n_rows + disjoint_block_rows - 1) / disjoint_block_rows);
    const dim3 grid(grid_for_waves(disjoint_block_rows), n_blocks);      // a workgroup works inside one block and holds that block's ids only
    if (disjoint_block_rows > kScatterMax)
        hipLaunchKernelGGL((batch_scatter_kernel<true, kScatterMaxWide>), grid, dim3(kBlockThreads), 0, static_cast<hipStream_t>(stream), rowgrad,
                           ld_rowgrad, width, rows, static_cast<int>(n_rows), static_cast<float*>(nullptr), int64_t{0}, 1, int64_t{0},
                           static_cast<float*>(nullptr), int64_t{0}, int64_t{0}, leader, static_cast<int>(disjoint_block_rows));
    else
        hipLaunchKernelGGL((batch_scatter_kernel<true>), grid, dim3(kBlockThreads), 0, static_cast<hipStream_t>(stream), rowgrad,
                           ld_rowgrad, width, rows, static_cast<int>(n_rows), static_cast<float*>(nullptr), int64_t{0}, 1, int64_t{0},
                           static_cast<float*>(nullptr), int64_t{0}, int64_t{0}, leader, static_cast<int>(disjoint_block_rows));
    return check_launch("ihg_batch_combine");
}

int ihg_batch_rows_add(const float* src, int64_t ld_src, int32_t width, const int64_t* rows, const int32_t* leader, int64_t n_rows, float* dense,
                       int64_t ld_dense, float* tail, int64_t tail_row_offset, int64_t tail_rows, ihg_stream_t stream) {
    if (n_rows < 0 || width <= 0 || ld_src < width) return fail(IHG_ERR_INVALID, "ihg_batch_rows_add: bad size");
    if (n_rows == 0) return IHG_OK;
    if (src == nullptr || rows == nullptr || leader == nullptr || (dense == nullptr && tail == nullptr) || (dense != nullptr && ld_dense < width))
        return fail(IHG_ERR_INVALID, "ihg_batch_rows_add: null pointer or short row stride");
    hipLaunchKernelGGL(batch_rows_add_kernel<false>, dim3(grid_for_waves(n_rows)), dim3(kBlockThreads), 0, static_cast<hipStream_t>(stream), src, ld_src, width,
                       rows, leader, static_cast<int>(n_rows), typed_rows_out(dense), int64_t{0}, int64_t{0}, ld_dense, tail, tail_row_offset, tail_rows);
    return check_launch("ihg_batch_rows_add");
}

int ihg_batch_rows_put(const float* src, int64_t ld_src, int32_t width, const int64_t* rows, const int32_t* leader, int64_t n_rows, float* const* dense_rows,
                       int64_t ld_dense, const int64_t* type_begin, int32_t assign, ihg_stream_t stream) {
    if (n_rows < 0 || width <= 0 || ld_src < width || ld_dense < width) return fail(IHG_ERR_INVALID, "ihg_batch_rows_put: bad size");
    if (n_rows == 0) return IHG_OK;
    if (src == nullptr || rows == nullptr || leader == nullptr || dense_rows == nullptr || type_begin == nullptr) return fail(IHG_ERR_INVALID, "ihg_batch_rows_put: null pointer");
    const TypedRowsOut dense = typed_rows_out(dense_rows, type_begin, ld_dense);
    if (assign)
        hipLaunchKernelGGL(batch_rows_add_kernel<true>, dim3(grid_for_waves(n_rows)), dim3(kBlockThreads), 0, static_cast<hipStream_t>(stream), src, ld_src, width, rows, leader,
                           static_cast<int>(n_rows), dense, type_begin[1], type_begin[2], ld_dense, static_cast<float*>(nullptr), int64_t{0}, int64_t{0});
    else
        hipLaunchKernelGGL(batch_rows_add_kernel<false>, dim3(grid_for_waves(n_rows)), dim3(kBlockThreads), 0, static_cast<hipStream_t>(stream), src, ld_src, width, rows, leader,
                           static_cast<int>(n_rows), dense, type_begin[1], type_begin[2], ld_dense, static_cast<float*>(nullptr), int64_t{0}, int64_t{0});
    return check_launch("ihg_batch_rows_put");
}

int ihg_zero_floats(float* p, int64_t n, ihg_stream_t stream) {
    if (n < 0 || (n > 0 && p == nullptr)) return fail(IHG_ERR_INVALID, "ihg_zero_floats: bad argument");
    launch_zero_floats(p, n, static_cast<hipStream_t>(stream));
    return check_launch("ihg_zero_floats");
}

int ihg_mark_rows(const int64_t* rows64, const int32_t* rows32, int64_t n, uint8_t* mask, int32_t value, ihg_stream_t stream) {
    if (n < 0 || (n > 0 && (mask == nullptr || (rows64 == nullptr) == (rows32 == nullptr)))) return fail(IHG_ERR_INVALID, "ihg_mark_rows: one of rows64 / rows32 and a mask");
    if (n == 0) return IHG_OK;
    const int grid = static_cast<int>(std::min<int64_t>((n + kBlockThreads - 1) / kBlockThreads, kMaxBlocks));
    hipLaunchKernelGGL(mark_rows_kernel, dim3(grid), dim3(kBlockThreads), 0, static_cast<hipStream_t>(stream), rows64, rows32, n, mask, static_cast<uint8_t>(value));
    return check_launch("ihg_mark_rows");
}

int ihg_zero_rows(float* base, int64_t ld, int32_t dim, const int64_t* rows, int64_t n, ihg_stream_t stream) {
    if (n < 0 || dim <= 0 || ld < dim || (n > 0 && (base == nullptr || rows == nullptr))) return fail(IHG_ERR_INVALID, "ihg_zero_rows: bad argument");
    if (n == 0) return IHG_OK;
    hipLaunchKernelGGL(zero_rows_kernel, dim3(grid_for_waves(n)), dim3(kBlockThreads), 0, static_cast<hipStream_t>(stream), base, ld, dim, rows, n);
    return check_launch("ihg_zero_rows");
}

int ihg_batch_node_rows(const int64_t* users, const int64_t* queries, const int64_t* items, int64_t batch, int64_t query_row0, int64_t item_row0, int64_t* rows,
                        int32_t* rows32, ihg_stream_t stream) {
    if (batch < 0 || (batch > 0 && (users == nullptr || queries == nullptr || items == nullptr || rows == nullptr))) return fail(IHG_ERR_INVALID, "ihg_batch_node_rows: bad argument");
    if (batch == 0) return IHG_OK;
    const int grid = static_cast<int>(std::min<int64_t>((3 * batch + kBlockThreads - 1) / kBlockThreads, kMaxBlocks));
    hipLaunchKernelGGL(batch_node_rows_kernel, dim3(grid), dim3(kBlockThreads), 0, static_cast<hipStream_t>(stream), users, queries, items, batch, query_row0, item_row0, rows, rows32);
    return check_launch("ihg_batch_node_rows");
}

static int adam_launch(const char* what, const ihg_adam_tensor* tensors, int32_t n_tensors, float beta1, float beta2, float eps, float weight_decay, float step_size,
                       float bias2_sqrt, const float* scalars, ihg_stream_t stream) {
    for (int t = 0; t < n_tensors;) {                        // one launch per kAdamMaxTensors NON-EMPTY tensors; t is the only cursor
        AdamTable tab{};
        int64_t chunks = 0;
        int used = 0;
        for (; t < n_tensors && used < kAdamMaxTensors; ++t) {
            const ihg_adam_tensor& a = tensors[t];
            if (a.count < 0 || (a.count > 0 && (a.param == nullptr || a.grad == nullptr || a.exp_avg == nullptr || a.exp_avg_sq == nullptr)))
                return fail(IHG_ERR_INVALID, "%s: tensor %d has a null pointer or a negative count", what, t);
            if (a.count == 0) continue;
            tab.param[used] = a.param; tab.grad[used] = a.grad; tab.exp_avg[used] = a.exp_avg; tab.exp_avg_sq[used] = a.exp_avg_sq;
            tab.count[used] = a.count;
            tab.chunk_begin[used] = chunks;
            chunks += (a.count + kAdamChunk - 1) / kAdamChunk;
            ++used;
        }
        tab.chunk_begin[used] = chunks;
        tab.n_tensors = used;
        if (used == 0) continue;
        const int grid = static_cast<int>(std::min<int64_t>(chunks, 256 * 16));
        hipLaunchKernelGGL(adam_kernel, dim3(grid), dim3(kBlockThreads), 0, static_cast<hipStream_t>(stream), tab, beta1, beta2, eps, weight_decay,
                           step_size, bias2_sqrt, scalars);
    }
    return check_launch(what);
}

int ihg_adam_step(const ihg_adam_tensor* tensors, int32_t n_tensors, float lr, float beta1, float beta2, float eps, float weight_decay,
                  int64_t step, ihg_stream_t stream) {
    if (n_tensors < 0 || (n_tensors > 0 && tensors == nullptr) || step < 1) return fail(IHG_ERR_INVALID, "ihg_adam_step: bad argument");
    const double bias1 = 1.0 - std::pow(static_cast<double>(beta1), static_cast<double>(step));
    const double bias2 = 1.0 - std::pow(static_cast<double>(beta2), static_cast<double>(step));
    return adam_launch("ihg_adam_step", tensors, n_tensors, beta1, beta2, eps, weight_decay, static_cast<float>(static_cast<double>(lr) / bias1),
                       static_cast<float>(std::sqrt(bias2)), nullptr, stream);
}

int ihg_adam_step_device_scalars(const ihg_adam_tensor* tensors, int32_t n_tensors, float beta1, float beta2, float eps, float weight_decay,
                                 const float* step_scalars, ihg_stream_t stream) {
    if (n_tensors < 0 || (n_tensors > 0 && tensors == nullptr) || step_scalars == nullptr) return fail(IHG_ERR_INVALID, "ihg_adam_step_device_scalars: bad argument");
    return adam_launch("ihg_adam_step_device_scalars", tensors, n_tensors, beta1, beta2, eps, weight_decay, 0.f, 1.f, step_scalars, stream);
}
}  // extern "C"
