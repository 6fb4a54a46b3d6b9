// interact.hip - the interactive (order 2 / 3) node -> hyperedge step and its backward on the matrix cores (exact fp32 MFMA),
// plus the one-thread-per-output kernels that take every other shape.
#include "common.hpp"
#include "narrow.hpp"
#include "split.hpp"

namespace {

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

// Position swizzle of an unpadded LDS row image read with ds_read_b128 as an MFMA A operand: 16-byte chunk c of row r sits at
// position c ^ tile_swizzle(r).  The 16 lanes one LDS cycle serves hold 16 different rows; the swizzle spreads them over all
// 64 banks (256-byte rows: r % 16; 128-byte rows alternate bank halves by themselves, so (r / 2) % 8).
template <int D> __device__ __forceinline__ int tile_swizzle(int r) { return D == 64 ? (r & 15) : ((r >> 1) & 7); }

// One 1-KiB piece of an LDS image filled straight from global memory (no staging registers, no ds_write): lane l's 16 bytes
// land at lds_piece + 16*l, the source address is per lane.
__device__ __forceinline__ void lds_dma16(const float* src, float* lds_piece) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)lds_piece, 16, 0, 0);
}

__device__ __forceinline__ void edge_row_store(float* p, v4f v) {
    store_stream4(p, v);
}

template <int D, int NBLK>
__global__ __launch_bounds__(kWsThreads, 2) void interact_fwd_ws_kernel(
    const float* __restrict__ h, int64_t ld_h, const float* __restrict__ p, int64_t ld_p, const int32_t* __restrict__ i3,
    const float* __restrict__ wp, float* __restrict__ out, int64_t ld_out, int64_t n_edges) {
    static_assert(D == 32 || D == 64, "wave-specialised form stages whole rows");
    using S = TileShape<D>;
    constexpr int V4 = D / 4, T_STEPS = D / 8;
    constexpr int RPP = kWave / V4;                       // member rows per 1-KiB piece
    constexpr int PIECES = 3 * S::TE / RPP / 4;           // pieces per loader wave per tile
    constexpr int PL = S::TE * V4 / kBlockThreads;        // result vectors per loader thread per tile
    constexpr int OSTRIDE = D + 8;                        // result rows: the two lane halves of one accumulator store land 32 banks apart
    // Member rows sit UNPADDED in LDS (a DMA piece is lane-linear), 16-byte chunk c of row r at position c ^ tile_swizzle(r):
    // the swizzle is applied to the source address here and to the ds_read_b128 address in the consumers.
    struct Buffer {
        float tile[3][S::TE][D];
        float prod[S::TE][OSTRIDE];     // product-block sums of the tile, handed back to the loaders for the epilogue
    };
    __shared__ __attribute__((aligned(16))) Buffer buf[2];
    const int64_t n_tiles = (n_edges + S::TE - 1) / S::TE;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t g = gridDim.x;
    // tiles this workgroup owns: blockIdx.x, +g, +2g ...; trip k lands tile k in LDS, multiplies tile k-1, writes out tile k-2
    const int n_my = blockIdx.x < n_tiles ? static_cast<int>((n_tiles - blockIdx.x + g - 1) / g) : 0;

    if (wave >= 4) {
        // ---------------- loaders (and epilogue) ----------------
        // Trip k: pick up the product sums of tile k-2 -> start the DMA of the member rows of tile k into the buffer tile k-2 just
        // left -> issue its first-order rows and the ids of tile k+1 -> store tile k-2 -> barrier (which waits for the DMA).
        // The consumers never touch global memory inside the loop and nothing is staged through ds_write_b128.
        // A loader shares its SIMD with a consumer that issues MFMAs back to back, and gets few issue slots: the stream below is
        // kept to one 32x32->64-bit multiply-add per address, wave-uniform values stay in scalar registers.
        const int lw = __builtin_amdgcn_readfirstlane(wave) - 4;
        const int sub = lane / V4, chunk = lane % V4;      // row inside a piece, 16-byte chunk inside the row
        constexpr int QP = S::TE / 4 / RPP;                // pieces per member per loader wave (its TE/4 rows, RPP per piece)
        static_assert(QP == PL && PIECES == 3 * QP, "piece and result mappings coincide");
        // Piece x = m * QP + q of this wave: member m of tile rows wrow0 + q * RPP ... + RPP - 1; this lane: row + sub, chunk.
        // The same thread therefore sees all three members of "its" rows: it also gathers their first-order rows, and
        // writes their results.
        const int wrow0 = lw * (S::TE / 4);                // wave-uniform
        const int row0 = wrow0 + sub;
        const uint32_t h_row_bytes = static_cast<uint32_t>(ld_h) * 4u, p_row_bytes = static_cast<uint32_t>(ld_p) * 4u;
        const char* hsrc[QP];                              // per-lane source of chunk position `chunk` of a row of piece q (swizzled)
#pragma unroll
        for (int q = 0; q < QP; ++q) hsrc[q] = reinterpret_cast<const char*>(h) + ((chunk ^ tile_swizzle<D>(row0 + q * RPP)) * 16);
        const char* psrc = reinterpret_cast<const char*>(p) + chunk * 16;
        int node[PIECES];
        v4f pr[PIECES], first_a[PL], first_b[PL];
        auto load_ids = [&](int64_t tile_id) {
            const int64_t e0 = tile_id * S::TE + row0;
            const int32_t* idp = i3 + e0 * 3;
            if (tile_id * S::TE + S::TE <= n_edges) {      // whole tile: constant offsets from one address
#pragma unroll
                for (int x = 0; x < PIECES; ++x) node[x] = idp[(x % QP) * RPP * 3 + x / QP];
            } else {
#pragma unroll
                for (int x = 0; x < PIECES; ++x) node[x] = e0 + (x % QP) * RPP < n_edges ? idp[(x % QP) * RPP * 3 + x / QP] : 0;
            }
        };
        auto start_loads = [&](Buffer& b) {
#pragma unroll
            for (int x = 0; x < PIECES; ++x)
                lds_dma16(reinterpret_cast<const float*>(hsrc[x % QP] + static_cast<uint64_t>(static_cast<uint32_t>(node[x])) * h_row_bytes),
                          &b.tile[x / QP][wrow0 + (x % QP) * RPP][0]);
#pragma unroll
            for (int x = 0; x < PIECES; ++x)
                pr[x] = *reinterpret_cast<const v4f*>(psrc + static_cast<uint64_t>(static_cast<uint32_t>(node[x])) * p_row_bytes);
        };
        auto write_out = [&](int64_t tile_id, const v4f (&prod)[PL], const v4f (&first)[PL]) {
            const int64_t e0 = tile_id * S::TE + row0;
            float* dst = out + e0 * ld_out + chunk * 4;
            if (tile_id * S::TE + S::TE <= n_edges) {
#pragma unroll
                for (int q = 0; q < PL; ++q) edge_row_store(dst + q * RPP * ld_out, prod[q] + first[q]);
            } else {
#pragma unroll
                for (int q = 0; q < PL; ++q)
                    if (e0 + q * RPP < n_edges) edge_row_store(dst + q * RPP * ld_out, prod[q] + first[q]);
            }
        };
        const int64_t t0 = blockIdx.x;
        if (n_my > 0) load_ids(t0);
        for (int k = 0; k <= n_my; ++k) {
            Buffer& b = buf[k & 1];
            v4f prod[PL];
#pragma unroll
            for (int q = 0; q < PL; ++q) prod[q] = *reinterpret_cast<const v4f*>(&b.prod[row0 + q * RPP][chunk * 4]);
            if (k < n_my) start_loads(b);                  // the ids of tile k arrived before the last barrier
            if (k + 1 < n_my) load_ids(t0 + (k + 1) * g);
            if (k >= 2) write_out(t0 + (k - 2) * g, prod, first_a);
#pragma unroll
            for (int q = 0; q < PL; ++q) {
                first_a[q] = first_b[q];
                first_b[q] = (pr[q] + pr[q + QP]) + pr[q + 2 * QP];
            }
            __syncthreads();
        }
        if (n_my >= 1) {
            const Buffer& b = buf[(n_my - 1) & 1];
            v4f prod[PL];
#pragma unroll
            for (int q = 0; q < PL; ++q) prod[q] = *reinterpret_cast<const v4f*>(&b.prod[row0 + q * RPP][chunk * 4]);
            write_out(t0 + (n_my - 1) * g, prod, first_a);
        }
        return;
    }
    // ---------------- consumers ----------------
    const int et = D == 32 ? wave : (wave & 1);
    const int jt = D == 32 ? 0 : (wave >> 1);
    const int row = et * 32 + (lane & 31), half = lane >> 5;
    const int swz = tile_swizzle<D>(row) ^ half;          // chunk 2 ts + half of this row sits at position swz ^ 2 ts
    const v4f* wfrag = reinterpret_cast<const v4f*>(wp) + static_cast<int64_t>(jt) * NBLK * T_STEPS * kWave + lane;
    v4f wreg[NBLK][T_STEPS];
#pragma unroll
    for (int b = 0; b < NBLK; ++b)
#pragma unroll
        for (int ts = 0; ts < T_STEPS; ++ts) wreg[b][ts] = wfrag[(b * T_STEPS + ts) * kWave];
    const int j = jt * 32 + (lane & 31);
    __syncthreads();
    for (int k = 0; k < n_my; ++k) {
        Buffer& b = buf[k & 1];
        v16f acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        // A operands one k-step ahead of the MFMAs that use them: the LDS round trip hides behind 16 MFMAs.  The scheduling
        // fences keep the compiler from sinking the reads down to their first use (it does, and exposes the LDS latency).
        v4f au = *reinterpret_cast<const v4f*>(&b.tile[0][row][4 * swz]);
        v4f aq = *reinterpret_cast<const v4f*>(&b.tile[1][row][4 * swz]);
        v4f ai = *reinterpret_cast<const v4f*>(&b.tile[2][row][4 * swz]);
#pragma unroll
        for (int ts = 0; ts < T_STEPS; ++ts) {
            v4f z[4];
            z[0] = au * aq;
            z[1] = aq * ai;
            z[2] = ai * au;
            z[3] = z[0] * ai;
            __builtin_amdgcn_sched_barrier(0);
            if (ts + 1 < T_STEPS) {
                const int col = 4 * (swz ^ (2 * (ts + 1)));
                au = *reinterpret_cast<const v4f*>(&b.tile[0][row][col]);
                aq = *reinterpret_cast<const v4f*>(&b.tile[1][row][col]);
                ai = *reinterpret_cast<const v4f*>(&b.tile[2][row][col]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int bk = 0; bk < NBLK; ++bk)
#pragma unroll
                for (int s2 = 0; s2 < 4; ++s2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(z[bk][s2], wreg[bk][ts][s2], acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) b.prod[et * 32 + acc_row(r, lane)][j] = acc[r];
        __syncthreads();
    }
}

template <int D, int NBLK>
__global__ __launch_bounds__(kWsThreads, 2) void interact_bwd_members_ws_kernel(
    const float* __restrict__ h, int64_t ld_h, const int32_t* __restrict__ i3, const float* __restrict__ wq,
    const float* __restrict__ dout, int64_t ld_dout, float* __restrict__ g, int64_t n_edges) {
    static_assert(D == 32 || D == 64, "wave-specialised form stages whole rows");
    constexpr int ET = D == 32 ? 4 : 2, TE = ET * 32, V4 = D / 4, T_STEPS = D / 8;
    constexpr int RPP = kWave / V4;                       // rows per 1-KiB DMA piece
    constexpr int QP = TE / 4 / RPP;                      // pieces per row block of one loader wave
    // Same trip structure as the forward kernel: the loaders fill dtile (dout rows, swizzled like the forward's member rows:
    // it is the MFMA A operand) and htile (member rows, plain) by DMA; the consumers overwrite htile IN PLACE with the member
    // gradients of the product rule (every (row, column) element is read and written by the one lane that owns it), and the
    // loaders stream those rows out as 16-byte vectors two trips later.
    struct Buffer {
        float dtile[TE][D];
        float htile[3][TE][D];
    };
    __shared__ __attribute__((aligned(16))) Buffer buf[2];
    const int64_t n_tiles = (n_edges + TE - 1) / TE;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t grid = gridDim.x;
    const int n_my = blockIdx.x < n_tiles ? static_cast<int>((n_tiles - blockIdx.x + grid - 1) / grid) : 0;

    if (wave >= 4) {
        const int lw = __builtin_amdgcn_readfirstlane(wave) - 4;
        const int sub = lane / V4, chunk = lane % V4;
        const int wrow0 = lw * (TE / 4), row0 = wrow0 + sub;
        const uint32_t h_row_bytes = static_cast<uint32_t>(ld_h) * 4u;
        const char* hsrc = reinterpret_cast<const char*>(h) + chunk * 16;
        int node[3 * QP];
        auto load_ids = [&](int64_t tile_id) {
            const int64_t e0 = tile_id * TE + row0;
            const int32_t* idp = i3 + e0 * 3;
            if (tile_id * TE + TE <= n_edges) {
#pragma unroll
                for (int x = 0; x < 3 * QP; ++x) node[x] = idp[(x % QP) * RPP * 3 + x / QP];
            } else {
#pragma unroll
                for (int x = 0; x < 3 * QP; ++x) node[x] = e0 + (x % QP) * RPP < n_edges ? idp[(x % QP) * RPP * 3 + x / QP] : 0;
            }
        };
        auto start_loads = [&](Buffer& b, int64_t tile_id) {
            const int64_t e0 = tile_id * TE + row0;
#pragma unroll
            for (int q = 0; q < QP; ++q) {                 // dout rows: a stream; rows past the end re-read the last one
                const int r = row0 + q * RPP;
                int64_t e = e0 + q * RPP;
                e = e < n_edges ? e : n_edges - 1;
                lds_dma16(dout + e * ld_dout + (chunk ^ tile_swizzle<D>(r)) * 4, &b.dtile[wrow0 + q * RPP][0]);
            }
#pragma unroll
            for (int x = 0; x < 3 * QP; ++x)
                lds_dma16(reinterpret_cast<const float*>(hsrc + static_cast<uint64_t>(static_cast<uint32_t>(node[x])) * h_row_bytes),
                          &b.htile[x / QP][wrow0 + (x % QP) * RPP][0]);
        };
        v4f gout[3 * QP];
        auto pick_up = [&](const Buffer& b) {
#pragma unroll
            for (int x = 0; x < 3 * QP; ++x) gout[x] = *reinterpret_cast<const v4f*>(&b.htile[x / QP][row0 + (x % QP) * RPP][chunk * 4]);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the rows are in registers before a DMA may overwrite them
        };
        auto write_out = [&](int64_t tile_id) {
            const int64_t e0 = tile_id * TE + row0;
            float* dst = g + e0 * (3 * D) + chunk * 4;
            const bool full = tile_id * TE + TE <= n_edges;
#pragma unroll
            for (int x = 0; x < 3 * QP; ++x)
                if (full || e0 + (x % QP) * RPP < n_edges) store_stream4(dst + (x % QP) * RPP * (3 * D) + (x / QP) * D, gout[x]);
        };
        const int64_t t0 = blockIdx.x;
        if (n_my > 0) load_ids(t0);
        for (int k = 0; k <= n_my; ++k) {
            Buffer& b = buf[k & 1];
            if (k >= 2) pick_up(b);
            if (k < n_my) start_loads(b, t0 + k * grid);   // the ids of tile k arrived before the last barrier
            if (k + 1 < n_my) load_ids(t0 + (k + 1) * grid);
            if (k >= 2) write_out(t0 + (k - 2) * grid);
            __syncthreads();
        }
        if (n_my >= 1) {
            pick_up(buf[(n_my - 1) & 1]);
            write_out(t0 + (n_my - 1) * grid);
        }
        return;
    }
    const int et = wave % ET, ct = wave / ET;
    const int row = et * 32 + (lane & 31), half = lane >> 5;
    const int swz = tile_swizzle<D>(row) ^ half;          // chunk 2 ts + half of this row sits at position swz ^ 2 ts
    const v4f* wfrag = reinterpret_cast<const v4f*>(wq) + static_cast<int64_t>(ct) * NBLK * T_STEPS * kWave + lane;
    v4f wreg[NBLK][T_STEPS];
#pragma unroll
    for (int b = 0; b < NBLK; ++b)
#pragma unroll
        for (int ts = 0; ts < T_STEPS; ++ts) wreg[b][ts] = wfrag[(b * T_STEPS + ts) * kWave];
    const int c = ct * 32 + (lane & 31);
    __syncthreads();
    for (int k = 0; k < n_my; ++k) {
        Buffer& b = buf[k & 1];
        v16f acc[NBLK];
#pragma unroll
        for (int bk = 0; bk < NBLK; ++bk)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[bk][r] = 0.f;
        v4f a = *reinterpret_cast<const v4f*>(&b.dtile[row][4 * swz]);
#pragma unroll
        for (int ts = 0; ts < T_STEPS; ++ts) {
            const v4f a_now = a;
            __builtin_amdgcn_sched_barrier(0);
            if (ts + 1 < T_STEPS) a = *reinterpret_cast<const v4f*>(&b.dtile[row][4 * (swz ^ (2 * (ts + 1)))]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2)
#pragma unroll
                for (int bk = 0; bk < NBLK; ++bk) acc[bk] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_now[s2], wreg[bk][ts][s2], acc[bk], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        // epilogue in batches of four rows: 12 LDS reads in flight, then the product rule, results back over the member values
#pragma unroll
        for (int r0 = 0; r0 < 16; r0 += 4) {
            float hu[4], hq[4], hi[4];
#pragma unroll
            for (int k4 = 0; k4 < 4; ++k4) {
                const int er = et * 32 + acc_row(r0 + k4, lane);
                hu[k4] = b.htile[0][er][c];
                hq[k4] = b.htile[1][er][c];
                hi[k4] = b.htile[2][er][c];
            }
#pragma unroll
            for (int k4 = 0; k4 < 4; ++k4) {
                const int r = r0 + k4;
                const int er = et * 32 + acc_row(r, lane);
                const float z_uq = acc[0][r], z_qi = acc[1][r], z_iu = acc[2][r];
                const float z_uqi = NBLK == 4 ? acc[NBLK - 1][r] : 0.f;
                b.htile[0][er][c] = z_uq * hq[k4] + z_iu * hi[k4] + z_uqi * (hq[k4] * hi[k4]);
                b.htile[1][er][c] = z_uq * hu[k4] + z_qi * hi[k4] + z_uqi * (hu[k4] * hi[k4]);
                b.htile[2][er][c] = z_qi * hq[k4] + z_iu * hu[k4] + z_uqi * (hu[k4] * hq[k4]);
            }
        }
        __syncthreads();
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

// ------------------------------------------------------------------------------------------------
// Strip kernels (D = 128: configs C3 / C4).  Eight waves and no loader role: wave w owns the 16-column strip w of the result with
// the WHOLE contraction index, so its weight fragments (NBLK * D/16 float4 = 128 VGPRs) are loaded once per kernel and stay in
// registers - nothing is streamed from L2 inside the loop (the plain tilings above re-fetch 256 KB of packed weights per tile).
// v_mfma_f32_16x16x4_f32 (A[l&15][k = l>>4], B[k = l>>4][l&15], D: column l&15, rows 4 (l>>4) + r): lane group kq = l>>4 of MFMA
// step s of k-group g is given k = 16 g + 4 kq + s, so a lane needs 4 CONSECUTIVE k per operand: one ds_read_b128 per staged row
// per k-group feeds 4 x NBLK MFMAs.  Tiles of 32 hyperedges (two 16-row MFMA tiles = two independent accumulator chains per
// wave), double-buffered LDS image filled by LDS-DMA which every wave issues for itself right after the barrier that opens the
// PREVIOUS tile's MFMA phase, so a fill has a whole MFMA phase (~16 k cycles) to land; two workgroup barriers per tile.
// Rows sit unpadded in LDS, 16-byte chunk c of row r at position c ^ (r & 15) (conflict-free ds_read_b128 for this lane map;
// the swizzle is applied to the DMA's per-lane SOURCE address).
// ------------------------------------------------------------------------------------------------
constexpr int kStripTE = 32;

// ws_fwd[strip][b][g][lane][4] = W[16 strip + (lane&15)][(3+b)d + 16 g + 4 (lane>>4) + s]        (k runs along c)
// ws_bwd[strip][b][g][lane][4] = W[16 g + 4 (lane>>4) + s][(3+b)d + 16 strip + (lane&15)]        (k runs along j)
__global__ __launch_bounds__(kBlockThreads) void pack_weights_strip_kernel(const float* __restrict__ w, int64_t ld_w, int d, int nblk,
                                                                           float* __restrict__ ws_fwd, float* __restrict__ ws_bwd) {
    const int kg = d / 16;
    const int total = (d / 16) * nblk * kg * kWave;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int lane = idx & (kWave - 1);
        const int g = (idx >> 6) % kg;
        const int b = ((idx >> 6) / kg) % nblk;
        const int strip = (idx >> 6) / (kg * nblk);
        const int c = lane & 15, kq = lane >> 4;
        if (ws_fwd != nullptr) {
            const float* src = w + static_cast<int64_t>(16 * strip + c) * ld_w + static_cast<int64_t>(3 + b) * d + 16 * g + 4 * kq;
            *reinterpret_cast<float4*>(ws_fwd + static_cast<int64_t>(idx) * 4) = make_float4(src[0], src[1], src[2], src[3]);
        }
        if (ws_bwd != nullptr) {
            const float* src = w + static_cast<int64_t>(16 * g + 4 * kq) * ld_w + static_cast<int64_t>(3 + b) * d + 16 * strip + c;
            *reinterpret_cast<float4*>(ws_bwd + static_cast<int64_t>(idx) * 4) = make_float4(src[0], src[ld_w], src[2 * ld_w], src[3 * ld_w]);
        }
    }
}

template <int D, int NBLK>
__global__ __launch_bounds__(kWsThreads) void interact_fwd_strip_kernel(
    const float* __restrict__ h, int64_t ld_h, const float* __restrict__ p, int64_t ld_p, const int32_t* __restrict__ i3,
    const float* __restrict__ wp, float* __restrict__ out, int64_t ld_out, int64_t n_edges) {
    static_assert(D == 128, "eight 16-column strips");
    constexpr int TE = kStripTE, V4 = D / 4, KG = D / 16, RT = TE / 16, STEPS = KG * RT;
    constexpr int ROWS_PER_PIECE = kWave / V4;                            // 2 rows of 512 B per 1-KiB DMA piece
    constexpr int PIECES_PER_MEMBER = TE / ROWS_PER_PIECE;                // 16
    constexpr int PIECES = 3 * PIECES_PER_MEMBER / 8;                     // 6 per wave per tile
    constexpr int OSTRIDE = D + 4;                                        // accumulator rows r and r + 4 of one store land 16 banks apart
    struct Buffer { float tile[3][TE][D]; };
    __shared__ __attribute__((aligned(16))) Buffer buf[2];
    // product sums of a tile, handed from the MFMA layout to the row-wise epilogue; two images because the epilogue of tile k runs
    // INSIDE the MFMA phase of tile k + 1 (one barrier per tile, no phase in which the matrix pipe only waits for stores)
    __shared__ __attribute__((aligned(16))) float prod[2][TE][OSTRIDE];
    __shared__ int ids[3][3 * TE];                                        // ring over tiles (tile_id % 3)
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t n_tiles = (n_edges + TE - 1) / TE;
    const int64_t grid = gridDim.x;
    const int n_my = blockIdx.x < n_tiles ? static_cast<int>((n_tiles - blockIdx.x + grid - 1) / grid) : 0;
    if (n_my == 0) return;
    const int64_t t0 = blockIdx.x;

    // this wave's weight fragments, resident for the whole kernel
    const v4f* wfrag = reinterpret_cast<const v4f*>(wp) + static_cast<int64_t>(wave) * NBLK * KG * kWave + (tid & 63);
    v4f wreg[NBLK][KG];
#pragma unroll
    for (int b = 0; b < NBLK; ++b)
#pragma unroll
        for (int gk = 0; gk < KG; ++gk) wreg[b][gk] = wfrag[(b * KG + gk) * kWave];

    // DMA pieces of this wave: piece x = wave * PIECES + k -> member x / 16, rows 2 (x % 16) and + 1; lane: row + (lane >> 5), chunk lane & 31.
    // Lane-dependent values are re-derived from `tl` (an opaque copy of the thread id) wherever they are used: kept in registers
    // across the MFMA phase they push the weight fragments out.
    const uint32_t h_row_bytes = static_cast<uint32_t>(ld_h) * 4u;
    const char* hbytes = reinterpret_cast<const char*>(h);
    int tl = tid;
    // Ids: the 3 TE member ids of a tile are contiguous in i3; waves 0 and 1 fetch them with one coalesced load each two tiles ahead
    // and park them in a small LDS ring (slot = local tile number % 3), every wave then picks the handful it needs with ds_read_b32:
    // 2 vector-memory instructions per tile and workgroup instead of 9 scattered ones per WAVE (each memory instruction issued
    // beside the MFMA stream costs the matrix pipe time).
    auto ring_of = [&](int k) { return ids[k % 3]; };
    auto fetch_ids = [&](int k) {
        const int lane = tl & 63;
        if (wave < 2 && (wave == 0 || lane < 32)) {
            const int j = wave * 64 + lane;
            const int64_t pos = (t0 + k * grid) * (3 * TE) + j;
            ring_of(k)[j] = pos < n_edges * 3 ? i3[pos] : 0;
        }
    };
    auto issue_dma = [&](Buffer& b, int k) {
        const int prow = (tl >> 5) & 1, pchunk = tl & 31;
        const int* idk = ring_of(k);
#pragma unroll
        for (int kk = 0; kk < PIECES; ++kk) {
            const int x = wave * PIECES + kk;
            const int r0 = 2 * (x & (PIECES_PER_MEMBER - 1));
            const int swz = (pchunk ^ ((r0 + prow) & 15)) * 16;
            const int node = idk[(r0 + prow) * 3 + (x >> 4)];
            lds_dma16(reinterpret_cast<const float*>(hbytes + static_cast<uint64_t>(static_cast<uint32_t>(node)) * h_row_bytes + swz), &b.tile[x >> 4][r0][0]);
        }
    };
    // epilogue mapping: thread -> hyperedge row tid >> 4, columns 4 (tid & 15) .. + 3 and 64 + the same
    v4f first[2], first_prev[2], pr[3][2];
    auto load_first_order_rows = [&](int k) {
        const int* idk = ring_of(k) + (tl >> 4) * 3;
        const int ecol = (tl & 15) * 4;
#pragma unroll
        for (int m = 0; m < 3; ++m) {
            const float* src = p + static_cast<int64_t>(idk[m]) * ld_p + ecol;
            pr[m][0] = *reinterpret_cast<const v4f*>(src);
            pr[m][1] = *reinterpret_cast<const v4f*>(src + 64);
        }
    };
    auto write_out = [&](int64_t tile_id, const float (*pimg)[OSTRIDE]) {
        const int erow = tl >> 4, ecol = (tl & 15) * 4;
        const int64_t e = tile_id * TE + erow;
        const v4f s0 = *reinterpret_cast<const v4f*>(&pimg[erow][ecol]) + first_prev[0];
        const v4f s1 = *reinterpret_cast<const v4f*>(&pimg[erow][64 + ecol]) + first_prev[1];
        if (e < n_edges) {
            float* dst = out + e * ld_out + ecol;
            store_stream4(dst, s0);
            store_stream4(dst + 64, s1);
        }
    };

    fetch_ids(0);
    if (n_my > 1) fetch_ids(1);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    issue_dma(buf[0], 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // Phase k (between two barriers) of a wave = the MFMAs of tile k with, somewhere inside, its SERVICE block: start the fill of
    // tile k + 1 -> finish tile k - 1 (product sums from LDS + first-order rows, 16-byte stores) -> request the first-order rows of
    // tile k (-> fetch the ids of tile k + 2).  The two waves of a SIMD run the service block half a phase apart (waves 0-3 at the
    // start, waves 4-7 after half of their MFMAs), so one wave's memory instructions issue beside the other wave's MFMAs instead
    // of both leaving the matrix pipe idle at the same time.  At the end: product sums to LDS, wait for everything this wave has
    // in flight, barrier.
    const bool late = wave >= 4;
    for (int k = 0; k < n_my; ++k) {
        const int64_t tile_id = t0 + k * grid;
        Buffer& b = buf[k & 1];
        asm volatile("" : "+v"(tl));
        const int arow = tl & 15, kq = (tl >> 4) & 3;
        v4f acc[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc[rt] = v4f{0.f, 0.f, 0.f, 0.f};
        // One (k-group, row tile) step = 3 ds_read_b128 + 4 products + 4 NBLK MFMAs on one accumulator chain; the chain's 40-cycle
        // dependent latency against the 32-cycle issue interval is covered by the SIMD's other wave.  The member rows of the NEXT
        // step are read before this step's MFMAs (the scheduling fences keep the compiler from sinking or hoisting them further,
        // which costs either the LDS latency or the registers of several steps' operands).
        // byte offset of chunk (4 g + kq) ^ arow of row arow = lane_off ^ (g << 6): ONE address register, re-derived per step (the
        // empty asm keeps the compiler from hoisting sixteen of them out of the tile loop, which spills the weight fragments)
        int lane_off = arow * (D * 4) + ((kq ^ arow) << 4);
        asm volatile("" : "+v"(lane_off));
        const char* tbase = reinterpret_cast<const char*>(&b.tile[0][0][0]);
        auto member_chunk = [&](int m, int rt, int gk) {
            return *reinterpret_cast<const v4f*>(tbase + (lane_off ^ (gk << 6)) + (m * TE + rt * 16) * (D * 4));
        };
        v4f au = member_chunk(0, 0, 0), aq = member_chunk(1, 0, 0), ai = member_chunk(2, 0, 0);
#pragma unroll
        for (int step = 0; step < STEPS; ++step) {
            const int gk = step / RT, rt = step % RT;
            if (step == 0 || step == STEPS / 2) {
                if (late == (step != 0)) {                                  // wave-uniform
                    if (k + 1 < n_my) issue_dma(buf[(k + 1) & 1], k + 1);   // that buffer's last reader was tile k - 1's MFMA phase
                    if (k > 0) write_out(tile_id - grid, prod[(k - 1) & 1]);
                    load_first_order_rows(k);
                    if (k + 2 < n_my) fetch_ids(k + 2);                     // slot (k + 2) % 3 = (k - 1) % 3: last read in phase k - 1
                }
            }
            v4f z[4];
            z[0] = au * aq;
            z[1] = aq * ai;
            z[2] = ai * au;
            z[3] = z[0] * ai;
            __builtin_amdgcn_sched_barrier(0);
            if (step + 1 < STEPS) {
                const int gn = (step + 1) / RT, rn = (step + 1) % RT;
                au = member_chunk(0, rn, gn);
                aq = member_chunk(1, rn, gn);
                ai = member_chunk(2, rn, gn);
            }
            if (step == 5 || step == STEPS / 2 + 5) {
                if (late == (step != 5)) {                                  // the first-order rows have arrived by now: 24 registers -> 8
                    first[0] = (pr[0][0] + pr[1][0]) + pr[2][0];
                    first[1] = (pr[0][1] + pr[1][1]) + pr[2][1];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int bk = 0; bk < NBLK; ++bk)
#pragma unroll
                for (int s2 = 0; s2 < 4; ++s2) acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(z[bk][s2], wreg[bk][gk][s2], acc[rt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        float (*pimg)[OSTRIDE] = prod[k & 1];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) pimg[rt * 16 + 4 * kq + r][16 * wave + arow] = acc[rt][r];
        first_prev[0] = first[0];
        first_prev[1] = first[1];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    asm volatile("" : "+v"(tl));
    write_out(t0 + (n_my - 1) * grid, prod[(n_my - 1) & 1]);
}

// Member gradients, same layout: the loaders' job of the D <= 64 kernel is done by every wave for its own rows.  A operand = the
// dout rows (a stream, swizzled like the forward's member rows), the member rows sit beside them as they lie; every wave owns a
// 16-column strip of all NBLK product blocks (NBLK x 2 accumulators), applies the product rule to its own (row, column) elements
// IN PLACE in the member tile, and - one barrier later, inside the next tile's MFMA phase - streams the rows it will refill out
// as 16-byte vectors.
// UR ("user reduced", needs hyperedges sorted by user - the layout's numbering): the user-slot gradients are NOT written to the
// member buffer.  Consecutive hyperedges of one user are summed on chip - waves 6 and 7 (one thread per column) walk the user-slot
// rows of every finished tile in hyperedge order, carrying (user, running sum) in registers from tile to tile, which is why a
// workgroup then takes a CONTIGUOUS range of tiles - and a finished run goes straight to dh[user]; only the first and the last run
// of a workgroup's range (which may continue in the neighbours) go to a small boundary table that user_boundary_fixup_kernel adds
// up in workgroup order.  The member buffer shrinks to [E, 2, d] (query, item): a third less stored here and a third less read by
// the K7 pass that follows.  Same sums as K7 over the user's incidence list, in the same (hyperedge) order.
template <int D, int NBLK, bool UR>
__global__ __launch_bounds__(kWsThreads) void interact_bwd_members_strip_kernel(
    const float* __restrict__ h, int64_t ld_h, const int32_t* __restrict__ i3, const float* __restrict__ wq,
    const float* __restrict__ dout, int64_t ld_dout, float* __restrict__ g_out, int64_t n_edges,
    float* __restrict__ dh_user, int64_t ld_dh, float* __restrict__ bnd_val, int32_t* __restrict__ bnd_user) {
    static_assert(D == 128, "eight 16-column strips");
    constexpr int TE = kStripTE, V4 = D / 4, KG = D / 16, RT = TE / 16;
    constexpr int ROWS_PER_PIECE = kWave / V4;                            // 2
    constexpr int PIECES_PER_MEMBER = TE / ROWS_PER_PIECE;                // 16
    constexpr int NP = UR ? 11 : 8;                                       // DMA pieces per wave per tile: 16 dout + 48 member pieces over 8 (UR: 6) waves
    // plain: member rows of a tile (3 images) beside its dout rows, double-buffered.  UR: the user-slot image is read by the two walker
    // waves half a phase after the barrier and over ALL rows, so it cannot be refilled by either of them while the other may still be
    // walking: it lives in a ring of three of its own (slot = local tile % 3, refilled one phase after it was walked).
    constexpr int HM = UR ? 2 : 3;
    struct Buffer {
        float dtile[TE][D];
        float htile[HM][TE][D];
    };
    __shared__ __attribute__((aligned(16))) Buffer buf[2];
    __shared__ __attribute__((aligned(16))) float utile[UR ? 3 : 1][TE][D];
    auto member_rows = [&](int m, int k) -> float (*)[D] {               // image of member m of local tile k
        if (UR) return m == 0 ? utile[k % 3] : buf[k & 1].htile[m - 1];
        return buf[k & 1].htile[m];
    };
    __shared__ int ids[4][3 * TE];                                        // ring over this workgroup's tiles (local tile number & 3)
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t n_tiles = (n_edges + TE - 1) / TE;
    const int64_t grid = gridDim.x;
    // plain: tiles blockIdx.x, + grid, ...;  UR: the contiguous range [blockIdx.x * per, ...) so that a user's run stays in one workgroup
    const int64_t per = (n_tiles + grid - 1) / grid;
    const int64_t t0 = UR ? static_cast<int64_t>(blockIdx.x) * per : blockIdx.x;
    const int64_t t_step = UR ? 1 : grid;
    const int n_my = UR ? static_cast<int>(std::max<int64_t>(0, std::min<int64_t>(per, n_tiles - t0)))
                        : (blockIdx.x < n_tiles ? static_cast<int>((n_tiles - blockIdx.x + grid - 1) / grid) : 0);
    if (n_my == 0) {
        if (UR && tid == 0) bnd_user[2 * blockIdx.x] = bnd_user[2 * blockIdx.x + 1] = -1;
        return;
    }

    const v4f* wfrag = reinterpret_cast<const v4f*>(wq) + static_cast<int64_t>(wave) * NBLK * KG * kWave + (tid & 63);
    v4f wreg[NBLK][KG];
#pragma unroll
    for (int b = 0; b < NBLK; ++b)
#pragma unroll
        for (int gk = 0; gk < KG; ++gk) wreg[b][gk] = wfrag[(b * KG + gk) * kWave];

    const uint32_t h_row_bytes = static_cast<uint32_t>(ld_h) * 4u;
    int tl = tid;                                                        // opaque copy of the thread id, see the forward kernel
    auto fetch_ids = [&](int k) {                                        // see the forward kernel
        const int lane = tl & 63;
        if (wave < 2 && (wave == 0 || lane < 32)) {
            const int j = wave * 64 + lane;
            const int64_t pos = (t0 + k * t_step) * (3 * TE) + j;
            ids[k & 3][j] = pos < n_edges * 3 ? i3[pos] : 0;
        }
    };
    // Piece kk of this wave: member m (-1: a dout piece, -2: none), rows r0, r0 + 1.  plain: 2 dout + 6 member pieces per wave;
    // UR: waves 6 and 7 walk the user slot instead and move nothing, waves 0-5 share all 64 pieces (the user image's ring slot of tile
    // k + 1 was last walked a phase ago, so anybody may refill it).
    auto piece = [&](int kk, int& m, int& r0) {
        if (UR) {
            const int y = kk * 6 + wave;
            if (wave >= 6 || y >= 4 * PIECES_PER_MEMBER) {
                m = -2;
                r0 = 0;
            } else {
                m = (y >> 4) - 1;
                r0 = 2 * (y & (PIECES_PER_MEMBER - 1));
            }
        } else if (kk < 2) {
            m = -1;
            r0 = 2 * (wave * 2 + kk);
        } else {
            const int x = wave * 6 + (kk - 2);
            m = x >> 4;
            r0 = 2 * (x & (PIECES_PER_MEMBER - 1));
        }
    };
    auto issue_dma = [&](int k) {
        Buffer& b = buf[k & 1];
        const int64_t tile_id = t0 + k * t_step;
        const int prow = (tl >> 5) & 1, pchunk = tl & 31;
        const char* hbytes = reinterpret_cast<const char*>(h) + pchunk * 16;
        const int* idk = ids[k & 3];
#pragma unroll
        for (int kk = 0; kk < NP; ++kk) {
            int m, r0;
            piece(kk, m, r0);
            if (m == -2) continue;
            if (m < 0) {                                                 // dout rows: a stream; rows past the end re-read the last one
                int64_t e = tile_id * TE + r0 + prow;
                e = e < n_edges ? e : n_edges - 1;
                lds_dma16(dout + e * ld_dout + (pchunk ^ ((r0 + prow) & 15)) * 4, &b.dtile[r0][0]);
            } else {
                const int node = idk[(r0 + prow) * 3 + m];
                lds_dma16(reinterpret_cast<const float*>(hbytes + static_cast<uint64_t>(static_cast<uint32_t>(node)) * h_row_bytes), &member_rows(m, k)[r0][0]);
            }
        }
    };
    // the member-gradient rows of a finished tile: picked up from LDS by the wave that refills exactly these rows (so its own DMA
    // may follow its own reads without a barrier) and stored as they lie; two batches of four pieces keep the register cost at 16
    auto stream_out = [&](int k) {
        const int64_t tile_id = t0 + k * t_step;
        const int prow = (tl >> 5) & 1, pchunk = tl & 31;
        const bool full = tile_id * TE + TE <= n_edges;
#pragma unroll
        for (int k0 = 0; k0 < NP; k0 += 4) {
            v4f gv[4];
#pragma unroll
            for (int kk = 0; kk < 4 && k0 + kk < NP; ++kk) {
                int m, r0;
                piece(k0 + kk, m, r0);
                if (m >= (UR ? 1 : 0)) gv[kk] = *reinterpret_cast<const v4f*>(&member_rows(m, k)[r0 + prow][pchunk * 4]);
            }
#pragma unroll
            for (int kk = 0; kk < 4 && k0 + kk < NP; ++kk) {
                int m, r0;
                piece(k0 + kk, m, r0);
                const int64_t e = tile_id * TE + r0 + prow;
                if (m >= (UR ? 1 : 0) && (full || e < n_edges))
                    store_stream4(UR ? g_out + e * (2 * D) + (m - 1) * D + pchunk * 4 : g_out + e * (3 * D) + m * D + pchunk * 4, gv[kk]);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");              // every row is in registers (or gone) before a DMA may overwrite it
    };
    // UR, waves 6-7: thread -> column; the running sum of the current user's run survives from tile to tile in registers.  The run's
    // destination row is a (scalar) pointer: dh[user], or - for the first run of the range, which may have begun in the previous
    // workgroup - the workgroup's first slot of the boundary table.
    int cur_user = -1, first_user = -1;
    float run_sum = 0.f;
    float* run_dst = UR ? bnd_val + (2 * blockIdx.x) * D : nullptr;
    auto walk_user_slot = [&](int k) {
        const float (*urows)[D] = utile[k % 3];
        const int c = tl - 6 * kWave, lane = tl & 63;
        const int* idk = ids[k & 3];
        const int64_t e0 = (t0 + k * t_step) * TE;
        const int rows = static_cast<int>(std::min<int64_t>(TE, n_edges - e0));
        // where the runs start is the same for every column: one ballot over the tile's user ids gives it as a scalar bit mask, so the
        // per-column walk is straight-line code (one scalar branch per row, taken only where a run ends), all 32 values in flight at once
        const int r = lane < rows ? lane : rows - 1;
        const int my_uid = idk[r * 3];
        const int prev_uid = r == 0 ? cur_user : idk[(r - 1) * 3];
        const uint64_t new_run = __ballot(lane < rows && my_uid != prev_uid);
        float v[TE];
#pragma unroll
        for (int x = 0; x < TE; ++x) v[x] = x < rows ? urows[x][c] : 0.f;
#pragma unroll
        for (int x = 0; x < TE; ++x) {
            if ((new_run >> x) & 1) {
                const int user = __builtin_amdgcn_readlane(my_uid, x);
                if (cur_user >= 0) run_dst[c] = run_sum;                 // the finished run
                else first_user = user;
                if (cur_user >= 0) run_dst = dh_user + static_cast<int64_t>(user) * ld_dh;
                cur_user = user;
                run_sum = 0.f;
            }
            run_sum += v[x];
        }
    };

    fetch_ids(0);
    if (n_my > 1) fetch_ids(1);
    if (n_my > 2) fetch_ids(2);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    issue_dma(0);
    if (n_my > 1) issue_dma(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // Phase k of a wave = the MFMAs of tile k with its service block inside (waves 0-3: at the start, waves 4-7: after half of the
    // MFMAs - see the forward kernel): stream out tile k - 1, refill its buffer with tile k + 1, fetch the ids of tile k + 2.  Then the
    // product rule in place, wait for this wave's outstanding operations, barrier.
    const bool late = wave >= 4;
    for (int k = 0; k < n_my; ++k) {
        Buffer& b = buf[k & 1];
        asm volatile("" : "+v"(tl));
        const int arow = tl & 15, kq = (tl >> 4) & 3;
        const int col = 16 * wave + arow;
        v4f acc[RT][NBLK];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int bk = 0; bk < NBLK; ++bk) acc[rt][bk] = v4f{0.f, 0.f, 0.f, 0.f};
        int lane_off = arow * (D * 4) + ((kq ^ arow) << 4);              // see the forward kernel
        asm volatile("" : "+v"(lane_off));
        const char* dbase = reinterpret_cast<const char*>(&b.dtile[0][0]);
        v4f a[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) a[rt] = *reinterpret_cast<const v4f*>(dbase + lane_off + rt * 16 * (D * 4));
#pragma unroll
        for (int gk = 0; gk < KG; ++gk) {
            if (gk == 0 && !late && k > 0) {                               // wave-uniform
                stream_out(k - 1);
                if (k + 1 < n_my) issue_dma(k + 1);                        // the buffer tile k - 1 leaves
                if (k + 2 < n_my) fetch_ids(k + 2);                        // ring slot (k + 2) & 3 = (k - 2) & 3: tile k - 2 is long finished
            }
            if (gk == KG / 2 && late && k > 0) {
                if (UR && wave >= 6) {
                    walk_user_slot(k - 1);
                } else {
                    stream_out(k - 1);
                    if (k + 1 < n_my) issue_dma(k + 1);
                }
            }
            v4f a_now[RT];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) a_now[rt] = a[rt];
            __builtin_amdgcn_sched_barrier(0);
            if (gk + 1 < KG) {
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) a[rt] = *reinterpret_cast<const v4f*>(dbase + (lane_off ^ ((gk + 1) << 6)) + rt * 16 * (D * 4));
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2)
#pragma unroll
                for (int bk = 0; bk < NBLK; ++bk)
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) acc[rt][bk] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_now[rt][s2], wreg[bk][gk][s2], acc[rt][bk], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        // product rule on this lane's own (row, column) elements, results over the member values
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            float hu[4], hq[4], hi[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int er = rt * 16 + 4 * kq + r;
                hu[r] = member_rows(0, k)[er][col];
                hq[r] = member_rows(1, k)[er][col];
                hi[r] = member_rows(2, k)[er][col];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int er = rt * 16 + 4 * kq + r;
                const float z_uq = acc[rt][0][r], z_qi = acc[rt][1][r], z_iu = acc[rt][2][r];
                const float z_uqi = NBLK == 4 ? acc[rt][NBLK - 1][r] : 0.f;
                member_rows(0, k)[er][col] = z_uq * hq[r] + z_iu * hi[r] + z_uqi * (hq[r] * hi[r]);
                member_rows(1, k)[er][col] = z_uq * hu[r] + z_qi * hi[r] + z_uqi * (hu[r] * hi[r]);
                member_rows(2, k)[er][col] = z_qi * hq[r] + z_iu * hu[r] + z_uqi * (hu[r] * hq[r]);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    asm volatile("" : "+v"(tl));
    if (UR && wave >= 6) {
        const int c = tl - 6 * kWave;
        walk_user_slot(n_my - 1);
        // The last run of the range may continue in the next workgroup: it goes to the second boundary slot - unless it IS the first
        // run (a range inside one user's run), which still points at the first slot.
        const bool one_run = run_dst == bnd_val + (2 * blockIdx.x) * D;
        if (cur_user >= 0) {
            if (one_run) run_dst[c] = run_sum;
            else bnd_val[(2 * blockIdx.x + 1) * D + c] = run_sum;
        }
        if (c == 0) {
            bnd_user[2 * blockIdx.x] = first_user;
            bnd_user[2 * blockIdx.x + 1] = (cur_user >= 0 && !one_run) ? cur_user : -1;
        }
    } else {
        stream_out(n_my - 1);
    }
}

// Adds up the boundary runs of the user-reduced member-gradient kernels in workgroup order.  One workgroup per table entry: the entry
// that opens a user's run (the valid entry before it belongs to another user) walks on while the following entries carry the same
// user - nearly always one or two - and writes dh[user]; every other workgroup leaves at once.  Same order of additions as a single
// walk over the table.
__global__ __launch_bounds__(256) void user_boundary_fixup_kernel(const float* __restrict__ bnd_val, const int32_t* __restrict__ bnd_user, int n_entries, int d,
                                                                  float* __restrict__ dh_user, int64_t ld_dh) {
    const int k0 = blockIdx.x, c = threadIdx.x;
    const int user = bnd_user[k0];
    if (user < 0 || c >= d) return;
    for (int k = k0 - 1; k >= 0; --k) {
        const int u = bnd_user[k];
        if (u == user) return;                    // an earlier entry opens this run
        if (u >= 0) break;
    }
    float acc = bnd_val[static_cast<int64_t>(k0) * d + c];
    for (int k = k0 + 1; k < n_entries; ++k) {
        const int u = bnd_user[k];
        if (u < 0) continue;
        if (u != user) break;
        acc += bnd_val[static_cast<int64_t>(k) * d + c];
    }
    dh_user[static_cast<int64_t>(user) * ld_dh + c] = acc;
}

// ------------------------------------------------------------------------------------------------
// Strip kernels for D = 256 (config C5).  A 16-column strip with the whole contraction index would need 256 weight registers, so
// a workgroup takes a QUARTER of the output columns (blockIdx.y; 4 strips) and splits the contraction index in two: wave = (strip,
// k-half), 128 weight registers each, partial sums meet in the epilogue (the product rule of the member-gradient kernel is linear
// in the contracted values, so its partial results simply add).  Tiles of 16 hyperedges (48 KB of member rows per buffer).  The
// four column quarters of one tile range sit on one XCD (grid.x is a multiple of 8 and workgroups are dealt round-robin by their
// linear id), so three of the four fetches of a member row hit that XCD's L2.
// ------------------------------------------------------------------------------------------------
constexpr int kStrip256TE = 16;
constexpr int kStrip256Grid = 64;           // workgroups per column quarter (x 4 quarters = one per CU)

template <int NBLK>
__global__ __launch_bounds__(kWsThreads) void interact_fwd_strip256_kernel(
    const float* __restrict__ h, int64_t ld_h, const float* __restrict__ p, int64_t ld_p, const int32_t* __restrict__ i3,
    const float* __restrict__ wp, float* __restrict__ out, int64_t ld_out, int64_t n_edges) {
    constexpr int D = 256, TE = kStrip256TE, CW = 64, KG = D / 16, KGW = KG / 2, OSTRIDE = CW + 4, PIECES = 3 * TE / 8;
    struct Buffer { float tile[3][TE][D]; };
    __shared__ __attribute__((aligned(16))) Buffer buf[2];
    __shared__ __attribute__((aligned(16))) float prod[2][2][TE][OSTRIDE];       // [tile parity][k-half]
    __shared__ int ids[3][3 * TE];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int strip = wave & 3, kpart = wave >> 2;
    const int cbase = blockIdx.y * CW;
    const int64_t n_tiles = (n_edges + TE - 1) / TE;
    const int64_t grid = gridDim.x;
    const int n_my = blockIdx.x < n_tiles ? static_cast<int>((n_tiles - blockIdx.x + grid - 1) / grid) : 0;
    if (n_my == 0) return;
    const int64_t t0 = blockIdx.x;

    const v4f* wfrag = reinterpret_cast<const v4f*>(wp) + (static_cast<int64_t>(blockIdx.y * 4 + strip) * NBLK * KG + kpart * KGW) * kWave + (tid & 63);
    v4f wreg[NBLK][KGW];
#pragma unroll
    for (int b = 0; b < NBLK; ++b)
#pragma unroll
        for (int gk = 0; gk < KGW; ++gk) wreg[b][gk] = wfrag[(b * KG + gk) * kWave];

    const uint32_t h_row_bytes = static_cast<uint32_t>(ld_h) * 4u;
    const char* hbytes = reinterpret_cast<const char*>(h);
    int tl = tid;
    auto fetch_ids = [&](int k) {
        const int lane = tl & 63;
        if (wave == 0 && lane < 3 * TE) {
            const int64_t pos = (t0 + k * grid) * (3 * TE) + lane;
            ids[k % 3][lane] = pos < n_edges * 3 ? i3[pos] : 0;
        }
    };
    auto issue_dma = [&](Buffer& b, int k) {                             // one 1-KiB piece = one member row
        const int pchunk = tl & 63;
        const int* idk = ids[k % 3];
#pragma unroll
        for (int kk = 0; kk < PIECES; ++kk) {
            const int x = wave * PIECES + kk, m = x >> 4, row = x & 15;
            const int node = idk[row * 3 + m];
            lds_dma16(reinterpret_cast<const float*>(hbytes + static_cast<uint64_t>(static_cast<uint32_t>(node)) * h_row_bytes + ((pchunk ^ row) << 4)),
                      &b.tile[m][row][0]);
        }
    };
    // epilogue (threads 0-255 = waves 0-3): hyperedge row tid >> 4, columns cbase + 4 (tid & 15) .. + 3
    v4f first, first_prev, pr[3];
    auto load_first_order_rows = [&](int k) {
        const int* idk = ids[k % 3] + (tl >> 4) * 3;
        const int ecol = cbase + (tl & 15) * 4;
#pragma unroll
        for (int m = 0; m < 3; ++m) pr[m] = *reinterpret_cast<const v4f*>(p + static_cast<int64_t>(idk[m]) * ld_p + ecol);
    };
    auto write_out = [&](int64_t tile_id, const float (*pimg)[TE][OSTRIDE]) {
        const int erow = tl >> 4, ecol = (tl & 15) * 4;
        const int64_t e = tile_id * TE + erow;
        const v4f sum = (*reinterpret_cast<const v4f*>(&pimg[0][erow][ecol]) + *reinterpret_cast<const v4f*>(&pimg[1][erow][ecol])) + first_prev;
        if (e < n_edges) edge_row_store(out + e * ld_out + cbase + ecol, sum);
    };

    fetch_ids(0);
    if (n_my > 1) fetch_ids(1);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    issue_dma(buf[0], 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const bool late = wave >= 4;                                          // = the second k-half: its service block sits half a phase later
    for (int k = 0; k < n_my; ++k) {
        const int64_t tile_id = t0 + k * grid;
        Buffer& b = buf[k & 1];
        asm volatile("" : "+v"(tl));
        const int arow = tl & 15, kq = (tl >> 4) & 3;
        v4f acc = v4f{0.f, 0.f, 0.f, 0.f};
        int lane_off = (arow * (D * 4) + ((kq ^ arow) << 4)) ^ (kpart * KGW << 6);
        asm volatile("" : "+v"(lane_off));
        const char* tbase = reinterpret_cast<const char*>(&b.tile[0][0][0]);
        auto member_chunk = [&](int m, int gk) { return *reinterpret_cast<const v4f*>(tbase + (lane_off ^ (gk << 6)) + m * TE * (D * 4)); };
        v4f au = member_chunk(0, 0), aq = member_chunk(1, 0), ai = member_chunk(2, 0);
#pragma unroll
        for (int gk = 0; gk < KGW; ++gk) {
            if (gk == 0 || gk == KGW / 2) {
                if (late == (gk != 0)) {
                    if (k + 1 < n_my) issue_dma(buf[(k + 1) & 1], k + 1);
                    if (!late) {
                        if (k > 0) write_out(tile_id - grid, prod[(k - 1) & 1]);
                        load_first_order_rows(k);
                        if (k + 2 < n_my) fetch_ids(k + 2);
                    }
                }
            }
            v4f z[4];
            z[0] = au * aq;
            z[1] = aq * ai;
            z[2] = ai * au;
            z[3] = z[0] * ai;
            __builtin_amdgcn_sched_barrier(0);
            if (gk + 1 < KGW) {
                au = member_chunk(0, gk + 1);
                aq = member_chunk(1, gk + 1);
                ai = member_chunk(2, gk + 1);
            }
            if (gk == 4 && !late) first = (pr[0] + pr[1]) + pr[2];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int bk = 0; bk < NBLK; ++bk)
#pragma unroll
                for (int s2 = 0; s2 < 4; ++s2) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(z[bk][s2], wreg[bk][gk][s2], acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) prod[k & 1][kpart][4 * kq + r][16 * strip + arow] = acc[r];
        first_prev = first;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    asm volatile("" : "+v"(tl));
    if (!late) write_out(t0 + (n_my - 1) * grid, prod[(n_my - 1) & 1]);
}

template <int NBLK>
__global__ __launch_bounds__(kWsThreads) void interact_bwd_members_strip256_kernel(
    const float* __restrict__ h, int64_t ld_h, const int32_t* __restrict__ i3, const float* __restrict__ wq,
    const float* __restrict__ dout, int64_t ld_dout, float* __restrict__ g_out, int64_t n_edges) {
    constexpr int D = 256, TE = kStrip256TE, CW = 64, KG = D / 16, KGW = KG / 2;
    struct Buffer {
        float dtile[TE][D];              // dout rows, whole (the contraction runs over all 256 columns), swizzled
        float htile[3][TE][CW];          // member rows, this workgroup's column quarter only
    };
    __shared__ __attribute__((aligned(16))) Buffer buf[2];
    __shared__ __attribute__((aligned(16))) float gimg[2][2][3][TE][CW];  // [tile parity][k-half]: partial member gradients
    __shared__ int ids[3][3 * TE];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int strip = wave & 3, kpart = wave >> 2;
    const int cbase = blockIdx.y * CW;
    const int64_t n_tiles = (n_edges + TE - 1) / TE;
    const int64_t grid = gridDim.x;
    const int n_my = blockIdx.x < n_tiles ? static_cast<int>((n_tiles - blockIdx.x + grid - 1) / grid) : 0;
    if (n_my == 0) return;
    const int64_t t0 = blockIdx.x;

    const v4f* wfrag = reinterpret_cast<const v4f*>(wq) + (static_cast<int64_t>(blockIdx.y * 4 + strip) * NBLK * KG + kpart * KGW) * kWave + (tid & 63);
    v4f wreg[NBLK][KGW];
#pragma unroll
    for (int b = 0; b < NBLK; ++b)
#pragma unroll
        for (int gk = 0; gk < KGW; ++gk) wreg[b][gk] = wfrag[(b * KG + gk) * kWave];

    const uint32_t h_row_bytes = static_cast<uint32_t>(ld_h) * 4u;
    int tl = tid;
    auto fetch_ids = [&](int k) {
        const int lane = tl & 63;
        if (wave == 0 && lane < 3 * TE) {
            const int64_t pos = (t0 + k * grid) * (3 * TE) + lane;
            ids[k % 3][lane] = pos < n_edges * 3 ? i3[pos] : 0;
        }
    };
    // 28 pieces per tile: 16 dout rows (1 KiB each) and 12 pieces of 4 member-row quarters (4 x 256 B); piece x = wave + 8 kk
    auto issue_dma = [&](Buffer& b, int k) {
        const int64_t tile_id = t0 + k * grid;
        const int lane = tl & 63;
        const int* idk = ids[k % 3];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int x = wave + 8 * kk;
            if (x < TE) {
                int64_t e = tile_id * TE + x;
                e = e < n_edges ? e : n_edges - 1;
                lds_dma16(dout + e * ld_dout + ((lane ^ (x & 15)) << 2), &b.dtile[x][0]);
            } else if (x < TE + 12) {
                const int y = x - TE, m = y >> 2, row = 4 * (y & 3) + (lane >> 4);
                const int node = idk[row * 3 + m];
                lds_dma16(reinterpret_cast<const float*>(reinterpret_cast<const char*>(h) + static_cast<uint64_t>(static_cast<uint32_t>(node)) * h_row_bytes +
                                                         (cbase + (lane & 15) * 4) * 4), &b.htile[m][4 * (y & 3)][0]);
            }
        }
    };
    // finished tile: sum of the two k-halves, 768 16-byte vectors -> threads take idx = tid and tid + 512 (< 768)
    auto store_out = [&](int64_t tile_id, const float (*img)[3][TE][CW]) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int idx = tl + 512 * half;
            if (idx < 3 * TE * (CW / 4)) {
                const int m = idx / (TE * (CW / 4)), row = (idx / (CW / 4)) % TE, c4 = idx % (CW / 4);
                const v4f v = *reinterpret_cast<const v4f*>(&img[0][m][row][c4 * 4]) + *reinterpret_cast<const v4f*>(&img[1][m][row][c4 * 4]);
                const int64_t e = tile_id * TE + row;
                if (e < n_edges) store_stream4(g_out + e * (3 * D) + m * D + cbase + c4 * 4, v);
            }
        }
    };

    fetch_ids(0);
    if (n_my > 1) fetch_ids(1);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    issue_dma(buf[0], 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const bool late = wave >= 4;
    for (int k = 0; k < n_my; ++k) {
        Buffer& b = buf[k & 1];
        asm volatile("" : "+v"(tl));
        const int arow = tl & 15, kq = (tl >> 4) & 3;
        const int col = 16 * strip + arow;
        v4f acc[NBLK];
#pragma unroll
        for (int bk = 0; bk < NBLK; ++bk) acc[bk] = v4f{0.f, 0.f, 0.f, 0.f};
        int lane_off = (arow * (D * 4) + ((kq ^ arow) << 4)) ^ (kpart * KGW << 6);
        asm volatile("" : "+v"(lane_off));
        const char* dbase = reinterpret_cast<const char*>(&b.dtile[0][0]);
        v4f a = *reinterpret_cast<const v4f*>(dbase + lane_off);
#pragma unroll
        for (int gk = 0; gk < KGW; ++gk) {
            if (gk == 0 || gk == KGW / 2) {
                if (late == (gk != 0)) {
                    if (k + 1 < n_my) issue_dma(buf[(k + 1) & 1], k + 1);
                    if (k > 0) store_out(t0 + (k - 1) * grid, gimg[(k - 1) & 1]);
                    if (k + 2 < n_my) fetch_ids(k + 2);
                }
            }
            const v4f a_now = a;
            __builtin_amdgcn_sched_barrier(0);
            if (gk + 1 < KGW) a = *reinterpret_cast<const v4f*>(dbase + (lane_off ^ ((gk + 1) << 6)));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2)
#pragma unroll
                for (int bk = 0; bk < NBLK; ++bk) acc[bk] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_now[s2], wreg[bk][gk][s2], acc[bk], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        // product rule on this wave's PARTIAL contractions (it is linear in them): partial member gradients of its (row, column) elements
        float (*img)[TE][CW] = gimg[k & 1][kpart];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int er = 4 * kq + r;
            const float hu = b.htile[0][er][col], hq = b.htile[1][er][col], hi = b.htile[2][er][col];
            const float z_uq = acc[0][r], z_qi = acc[1][r], z_iu = acc[2][r];
            const float z_uqi = NBLK == 4 ? acc[NBLK - 1][r] : 0.f;
            img[0][er][col] = z_uq * hq + z_iu * hi + z_uqi * (hq * hi);
            img[1][er][col] = z_uq * hu + z_qi * hi + z_uqi * (hu * hi);
            img[2][er][col] = z_qi * hq + z_iu * hu + z_uqi * (hu * hq);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    asm volatile("" : "+v"(tl));
    store_out(t0 + (n_my - 1) * grid, gimg[(n_my - 1) & 1]);
}

// Weight gradient in the strip style (D = 128): dW[j][(3+b)d + c] = sum_e dout[e][j] z_b[e][c] - the contraction index is the
// hyperedge, so the whole [d x NBLK d] gradient of a workgroup's hyperedges stays in accumulators for the entire sweep: wave w owns
// rows 16 w .. 16 w + 15 and all NBLK * d columns (NBLK * 8 accumulator tiles = 128 VGPRs), there is no per-tile epilogue and
// nothing leaves the CU until the end (one [d x NBLK d] slab per workgroup, summed by slab_reduce_kernel in a fixed order).
// MFMA step (g, s) takes the 4 hyperedges 16 g + 4 s + kq (kq = lane >> 4): A = dout[e][16 w + (lane & 15)] (one ds_read_b32 for
// 32 MFMAs), B = z_b[e][8 c + m] with c = lane & 15 for accumulator tile (b, m): a lane reads columns 8 c .. 8 c + 7 of the three
// member rows of ITS hyperedge (two ds_read_b128 each) and forms the 8 x NBLK products from them.  Member rows are stored with
// 16-byte chunk p at position p ^ (row & 1): the four rows of a step then cover all 64 banks in every ds_read_b128 phase.
template <int D, int NBLK>
__global__ __launch_bounds__(kWsThreads) void interact_bwd_weight_strip_kernel(
    const float* __restrict__ h, int64_t ld_h, const int32_t* __restrict__ i3, const float* __restrict__ dout, int64_t ld_dout,
    float* __restrict__ slabs, int64_t n_edges) {
    static_assert(D == 128, "eight 16-row strips");
    constexpr int TE = kStripTE, V4 = D / 4;
    constexpr int ROWS_PER_PIECE = kWave / V4;                            // 2
    constexpr int PIECES_PER_MEMBER = TE / ROWS_PER_PIECE;                // 16
    constexpr int HP = 3 * PIECES_PER_MEMBER / 8;                         // 6 member-row pieces per wave per tile
    constexpr int DP = PIECES_PER_MEMBER / 8;                             // 2 dout pieces per wave per tile
    constexpr int MT = D / 16;                                            // 8 accumulator tiles per product block
    struct Buffer {
        float dtile[TE][D];
        float mtile[3][TE][D];
    };
    __shared__ __attribute__((aligned(16))) Buffer buf[2];
    __shared__ int ids[3][3 * TE];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t n_tiles = (n_edges + TE - 1) / TE;
    const int64_t grid = gridDim.x;
    const int n_my = blockIdx.x < n_tiles ? static_cast<int>((n_tiles - blockIdx.x + grid - 1) / grid) : 0;
    const int64_t t0 = blockIdx.x;
    const uint32_t h_row_bytes = static_cast<uint32_t>(ld_h) * 4u;
    int tl = tid;
    auto fetch_ids = [&](int k) {                                        // see interact_fwd_strip_kernel
        const int lane = tl & 63;
        if (wave < 2 && (wave == 0 || lane < 32)) {
            const int j = wave * 64 + lane;
            const int64_t pos = (t0 + k * grid) * (3 * TE) + j;
            ids[k % 3][j] = pos < n_edges * 3 ? i3[pos] : 0;
        }
    };
    auto issue_dma = [&](Buffer& b, int k) {
        const int64_t tile_id = t0 + k * grid;
        const int prow = (tl >> 5) & 1, pchunk = tl & 31;
        const int* idk = ids[k % 3];
#pragma unroll
        for (int kk = 0; kk < DP; ++kk) {                                // dout rows as they lie; rows past the end re-read the last one (masked at use)
            const int r0 = 2 * (wave * DP + kk);
            int64_t e = tile_id * TE + r0 + prow;
            e = e < n_edges ? e : n_edges - 1;
            lds_dma16(dout + e * ld_dout + pchunk * 4, &b.dtile[r0][0]);
        }
#pragma unroll
        for (int kk = 0; kk < HP; ++kk) {
            const int x = wave * HP + kk;
            const int r0 = 2 * (x & (PIECES_PER_MEMBER - 1));
            const int node = idk[(r0 + prow) * 3 + (x >> 4)];
            const char* src = reinterpret_cast<const char*>(h) + static_cast<uint64_t>(static_cast<uint32_t>(node)) * h_row_bytes + ((pchunk ^ prow) << 4);
            lds_dma16(reinterpret_cast<const float*>(src), &b.mtile[x >> 4][r0][0]);     // r0 even: (row & 1) == prow
        }
    };
    v4f acc[NBLK][MT];
#pragma unroll
    for (int bk = 0; bk < NBLK; ++bk)
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[bk][m] = v4f{0.f, 0.f, 0.f, 0.f};

    if (n_my > 0) {
        fetch_ids(0);
        if (n_my > 1) fetch_ids(1);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __syncthreads();
        issue_dma(buf[0], 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    const bool late = wave >= 4;
    for (int k = 0; k < n_my; ++k) {
        const int64_t tile_id = t0 + k * grid;
        const Buffer& b = buf[k & 1];
        asm volatile("" : "+v"(tl));
        const int c = tl & 15, kq = (tl >> 4) & 3;
        const bool full = tile_id * TE + TE <= n_edges;
        // per step the lane's hyperedge row is 4 step + kq: its dout element and the 32 bytes 8 c .. 8 c + 7 of each member row
        int a_off = kq * (D * 4) + (16 * wave + c) * 4;
        int m_off = kq * (D * 4) + (((2 * c) ^ (kq & 1)) << 4);          // chunk 2 c of a row of parity kq & 1; chunk 2 c + 1 sits at the position ^ 1
        asm volatile("" : "+v"(a_off), "+v"(m_off));
        const char* dbase = reinterpret_cast<const char*>(&b.dtile[0][0]);
        const char* mbase = reinterpret_cast<const char*>(&b.mtile[0][0][0]);
#pragma unroll
        for (int step = 0; step < TE / 4; ++step) {
            if (step == 0 || step == TE / 8) {
                if (late == (step != 0)) {                                  // service block, half a phase apart on the two waves of a SIMD
                    if (k + 1 < n_my) issue_dma(buf[(k + 1) & 1], k + 1);
                    if (k + 2 < n_my) fetch_ids(k + 2);
                }
            }
            const int row_bytes = step * 4 * (D * 4);
            float a = *reinterpret_cast<const float*>(dbase + a_off + row_bytes);
            if (!full && tile_id * TE + 4 * step + kq >= n_edges) a = 0.f;  // rows past the end contribute exact zeros
            v4f mu[2], mq[2], mi[2];
#pragma unroll
            for (int x = 0; x < 2; ++x) {
                mu[x] = *reinterpret_cast<const v4f*>(mbase + ((m_off + row_bytes) ^ (x << 4)));
                mq[x] = *reinterpret_cast<const v4f*>(mbase + ((m_off + row_bytes) ^ (x << 4)) + TE * D * 4);
                mi[x] = *reinterpret_cast<const v4f*>(mbase + ((m_off + row_bytes) ^ (x << 4)) + 2 * TE * D * 4);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int bk = 0; bk < NBLK; ++bk) {
                v4f z[2];
#pragma unroll
                for (int x = 0; x < 2; ++x)
                    z[x] = bk == 0 ? mu[x] * mq[x] : bk == 1 ? mq[x] * mi[x] : bk == 2 ? mi[x] * mu[x] : (mu[x] * mq[x]) * mi[x];
#pragma unroll
                for (int m = 0; m < MT; ++m) acc[bk][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, z[m >> 2][m & 3], acc[bk][m], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    // slab of this workgroup: element (j, b * D + col); accumulator tile (b, m): column lane & 15 -> col 8 c + m, rows 4 kq + r
    float* slab = slabs + static_cast<int64_t>(blockIdx.x) * D * NBLK * D;
    const int c = tid & 15, kq = (tid >> 4) & 3;
#pragma unroll
    for (int bk = 0; bk < NBLK; ++bk)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                slab[static_cast<int64_t>(16 * wave + 4 * kq + r) * NBLK * D + bk * D + 8 * c + m] = acc[bk][m][r];
}

// dW, same roles: workgroup (x, y) owns the 64 x 64 x NBLK sub-block y = (js, cs) of the d x NBLK*d gradient (d a multiple of 64)
// for the hyperedge tiles x, x + gridDim.x, ...; its loaders fetch the matching 64-column slices of dout and of the member rows.
template <int NBLK>
__global__ __launch_bounds__(kWsThreads, 2) void interact_bwd_weight_ws_kernel(
    const float* __restrict__ h, int64_t ld_h, const int32_t* __restrict__ i3, const float* __restrict__ dout, int64_t ld_dout,
    float* __restrict__ slabs, int64_t n_edges, int d) {
    constexpr int D = 64, TE = 64, V4 = D / 4;
    constexpr int RPP = kWave / V4;                       // rows per 1-KiB DMA piece
    constexpr int QP = TE / 4 / RPP;                      // pieces per row block of one loader wave
    // Both tiles are read by the consumers one float per lane with the lanes running over columns (conflict-free as they lie):
    // plain row images, filled by DMA.  Trip structure as in the forward kernel, with nothing to hand back - the accumulators
    // stay in the consumers' registers for the whole sweep.
    struct Buffer {
        float dtile[TE][D];
        float mtile[3][TE][D];
    };
    __shared__ __attribute__((aligned(16))) Buffer buf[2];
    const int64_t n_tiles = (n_edges + TE - 1) / TE;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int subs = d / D;
    const int js = blockIdx.y / subs, cs = blockIdx.y % subs;
    const int64_t grid = gridDim.x;
    const int n_my = blockIdx.x < n_tiles ? static_cast<int>((n_tiles - blockIdx.x + grid - 1) / grid) : 0;

    if (wave >= 4) {
        // loaders: start the DMA of tile k (ids fetched a trip ago) -> fetch the ids of tile k+1 -> barrier (waits for the DMA)
        const int lw = __builtin_amdgcn_readfirstlane(wave) - 4;
        const int sub = lane / V4, chunk = lane % V4;
        const int wrow0 = lw * (TE / 4), row0 = wrow0 + sub;
        const uint32_t h_row_bytes = static_cast<uint32_t>(ld_h) * 4u;
        const char* hsrc = reinterpret_cast<const char*>(h + cs * D) + chunk * 16;
        const float* dsrc = dout + js * D + chunk * 4;
        int node[3 * QP];
        auto load_ids = [&](int64_t tile_id) {
            const int64_t e0 = tile_id * TE + row0;
            const int32_t* idp = i3 + e0 * 3;
            if (tile_id * TE + TE <= n_edges) {
#pragma unroll
                for (int x = 0; x < 3 * QP; ++x) node[x] = idp[(x % QP) * RPP * 3 + x / QP];
            } else {
#pragma unroll
                for (int x = 0; x < 3 * QP; ++x) node[x] = e0 + (x % QP) * RPP < n_edges ? idp[(x % QP) * RPP * 3 + x / QP] : -1;
            }
        };
        auto start_loads = [&](Buffer& b, int64_t tile_id) {
            const int64_t e0 = tile_id * TE + row0;
            const bool full = tile_id * TE + TE <= n_edges;
#pragma unroll
            for (int q = 0; q < QP; ++q) {
                const int64_t e = e0 + q * RPP;
                if (full || e < n_edges) lds_dma16(dsrc + e * ld_dout, &b.dtile[wrow0 + q * RPP][0]);
            }
#pragma unroll
            for (int x = 0; x < 3 * QP; ++x)
                if (full || node[x] >= 0)
                    lds_dma16(reinterpret_cast<const float*>(hsrc + static_cast<uint64_t>(static_cast<uint32_t>(node[x])) * h_row_bytes),
                              &b.mtile[x / QP][wrow0 + (x % QP) * RPP][0]);
            if (!full) {
                // rows past the end must contribute exact zeros to the contraction over hyperedges: lanes of missing rows (EXEC-masked
                // out of the DMA above) clear their 16 bytes of the dout image by hand
#pragma unroll
                for (int q = 0; q < QP; ++q)
                    if (e0 + q * RPP >= n_edges) *reinterpret_cast<v4f*>(&b.dtile[row0 + q * RPP][chunk * 4]) = v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int x = 0; x < 3 * QP; ++x)
                    if (node[x] < 0) *reinterpret_cast<v4f*>(&b.mtile[x / QP][row0 + (x % QP) * RPP][chunk * 4]) = v4f{0.f, 0.f, 0.f, 0.f};
            }
        };
        const int64_t t0 = blockIdx.x;
        if (n_my > 0) load_ids(t0);
        for (int k = 0; k <= n_my; ++k) {
            if (k < n_my) start_loads(buf[k & 1], t0 + k * grid);
            if (k + 1 < n_my) load_ids(t0 + (k + 1) * grid);
            __syncthreads();
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
    __syncthreads();
    for (int k = 0; k < n_my; ++k) {
        const Buffer& b = buf[k & 1];
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
        __syncthreads();
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

constexpr int kPipeGridSlabs = 256;
inline int64_t packed_weight_floats(int dim, int order) { return static_cast<int64_t>(order == 3 ? 4 : 3) * dim * dim; }
inline int weight_slabs(int dim) {
    if (dim == 128) return kPipeGridSlabs;                  // interact_bwd_weight_strip_kernel: one full slab per workgroup
    const int subs = dim >= 64 ? (dim / 64) * (dim / 64) : 1;
    int n = 512 / subs;
    return n < 8 ? 8 : n;
}
constexpr int kFwdGrid = 256 * 3;
constexpr int kPipeGrid = 256;          // wave-specialised and strip kernels: one 512-thread workgroup per CU
// tile ranges of the user-reduced member-gradient kernels (two boundary-table entries each): one per workgroup at d = 64 / 128, one per wave at d = 32 (narrow.hip)
inline int64_t boundary_ranges(int dim) { return dim == 128 || dim == 64 || dim == 256 ? kPipeGrid : (dim == kNarrowDim ? kNarrowMemberRanges : 0); }

// Persistent grid of a plain (one role) tiling: as many workgroups as are resident at once - a larger grid runs in rounds, and
// the workgroups of the last round start when the others have already walked their whole share of the tiles.
template <typename Kernel>
int resident_grid(Kernel kernel) {
    int per_cu = 0, device = 0, cus = 256;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, kBlockThreads, 0) != hipSuccess || per_cu < 1) return kFwdGrid;
    hipDeviceProp_t prop;
    if (hipGetDevice(&device) == hipSuccess && hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0)
        cus = prop.multiProcessorCount;
    return per_cu * cus;
}


// the strip kernels (D = 128) move rows as 16-byte vectors and form addresses as 32 x 32-bit products
inline bool strip_fwd_ok(int dim, const float* p, int64_t ld_p, const float* out, int64_t ld_out, int64_t ld_h) {
    return (dim == 128 || dim == 256) && aligned16(p) && aligned16(out) && ld_p % 4 == 0 && ld_out % 4 == 0 && ld_h < (int64_t{1} << 30);
}
inline bool strip_bwd_ok(int dim, const float* g, int64_t ld_h) { return (dim == 128 || dim == 256) && aligned16(g) && ld_h < (int64_t{1} << 30); }

template <int NBLK>
void launch_interact_fwd_mfma(int dim, const float* h, int64_t ld_h, const float* p, int64_t ld_p, const int32_t* i3, const float* wp,
                              float* out, int64_t ld_out, int64_t n_edges, hipStream_t s) {
    if (strip_fwd_ok(dim, p, ld_p, out, ld_out, ld_h)) {                    // wp is strip-packed (the caller asked strip_fwd_ok too)
        if (dim == 256) {
            const int grid = static_cast<int>(std::min<int64_t>((n_edges + kStrip256TE - 1) / kStrip256TE, kStrip256Grid));
            hipLaunchKernelGGL((interact_fwd_strip256_kernel<NBLK>), dim3(grid, 4), dim3(kWsThreads), 0, s, h, ld_h, p, ld_p, i3, wp, out, ld_out, n_edges);
            return;
        }
        const int grid = static_cast<int>(std::min<int64_t>((n_edges + kStripTE - 1) / kStripTE, kPipeGrid));
        hipLaunchKernelGGL((interact_fwd_strip_kernel<128, NBLK>), dim3(grid), dim3(kWsThreads), 0, s, h, ld_h, p, ld_p, i3, wp, out, ld_out, n_edges);
        return;
    }
#define IHG_FWD(D)                                                                                                          \
    {                                                                                                                       \
        const int64_t tiles = (n_edges + TileShape<D>::TE - 1) / TileShape<D>::TE;                                          \
        static const int resident = resident_grid(interact_fwd_mfma_kernel<D, NBLK>);                                       \
        const int grid = static_cast<int>(std::min<int64_t>(tiles, resident));                                              \
        hipLaunchKernelGGL((interact_fwd_mfma_kernel<D, NBLK>), dim3(grid), dim3(kBlockThreads), 0, s, h, ld_h, p, ld_p, i3, wp, out, ld_out, n_edges); \
    }
#define IHG_FWD_PIPE(D)                                                                                                     \
    {                                                                                                                       \
        const int64_t tiles = (n_edges + TileShape<D>::TE - 1) / TileShape<D>::TE;                                          \
        const int grid = static_cast<int>(std::min<int64_t>(tiles, kPipeGrid));                                             \
        hipLaunchKernelGGL((interact_fwd_ws_kernel<D, NBLK>), dim3(grid), dim3(kWsThreads), 0, s, h, ld_h, p, ld_p, i3, wp, out, ld_out, n_edges); \
    }
    // the wave-specialised form moves first-order rows and results as 16-byte vectors and forms addresses as 32x32-bit products
    const bool vector_io = aligned16(p) && aligned16(out) && ld_p % 4 == 0 && ld_out % 4 == 0 && ld_h < (int64_t{1} << 30) && ld_p < (int64_t{1} << 30);
    switch (dim) {
        case 32:
            if (vector_io) IHG_FWD_PIPE(32) else IHG_FWD(32)
            break;
        case 64:
            if (vector_io) IHG_FWD_PIPE(64) else IHG_FWD(64)
            break;
        case 128: IHG_FWD(128) break;
        default: IHG_FWD(256) break;
    }
#undef IHG_FWD
#undef IHG_FWD_PIPE
}

template <int NBLK>
void launch_interact_bwd_mfma(int dim, const float* h, int64_t ld_h, const int32_t* i3, const float* wq, const float* dout, int64_t ld_dout,
                              float* g, float* slabs, float* dw, int64_t ld_dw, int64_t n_edges, hipStream_t s,
                              float* dh_user = nullptr, int64_t ld_dh = 0, float* bnd_val = nullptr, int32_t* bnd_user = nullptr,
                              const float* w_raw = nullptr, int64_t ld_w = 0, void* planes = nullptr,
                              const float* dy_scale = nullptr, float* dout_store = nullptr, int64_t ld_store = 0) {
    // dout_store != nullptr (the caller has checked the split kernels take the shape): `dout` is the node-level cotangent, the member-gradient
    // kernel forms the hyperedges' cotangents from it and leaves them in dout_store for everything after it
#define IHG_MEM(D)                                                                                                          \
    {                                                                                                                       \
        constexpr int TE = D == 32 ? 128 : 64;                                                                              \
        static const int resident = resident_grid(interact_bwd_members_mfma_kernel<D, NBLK>);                               \
        const int grid = static_cast<int>(std::min<int64_t>((n_edges + TE - 1) / TE, resident));                            \
        hipLaunchKernelGGL((interact_bwd_members_mfma_kernel<D, NBLK>), dim3(grid), dim3(kBlockThreads), 0, s, h, ld_h, i3, wq, dout, ld_dout, g, n_edges); \
    }
#define IHG_MEM_PIPE(D)                                                                                                     \
    {                                                                                                                       \
        constexpr int TE = D == 32 ? 128 : 64;                                                                              \
        const int grid = static_cast<int>(std::min<int64_t>((n_edges + TE - 1) / TE, kPipeGrid));                           \
        hipLaunchKernelGGL((interact_bwd_members_ws_kernel<D, NBLK>), dim3(grid), dim3(kWsThreads), 0, s, h, ld_h, i3, wq, dout, ld_dout, g, n_edges); \
    }
    const bool vector_io = aligned16(g) && ld_h < (int64_t{1} << 30);     // the pipelined form stores g as 16-byte vectors
    if (dim == kNarrowDim && dh_user != nullptr && planes != nullptr && narrow_members_ok(dim, NBLK == 4 ? 3 : 2, g, ld_h, ld_dout, dout)) {
        // d = 32, user-reduced (g is [E, 2, d]): one wave per 16-hyperedge tile on fp32 MFMA, gathering the node-level cotangent when dy_scale / dout_store say so
        const bool gather = dout_store != nullptr || ld_store < 0;
        int entries = 0;
        launch_members_narrow(NBLK == 4 ? 3 : 2, gather ? 1 : 0, h, ld_h, i3, w_raw, ld_w, static_cast<float*>(planes), dout, ld_dout, dy_scale, ld_store > 0 ? dout_store : nullptr,
                              ld_store, g, n_edges, dh_user, ld_dh, bnd_val, bnd_user, &entries, s);
        if (gather && ld_store > 0) {
            dout = dout_store;
            ld_dout = ld_store;
        }
    } else if (planes != nullptr && split_members_ok(dim, NBLK == 4 ? 3 : 2, g, ld_h, ld_dout, dout, dh_user != nullptr)) {   // bf16-split contraction
        int entries = 0;
        launch_members_split(dim, NBLK == 4 ? 3 : 2, h, ld_h, i3, w_raw, ld_w, planes, dout, ld_dout, g, n_edges, dh_user, ld_dh, bnd_val, bnd_user, &entries, s,
                             dy_scale, dout_store, ld_store);
        if (dh_user != nullptr) hipLaunchKernelGGL(user_boundary_fixup_kernel, dim3(entries), dim3(std::max(128, dim)), 0, s, bnd_val, bnd_user, entries, dim, dh_user, ld_dh);
        if (dout_store != nullptr) {
            dout = dout_store;
            ld_dout = ld_store;
        }
    } else if (strip_bwd_ok(dim, g, ld_h) && dim == 256) {               // wq is strip-packed
        const int grid = static_cast<int>(std::min<int64_t>((n_edges + kStrip256TE - 1) / kStrip256TE, kStrip256Grid));
        hipLaunchKernelGGL((interact_bwd_members_strip256_kernel<NBLK>), dim3(grid, 4), dim3(kWsThreads), 0, s, h, ld_h, i3, wq, dout, ld_dout, g, n_edges);
    } else if (strip_bwd_ok(dim, g, ld_h) && dh_user != nullptr) {       // user-reduced form: g is [E, 2, d]
        const int grid = static_cast<int>(std::min<int64_t>((n_edges + kStripTE - 1) / kStripTE, kPipeGrid));
        hipLaunchKernelGGL((interact_bwd_members_strip_kernel<128, NBLK, true>), dim3(grid), dim3(kWsThreads), 0, s, h, ld_h, i3, wq, dout, ld_dout, g, n_edges,
                           dh_user, ld_dh, bnd_val, bnd_user);
        hipLaunchKernelGGL(user_boundary_fixup_kernel, dim3(2 * grid), dim3(128), 0, s, bnd_val, bnd_user, 2 * grid, dim, dh_user, ld_dh);
    } else if (strip_bwd_ok(dim, g, ld_h)) {
        const int grid = static_cast<int>(std::min<int64_t>((n_edges + kStripTE - 1) / kStripTE, kPipeGrid));
        hipLaunchKernelGGL((interact_bwd_members_strip_kernel<128, NBLK, false>), dim3(grid), dim3(kWsThreads), 0, s, h, ld_h, i3, wq, dout, ld_dout, g, n_edges,
                           static_cast<float*>(nullptr), int64_t{0}, static_cast<float*>(nullptr), static_cast<int32_t*>(nullptr));
    } else
    switch (dim) {
        case 32:
            if (vector_io) IHG_MEM_PIPE(32) else IHG_MEM(32)
            break;
        case 64:
            if (vector_io) IHG_MEM_PIPE(64) else IHG_MEM(64)
            break;
        case 128: {
            const int grid = static_cast<int>(std::min<int64_t>((n_edges + 63) / 64, kPipeGrid));
            hipLaunchKernelGGL((interact_bwd_members_wsbig_kernel<128, NBLK>), dim3(grid), dim3(kWsThreads), 0, s, h, ld_h, i3, wq, dout, ld_dout, g, n_edges);
        } break;
        default: IHG_MEM(256) break;
    }
#undef IHG_MEM
#undef IHG_MEM_PIPE
    if (dw == nullptr) return;                               // member gradients only (the caller takes d w from the node-level kernel)
    const int subs_ws = (dim / 64) * (dim / 64);
    int n_slabs = static_cast<int>(std::min<int64_t>(dim >= 64 ? std::max(kPipeGrid / subs_ws, 8) : weight_slabs(dim), (n_edges + 63) / 64));
    if (split_weight_ok(dim, NBLK == 4 ? 3 : 2, ld_h, ld_dout, dout)) {      // bf16-split contraction
        n_slabs = launch_weight_split(dim, NBLK == 4 ? 3 : 2, h, ld_h, i3, dout, ld_dout, slabs, n_edges, s);
    } else if (dim == 128 && ld_h < (int64_t{1} << 30)) {
        n_slabs = static_cast<int>(std::min<int64_t>((n_edges + kStripTE - 1) / kStripTE, kPipeGridSlabs));
        hipLaunchKernelGGL((interact_bwd_weight_strip_kernel<128, NBLK>), dim3(n_slabs), dim3(kWsThreads), 0, s, h, ld_h, i3, dout, ld_dout, slabs, n_edges);
    } else if (dim >= 64 && ld_h < (int64_t{1} << 30)) {
        hipLaunchKernelGGL((interact_bwd_weight_ws_kernel<NBLK>), dim3(n_slabs, subs_ws), dim3(kWsThreads), 0, s, h, ld_h, i3, dout, ld_dout, slabs, n_edges, dim);
    } else if (dim >= 64) {                                  // row strides beyond 32-bit byte offsets: the plain tiling
        hipLaunchKernelGGL((interact_bwd_weight_mfma_kernel<64, NBLK>), dim3(n_slabs, subs_ws), dim3(kBlockThreads), 0, s, h, ld_h, i3, dout, ld_dout, slabs, n_edges, dim);
    } else {
        hipLaunchKernelGGL((interact_bwd_weight_mfma_kernel<32, NBLK>), dim3(n_slabs, 1), dim3(kBlockThreads), 0, s, h, ld_h, i3, dout, ld_dout, slabs, n_edges, dim);
    }
    const int total = dim * NBLK * dim;
    hipLaunchKernelGGL(slab_reduce_kernel, dim3((total + kWave - 1) / kWave), dim3(kBlockThreads), 0, s, slabs, n_slabs, dim, NBLK, dw, ld_dw);
}

}  // namespace

extern "C" {

int64_t ihg_interact_fwd_workspace_bytes(int64_t n_edges, int32_t dim, int32_t order) {
    (void)n_edges;
    if (!mfma_dim(dim) || (order != 2 && order != 3)) return 0;
    return (packed_weight_floats(dim, order) + split_plane_floats(dim, order)) * static_cast<int64_t>(sizeof(float));
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
        if (split_fwd_ok(dim, order, p, ld_p, out, ld_out, ld_h)) {       // bf16-split contraction; its planes sit behind the fp32-packed weights
            launch_fwd_split(dim, order, h, ld_h, p, ld_p, i3, w, ld_w, wp + packed_weight_floats(dim, order), out, ld_out, n_edges, s);
            return check_launch("ihg_interact_fwd");
        }
        const int pack_items = (dim / 32) * nblk * (dim / 8) * kWave;
        if (strip_fwd_ok(dim, p, ld_p, out, ld_out, ld_h))
            hipLaunchKernelGGL(pack_weights_strip_kernel, dim3((pack_items + kBlockThreads - 1) / kBlockThreads), dim3(kBlockThreads), 0, s, w, ld_w, dim, nblk, wp,
                               static_cast<float*>(nullptr));
        else
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

int32_t ihg_node_interact_fwd_supported(int32_t dim, int32_t order, int64_t ld_h, int64_t ld_sums, int64_t ld_out) {
    static float aligned_probe __attribute__((aligned(16)));
    return (narrow_node_fwd_ok(dim, order, ld_h, ld_sums, &aligned_probe, ld_out, nullptr) || split_node_fwd_ok(dim, order, ld_h, ld_sums, &aligned_probe, ld_out, nullptr)) &&
           ld_h >= dim && ld_sums >= 3LL * dim && ld_out >= dim ? 1 : 0;
}

int64_t ihg_node_interact_fwd_workspace_bytes(int32_t dim) {
    return (dim == kNarrowDim ? narrow_node_fwd_floats() : split_node_fwd_plane_floats(dim)) * static_cast<int64_t>(sizeof(float));
}

int ihg_node_interact_fwd(const float* h, int64_t ld_h, const float* sums, int64_t ld_sums, const float* degree, const float* out_scale, const float* bias,
                          const float* w, int64_t ld_w, int32_t order, const int64_t* type_begin, float* out, int64_t ld_out, void* workspace,
                          int64_t workspace_bytes, int32_t dim, ihg_stream_t stream) {
    if (order != 2 && order != 3) return fail(IHG_ERR_INVALID, "ihg_node_interact_fwd: order must be 2 or 3, got %d", order);
    if (type_begin == nullptr || type_begin[0] != 0 || type_begin[1] < type_begin[0] || type_begin[2] < type_begin[1] || type_begin[3] < type_begin[2])
        return fail(IHG_ERR_INVALID, "ihg_node_interact_fwd: bad type_begin");
    if (dim <= 0 || ld_h < dim || ld_out < dim || ld_sums < 3LL * dim || ld_w < static_cast<int64_t>(order == 3 ? 7 : 6) * dim)
        return fail(IHG_ERR_INVALID, "ihg_node_interact_fwd: bad size");
    if (type_begin[3] == 0) return IHG_OK;
    if (h == nullptr || sums == nullptr || degree == nullptr || w == nullptr || out == nullptr) return fail(IHG_ERR_INVALID, "ihg_node_interact_fwd: null pointer");
    const bool narrow = narrow_node_fwd_ok(dim, order, ld_h, ld_sums, out, ld_out, bias);
    if ((!narrow && !split_node_fwd_ok(dim, order, ld_h, ld_sums, out, ld_out, bias)) || !aligned16(h) || !aligned16(sums))
        return fail(IHG_ERR_INVALID, "ihg_node_interact_fwd: dim %d / order %d / alignment not supported (ihg_node_interact_fwd_supported)", dim, order);
    if (workspace == nullptr || !aligned16(workspace) || workspace_bytes < ihg_node_interact_fwd_workspace_bytes(dim))
        return fail(IHG_ERR_WORKSPACE, "ihg_node_interact_fwd: workspace too small");
    if (narrow) {                                                        // d = 32: plain fp32 MFMA, one wave per 16-row tile (narrow.hip)
        launch_node_fwd_narrow(order, h, ld_h, sums, ld_sums, degree, out_scale, bias, w, ld_w, type_begin, out, ld_out, static_cast<float*>(workspace), static_cast<hipStream_t>(stream));
        return check_launch("ihg_node_interact_fwd");
    }
    launch_node_fwd_split(dim, order, h, ld_h, sums, ld_sums, degree, out_scale, bias, w, ld_w, type_begin, out, ld_out, workspace, static_cast<hipStream_t>(stream));
    return check_launch("ihg_node_interact_fwd");
}

int32_t ihg_node_interact_bwd_weight_supported(int32_t dim, int32_t order, int64_t ld_h, int64_t ld_sums, int64_t ld_dy) {
    static float aligned_probe __attribute__((aligned(16)));
    return (narrow_node_weight_ok(dim, order, ld_h, ld_sums, ld_dy, &aligned_probe) || split_node_weight_ok(dim, order, ld_h, ld_sums, ld_dy, &aligned_probe)) && ld_h >= dim &&
           ld_sums >= 3LL * dim && ld_dy >= dim ? 1 : 0;
}

int64_t ihg_node_interact_bwd_weight_workspace_bytes(int32_t dim, int32_t order) {
    return (dim == kNarrowDim ? narrow_node_weight_floats(order) : split_node_weight_slab_floats(dim, order)) * static_cast<int64_t>(sizeof(float));
}

int ihg_node_interact_bwd_weight(const float* h, int64_t ld_h, const float* sums, int64_t ld_sums, const float* dy, int64_t ld_dy, const float* dy_scale, int32_t order,
                                 const int64_t* type_begin, float* dw, int64_t ld_dw, void* workspace, int64_t workspace_bytes, int32_t dim, ihg_stream_t stream) {
    if (order != 2 && order != 3) return fail(IHG_ERR_INVALID, "ihg_node_interact_bwd_weight: order must be 2 or 3, got %d", order);
    if (type_begin == nullptr || type_begin[0] != 0 || type_begin[1] < type_begin[0] || type_begin[2] < type_begin[1] || type_begin[3] < type_begin[2])
        return fail(IHG_ERR_INVALID, "ihg_node_interact_bwd_weight: bad type_begin");
    if (dim <= 0 || ld_h < dim || ld_dy < dim || ld_sums < 3LL * dim || ld_dw < static_cast<int64_t>(order == 3 ? 7 : 6) * dim)
        return fail(IHG_ERR_INVALID, "ihg_node_interact_bwd_weight: bad size");
    if (h == nullptr || sums == nullptr || dy == nullptr || dw == nullptr) return fail(IHG_ERR_INVALID, "ihg_node_interact_bwd_weight: null pointer");
    const bool narrow = narrow_node_weight_ok(dim, order, ld_h, ld_sums, ld_dy, dy);
    if ((!narrow && !split_node_weight_ok(dim, order, ld_h, ld_sums, ld_dy, dy)) || !aligned16(h) || !aligned16(sums))
        return fail(IHG_ERR_INVALID, "ihg_node_interact_bwd_weight: dim %d / order %d / alignment not supported (ihg_node_interact_bwd_weight_supported)", dim, order);
    if (workspace == nullptr || !aligned16(workspace) || workspace_bytes < ihg_node_interact_bwd_weight_workspace_bytes(dim, order))
        return fail(IHG_ERR_WORKSPACE, "ihg_node_interact_bwd_weight: workspace too small");
    if (narrow) {
        launch_node_weight_narrow(order, h, ld_h, sums, ld_sums, dy, ld_dy, dy_scale, type_begin, static_cast<float*>(workspace), dw, ld_dw, static_cast<hipStream_t>(stream));
        return check_launch("ihg_node_interact_bwd_weight");
    }
    launch_node_weight_split(dim, order, h, ld_h, sums, ld_sums, dy, ld_dy, dy_scale, type_begin, static_cast<float*>(workspace), dw, ld_dw, static_cast<hipStream_t>(stream));
    return check_launch("ihg_node_interact_bwd_weight");
}

int64_t ihg_interact_bwd_workspace_bytes(int64_t n_edges, int32_t dim, int32_t order) {
    (void)n_edges;
    if (!mfma_dim(dim) || (order != 2 && order != 3)) return 0;
    const int64_t w_floats = packed_weight_floats(dim, order);
    const int64_t boundary = 2LL * boundary_ranges(dim) * dim + 2LL * boundary_ranges(dim);               // user-reduced form: boundary runs + their users
    const int64_t planes = dim == kNarrowDim ? narrow_members_floats(order) : split_plane_floats(dim, order);
    return (w_floats + static_cast<int64_t>(weight_slabs(dim)) * w_floats + boundary + planes) * static_cast<int64_t>(sizeof(float));
}

int32_t ihg_interact_bwd_user_reduced_supported(int32_t dim, int32_t order, int64_t ld_h) {
    // d = 128: the split kernel or the fp32 strip kernel; d = 64, 256: the split kernel only (their fp32-MFMA kernels have no user-reduced form); d = 32: narrow.hip (fp32 MFMA)
    return (dim == 128 || dim == kNarrowDim || ((dim == 64 || dim == 256) && split_arith_enabled())) && (order == 2 || order == 3) && ld_h % 4 == 0 && ld_h < (int64_t{1} << 30) ? 1 : 0;
}

int ihg_interact_bwd_user_reduced(const float* h, int64_t ld_h, const int32_t* i3, const float* w, int64_t ld_w, int32_t order,
                                  const float* dout, int64_t ld_dout, float* g2, float* dh, int64_t ld_dh, float* dw, int64_t ld_dw,
                                  void* workspace, int64_t workspace_bytes, int64_t n_edges, int32_t dim, ihg_stream_t stream) {
    if (!ihg_interact_bwd_user_reduced_supported(dim, order, ld_h)) return fail(IHG_ERR_INVALID, "ihg_interact_bwd_user_reduced: shape not supported (ask ihg_interact_bwd_user_reduced_supported)");
    const int k = order == 3 ? 7 : 6;
    if (n_edges <= 0 || ld_h < dim || ld_dout < dim || ld_dh < dim || ld_w < static_cast<int64_t>(k) * dim || (dw != nullptr && ld_dw < static_cast<int64_t>(k) * dim))
        return fail(IHG_ERR_INVALID, "ihg_interact_bwd_user_reduced: bad size");
    if (h == nullptr || i3 == nullptr || w == nullptr || dout == nullptr || g2 == nullptr || dh == nullptr || workspace == nullptr)
        return fail(IHG_ERR_INVALID, "ihg_interact_bwd_user_reduced: null pointer");
    if (ld_w % 4 || ld_dout % 4 || !aligned16(h) || !aligned16(w) || !aligned16(dout) || !aligned16(g2) || !aligned16(workspace))
        return fail(IHG_ERR_INVALID, "ihg_interact_bwd_user_reduced: rows must be 16-byte aligned");
    if (workspace_bytes < ihg_interact_bwd_workspace_bytes(n_edges, dim, order)) return fail(IHG_ERR_WORKSPACE, "ihg_interact_bwd_user_reduced: workspace too small");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int nblk = order == 3 ? 4 : 3;
    float* wq = static_cast<float*>(workspace);
    float* slabs = wq + packed_weight_floats(dim, order);
    float* bnd_val = slabs + static_cast<int64_t>(weight_slabs(dim)) * packed_weight_floats(dim, order);
    int32_t* bnd_user = reinterpret_cast<int32_t*>(bnd_val + 2LL * boundary_ranges(dim) * dim);
    void* planes = bnd_user + 2LL * boundary_ranges(dim);
    const int pack_items = (dim / 16) * nblk * (dim / 16) * kWave;
    hipLaunchKernelGGL(pack_weights_strip_kernel, dim3((pack_items + kBlockThreads - 1) / kBlockThreads), dim3(kBlockThreads), 0, s, w, ld_w, dim, nblk,
                       static_cast<float*>(nullptr), wq);
    if (nblk == 4) launch_interact_bwd_mfma<4>(dim, h, ld_h, i3, wq, dout, ld_dout, g2, slabs, dw, ld_dw, n_edges, s, dh, ld_dh, bnd_val, bnd_user, w, ld_w, planes);
    else launch_interact_bwd_mfma<3>(dim, h, ld_h, i3, wq, dout, ld_dout, g2, slabs, dw, ld_dw, n_edges, s, dh, ld_dh, bnd_val, bnd_user, w, ld_w, planes);
    return check_launch("ihg_interact_bwd_user_reduced");
}

int32_t ihg_interact_bwd_user_reduced_planes_supported(int32_t dim, int32_t order, int64_t ld_h) {
    return dim == 256 && split_arith_enabled() && ihg_interact_bwd_user_reduced_supported(dim, order, ld_h) ? 1 : 0;
}

int ihg_interact_bwd_user_reduced_planes(const float* h, int64_t ld_h, const int32_t* i3, const float* w, int64_t ld_w, int32_t order, const void* planes_rows,
                                         const float* inv_scale, float* g2, float* dh, int64_t ld_dh, void* workspace, int64_t workspace_bytes, int64_t n_edges,
                                         int32_t dim, ihg_stream_t stream) {
    if (!ihg_interact_bwd_user_reduced_planes_supported(dim, order, ld_h))
        return fail(IHG_ERR_INVALID, "ihg_interact_bwd_user_reduced_planes: shape or arithmetic mode not supported (ask ihg_interact_bwd_user_reduced_planes_supported)");
    const int k = order == 3 ? 7 : 6;
    if (n_edges <= 0 || ld_h < dim || ld_dh < dim || ld_w < static_cast<int64_t>(k) * dim) return fail(IHG_ERR_INVALID, "ihg_interact_bwd_user_reduced_planes: bad size");
    if (h == nullptr || i3 == nullptr || w == nullptr || planes_rows == nullptr || inv_scale == nullptr || g2 == nullptr || dh == nullptr || workspace == nullptr)
        return fail(IHG_ERR_INVALID, "ihg_interact_bwd_user_reduced_planes: null pointer");
    if (ld_w % 4 || !aligned16(h) || !aligned16(w) || !aligned16(planes_rows) || !aligned16(g2) || !aligned16(workspace))
        return fail(IHG_ERR_INVALID, "ihg_interact_bwd_user_reduced_planes: rows must be 16-byte aligned");
    if (workspace_bytes < ihg_interact_bwd_workspace_bytes(n_edges, dim, order)) return fail(IHG_ERR_WORKSPACE, "ihg_interact_bwd_user_reduced_planes: workspace too small");
    hipStream_t s = static_cast<hipStream_t>(stream);
    float* wq = static_cast<float*>(workspace);
    float* slabs = wq + packed_weight_floats(dim, order);
    float* bnd_val = slabs + static_cast<int64_t>(weight_slabs(dim)) * packed_weight_floats(dim, order);
    int32_t* bnd_user = reinterpret_cast<int32_t*>(bnd_val + 2LL * boundary_ranges(dim) * dim);
    void* planes = bnd_user + 2LL * boundary_ranges(dim);
    int entries = 0;
    launch_members_split(dim, order, h, ld_h, i3, w, ld_w, planes, static_cast<const float*>(planes_rows), dim, g2, n_edges, dh, ld_dh, bnd_val, bnd_user, &entries, s, nullptr,
                         nullptr, 0, inv_scale);
    hipLaunchKernelGGL(user_boundary_fixup_kernel, dim3(entries), dim3(std::max(128, dim)), 0, s, bnd_val, bnd_user, entries, dim, dh, ld_dh);
    return check_launch("ihg_interact_bwd_user_reduced_planes");
}

int32_t ihg_interact_bwd_gathered_supported(int32_t dim, int32_t order, int64_t ld_h, int64_t ld_dy) {
    return dim != 256 && ihg_interact_bwd_user_reduced_supported(dim, order, ld_h) && (split_arith_enabled() || dim == kNarrowDim) && ld_dy >= dim && ld_dy % 4 == 0 && ld_dy < (int64_t{1} << 30) ? 1 : 0;
}

int ihg_interact_bwd_gathered(const float* h, int64_t ld_h, const int32_t* i3, const float* w, int64_t ld_w, int32_t order,
                              const float* dy, int64_t ld_dy, const float* dy_scale, float* dout, int64_t ld_dout, float* g2, float* dh, int64_t ld_dh,
                              float* dw, int64_t ld_dw, void* workspace, int64_t workspace_bytes, int64_t n_edges, int32_t dim, ihg_stream_t stream) {
    if (!ihg_interact_bwd_gathered_supported(dim, order, ld_h, ld_dy))
        return fail(IHG_ERR_INVALID, "ihg_interact_bwd_gathered: shape or arithmetic mode not supported (ask ihg_interact_bwd_gathered_supported)");
    const int k = order == 3 ? 7 : 6;
    if (n_edges <= 0 || ld_h < dim || (dout != nullptr && ld_dout < dim) || ld_dh < dim || ld_w < static_cast<int64_t>(k) * dim || (dw != nullptr && ld_dw < static_cast<int64_t>(k) * dim))
        return fail(IHG_ERR_INVALID, "ihg_interact_bwd_gathered: bad size");
    if (h == nullptr || i3 == nullptr || w == nullptr || dy == nullptr || (dout == nullptr && dw != nullptr) || g2 == nullptr || dh == nullptr || workspace == nullptr)
        return fail(IHG_ERR_INVALID, "ihg_interact_bwd_gathered: null pointer");
    if (dout == nullptr) {                                               // the hyperedges' cotangents are formed and used on chip only
        dout = g2;
        ld_dout = -1;
    }
    if (ld_w % 4 || (ld_dout > 0 && ld_dout % 4) || !aligned16(h) || !aligned16(w) || !aligned16(dy) || !aligned16(dout) || !aligned16(g2) || !aligned16(workspace))
        return fail(IHG_ERR_INVALID, "ihg_interact_bwd_gathered: rows must be 16-byte aligned");
    if (workspace_bytes < ihg_interact_bwd_workspace_bytes(n_edges, dim, order)) return fail(IHG_ERR_WORKSPACE, "ihg_interact_bwd_gathered: workspace too small");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int nblk = order == 3 ? 4 : 3;
    float* wq = static_cast<float*>(workspace);                          // (the fp32 fragment image of w is not needed by the split kernels)
    float* slabs = wq + packed_weight_floats(dim, order);
    float* bnd_val = slabs + static_cast<int64_t>(weight_slabs(dim)) * packed_weight_floats(dim, order);
    int32_t* bnd_user = reinterpret_cast<int32_t*>(bnd_val + 2LL * boundary_ranges(dim) * dim);
    void* planes = bnd_user + 2LL * boundary_ranges(dim);
    if (dim == kNarrowDim ? (!narrow_members_ok(dim, order, g2, ld_h, ld_dy, dy) || (dw != nullptr && ld_dout <= 0))
                          : (!split_members_ok(dim, order, g2, ld_h, ld_dy, dy, true) || (dw != nullptr && !split_weight_ok(dim, order, ld_h, ld_dout, dout))))
        return fail(IHG_ERR_INVALID, "ihg_interact_bwd_gathered: the kernels do not take these strides / alignments");
    if (nblk == 4) launch_interact_bwd_mfma<4>(dim, h, ld_h, i3, wq, dy, ld_dy, g2, slabs, dw, ld_dw, n_edges, s, dh, ld_dh, bnd_val, bnd_user, w, ld_w, planes, dy_scale, dout, ld_dout);
    else launch_interact_bwd_mfma<3>(dim, h, ld_h, i3, wq, dy, ld_dy, g2, slabs, dw, ld_dw, n_edges, s, dh, ld_dh, bnd_val, bnd_user, w, ld_w, planes, dy_scale, dout, ld_dout);
    return check_launch("ihg_interact_bwd_gathered");
}

int ihg_interact_bwd(const float* h, int64_t ld_h, const int32_t* i3, const float* w, int64_t ld_w, int32_t order,
                     const float* dout, int64_t ld_dout, float* g, float* dw, int64_t ld_dw, void* workspace,
                     int64_t workspace_bytes, int64_t n_edges, int32_t dim, ihg_stream_t stream) {
    if (order != 2 && order != 3) return fail(IHG_ERR_INVALID, "ihg_interact_bwd: order must be 2 or 3, got %d", order);
    const int k = order == 3 ? 7 : 6;
    if (n_edges < 0 || dim <= 0 || ld_h < dim || ld_dout < dim || ld_w < static_cast<int64_t>(k) * dim || (dw != nullptr && ld_dw < static_cast<int64_t>(k) * dim))
        return fail(IHG_ERR_INVALID, "ihg_interact_bwd: bad size");
    if (h == nullptr || i3 == nullptr || w == nullptr || dout == nullptr || g == nullptr)
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
        if (strip_bwd_ok(dim, g, ld_h))
            hipLaunchKernelGGL(pack_weights_strip_kernel, dim3((pack_items + kBlockThreads - 1) / kBlockThreads), dim3(kBlockThreads), 0, s, w, ld_w, dim, nblk,
                               static_cast<float*>(nullptr), wq);
        else
            hipLaunchKernelGGL(pack_weights_kernel, dim3((pack_items + kBlockThreads - 1) / kBlockThreads), dim3(kBlockThreads), 0, s, w, ld_w, dim, nblk,
                               static_cast<float*>(nullptr), wq);
        // the bf16 planes of the split contraction sit behind the boundary table (only d = 128 has either)
        void* planes = split_plane_floats(dim, order) == 0 ? nullptr :
                       static_cast<void*>(slabs + static_cast<int64_t>(weight_slabs(dim)) * packed_weight_floats(dim, order) + 2LL * boundary_ranges(dim) * dim + 2LL * boundary_ranges(dim));
        if (nblk == 4) launch_interact_bwd_mfma<4>(dim, h, ld_h, i3, wq, dout, ld_dout, g, slabs, dw, ld_dw, n_edges, s, nullptr, 0, nullptr, nullptr, w, ld_w, planes);
        else launch_interact_bwd_mfma<3>(dim, h, ld_h, i3, wq, dout, ld_dout, g, slabs, dw, ld_dw, n_edges, s, nullptr, 0, nullptr, nullptr, w, ld_w, planes);
        return check_launch("ihg_interact_bwd");
    }
    if (n_edges > 0) {
        const int64_t total = n_edges * dim;
        const int grid = static_cast<int>(std::min<int64_t>((total + kBlockThreads - 1) / kBlockThreads, kMaxBlocks * 4));
        hipLaunchKernelGGL(interact_bwd_members_generic_kernel, dim3(grid), dim3(kBlockThreads), 0, s, h, ld_h, i3, w, ld_w, order, dout, ld_dout, g, n_edges, dim);
    }
    if (dw == nullptr) return check_launch("ihg_interact_bwd");
    const int64_t wtotal = static_cast<int64_t>(dim) * (order == 3 ? 4 : 3) * dim;
    const int wgrid = static_cast<int>(std::min<int64_t>((wtotal + kBlockThreads - 1) / kBlockThreads, kMaxBlocks * 4));
    hipLaunchKernelGGL(interact_bwd_weight_generic_kernel, dim3(wgrid), dim3(kBlockThreads), 0, s, h, ld_h, i3, order, dout, ld_dout, dw, ld_dw, n_edges, dim);
    return check_launch("ihg_interact_bwd");
}

}  // extern "C"
