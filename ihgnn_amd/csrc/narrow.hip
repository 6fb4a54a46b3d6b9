// narrow.hip - the interactive layer at d = 32 (the reference's default width: Helpers/GlobalSettings.py:30, Main.py:23) in its node-level form, with the gathering /
// user-reduced member gradients (narrow.hpp; DESIGN.md section 4, "Round 5: the reference's default width").  Replaces, at that width, the hyperedge-form kernels of
// rounds 1 - 2 (interact_fwd_ws_kernel<32>, interact_bwd_members_ws_kernel<32>, interact_bwd_weight_mfma_kernel<32>: an [E, d] forward tensor, an [E, 3, d] member
// buffer, a K5 launch) - Models/CommonLayers.py:58-87 and Models/GnnLayers.py:221-236 re-associated as in split_node.hip.
//
// One wave = one tile of 16 rows (nodes or hyperedges) at a time, a contiguous range of tiles per wave, no LDS, no barriers.  v_mfma_f32_16x16x4_f32 is issued as
// W^T x rows^T: the B operand's lane (row = lane & 15, q = lane >> 4) supplies values of ITS row, the result's lane (row = lane & 15, q) receives four consecutive output
// columns 4 q .. of that row per 16-column tile.  The contraction index is free to be enumerated in any order as long as both operands agree, so lane group q supplies the
// columns { 4 q .. 4 q + 3 } and { 16 + 4 q .. 16 + 4 q + 3 } of a 32-wide row - exactly the two 16-byte pieces it loads: one load instruction moves the contiguous 64-byte
// halves of 16 rows, and what a lane loads, multiplies element-wise (the product rule, the node-level products) and stores are the same eight columns.  Weights sit in
// registers in A-fragment order (packed once per call).  Rows are requested one tile ahead (member ids two), unconditionally and with clamped indices - a branch around a
// request makes the compiler wait for everything at the join.
#include "narrow.hpp"

namespace {

constexpr int ND = kNarrowDim;
constexpr int NT = 16;      // rows per tile
constexpr int kNarrowDumpFloats = 64 * 4 + 64;      // a wave's dump region: 16 bytes per lane, and the + 16 / 32 / 48-float offsets of a row's further pieces stay inside it

// column of a 32-wide row that lane group q holds in register s < 8 (= contraction index of step s for lane group q)
__host__ __device__ constexpr int ncol(int q, int s) { return 16 * (s >> 2) + 4 * q + (s & 3); }

// X block xb of the node-level contraction ([deg h | S_a | h S_a | S_b | h S_b | S_ab | h S_ab]) -> block of w = [A_u | A_q | A_i | W_uq | W_qi | W_iu | W_uqi] it meets
// at a node of `type` (the table of the node-level form; split_node.hip's node_xblock_weight)
__device__ __forceinline__ int narrow_xblock_weight(int type, int xb) {
    constexpr signed char kBlock[3][7] = {{0, 1, 3, 2, 5, 4, 6}, {1, 0, 3, 2, 4, 5, 6}, {2, 0, 5, 1, 4, 3, 6}};
    return kBlock[type][xb];
}

struct NarrowTiles {
    int64_t begin[4];      // first row of every node type
    int tile_prefix[4];    // 16-row tiles before type t
    int wave_prefix[4];    // waves before type t, where a launch gives every wave ONE type (narrow_tiles(type_begin, waves))
};
struct NarrowRanges {
    int64_t begin[4];
    int range_prefix[4];   // row ranges (one per workgroup) before type t
};

// ------------------------------------------------------------------------------------------------
// Node-level forward (ihg_node_interact_fwd at d = 32): out[v] = scale[v] (X[v] Wt_type(v) + deg[v] bias), X = the seven 32-wide blocks above formed in registers.
// packed[type][xb][s][ct][lane] = w[16 ct + (lane & 15)][32 block(type, xb) + ncol(lane >> 4, s)]   (A fragment of the W^T tile ct at contraction step (xb, s))
// 112 MFMAs per 16 rows beside 8 KB of rows in and 2 KB out: the matrix pipe needs ~ 25 us for C2's 256 k rows, the rows ~ 35 us at 5 TB/s.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlockThreads) void pack_node_fwd_narrow_kernel(const float* __restrict__ w, int64_t ld_w, int n_xb, float* __restrict__ packed) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= 3 * n_xb * 8 * 2 * kWave) return;
    const int lane = idx & 63, ct = (idx >> 6) & 1, s = (idx >> 7) & 7, rest = idx >> 10;
    const int xb = rest % n_xb, type = rest / n_xb;
    packed[idx] = w[static_cast<int64_t>(16 * ct + (lane & 15)) * ld_w + ND * narrow_xblock_weight(type, xb) + ncol(lane >> 4, s)];
}

struct FwdRows {
    v4f h[2], a[2], b[2], ab[2];      // the lane's two pieces of h[v] and of the three pair sums of v
    float d, sc;
    int64_t v;
};

template <int ORDER>
__global__ __launch_bounds__(kBlockThreads, 2) void node_interact_fwd_narrow_kernel(const float* __restrict__ h, int64_t ld_h, const float* __restrict__ sums, int64_t ld_s,
                                                                                     const float* __restrict__ deg, const float* __restrict__ scale,
                                                                                     const float* __restrict__ bias, const float* __restrict__ packed, NarrowTiles plan,
                                                                                     float* __restrict__ out, int64_t ld_out) {
    constexpr int XB = ORDER == 3 ? 7 : 6;
    const int lane = threadIdx.x & 63, q = lane >> 4, i = lane & 15;
    // a wave works inside ONE node type (plan.wave_prefix: waves in proportion to the types' tiles): its weights are loaded once, in front of the loop - a reload under a
    // branch inside the loop would turn the loop's waits into vmcnt(0)
    const int wid = static_cast<int>(global_wave_id());
    if (wid >= plan.wave_prefix[3]) return;
    const int type = wid >= plan.wave_prefix[2] ? 2 : (wid >= plan.wave_prefix[1] ? 1 : 0);
    const int n_w = plan.wave_prefix[type + 1] - plan.wave_prefix[type];
    const int tiles_t = plan.tile_prefix[type + 1] - plan.tile_prefix[type];
    const int per = (tiles_t + n_w - 1) / n_w;
    const int t0 = (wid - plan.wave_prefix[type]) * per;                  // first tile, counted inside the type
    const int n_my = std::max(0, std::min(per, tiles_t - t0));
    if (n_my == 0) return;
    const float* const scale_src = scale != nullptr ? scale : deg;
    auto load = [&](int k, FwdRows& r) {
        const int tile = std::min(t0 + k, tiles_t - 1);
        const int64_t row = plan.begin[type] + static_cast<int64_t>(tile) * NT + i;
        r.v = std::min(row, plan.begin[type + 1] - 1);                   // (a lane past the type's last row: that row again, the same values stored to the same place)
        const float* hp = h + r.v * ld_h + 4 * q;
        const float* sp = sums + r.v * ld_s + 4 * q;
        r.d = deg[r.v];
        r.sc = scale_src[r.v];                                             // (no scale: the degree is read again and replaced by 1 at use - a request under a branch is what costs,
                                                                           //  and a select on the loaded value HERE would wait for it in front of the row requests)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            r.h[hf] = *reinterpret_cast<const v4f*>(hp + 16 * hf);
            r.a[hf] = *reinterpret_cast<const v4f*>(sp + 16 * hf);
            r.b[hf] = *reinterpret_cast<const v4f*>(sp + ND + 16 * hf);
            r.ab[hf] = *reinterpret_cast<const v4f*>(sp + 2 * ND + 16 * hf);
        }
    };
    v4f bias_c[2];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) bias_c[ct] = bias != nullptr ? *reinterpret_cast<const v4f*>(bias + 16 * ct + 4 * q) : v4f{0.f, 0.f, 0.f, 0.f};
    float wreg[XB][8][2];
    {
        const float* wp = packed + static_cast<int64_t>(type) * XB * 8 * 2 * kWave + lane;
#pragma unroll
        for (int xb = 0; xb < XB; ++xb)
#pragma unroll
            for (int s = 0; s < 8; ++s)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) wreg[xb][s][ct] = wp[((xb * 8 + s) * 2 + ct) * kWave];
    }
    auto step = [&](int k, const FwdRows& use, FwdRows& fill) {
        load(k + 1, fill);
        __builtin_amdgcn_sched_barrier(0);
        v4f acc[2] = {v4f{0.f, 0.f, 0.f, 0.f}, v4f{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int xb = 0; xb < XB; ++xb) {
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const float hv = use.h[s >> 2][s & 3];
                const float z = xb == 0 ? hv * use.d : xb == 1 ? use.a[s >> 2][s & 3] : xb == 2 ? hv * use.a[s >> 2][s & 3] : xb == 3 ? use.b[s >> 2][s & 3]
                                : xb == 4 ? hv * use.b[s >> 2][s & 3] : xb == 5 ? use.ab[s >> 2][s & 3] : hv * use.ab[s >> 2][s & 3];
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[xb][s][ct], z, acc[ct], 0, 0, 0);
            }
        }
        const float sc = scale != nullptr ? use.sc : 1.f;
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) *reinterpret_cast<v4f*>(out + use.v * ld_out + 16 * ct + 4 * q) = (acc[ct] + bias_c[ct] * use.d) * sc;
    };
    FwdRows ra, rb;
    load(0, ra);
    int k = 0;
#pragma clang loop unroll(disable)
    for (; k + 1 < n_my; k += 2) {
        step(k, ra, rb);
        step(k + 1, rb, ra);
    }
    if (k < n_my) step(k, ra, rb);
}

// ------------------------------------------------------------------------------------------------
// Node-level weight gradients of the product blocks (ihg_node_interact_bwd_weight at d = 32): G_type[j][X col] = sum over the type's rows of (dy_scale dy)[v][j] X[v][col],
// X = [h S_a | h S_b | S_ab | h S_ab].  The contraction index is the ROW: a step takes four rows (lane group rq = lane >> 4 -> row), lane ci = lane & 15 supplies columns
// 2 ci and 2 ci + 1 of every operand row (8-byte loads, 16 lanes = one whole 128-byte row) - output index i of tile mt / nt stands for column 2 i + mt / nt.  A wave keeps
// the whole [32 x NBLK 32] gradient of its row range in 8 NBLK accumulator tiles and writes ONE slab; narrow_weight_reduce_kernel adds the slabs of each type, in range
// order, into the w blocks that type's X blocks stand for (split_node.hip's node_weight_reduce_kernel at this width).
// ------------------------------------------------------------------------------------------------
template <int NBLK>
__global__ __launch_bounds__(kBlockThreads, 2) void node_interact_weight_narrow_kernel(const float* __restrict__ h, int64_t ld_h, const float* __restrict__ sums, int64_t ld_s,
                                                                                        const float* __restrict__ dy, int64_t ld_dy, const float* __restrict__ dy_scale,
                                                                                        NarrowRanges plan, float* __restrict__ slabs) {
    typedef float v2f __attribute__((ext_vector_type(2)));
    __shared__ float red[3][NBLK * 16][kWave];                           // the partial gradients of waves 1 - 3, register by register (48 KB at order 3)
    const int range = blockIdx.x;                                        // a row range = a workgroup = one slab; its four waves take a quarter of the rows each
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, ci = lane & 15, rq = lane >> 4;
    const int type = range >= plan.range_prefix[2] ? 2 : (range >= plan.range_prefix[1] ? 1 : 0);
    const int n_r = plan.range_prefix[type + 1] - plan.range_prefix[type];
    const int64_t rows_t = plan.begin[type + 1] - plan.begin[type];
    const int64_t per = ((rows_t + n_r - 1) / n_r + 63) / 64 * 64;
    const int64_t b0 = plan.begin[type] + static_cast<int64_t>(range - plan.range_prefix[type]) * per;
    const int64_t r0 = std::min(b0 + wave * (per / 4), plan.begin[type + 1]);
    const int64_t r1 = std::min(b0 + (wave + 1) * (per / 4), plan.begin[type + 1]);
    v4f acc[2][NBLK * 2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int t = 0; t < NBLK * 2; ++t) acc[mt][t] = v4f{0.f, 0.f, 0.f, 0.f};
    struct Four { v2f dyv[4], hv[4], a[4], b[4], ab[4]; float sc[4]; };
    auto load = [&](int64_t row, Four& f) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t v = row + 4 * u + rq;
            const int64_t vc = std::max<int64_t>(std::min(v, r1 - 1), plan.begin[type]);
            f.sc[u] = v < r1 ? (dy_scale != nullptr ? dy_scale[vc] : 1.f) : 0.f;       // rows past the range: zero cotangent
            f.dyv[u] = *reinterpret_cast<const v2f*>(dy + vc * ld_dy + 2 * ci);
            f.hv[u] = *reinterpret_cast<const v2f*>(h + vc * ld_h + 2 * ci);
            f.a[u] = *reinterpret_cast<const v2f*>(sums + vc * ld_s + 2 * ci);
            f.b[u] = *reinterpret_cast<const v2f*>(sums + vc * ld_s + ND + 2 * ci);
            f.ab[u] = *reinterpret_cast<const v2f*>(sums + vc * ld_s + 2 * ND + 2 * ci);
        }
    };
    auto contract = [&](const Four& f) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const v2f a = f.dyv[u] * f.sc[u];
            const v2f z[4] = {f.hv[u] * f.a[u], f.hv[u] * f.b[u], f.ab[u], f.hv[u] * f.ab[u]};
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int xb = 0; xb < NBLK; ++xb)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) acc[mt][2 * xb + nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt], z[xb][nt], acc[mt][2 * xb + nt], 0, 0, 0);
        }
    };
    if (r0 < r1) {
        Four fa, fb;
        load(r0, fa);
        int64_t row = r0;
#pragma clang loop unroll(disable)
        for (; row + 16 < r1; row += 32) {
            load(row + 16, fb);
            contract(fa);
            load(row + 32, fa);
            contract(fb);
        }
        if (row < r1) contract(fa);
    }
    if (wave > 0) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int t = 0; t < NBLK * 2; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) red[wave - 1][(mt * NBLK * 2 + t) * 4 + r][lane] = acc[mt][t][r];
    }
    __syncthreads();
    if (wave != 0) return;
    float* slab = slabs + static_cast<int64_t>(range) * ND * NBLK * ND;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int xb = 0; xb < NBLK; ++xb)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int slot = (mt * NBLK * 2 + 2 * xb + nt) * 4 + r;
                    slab[(2 * (4 * rq + r) + mt) * (NBLK * ND) + xb * ND + 2 * ci + nt] = ((acc[mt][2 * xb + nt][r] + red[0][slot][lane]) + red[1][slot][lane]) + red[2][slot][lane];
                }
}

__global__ __launch_bounds__(kBlockThreads) void narrow_weight_reduce_kernel(const float* __restrict__ slabs, NarrowRanges plan, int nblk, float* __restrict__ dw, int64_t ld_dw) {
    constexpr signed char kPos[3][4] = {{0, 2, 1, 3}, {0, 1, 2, 3}, {2, 1, 0, 3}};      // [type][w block uq / qi / iu / uqi] -> X block
    const int width = nblk * ND;
    const int64_t total = static_cast<int64_t>(ND) * width;
    const int64_t idx = static_cast<int64_t>(blockIdx.x) * kWave + (threadIdx.x & 63);
    const bool live = idx < total;
    const int j = static_cast<int>((live ? idx : 0) / width), col = static_cast<int>((live ? idx : 0) - static_cast<int64_t>(j) * width);
    const int b = col / ND, c = col - b * ND;
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const int n = plan.range_prefix[t + 1] - plan.range_prefix[t];
        sum += slab_sum(slabs + static_cast<int64_t>(plan.range_prefix[t]) * total, n, total, static_cast<int64_t>(j) * width + kPos[t][b] * ND + c, live);
    }
    if ((threadIdx.x >> 6) == 0 && live) dw[static_cast<int64_t>(j) * ld_dw + 3 * ND + col] = sum;
}

// ------------------------------------------------------------------------------------------------
// Member gradients (ihg_interact_bwd_user_reduced / ihg_interact_bwd_gathered at d = 32): dz_b = dout W_b, then the product rule; the user slot summed on chip.
// packed[nt = 2 b + hf][s][lane] = w[ncol(lane >> 4, s)][(3 + b) 32 + 16 hf + (lane & 15)]: A fragment of the W_b^T tile (b, hf) at step s; 64 registers, 64 MFMAs per tile.
// Lane (e = lane & 15, q = lane >> 4) of a tile of 16 CONSECUTIVE hyperedges: requests the member ids of e two tiles ahead and - GATHER - the three members' rows of the
// node-level cotangent dy ([N, d]: dout[e] = sum over the members of dy_scale[m] dy[m], K5's order) or the row of dout, and the three members' rows of h, one tile ahead;
// its eight values of dout are the B operand, its eight results per block the columns it holds of h - the product rule needs no data from another lane.
// User slot: hyperedges are numbered by user, so the tile's 16 user-slot gradients are runs of equal user along e = the 16 lanes of a DPP row: a segmented inclusive scan
// (row_shr 1 / 2 / 4 / 8, a lane adds what lies inside its run) leaves every run's sum in its last lane, which stores it to dh[user] - unless the run is the FIRST of
// the wave's tile range (it may continue a run of the previous range) or the LAST (carried from tile to tile in lane 15's registers; at the end of the range it may continue
// in the next): those go to the boundary table that interact.hip's user_boundary_fixup_kernel adds up in range order (two entries per range).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlockThreads) void pack_members_narrow_kernel(const float* __restrict__ w, int64_t ld_w, int nblk, float* __restrict__ packed) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= nblk * 2 * 8 * kWave) return;
    const int lane = idx & 63, s = (idx >> 6) & 7, nt = idx >> 9;
    packed[idx] = w[static_cast<int64_t>(ncol(lane >> 4, s)) * ld_w + (3 + (nt >> 1)) * ND + 16 * (nt & 1) + (lane & 15)];
}

template <int CTRL>
__device__ __forceinline__ float dpp_row_f(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, true));      // lanes whose source lies outside the row read 0
}
template <int CTRL>
__device__ __forceinline__ int dpp_row_i(int x) {
    return __builtin_amdgcn_update_dpp(0, x, CTRL, 0xf, 0xf, true);
}

// stores through pointers that are KNOWN to be global (a pointer rebuilt from an integer would be a flat one)
typedef __attribute__((address_space(1))) float GlobalFloat;
typedef __attribute__((address_space(1))) v4f GlobalV4;
__device__ __forceinline__ void gstore4(GlobalFloat* p, v4f v) { *reinterpret_cast<GlobalV4*>(p) = v; }
__device__ __forceinline__ void gstore_stream4(GlobalFloat* p, v4f v) { __builtin_nontemporal_store(v, reinterpret_cast<GlobalV4*>(p)); }

struct MemberIds {
    int u, q, i;
};
template <bool GATHER>
struct MemberRows {
    v4f hu[2], hq[2], hi[2];
    v4f du[2], dq[GATHER ? 2 : 1], di[GATHER ? 2 : 1];      // GATHER: the three members' dy pieces; otherwise du = the dout pieces
    float su, sq, si;
    int user;                                                // user of the lane's hyperedge (-2: past the end of the list)
    int64_t e;
};

// EVERY vector-memory instruction of the loop is issued unconditionally.  The memory counter (vmcnt) is in order and the compiler waits by COUNT: "all but the N youngest
// requests have returned".  A request inside a branch makes that count unknown, and the wait in front of the next use of ANY loaded value becomes vmcnt(0) - the first form
// of this kernel (stores under `if (live)` / `if (tail)`, the scale loads under `dy_scale != nullptr`) drained the queue twice per tile, the tile's own stores included:
// 249 us whatever was taken out of its instruction stream (- 34 % instructions: 249 us; no gathers AND no MFMAs: 150 us).  Here a lane that has nothing to store stores
// to a dump row of its own (`dump`: 8 floats per lane and wave, never read: it stays in L2), absent scales are read from a vector of ones' stand-in and replaced by a
// select, and the weights are loaded before the loop: every wait is an exact count and leaves the younger requests - the next tile's rows, this tile's stores - in flight.
template <int NBLK, bool GATHER, bool STORE_DOUT>
__global__ __launch_bounds__(kBlockThreads, 2) void members_narrow_kernel(const float* __restrict__ h, int64_t ld_h, const int32_t* __restrict__ i3,
                                                                           const float* __restrict__ packed, const float* __restrict__ dsrc, int64_t ld_d,
                                                                           const float* __restrict__ dy_scale, float* __restrict__ dout_store, int64_t ld_store,
                                                                           float* __restrict__ g2, int64_t n_edges, float* __restrict__ dh_user, int64_t ld_dh,
                                                                           float* __restrict__ bnd_val, int32_t* __restrict__ bnd_user, int n_ranges, float* __restrict__ dump_base) {
    const int range = static_cast<int>(global_wave_id());
    if (range >= n_ranges) return;
    const int lane = threadIdx.x & 63, q = lane >> 4, el = lane & 15;
    const int64_t n_tiles = (n_edges + NT - 1) / NT;
    const int64_t per = (n_tiles + n_ranges - 1) / n_ranges;
    const int64_t t0 = static_cast<int64_t>(range) * per;
    const int n_my = static_cast<int>(std::max<int64_t>(0, std::min<int64_t>(per, n_tiles - t0)));
    if (n_my == 0) {
        if (lane == 0) bnd_user[2 * range] = bnd_user[2 * range + 1] = -1;
        return;
    }
    float wreg[NBLK * 2][8];
#pragma unroll
    for (int nt = 0; nt < NBLK * 2; ++nt)
#pragma unroll
        for (int s = 0; s < 8; ++s) wreg[nt][s] = packed[(nt * 8 + s) * kWave + lane];
    float* const dump = dump_base + static_cast<int64_t>(range) * kNarrowDumpFloats + lane * 4;  // this lane's 16 bytes of nowhere (+ 16 / 32 / 48 floats: the pieces a row store adds; 12 floats below: the dout pieces)
    const bool scaled = GATHER && dy_scale != nullptr;
    const float* const scale_src = scaled ? dy_scale : reinterpret_cast<const float*>(i3);   // (unscaled: any readable word; the value is replaced by 1)

    // row r of a table whose rows are ld floats apart (ld < 2^31, checked by narrow_members_ok): ONE v_mad_u64_u32 where the 64 x 64-bit product costs four instructions
    const uint32_t ldh = static_cast<uint32_t>(ld_h), ldd = static_cast<uint32_t>(ld_d);
    const float* const hq4 = h + 4 * q;
    const float* const dq4 = dsrc + 4 * q;
    auto row_of = [](const float* base, int id, uint32_t ld) { return base + static_cast<uint64_t>(static_cast<uint32_t>(id)) * static_cast<uint64_t>(ld); };
    auto edge_of = [&](int k) { return (t0 + std::min(k, n_my - 1)) * NT + el; };
    auto load_ids = [&](int k, MemberIds& m) {
        const int64_t ec = std::min(edge_of(k), n_edges - 1);
        m.u = i3[3 * ec];
        m.q = i3[3 * ec + 1];
        m.i = i3[3 * ec + 2];
    };
    auto load_rows = [&](int k, const MemberIds& m, MemberRows<GATHER>& r) {
        r.e = edge_of(k);
        r.user = (r.e < n_edges && k < n_my) ? m.u : -2;
        const int64_t ec = std::min(r.e, n_edges - 1);
        const float *pu = row_of(hq4, m.u, ldh), *pq = row_of(hq4, m.q, ldh), *pi = row_of(hq4, m.i, ldh);
        const float *du = GATHER ? row_of(dq4, m.u, ldd) : dq4 + ec * ld_d, *dq = row_of(dq4, m.q, ldd), *di = row_of(dq4, m.i, ldd);
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            r.hu[hf] = *reinterpret_cast<const v4f*>(pu + 16 * hf);
            r.hq[hf] = *reinterpret_cast<const v4f*>(pq + 16 * hf);
            r.hi[hf] = *reinterpret_cast<const v4f*>(pi + 16 * hf);
            r.du[hf] = *reinterpret_cast<const v4f*>(du + 16 * hf);
            if (GATHER) {
                r.dq[GATHER ? hf : 0] = *reinterpret_cast<const v4f*>(dq + 16 * hf);
                r.di[GATHER ? hf : 0] = *reinterpret_cast<const v4f*>(di + 16 * hf);
            }
        }
        if (GATHER) {
            r.su = scale_src[scaled ? m.u : 0];                              // (replaced by 1 at use when there are no scales)
            r.sq = scale_src[scaled ? m.q : 0];
            r.si = scale_src[scaled ? m.i : 0];
        }
    };

    // the run that is open when a tile begins: whether it continues the previous tile's last run (wave-uniform) and that run's sum so far (in the lanes el == 15);
    // `first_run`: the run that is open at the tile's start is still the range's first run
    bool continuing = false, first_run = true, last_to_slot1 = false;
    float carry[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) carry[s] = 0.f;
    float* const slot0 = bnd_val + static_cast<int64_t>(2 * range) * ND;

    // (an address chosen between a real row and the dump row, made opaque: left visible, the choice is compiled into two stores under complementary branches -
    // and kept in the GLOBAL address space: a pointer of unknown origin becomes a flat one, and flat stores drain the memory counter like a branch does)
    auto pick = [](bool real, float* a, float* b) {
        uint64_t p = real ? reinterpret_cast<uint64_t>(a) : reinterpret_cast<uint64_t>(b);
        asm("" : "+v"(p));
        return reinterpret_cast<GlobalFloat*>(p);
    };
    auto step = [&](int k, const MemberRows<GATHER>& use, MemberRows<GATHER>& fill, const MemberIds& ids_next, MemberIds& ids_fill) {
        load_ids(k + 2, ids_fill);
        load_rows(k + 1, ids_next, fill);
        __builtin_amdgcn_sched_barrier(0);                                   // the requests stay in front of the tile's work (the scheduler sinks them into the MFMA phase otherwise)
        const bool live = use.user >= 0;
        // the hyperedge's cotangent at this lane's eight columns
        float dout[8];
        const float su = scaled ? use.su : 1.f, sq = scaled ? use.sq : 1.f, si = scaled ? use.si : 1.f;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            if (GATHER) dout[s] = ((0.f + su * use.du[s >> 2][s & 3]) + sq * use.dq[GATHER ? s >> 2 : 0][s & 3]) + si * use.di[GATHER ? s >> 2 : 0][s & 3];
            else dout[s] = use.du[s >> 2][s & 3];
        }
        if (STORE_DOUT) {
            GlobalFloat* dp = pick(live, dout_store + use.e * ld_store + 4 * q, dump - 12);
            gstore_stream4(dp, v4f{dout[0], dout[1], dout[2], dout[3]});
            gstore_stream4(dp + 16, v4f{dout[4], dout[5], dout[6], dout[7]});
        }
        v4f acc[NBLK * 2];                                                   // (the first product takes the constant 0 as its addend: no 32 zero moves per tile)
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int nt = 0; nt < NBLK * 2; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[nt][s], dout[s], s == 0 ? v4f{0.f, 0.f, 0.f, 0.f} : acc[nt], 0, 0, 0);
        // product rule at the lane's columns
        float gu[8], gq[8], gi[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const int hf = s >> 2, r = s & 3;
            const float hu = use.hu[hf][r], hq = use.hq[hf][r], hi = use.hi[hf][r];
            const float z_uq = acc[0 + hf][r], z_qi = acc[2 + hf][r], z_iu = acc[4 + hf][r];
            const float z_uqi = NBLK == 4 ? acc[2 * (NBLK - 1) + hf][r] : 0.f;
            gu[s] = live ? z_uq * hq + z_iu * hi + z_uqi * (hq * hi) : 0.f;
            gq[s] = z_uq * hu + z_qi * hi + z_uqi * (hu * hi);
            gi[s] = z_qi * hq + z_iu * hu + z_uqi * (hu * hq);
        }
        {
            GlobalFloat* gp = pick(live, g2 + use.e * (2 * ND) + 4 * q, dump);
            gstore_stream4(gp, v4f{gq[0], gq[1], gq[2], gq[3]});
            gstore_stream4(gp + 16, v4f{gq[4], gq[5], gq[6], gq[7]});
            gstore_stream4(gp + ND, v4f{gi[0], gi[1], gi[2], gi[3]});
            gstore_stream4(gp + ND + 16, v4f{gi[4], gi[5], gi[6], gi[7]});
        }
        // ---- user slot.  The tile's first run continues the previous tile's last one: lane 0 takes the carried sum in
        const int user = use.user;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const float from15 = __shfl(carry[s], (lane & 48) | 15);
            if (continuing && el == 0) gu[s] += from15;
        }
        const int prev_user = dpp_row_i<0x111>(user);                        // row_shr:1
        const int next_in_tile = dpp_row_i<0x101>(user);                     // row_shl:1
        // the user of the NEXT tile's first hyperedge (its ids came in a step ago) closes or continues lane 15's run; behind the range's last tile: nobody's
        const int next_first = k + 1 < n_my ? __builtin_amdgcn_readfirstlane(ids_next.u) : -3;
        const int next_user = el == 15 ? next_first : next_in_tile;
        const bool head = el == 0 || prev_user != user;
        int hp = head ? el : 0;                                              // position of the lane's run start inside the tile: a max-scan
        hp = std::max(hp, dpp_row_i<0x111>(hp));
        hp = std::max(hp, dpp_row_i<0x112>(hp));
        hp = std::max(hp, dpp_row_i<0x114>(hp));
        hp = std::max(hp, dpp_row_i<0x118>(hp));
        // (a lane adds what lies inside its run: the shifted value times 1 or 0 - one fused multiply-add with a DPP operand per value and offset where a select and an
        // add took three instructions; the values are finite, so 0 x anything is 0)
        const float m1 = el - 1 >= hp ? 1.f : 0.f, m2 = el - 2 >= hp ? 1.f : 0.f, m4 = el - 4 >= hp ? 1.f : 0.f, m8 = el - 8 >= hp ? 1.f : 0.f;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            float v = gu[s];
            v = __builtin_fmaf(dpp_row_f<0x111>(v), m1, v);
            v = __builtin_fmaf(dpp_row_f<0x112>(v), m2, v);
            v = __builtin_fmaf(dpp_row_f<0x114>(v), m4, v);
            v = __builtin_fmaf(dpp_row_f<0x118>(v), m8, v);
            gu[s] = v;
        }
        // a run ends in this lane: its sum goes to dh[user] - the range's first run (it may continue a run of the previous range) and its last (it may continue in the next)
        // go to the boundary table instead; every other lane stores to its dump row
        const bool tail = user >= 0 && next_user != user;
        const bool to_slot0 = tail && first_run && hp == 0;
        const bool to_slot1 = tail && !to_slot0 && el == 15 && k + 1 >= n_my;
        GlobalFloat* dst = pick(tail, to_slot0 ? slot0 + 4 * q : (to_slot1 ? slot0 + ND + 4 * q : dh_user + static_cast<int64_t>(user) * ld_dh + 4 * q), dump);
        gstore4(dst, v4f{gu[0], gu[1], gu[2], gu[3]});
        gstore4(dst + 16, v4f{gu[4], gu[5], gu[6], gu[7]});
        GlobalFloat* udst = pick((to_slot0 || to_slot1) && q == 0, reinterpret_cast<float*>(bnd_user + 2 * range + (to_slot1 ? 1 : 0)), dump);
        *reinterpret_cast<__attribute__((address_space(1))) int*>(udst) = user;
        last_to_slot1 = __ballot(to_slot1) != 0;
        const bool any_head = __ballot(head && el > 0) != 0;
        const int user15 = __builtin_amdgcn_readlane(user, 15);
        continuing = user15 >= 0 && next_first == user15;
        first_run = first_run && !any_head && continuing;
#pragma unroll
        for (int s = 0; s < 8; ++s) carry[s] = gu[s];
    };

    MemberIds ia, ib;
    MemberRows<GATHER> ra, rb;
    load_ids(0, ia);
    load_ids(1, ib);
    __builtin_amdgcn_sched_barrier(0);                                     // (the order of the steady state: the ids of the tile after next are OLDER than the next tile's rows)
    load_rows(0, ia, ra);
    __builtin_amdgcn_sched_barrier(0);
    int k = 0;
#pragma clang loop unroll(disable)
    for (; k + 1 < n_my; k += 2) {
        step(k, ra, rb, ib, ia);
        step(k + 1, rb, ra, ia, ib);
    }
    if (k < n_my) step(k, ra, rb, ib, ia);
    if (!last_to_slot1 && lane == 0) bnd_user[2 * range + 1] = -1;         // the range's last run was its first, or ended inside the range
}

// ------------------------------------------------------------------------------------------------
// Node-level linear maps at d = 32 and d = 64 (ihg_node_linear_fwd / _bwd_input / _bwd_weight and their typed forms; template <int D>): streams over [N, d] - 4 d bytes in
// and out per row beside d^2 multiply-adds, which the fp32 matrix pipe keeps up with at these widths (d = 64: 28 us of MFMAs for C2's 256 k rows beside ~ 40 us of rows).
// Forward: out[v] = in[v] Wt_type(v) (+ bias of the type) (+ out[v]), Wt[k][c] = W_t[c][k] (transpose == 0) or W_t[k][c].  One node type per wave (plan.wave_prefix), its W^T
// fragments loaded once in front of the loop (d^2 / 64 registers).  Rows by typed base pointers (TypedRows: the embedding tables read in place).  A lane holds the d / 16
// pieces 16 p + 4 q .. of its row.
// ------------------------------------------------------------------------------------------------
template <int W>
using vecf = float __attribute__((ext_vector_type(W)));

// pk[t][s][ct][lane] = Wt_t[ncol(lane >> 4, s)][16 ct + (lane & 15)], s < d / 4, ct < d / 16: the A fragments of every node type's W^T
__global__ __launch_bounds__(kBlockThreads) void pack_linear_narrow_kernel(const float* __restrict__ w, int64_t ld_w, int64_t w_type_stride, int n_types, int transpose, int d,
                                                                           float* __restrict__ pk) {
    const int steps = d / 4, tiles = d / 16;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_types * steps * tiles * kWave) return;
    const int lane = idx & 63, ct = (idx >> 6) % tiles, s = ((idx >> 6) / tiles) % steps, t = (idx >> 6) / (tiles * steps);
    const float* wt = w + t * w_type_stride;
    const int c = 16 * ct + (lane & 15), k = ncol(lane >> 4, s);
    pk[idx] = transpose == 0 ? wt[static_cast<int64_t>(c) * ld_w + k] : wt[static_cast<int64_t>(k) * ld_w + c];
}

template <int D, bool ACC>
__global__ __launch_bounds__(kBlockThreads) void row_gemm_narrow_kernel(TypedRows in, int64_t ld_in, const float* __restrict__ pk, int single_weight,
                                                                        const float* __restrict__ bias, int bias_mask, int64_t bias_type_stride, NarrowTiles plan, TypedRowsOut out,
                                                                        int64_t ld_out) {
    constexpr int P = D / 16, S = D / 4;
    const int lane = threadIdx.x & 63, q = lane >> 4, i = lane & 15;
    const int wid = static_cast<int>(global_wave_id());
    if (wid >= plan.wave_prefix[3]) return;
    const int type = wid >= plan.wave_prefix[2] ? 2 : (wid >= plan.wave_prefix[1] ? 1 : 0);
    const int n_w = plan.wave_prefix[type + 1] - plan.wave_prefix[type];
    const int tiles_t = plan.tile_prefix[type + 1] - plan.tile_prefix[type];
    const int per = (tiles_t + n_w - 1) / n_w;
    const int t0 = (wid - plan.wave_prefix[type]) * per;
    const int n_my = std::max(0, std::min(per, tiles_t - t0));
    if (n_my == 0) return;
    float wreg[S][P];
#pragma unroll
    for (int s = 0; s < S; ++s)
#pragma unroll
        for (int ct = 0; ct < P; ++ct) wreg[s][ct] = pk[(((single_weight ? 0 : type) * S + s) * P + ct) * kWave + lane];
    const float* const in_rows = typed_base(in, type);
    float* const out_rows = typed_base(out, type);
    const bool with_bias = bias != nullptr && ((bias_mask >> type) & 1);
    v4f bias_c[P];
#pragma unroll
    for (int ct = 0; ct < P; ++ct) bias_c[ct] = with_bias ? *reinterpret_cast<const v4f*>(bias + type * bias_type_stride + 16 * ct + 4 * q) : v4f{0.f, 0.f, 0.f, 0.f};
    struct Rows { v4f x[P], old[P]; int64_t v; };
    auto load = [&](int k, Rows& r) {
        const int tile = std::min(t0 + k, tiles_t - 1);
        const int64_t row = plan.begin[type] + static_cast<int64_t>(tile) * NT + i;
        r.v = std::min(row, plan.begin[type + 1] - 1);                   // (a lane past the type's last row: that row again, the same values)
        const float* ip = in_rows + r.v * ld_in + 4 * q;
#pragma unroll
        for (int p = 0; p < P; ++p) r.x[p] = *reinterpret_cast<const v4f*>(ip + 16 * p);
        if (ACC) {
            const float* op = out_rows + r.v * ld_out + 4 * q;
#pragma unroll
            for (int p = 0; p < P; ++p) r.old[p] = *reinterpret_cast<const v4f*>(op + 16 * p);
        }
    };
    auto step = [&](int k, const Rows& use, Rows& fill) {
        load(k + 1, fill);
        __builtin_amdgcn_sched_barrier(0);
        v4f acc[P];
#pragma unroll
        for (int ct = 0; ct < P; ++ct) acc[ct] = bias_c[ct];
#pragma unroll
        for (int s = 0; s < S; ++s)
#pragma unroll
            for (int ct = 0; ct < P; ++ct) acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[s][ct], use.x[s >> 2][s & 3], acc[ct], 0, 0, 0);
        float* op = out_rows + use.v * ld_out + 4 * q;
#pragma unroll
        for (int ct = 0; ct < P; ++ct) *reinterpret_cast<v4f*>(op + 16 * ct) = ACC ? acc[ct] + use.old[ct] : acc[ct];
    };
    Rows ra, rb;
    load(0, ra);
    int k = 0;
#pragma clang loop unroll(disable)
    for (; k + 1 < n_my; k += 2) {
        step(k, ra, rb);
        step(k + 1, rb, ra);
    }
    if (k < n_my) step(k, ra, rb);
}

// Backward of the same maps in ONE pass over (dout, x): d W_t = dout^T x and the bias gradient over the rows of type t, and - DX - the input gradient dx = dout W_t
// (+ dx) of the same rows.  grid = (slabs, 1, weight types); a workgroup's four waves take the 16-row tiles 4 sx + wave, + 4 slabs, ... of the type's rows.  The row
// contraction reads its operands as in node_interact_weight_narrow_kernel (lane ci = the d / 16 consecutive columns (d / 16) ci ..; four rows a step), the input gradient
// as in the forward; the second read of a row hits the CU's cache.  The waves' partial gradients meet in LDS in wave order; slab layout = dense.hip's
// ([type][slab][d][d], [type][slab][d]), reduced by its dense_slab_reduce_kernel.
template <int D, bool DX, bool ACC>
__global__ __launch_bounds__(kBlockThreads) void dense_weight_grad_narrow_kernel(const float* __restrict__ dout, int64_t ld_dout, TypedRows x, int64_t ld_x, NarrowTiles plan,
                                                                                 int single_weight, float* __restrict__ slabs, float* __restrict__ bias_slabs,
                                                                                 const float* __restrict__ pk, TypedRowsOut dx, int64_t ld_dx) {
    constexpr int P = D / 16, S = D / 4, W = D / 16;
    typedef vecf<W> vw;
    constexpr int RED = W * W * 4 + W;
    __shared__ float red[3][RED][kWave];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, q = lane >> 4, i = lane & 15;
    const int wtype = blockIdx.z;
    const int64_t r_begin = single_weight ? plan.begin[0] : plan.begin[wtype];
    const int64_t r_end = single_weight ? plan.begin[3] : plan.begin[wtype + 1];
    const int64_t n_tiles = (r_end - r_begin + NT - 1) / NT;
    const int64_t stride = static_cast<int64_t>(gridDim.x) * kWavesPerBlock;
    auto type_of = [&](int64_t v) { return v >= plan.begin[2] ? 2 : (v >= plan.begin[1] ? 1 : 0); };
    float wreg[DX ? S : 1][P];
    if (DX) {                                                            // dx = dout W_t: the fragments of pack_linear_narrow_kernel with transpose = 1
#pragma unroll
        for (int s = 0; s < S; ++s)
#pragma unroll
            for (int ct = 0; ct < P; ++ct) wreg[DX ? s : 0][ct] = pk[((wtype * S + s) * P + ct) * kWave + lane];
    }
    struct Rows { v4f d[P], old[P]; vw dv[4], xv[4]; int64_t v; int type; };
    auto load = [&](int64_t tile, Rows& r) {
        const int64_t base = r_begin + std::min(tile, n_tiles - 1) * NT;
        const bool tile_live = tile < n_tiles;
        if (DX) {
            r.v = std::min(base + i, r_end - 1);
            r.type = type_of(r.v);
#pragma unroll
            for (int p = 0; p < P; ++p) r.d[p] = *reinterpret_cast<const v4f*>(dout + r.v * ld_dout + 16 * p + 4 * q);
            if (ACC) {
                const float* op = typed_base(dx, r.type) + r.v * ld_dx + 4 * q;
#pragma unroll
                for (int p = 0; p < P; ++p) r.old[p] = *reinterpret_cast<const v4f*>(op + 16 * p);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t row = base + 4 * u + q;
            const int64_t vc = std::min(row, r_end - 1);
            const float live = tile_live && row < r_end ? 1.f : 0.f;
            r.dv[u] = *reinterpret_cast<const vw*>(dout + vc * ld_dout + W * i) * live;
            r.xv[u] = *reinterpret_cast<const vw*>(typed_base(x, type_of(vc)) + vc * ld_x + W * i);
        }
    };
    v4f acc[W][W];
#pragma unroll
    for (int mt = 0; mt < W; ++mt)
#pragma unroll
        for (int nt = 0; nt < W; ++nt) acc[mt][nt] = v4f{0.f, 0.f, 0.f, 0.f};
    vw colsum = vw(0.f);
    auto work = [&](const Rows& r) {
        if (DX) {
            v4f g[P];
#pragma unroll
            for (int ct = 0; ct < P; ++ct) g[ct] = ACC ? r.old[ct] : v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < S; ++s)
#pragma unroll
                for (int ct = 0; ct < P; ++ct) g[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[DX ? s : 0][ct], r.d[s >> 2][s & 3], g[ct], 0, 0, 0);
            float* op = typed_base(dx, r.type) + r.v * ld_dx + 4 * q;        // (a lane past the last row: that row again, the same values)
#pragma unroll
            for (int ct = 0; ct < P; ++ct) *reinterpret_cast<v4f*>(op + 16 * ct) = g[ct];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            colsum += r.dv[u];
#pragma unroll
            for (int mt = 0; mt < W; ++mt)
#pragma unroll
                for (int nt = 0; nt < W; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(r.dv[u][mt], r.xv[u][nt], acc[mt][nt], 0, 0, 0);
        }
    };
    if (n_tiles > 0) {
        Rows ra, rb;
        int64_t tile = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + wave;
        load(tile, ra);
#pragma clang loop unroll(disable)
        for (; tile < n_tiles; tile += 2 * stride) {
            load(tile + stride, rb);
            work(ra);
            load(tile + 2 * stride, ra);
            if (tile + stride < n_tiles) work(rb);
        }
    }
    // the bias gradient's four row groups, then the four waves, in a fixed order
#pragma unroll
    for (int t = 0; t < W; ++t) {
        colsum[t] += __shfl_xor(colsum[t], 16);
        colsum[t] += __shfl_xor(colsum[t], 32);
    }
    if (wave > 0) {
#pragma unroll
        for (int mt = 0; mt < W; ++mt)
#pragma unroll
            for (int nt = 0; nt < W; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) red[wave - 1][(W * mt + nt) * 4 + r][lane] = acc[mt][nt][r];
#pragma unroll
        for (int t = 0; t < W; ++t) red[wave - 1][W * W * 4 + t][lane] = colsum[t];
    }
    __syncthreads();
    if (wave == 0) {
        const int n_slabs = gridDim.x;
        float* slab = slabs + (static_cast<int64_t>(wtype) * n_slabs + blockIdx.x) * D * D;
#pragma unroll
        for (int mt = 0; mt < W; ++mt)
#pragma unroll
            for (int nt = 0; nt < W; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int slot = (W * mt + nt) * 4 + r;
                    slab[(W * (4 * q + r) + mt) * D + W * i + nt] = ((acc[mt][nt][r] + red[0][slot][lane]) + red[1][slot][lane]) + red[2][slot][lane];
                }
        if (q == 0) {
            float* bs = bias_slabs + (static_cast<int64_t>(wtype) * n_slabs + blockIdx.x) * D;
#pragma unroll
            for (int t = 0; t < W; ++t) bs[W * i + t] = ((colsum[t] + red[0][W * W * 4 + t][lane]) + red[1][W * W * 4 + t][lane]) + red[2][W * W * 4 + t][lane];
        }
    }
}

// Adds up the boundary runs of members_narrow_kernel in range order and writes dh[user] (interact.hip's user_boundary_fixup_kernel at this width, eight table entries per
// workgroup - 32 threads an entry - instead of one: 4,096 workgroups of a few instructions each took 17 us).  The entry that opens a user's run walks on while the following
// entries carry the same user; every other entry's threads leave at once.  Same order of additions as a single walk over the table.
__global__ __launch_bounds__(kBlockThreads) void narrow_boundary_fixup_kernel(const float* __restrict__ bnd_val, const int32_t* __restrict__ bnd_user, int n_entries,
                                                                              float* __restrict__ dh_user, int64_t ld_dh) {
    const int k0 = blockIdx.x * (kBlockThreads / ND) + threadIdx.x / ND, c = threadIdx.x % ND;
    if (k0 >= n_entries) return;
    const int user = bnd_user[k0];
    if (user < 0) return;
    for (int k = k0 - 1; k >= 0; --k) {
        const int u = bnd_user[k];
        if (u == user) return;                    // an earlier entry opens this run
        if (u >= 0) break;
    }
    float acc = bnd_val[static_cast<int64_t>(k0) * ND + c];
    for (int k = k0 + 1; k < n_entries; ++k) {
        const int u = bnd_user[k];
        if (u < 0) continue;
        if (u != user) break;
        acc += bnd_val[static_cast<int64_t>(k) * ND + c];
    }
    dh_user[static_cast<int64_t>(user) * ld_dh + c] = acc;
}

NarrowTiles narrow_tiles(const int64_t* type_begin, int waves = 0) {
    NarrowTiles plan;
    int acc = 0;
    for (int t = 0; t < 4; ++t) plan.begin[t] = type_begin[t];
    for (int t = 0; t < 3; ++t) {
        plan.tile_prefix[t] = acc;
        acc += static_cast<int>((type_begin[t + 1] - type_begin[t] + NT - 1) / NT);
    }
    plan.tile_prefix[3] = acc;
    // `waves` dealt to the types in proportion to their tiles, every non-empty type at least one wave and none more waves than tiles
    int wacc = 0;
    for (int t = 0; t < 3; ++t) {
        const int tiles = plan.tile_prefix[t + 1] - plan.tile_prefix[t];
        plan.wave_prefix[t] = wacc;
        if (tiles > 0 && waves > 0) wacc += std::min(tiles, std::max(1, static_cast<int>(static_cast<int64_t>(tiles) * waves / std::max(acc, 1))));
    }
    plan.wave_prefix[3] = wacc;
    return plan;
}

}  // namespace

int64_t narrow_node_fwd_floats() { return 3LL * 7 * 8 * 2 * kWave; }

bool narrow_node_fwd_ok(int dim, int order, int64_t ld_h, int64_t ld_s, const float* out, int64_t ld_out, const float* bias) {
    return dim == ND && (order == 2 || order == 3) && ld_h % 4 == 0 && ld_s % 4 == 0 && ld_out % 4 == 0 && aligned16(out) && (bias == nullptr || aligned16(bias));
}

void launch_node_fwd_narrow(int order, const float* h, int64_t ld_h, const float* sums, int64_t ld_s, const float* deg, const float* scale, const float* bias, const float* w,
                            int64_t ld_w, const int64_t* type_begin, float* out, int64_t ld_out, float* packed, hipStream_t s) {
    const int n_xb = order == 3 ? 7 : 6;
    const int items = 3 * n_xb * 8 * 2 * kWave;
    hipLaunchKernelGGL(pack_node_fwd_narrow_kernel, dim3((items + kBlockThreads - 1) / kBlockThreads), dim3(kBlockThreads), 0, s, w, ld_w, n_xb, packed);
    const NarrowTiles plan = narrow_tiles(type_begin, 2048);
    if (plan.tile_prefix[3] == 0) return;
    const int grid = grid_for_waves(plan.wave_prefix[3]);
    if (order == 3) hipLaunchKernelGGL(node_interact_fwd_narrow_kernel<3>, dim3(grid), dim3(kBlockThreads), 0, s, h, ld_h, sums, ld_s, deg, scale, bias, packed, plan, out, ld_out);
    else hipLaunchKernelGGL(node_interact_fwd_narrow_kernel<2>, dim3(grid), dim3(kBlockThreads), 0, s, h, ld_h, sums, ld_s, deg, scale, bias, packed, plan, out, ld_out);
}

int64_t narrow_node_weight_floats(int order) { return static_cast<int64_t>(kNarrowWeightRanges) * ND * (order == 3 ? 4 : 3) * ND; }

bool narrow_node_weight_ok(int dim, int order, int64_t ld_h, int64_t ld_s, int64_t ld_dy, const float* dy) {
    return dim == ND && (order == 2 || order == 3) && ld_h % 2 == 0 && ld_s % 2 == 0 && ld_dy % 2 == 0 && (reinterpret_cast<uintptr_t>(dy) & 7u) == 0;
}

void launch_node_weight_narrow(int order, const float* h, int64_t ld_h, const float* sums, int64_t ld_s, const float* dy, int64_t ld_dy, const float* dy_scale,
                               const int64_t* type_begin, float* slabs, float* dw, int64_t ld_dw, hipStream_t s) {
    NarrowRanges plan;
    int64_t rows[3], total = 0;
    for (int t = 0; t < 4; ++t) plan.begin[t] = type_begin[t];
    for (int t = 0; t < 3; ++t) {
        rows[t] = type_begin[t + 1] - type_begin[t];
        total += rows[t];
    }
    // row ranges (= workgroups = slabs) by type in proportion to the rows (at least 64 rows a range, every non-empty type at least one, kNarrowWeightRanges in all at most)
    int acc = 0;
    for (int t = 0; t < 3; ++t) {
        int64_t n = rows[t] == 0 ? 0 : std::max<int64_t>(1, rows[t] * (kNarrowWeightRanges - 2) / std::max<int64_t>(total, 1));
        n = std::min<int64_t>(n, (rows[t] + 63) / 64);
        plan.range_prefix[t] = acc;
        acc += static_cast<int>(n);
    }
    plan.range_prefix[3] = acc;
    const int nblk = order == 3 ? 4 : 3;
    if (acc > 0) {
        if (order == 3) hipLaunchKernelGGL(node_interact_weight_narrow_kernel<4>, dim3(acc), dim3(kBlockThreads), 0, s, h, ld_h, sums, ld_s, dy, ld_dy, dy_scale, plan, slabs);
        else hipLaunchKernelGGL(node_interact_weight_narrow_kernel<3>, dim3(acc), dim3(kBlockThreads), 0, s, h, ld_h, sums, ld_s, dy, ld_dy, dy_scale, plan, slabs);
    }
    const int total_w = ND * nblk * ND;
    hipLaunchKernelGGL(narrow_weight_reduce_kernel, dim3((total_w + kWave - 1) / kWave), dim3(kBlockThreads), 0, s, slabs, plan, nblk, dw, ld_dw);
}

static int64_t narrow_members_packed(int order) { return static_cast<int64_t>(order == 3 ? 4 : 3) * 2 * 8 * kWave; }
int64_t narrow_members_floats(int order) { return narrow_members_packed(order) + 16 + static_cast<int64_t>(kNarrowMemberRanges) * kNarrowDumpFloats; }      // the packed weights + a dump region per range

bool narrow_members_ok(int dim, int order, const float* g2, int64_t ld_h, int64_t ld_d, const float* dsrc) {
    return dim == ND && (order == 2 || order == 3) && ld_h % 4 == 0 && ld_d % 4 == 0 && ld_h < (int64_t{1} << 31) && ld_d < (int64_t{1} << 31) && aligned16(g2) && aligned16(dsrc);
}

void launch_members_narrow(int order, int gather, const float* h, int64_t ld_h, const int32_t* i3, const float* w, int64_t ld_w, float* packed, const float* dsrc, int64_t ld_d,
                           const float* dy_scale, float* dout_store, int64_t ld_store, float* g2, int64_t n_edges, float* dh_user, int64_t ld_dh, float* bnd_val,
                           int32_t* bnd_user, int* n_boundary_entries, hipStream_t s) {
    const int nblk = order == 3 ? 4 : 3;
    const int items = nblk * 2 * 8 * kWave;
    hipLaunchKernelGGL(pack_members_narrow_kernel, dim3((items + kBlockThreads - 1) / kBlockThreads), dim3(kBlockThreads), 0, s, w, ld_w, nblk, packed);
    const int64_t n_tiles = (n_edges + NT - 1) / NT;
    const int n_ranges = static_cast<int>(std::min<int64_t>(n_tiles, kNarrowMemberRanges));
    *n_boundary_entries = 2 * n_ranges;
    const int grid = grid_for_waves(n_ranges);
    float* dump = packed + narrow_members_packed(order) + 16;
#define IHG_NARROW_MEMBERS(NBLK, GATHER, STORE)                                                                                                                            \
    hipLaunchKernelGGL((members_narrow_kernel<NBLK, GATHER, STORE>), dim3(grid), dim3(kBlockThreads), 0, s, h, ld_h, i3, packed, dsrc, ld_d, dy_scale, dout_store, ld_store, g2, \
                       n_edges, dh_user, ld_dh, bnd_val, bnd_user, n_ranges, dump)
    const bool store = gather && dout_store != nullptr;
    if (nblk == 4) {
        if (!gather) IHG_NARROW_MEMBERS(4, false, false);
        else if (store) IHG_NARROW_MEMBERS(4, true, true);
        else IHG_NARROW_MEMBERS(4, true, false);
    } else {
        if (!gather) IHG_NARROW_MEMBERS(3, false, false);
        else if (store) IHG_NARROW_MEMBERS(3, true, true);
        else IHG_NARROW_MEMBERS(3, true, false);
    }
#undef IHG_NARROW_MEMBERS
    const int per_block = kBlockThreads / ND;
    hipLaunchKernelGGL(narrow_boundary_fixup_kernel, dim3((2 * n_ranges + per_block - 1) / per_block), dim3(kBlockThreads), 0, s, bnd_val, bnd_user, 2 * n_ranges, dh_user, ld_dh);
}

bool narrow_linear_ok(int dim, int64_t ld_a, int64_t ld_b) { return (dim == 32 || dim == 64) && ld_a % 4 == 0 && ld_b % 4 == 0; }

static void pack_linear_narrow(int dim, const float* w, int64_t ld_w, int64_t w_type_stride, int transpose, float* pk, hipStream_t s) {
    const int n_types = w_type_stride == 0 ? 1 : 3;
    const int items = n_types * (dim / 4) * (dim / 16) * kWave;
    hipLaunchKernelGGL(pack_linear_narrow_kernel, dim3((items + kBlockThreads - 1) / kBlockThreads), dim3(kBlockThreads), 0, s, w, ld_w, w_type_stride, n_types, transpose, dim, pk);
}

void launch_row_gemm_narrow(int dim, TypedRows in, int64_t ld_in, const float* w, int64_t ld_w, int64_t w_type_stride, int transpose, const float* bias, int bias_mask,
                            int64_t bias_type_stride, const int64_t* type_begin, TypedRowsOut out, int64_t ld_out, int accumulate, float* pk, hipStream_t s) {
    const NarrowTiles plan = narrow_tiles(type_begin, 4096);
    if (plan.tile_prefix[3] == 0) return;
    pack_linear_narrow(dim, w, ld_w, w_type_stride, transpose, pk, s);
    const int grid = grid_for_waves(plan.wave_prefix[3]);
    const int single = w_type_stride == 0 ? 1 : 0;
#define IHG_NARROW_RG(D, ACC) hipLaunchKernelGGL((row_gemm_narrow_kernel<D, ACC>), dim3(grid), dim3(kBlockThreads), 0, s, in, ld_in, pk, single, bias, bias_mask, bias_type_stride, plan, out, ld_out)
    if (dim == 32) {
        if (accumulate) IHG_NARROW_RG(32, true);
        else IHG_NARROW_RG(32, false);
    } else {
        if (accumulate) IHG_NARROW_RG(64, true);
        else IHG_NARROW_RG(64, false);
    }
#undef IHG_NARROW_RG
}

int launch_dense_weight_narrow(int dim, const float* dout, int64_t ld_dout, TypedRows x, int64_t ld_x, const int64_t* type_begin, int n_types, float* slabs, float* bias_slabs,
                               const float* w, int64_t ld_w, int64_t w_type_stride, const TypedRowsOut* dx, int64_t ld_dx, int dx_accumulate, float* pk, hipStream_t s) {
    const NarrowTiles plan = narrow_tiles(type_begin);
    const int n_slabs = 256;                                             // = dense.hip's kDenseSlabs: the workspace holds that many per type
    const int single = n_types == 1 ? 1 : 0;
    const TypedRowsOut dxr = dx != nullptr ? *dx : typed_rows_out(nullptr);
    if (dx != nullptr) pack_linear_narrow(dim, w, ld_w, w_type_stride, 1, pk, s);
#define IHG_NARROW_DW(D, DXF, ACC) hipLaunchKernelGGL((dense_weight_grad_narrow_kernel<D, DXF, ACC>), dim3(n_slabs, 1, n_types), dim3(kBlockThreads), 0, s, dout, ld_dout, x, ld_x, plan, single, slabs, bias_slabs, pk, dxr, ld_dx)
    if (dim == 32) {
        if (dx == nullptr) IHG_NARROW_DW(32, false, false);
        else if (dx_accumulate) IHG_NARROW_DW(32, true, true);
        else IHG_NARROW_DW(32, true, false);
    } else {
        if (dx == nullptr) IHG_NARROW_DW(64, false, false);
        else if (dx_accumulate) IHG_NARROW_DW(64, true, true);
        else IHG_NARROW_DW(64, true, false);
    }
#undef IHG_NARROW_DW
    return n_slabs;
}
