// split_node.hip - the NODE-LEVEL kernels of the split arithmetic (d = 64 / 128 / 256): the interactive layer's contraction per node (one launch at d = 128, the 64-column
// pass kernel at d = 64 / 256) and its weight gradients, the node-level linear maps (typed row GEMM) and their weight / bias / input gradients.  The arithmetic
// (two fp16 terms per operand: per-row scales where the contraction runs along a row, a row's two operands balanced against each other where it runs over the rows), the wave roles and the helpers are split_common.hpp's;
// the contractions per hyperedge are in split_arith.hip.  DESIGN.md section 4.
#include "split_common.hpp"

namespace {

// ------------------------------------------------------------------------------------------------
// The interactive layer WITHOUT hyperedge rows (d = 128): node-level form of  out = scale * H FeatureInteractor(h).
// For a node v of type t with incident hyperedges e and their other two members (a_e, b_e), every term of the hyperedge feature is linear
// in the member features once v's own feature is held fixed, so the sum over v's hyperedges is a linear map of
//     deg(v) h[v],   S_a = sum h[a_e],   S_b = sum h[b_e],   S_ab = sum h[a_e] h[b_e]     (ihg_node_pair_sums; products elementwise)
// and of their products with h[v]:
//     sum_e F(e) = deg (A_t h + c) + L_a S_a + P_a (h S_a) + L_b S_b + P_b (h S_b) + L_ab S_ab + W_uqi (h S_ab)
// with the blocks of w = [A_u | A_q | A_i | W_uq | W_qi | W_iu | W_uqi] assigned by node type (kNodeBlocks).  That is a row GEMM over the
// N nodes with a contraction index of 7 d - E / N times fewer multiply-adds than the hyperedge form, no [E, d] tensor, no hyperedge -> node
// pass - run in PASSES like the forward above: a pass contracts one source block and its product with h (256 values: the 192 weight
// registers per matrix wave), adds onto what the earlier passes left in `out`; the last pass applies the output scale.  Matrix waves reload
// their weight registers where a workgroup's tile range crosses a node type.
// wnp[type][pass][m][jt < 2][kb < 8][plane][lane][8]: element i = plane of W[32 m + 16 jt + (lane & 15)][block(type, pass, kb >> 2) d + 32 (kb & 3) + 8 (lane >> 4) + i]
// passes: 0 = {deg h}, 1 = {S_a, h S_a}, 2 = {S_b, h S_b}, 3 = {S_ab, h S_ab}
// ------------------------------------------------------------------------------------------------
// rows grouped by node type in tiles of 32 that do not cross a type: begin[t] = first row of type t, tile_prefix[t] = tiles before type t
struct RowTiles {
    int64_t begin[4];
    int tile_prefix[4];
};

__device__ __forceinline__ int node_block(int type, int pass, int second) {
    // user: a = query, b = item; query: a = user, b = item; item: a = user, b = query     (-1: no such block)
    constexpr signed char kNodeBlocks[3][4][2] = {{{0, -1}, {1, 3}, {2, 5}, {4, 6}}, {{1, -1}, {0, 3}, {2, 4}, {5, 6}}, {{2, -1}, {0, 5}, {1, 4}, {3, 6}}};
    return kNodeBlocks[type][pass][second];
}

// scale of every weight ROW of every node type (= output column j of that type's contraction): wsc[type][j] = scale_up_for(max_k |W_t[j][k]|) over the type's seven blocks,
// winv[type][j] its inverse.  One wave per (type, j).
__global__ __launch_bounds__(kBlockThreads) void node_fwd_weight_scales_kernel(const float* __restrict__ w, int64_t ld_w, int d, int order, float* __restrict__ wsc,
                                                                               float* __restrict__ winv) {
    const int lane = threadIdx.x & 63;
    const int64_t unit = global_wave_id();
    if (unit >= 3 * d) return;
    const int type = static_cast<int>(unit) / d, j = static_cast<int>(unit) % d;
    float m = 0.f;
    for (int pass = 0; pass < 4; ++pass)
        for (int second = 0; second < 2; ++second) {
            int b = node_block(type, pass, second);
            if (b == 6 && order != 3) b = -1;
            if (b < 0) continue;
            const float* src = w + static_cast<int64_t>(j) * ld_w + b * d;
            for (int c = lane; c < d; c += kWave) m = fmaxf(m, fabsf(src[c]));
        }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    float inv;
    const float sc = scale_up_for(m, inv);
    if (lane == 0) {
        wsc[unit] = sc;
        winv[unit] = inv;
    }
}

// wnp[type][pass][m][jt < 2][kb < 8][plane < 2][lane][8 x fp16]: element i = plane of wsc[type][j] W[j][block(type, pass, kb >> 2) d + 32 (kb & 3) + 8 (lane >> 4) + i],
// j = 32 m + 16 jt + (lane & 15)
__global__ __launch_bounds__(kBlockThreads) void pack_planes_node_fwd_kernel(const float* __restrict__ w, int64_t ld_w, int order, const float* __restrict__ wsc,
                                                                             v4u* __restrict__ wnp) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= 3 * 4 * 4 * 2 * 8 * kWave) return;
    const int lane = idx & 63, kb = (idx >> 6) & 7, jt = (idx >> 9) & 1, m = (idx >> 10) & 3, pass = (idx >> 12) & 3, type = idx >> 14;
    int b = node_block(type, pass, kb >> 2);
    if (b == 6 && order != 3) b = -1;
    v4u hi = v4u{0, 0, 0, 0}, lo = v4u{0, 0, 0, 0};
    if (b >= 0) {
        const int j = 32 * m + 16 * jt + (lane & 15);
        const float sc = wsc[type * 128 + j];
        const float* src = w + static_cast<int64_t>(j) * ld_w + b * 128 + 32 * (kb & 3) + 8 * (lane >> 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            unsigned hh, ll;
            split_pair_h2(src[2 * i] * sc, src[2 * i + 1] * sc, hh, ll);
            hi[i] = hh;
            lo[i] = ll;
        }
    }
    wnp[(static_cast<int64_t>(idx >> 6) * 2 + 0) * kWave + lane] = hi;
    wnp[(static_cast<int64_t>(idx >> 6) * 2 + 1) * kWave + lane] = lo;
}

constexpr int kNodePassV4 = 4 * 2 * 8 * 2 * kWave;                      // v4u of one (type, pass)'s planes

// ------------------------------------------------------------------------------------------------
// The node-level contraction at d = 128: FOUR passes over the contraction index ({deg h}, {S_a, h S_a}, {S_b, h S_b}, {S_ab, h S_ab}) as ONE launch that touches
// `out` once.  A matrix wave's weight registers hold the planes of 256
// values of the contraction index - but they run GROUP by group of G = 8 row tiles: for each pass the matrix waves load that pass's planes once per group and contract
// the group's tiles, and the partial sums of the group's rows live in the SERVICE waves' registers between the passes (thread = one row x 16 columns of every tile:
// 16 G = 128 registers), not in `out`.  h and the pair sums are read once from memory (h four times out of L2: a group's rows are 128 KB), `out` is written once:
// 0.96 GB at C3 instead of the 2.8 GB of four launches that each read-modify-write `out` - the four launches are streams at 5.1 TB/s, this one is bound by its matrix
// waves.  No global load is issued and consumed inside one phase (the memory counter is in order: waiting for such a load also waits for the row requests in front of it,
// and the phase becomes as long as a memory round trip - what bounded each of the four launches' phases at ~3 us).
// Measured at C3 (us): four launches, three bf16 terms 560; this kernel with three bf16 terms 650 (matrix waves: 6 MFMAs per product); two fp16 terms: four launches 541,
// this kernel: see DESIGN.md section 4.
// ------------------------------------------------------------------------------------------------
constexpr int kNodeGroupTiles = 8;

struct NodeGroups {
    RowTiles tiles;                                                      // tiles of 32 rows that do not cross a node type
    int group_prefix[4];                                                 // groups of kNodeGroupTiles consecutive tiles of ONE type: groups before type t
};

template <int ORDER>
__global__ __launch_bounds__(kSplitThreads) void node_interact_fwd_grouped_kernel(const float* __restrict__ h, int64_t ld_h, const float* __restrict__ sums, int64_t ld_s,
                                                                                  const float* __restrict__ deg, const float* __restrict__ scale, const float* __restrict__ bias,
                                                                                  const v4u* __restrict__ wnp, const float* __restrict__ winv, NodeGroups plan,
                                                                                  float* __restrict__ out, int64_t ld_out) {
    constexpr int G = kNodeGroupTiles, TE = 32, RT = 2, CSTR = 32, ZRB = 512, ZPL = TE * ZRB, PS = 128 + 4, X = 4, PASSES = 4, KB = 8;
    __shared__ __attribute__((aligned(16))) unsigned char zplanes[2][2][TE][ZRB];
    __shared__ __attribute__((aligned(16))) float part[2][TE][PS];
    __shared__ __attribute__((aligned(16))) float swinv[3][128];
    __shared__ __attribute__((aligned(16))) float sbias[128];            // the aggregation's bias (zeros without one)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (tid < 3 * 128) (&swinv[0][0])[tid] = winv[tid];
    if (tid < 128) sbias[tid] = bias != nullptr ? bias[tid] : 0.f;
    const int total_groups = plan.group_prefix[3];
    const int per = (total_groups + static_cast<int>(gridDim.x) - 1) / static_cast<int>(gridDim.x);
    const int g_begin = static_cast<int>(blockIdx.x) * per;
    const int g_end = std::min(g_begin + per, total_groups);
    if (g_begin >= g_end) return;

    // cursor over this workgroup's phases: (group, pass, tile of the group), all wave-uniform
    struct Cur {
        int gi, p, t, n, type;
        int64_t row0, r_end;                                             // first row of the group, end of the type's rows
    };
    auto open_group = [&](Cur& c) {                                     // (c.gi < g_end)
        const int type = c.gi >= plan.group_prefix[2] ? 2 : (c.gi >= plan.group_prefix[1] ? 1 : 0);
        const int first_tile = (c.gi - plan.group_prefix[type]) * G;     // within the type
        const int type_tiles = plan.tiles.tile_prefix[type + 1] - plan.tiles.tile_prefix[type];
        c.type = type;
        c.n = std::min(G, type_tiles - first_tile);
        c.row0 = plan.tiles.begin[type] + static_cast<int64_t>(first_tile) * TE;
        c.r_end = plan.tiles.begin[type + 1];
        c.p = 0;
        c.t = 0;
    };
    auto advance = [&](Cur& c) {                                        // past the last phase the cursor repeats the last tile (read and dropped)
        if (c.t + 1 < c.n) {
            ++c.t;
        } else if (c.p + 1 < PASSES) {
            ++c.p;
            c.t = 0;
        } else if (c.gi + 1 < g_end) {
            ++c.gi;
            open_group(c);
        }
    };
    int n_phases = 0;
    {
        Cur c;
        for (int gi = g_begin; gi < g_end; ++gi) {
            c.gi = gi;
            open_group(c);
            n_phases += PASSES * c.n;
        }
    }
    // pass p contracts: 0 {deg h}; 1 {S_a, h S_a}; 2 {S_b, h S_b}; 3 {S_ab, h S_ab} (order 2: {S_ab})
    auto blocks_of = [&](int p) { return (p == 0 || (ORDER == 2 && p == 3)) ? 1 : 2; };

    role_priority(wave >= 4);
    if (wave >= 4) {
        // ---------------- service waves: thread -> node row of the tile, columns 4 o + 32 x .. (x < 4)
        const int st = tid - 256, row = st >> 3, o = st & 7;
        struct Piece { v4f hv[X], sv[X]; float d; };
        auto load_piece = [&](const Cur& c, Piece& pc) {
            const int64_t v = std::min(c.row0 + static_cast<int64_t>(c.t) * TE + row, c.r_end - 1);      // rows past the type's end re-read its last row (never stored)
            // (the {deg h} pass uses no pair sum: its request goes to h's address - the same cache lines a second time - instead of fetching a block nobody looks at)
            const float* sp = c.p > 0 ? sums + v * ld_s + (c.p - 1) * 128 + 4 * o : h + v * ld_h + 4 * o;
#pragma unroll
            for (int x = 0; x < X; ++x) {
                if (abl::n_no_first) {                                   // (ablation: no row loads)
                    pc.hv[x] = pc.sv[x] = v4f{1.f, 2.f, 3.f, 4.f} * static_cast<float>(v + x);
                    continue;
                }
                pc.hv[x] = *reinterpret_cast<const v4f*>(h + v * ld_h + 4 * o + CSTR * x);
                // (a block of the pair sums is read by exactly one pass: non-temporal, h keeps the caches)
                pc.sv[x] = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(sp + CSTR * x));
            }
            pc.d = deg[v];
        };
        // the phase's contraction values of this thread's row piece -> scaled by the row's power of two, two fp16 planes; returns the inverse of the scale
        auto split_tile = [&](const Cur& c, const Piece& pc, int buf) {
            typedef unsigned v2u __attribute__((ext_vector_type(2)));
            const int nb = blocks_of(c.p);
            if (abl::n_no_split) return 1.f;
            v4f z[X][2];
            float m = 0.f;
            // the pass kind is wave-uniform: a real branch per kind (the empty asm keeps the compiler from flattening it into 58 selects + the products of both kinds)
            if (c.p == 0) {
                asm volatile("" ::: "memory");
#pragma unroll
                for (int x = 0; x < X; ++x) {
                    z[x][0] = pc.hv[x] * pc.d;
                    z[x][1] = v4f{0.f, 0.f, 0.f, 0.f};
                }
            } else {
                asm volatile("" ::: "memory");
#pragma unroll
                for (int x = 0; x < X; ++x) {
                    z[x][0] = pc.sv[x];
                    z[x][1] = pc.hv[x] * pc.sv[x];
                }
            }
            {
                float mx[X];                                             // (four chains, not one of sixteen dependent maxima; the second block of a one-block pass is zeros)
#pragma unroll
                for (int x = 0; x < X; ++x) mx[x] = abs_max_of(z[x][0], z[x][1], 0.f);
                m = fmaxf(fmaxf(mx[0], mx[1]), fmaxf(mx[2], mx[3]));
            }
            if (!abl::n_no_shuffle) {
                m = row_lanes_max<8>(m);                                 // the row's eight threads are eight consecutive lanes
            }
            float inv;
            const float sc = scale_up_for(m, inv);
#pragma unroll
            for (int x = 0; x < X; ++x)
#pragma unroll
                for (int b2 = 0; b2 < 2; ++b2) {
                    if (b2 >= nb) continue;
                    unsigned h0, l0, h1, l1;
                    split_pair_h2(z[x][b2][0] * sc, z[x][b2][1] * sc, h0, l0);
                    split_pair_h2(z[x][b2][2] * sc, z[x][b2][3] * sc, h1, l1);
                    const int off = row * ZRB + (((16 * b2 + 4 * x + (o >> 1)) ^ (row & 15)) << 4) + 8 * (o & 1);
                    *reinterpret_cast<v2u*>(&zplanes[buf][0][0][0] + off) = v2u{h0, h1};
                    *reinterpret_cast<v2u*>(&zplanes[buf][0][0][0] + ZPL + off) = v2u{l0, l1};
                }
            return inv;
        };
        v4f acc[G][X];                                                   // partial sums of this thread's row piece of every tile of the open group
        // the partial sums of the phase before: unscaled and added to the tile's accumulators; after the last pass the row is finished and stored (no global loads in here)
        auto finish = [&](const Cur& c, int buf, float xinv, float d, float sc) {
            const float (*pp)[PS] = part[buf];
            const int64_t v = c.row0 + static_cast<int64_t>(c.t) * TE + row;
            v4f val[X];
#pragma unroll
            for (int x = 0; x < X; ++x) val[x] = *reinterpret_cast<const v4f*>(&pp[row][4 * o + CSTR * x]) * (*reinterpret_cast<const v4f*>(&swinv[c.type][4 * o + CSTR * x]) * xinv);
            if (c.p == 0) {
#pragma unroll
                for (int x = 0; x < X; ++x) val[x] += *reinterpret_cast<const v4f*>(&sbias[4 * o + CSTR * x]) * d;
            }
            // the open group's tile c.t (wave-uniform): a real branch per tile - flattened into selects (which the compiler does to bodies this small) the update costs
            // 2 x 16 x G v_cndmask per phase and thread, two thirds of the service waves' vector instructions; the empty asm keeps the bodies from being if-converted
#pragma unroll
            for (int t = 0; t < G; ++t) {
                if (c.t == t) {
                    asm volatile("" ::: "memory");
#pragma unroll
                    for (int x = 0; x < X; ++x) {
                        if (c.p != 0) val[x] += acc[t][x];
                        acc[t][x] = val[x];
                    }
                }
            }
            if (c.p == PASSES - 1 && v < c.r_end) {
#pragma unroll
                for (int x = 0; x < X; ++x) *reinterpret_cast<v4f*>(out + v * ld_out + 4 * o + CSTR * x) = val[x] * sc;
            }
        };
        Cur cprev, cnext, cnext2;                                        // phases s - 1, s + 1, s + 2
        cprev.gi = g_begin;
        open_group(cprev);
        cnext = cprev;
        advance(cnext);
        cnext2 = cnext;
        advance(cnext2);
        Piece pc0, pc1;                                                  // values of phase m in pc<m & 1>
        load_piece(cprev, pc0);
        load_piece(cnext, pc1);
        float inv_prev = 1.f, inv_cur = split_tile(cprev, pc0, 0), inv_next = 1.f;     // inverse row scales of phases s - 1, s, s + 1
        __syncthreads();
        const float* const sc_src = scale != nullptr ? scale : deg;
        // phase s: images of phase s + 1 (`use`); partial sums of phase s - 1 into the accumulators (its row finished after the last pass); request: values of phase s + 2 (`fill`)
        auto phase = [&](int s, const Piece& use, Piece& fill) {
            if (abl::n_no_service) {                                     // (one word of the partial sums read and - never - stored: the matrix waves' work stays observable)
                const float probe = part[s & 1][row][4 * o];
                if (probe == 1.2345e30f) out[0] = probe;
                __syncthreads();
                return;
            }
            // the finished phase's degree and output scale: requested FIRST and unconditionally (older than this phase's row requests: the wait for them in finish() leaves the
            // row requests in flight), consumed after the split
            const int64_t vp = std::min(cprev.row0 + static_cast<int64_t>(cprev.t) * TE + row, cprev.r_end - 1);
            const float d = deg[vp];
            float sc = sc_src[vp];
            load_piece(cnext2, fill);
            if (s + 1 < n_phases) inv_next = split_tile(cnext, use, (s + 1) & 1);
            if (scale == nullptr) sc = 1.f;
            if (s >= 1) {
                finish(cprev, (s - 1) & 1, inv_prev, d, sc);
                advance(cprev);
            }
            inv_prev = inv_cur;
            inv_cur = inv_next;
            cnext = cnext2;
            advance(cnext2);
            __syncthreads();
        };
        int s = 0;
#pragma clang loop unroll(disable)
        for (; s + 1 <= n_phases; s += 2) {
            phase(s, pc1, pc0);
            phase(s + 1, pc0, pc1);
        }
        if (s <= n_phases) phase(s, pc1, pc0);
        return;
    }

    // ---------------- matrix waves: wave m = output columns 32 m .. + 31, the pass's whole contraction index
    v8h wreg[2][KB][2];
    Cur c;
    c.gi = g_begin;
    open_group(c);
    __syncthreads();
    const int arow = lane & 15, kq = lane >> 4;
    auto plane_ptr = [&](const Cur& cw) { return wnp + static_cast<int64_t>(cw.type * 4 + cw.p) * kNodePassV4 + lane; };
    auto load_kb = [&](const v4u* wf, int kb) {                          // the planes of k-block kb of this wave's two column tiles
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) wreg[jt][kb][pl] = __builtin_bit_cast(v8h, wf[(static_cast<int64_t>((wave * 2 + jt) * 8 + kb) * 2 + pl) * kWave]);
    };
    {
        const v4u* wf = plane_ptr(c);
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) load_kb(wf, kb);
    }
    // One tile: step = (k-block, row tile), six MFMAs on two accumulators, the fragment of the next step requested in front of them; the two row tiles alternate, so an
    // accumulator's next product is a step away.
    // RELOAD - the last tile of a (type, pass): the planes of the NEXT pass are requested k-block by k-block, each right behind the last MFMAs that read its registers,
    // so that their round trips to L2 run beside the rest of this tile and the barrier instead of in front of the next pass's first MFMA (one request for all of them after
    // the tile: 92 exposed round trips per workgroup at C3; 455 -> 441 us, same box)
    auto tile = [&](bool reload, int s, const v4u* wf_next) {
        const unsigned char* zp = &zplanes[s & 1][0][0][0];
        const int kb_live = 4 * blocks_of(c.p);                          // a one-block pass skips the steps of its empty second block (uniform branches; one MFMA body)
        int ar = arow, kqq = kq;                                         // opaque copies: the sixteen fragment offsets are re-derived per tile (hoisted out of the phase loop
        asm volatile("" : "+v"(ar), "+v"(kqq));                          // they cost sixteen registers beside 128 of weights, and the weights spill)
        auto fragment = [&](int step, v8h (&f)[2]) {
            const int kb = step >> 1, rt = step & 1;
            const unsigned char* src = zp + (16 * rt + ar) * ZRB + (((4 * kb + kqq) ^ ar) << 4);
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) f[pl] = *reinterpret_cast<const v8h*>(src + pl * ZPL);
        };
        v4f acc[RT][2];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int jt = 0; jt < 2; ++jt) acc[rt][jt] = v4f{0.f, 0.f, 0.f, 0.f};
        v8h a[2], an[2];                                                 // (fragments two steps ahead instead of one: 438 - 448 us against 441 - 447, no gain)
        fragment(0, a);
#pragma unroll
        for (int step = 0; step < RT * KB; ++step) {
            const int kb = step >> 1, rt = step & 1;
            if (step + 1 < RT * KB && !abl::n_no_fragments) fragment(step + 1, an);      // (the fragment of a skipped step is read and dropped)
            IHG_PIN_ORDER();
            if (kb < kb_live && !abl::n_no_mfma) {
#pragma unroll
                for (int term = 0; term < 3; ++term)
#pragma unroll
                    for (int jt = 0; jt < 2; ++jt)
                        acc[rt][jt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wreg[jt][kb][kTermB2[term]], a[kTermA2[term]], acc[rt][jt], 0, 0, 0);
            }
            IHG_PIN_ORDER();
            if (rt == RT - 1 && reload) load_kb(wf_next, kb);             // (a uniform branch around four requests: one code path, so that old and new planes share their registers)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) a[pl] = an[pl];
        }
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int jt = 0; jt < 2; ++jt) *reinterpret_cast<v4f*>(&part[s & 1][16 * rt + arow][32 * wave + 16 * jt + 4 * kq]) = acc[rt][jt];
    };
    for (int s = 0; s <= n_phases; ++s) {
        if (s < n_phases) {
            Cur cn = c;
            advance(cn);
            tile(!abl::n_no_reload && s + 1 < n_phases && (cn.type != c.type || cn.p != c.p), s, plane_ptr(cn));
            c = cn;
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// The same one-launch scheme at d = 256.  A matrix wave's 128 weight registers hold the planes of 128 values of the contraction index for 64 output columns, so a pass
// is HALF a source block: 14 passes (order 2: 12) - {deg h} x 2 halves, then for each of the three pair sums S: {S half 0}, {h S half 0}, {S half 1}, {h S half 1} (the two
// passes that read one half of S are neighbours: its second read comes out of L2) - over groups of FOUR 32-row tiles: a service thread keeps one row x 32 output columns of
// every tile of the group (128 registers) and forms / splits 16 contraction values per phase.  Every value is split ONCE per row (the four-launch kernel below splits it in
// each of its four column parts) and `out` is written once (four launches: read-modify-written four times).
// wnx[type][pass][m][jt < 4][kb < 4][plane < 2][lane][8 x fp16]: element i = plane of wsc[type][j] W[j][block(type, xb) d + 128 half + 32 kb + 8 (lane >> 4) + i],
// j = 64 m + 16 jt + (lane & 15), (xb, half) = node_pass256(pass)
// ------------------------------------------------------------------------------------------------
constexpr int kNodeGroupTiles256 = 4;

__device__ __forceinline__ int node_xblock_weight(int type, int xb);

// pass -> source block xb of X = [deg h | S_a | h S_a | S_b | h S_b | S_ab | h S_ab] and half of its 256 columns (order 3: 14 passes; order 2 has no h S_ab: its passes
// 10, 11 are the two halves of S_ab - handled by the callers)
__device__ __forceinline__ void node_pass256(int p, int& xb, int& half) {
    if (p < 2) {
        xb = 0;
        half = p;
    } else {
        const int q = p - 2, sb = q >> 2, r = q & 3;
        xb = 1 + 2 * sb + (r & 1);
        half = r >> 1;
    }
}

template <int ORDER>
__global__ __launch_bounds__(kBlockThreads) void pack_planes_node_fwd256_kernel(const float* __restrict__ w, int64_t ld_w, const float* __restrict__ wsc, v4u* __restrict__ wnx) {
    constexpr int PASSES = ORDER == 3 ? 14 : 12;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= 3 * PASSES * 4 * 4 * 4 * kWave) return;
    const int lane = idx & 63, kb = (idx >> 6) & 3, jt = (idx >> 8) & 3, m = (idx >> 10) & 3, pass = (idx >> 12) % PASSES, type = (idx >> 12) / PASSES;
    int xb, half;
    if (ORDER == 3) {
        node_pass256(pass, xb, half);
    } else if (pass < 10) {
        node_pass256(pass, xb, half);
    } else {                                                             // order 2: S_ab half 0, half 1
        xb = 5;
        half = pass - 10;
    }
    const int b = node_xblock_weight(type, xb);
    const int j = 64 * m + 16 * jt + (lane & 15);
    const float sc = wsc[type * 256 + j];
    const float* src = w + static_cast<int64_t>(j) * ld_w + b * 256 + 128 * half + 32 * kb + 8 * (lane >> 4);
    v4u hi, lo;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        unsigned hh, ll;
        split_pair_h2(src[2 * i] * sc, src[2 * i + 1] * sc, hh, ll);
        hi[i] = hh;
        lo[i] = ll;
    }
    wnx[(static_cast<int64_t>(idx >> 6) * 2 + 0) * kWave + lane] = hi;
    wnx[(static_cast<int64_t>(idx >> 6) * 2 + 1) * kWave + lane] = lo;
}

template <int ORDER>
__global__ __launch_bounds__(kSplitThreads) void node_interact_fwd_grouped256_kernel(const float* __restrict__ h, int64_t ld_h, const float* __restrict__ sums, int64_t ld_s,
                                                                                     const float* __restrict__ deg, const float* __restrict__ scale,
                                                                                     const float* __restrict__ bias, const v4u* __restrict__ wnx,
                                                                                     const float* __restrict__ winv, NodeGroups plan, float* __restrict__ out, int64_t ld_out) {
    constexpr int D = 256, G = kNodeGroupTiles256, TE = 32, RT = 2, JT = 4, KB = 4, CSTR = 32, ZRB = 256, ZPL = TE * ZRB, PS = D + 4, X = 4, XO = 8;
    constexpr int PASSES = ORDER == 3 ? 14 : 12, PASS_V4 = 4 * JT * KB * 2 * kWave;
    __shared__ __attribute__((aligned(16))) unsigned char zplanes[2][2][TE][ZRB];
    __shared__ __attribute__((aligned(16))) float part[2][TE][PS];
    __shared__ __attribute__((aligned(16))) float swinv[3][D];
    __shared__ __attribute__((aligned(16))) float sbias[D];              // the aggregation's bias (zeros without one)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 3 * D; i += kSplitThreads) (&swinv[0][0])[i] = winv[i];
    if (tid < D) sbias[tid] = bias != nullptr ? bias[tid] : 0.f;
    const int total_groups = plan.group_prefix[3];
    const int per = (total_groups + static_cast<int>(gridDim.x) - 1) / static_cast<int>(gridDim.x);
    const int g_begin = static_cast<int>(blockIdx.x) * per;
    const int g_end = std::min(g_begin + per, total_groups);
    if (g_begin >= g_end) return;

    // cursor over this workgroup's phases: (group, pass, tile of the group), all wave-uniform; xb / half: the pass's source block and column half
    struct Cur {
        int gi, p, t, n, type, xb, half;
        int64_t row0, r_end;                                             // first row of the group, end of the type's rows
    };
    auto set_pass = [&](Cur& c) {
        if (ORDER == 2 && c.p >= 10) {
            c.xb = 5;
            c.half = c.p - 10;
        } else {
            node_pass256(c.p, c.xb, c.half);
        }
    };
    auto open_group = [&](Cur& c) {                                     // (c.gi < g_end)
        const int type = c.gi >= plan.group_prefix[2] ? 2 : (c.gi >= plan.group_prefix[1] ? 1 : 0);
        const int first_tile = (c.gi - plan.group_prefix[type]) * G;     // within the type
        const int type_tiles = plan.tiles.tile_prefix[type + 1] - plan.tiles.tile_prefix[type];
        c.type = type;
        c.n = std::min(G, type_tiles - first_tile);
        c.row0 = plan.tiles.begin[type] + static_cast<int64_t>(first_tile) * TE;
        c.r_end = plan.tiles.begin[type + 1];
        c.p = 0;
        c.t = 0;
        set_pass(c);
    };
    auto advance = [&](Cur& c) {                                        // past the last phase the cursor repeats the last tile (read and dropped)
        if (c.t + 1 < c.n) {
            ++c.t;
        } else if (c.p + 1 < PASSES) {
            ++c.p;
            c.t = 0;
            set_pass(c);
        } else if (c.gi + 1 < g_end) {
            ++c.gi;
            open_group(c);
        }
    };
    int n_phases = 0;
    {
        Cur c;
        for (int gi = g_begin; gi < g_end; ++gi) {
            c.gi = gi;
            open_group(c);
            n_phases += PASSES * c.n;
        }
    }

    role_priority(wave >= 4);
    if (wave >= 4) {
        // ---------------- service waves: thread -> node row of the tile; contraction values at columns 128 half + 4 o + 32 x (x < 4), output columns 4 o + 32 x (x < 8)
        const int st = tid - 256, row = st >> 3, o = st & 7;
        struct Piece { v4f hv[X], sv[X]; float d; };
        // both operands are requested in every pass (a branch around requests makes the compiler wait for all of them): a pass that does not use one reads bytes its
        // neighbours use anyway
        auto load_piece = [&](const Cur& c, Piece& pc) {
            const int64_t v = std::min(c.row0 + static_cast<int64_t>(c.t) * TE + row, c.r_end - 1);      // rows past the type's end re-read its last row (never stored)
            const int sb = c.xb > 0 ? (c.xb - 1) >> 1 : 0;
            const float* hp = h + v * ld_h + 128 * c.half + 4 * o;
            const float* sp = sums + v * ld_s + sb * D + 128 * c.half + 4 * o;
            // ... but the operand a pass does not use is requested at the OTHER operand's address (the same cache lines a second time: no traffic): 3 of a half's 7 passes
            // do not use h, the {deg h} pass no pair sum - 40 of 149 GB at C5 were rows nobody looked at
            const bool need_h = (c.xb & 1) == 0, need_s = c.xb > 0;
            const float* hq = need_h ? hp : sp;
            const float* sq = need_s ? sp : hp;
#pragma unroll
            for (int x = 0; x < X; ++x) {
                pc.hv[x] = *reinterpret_cast<const v4f*>(hq + CSTR * x);
                pc.sv[x] = *reinterpret_cast<const v4f*>(sq + CSTR * x);
            }
            pc.d = deg[v];
        };
        // the phase's contraction values of this thread's row piece -> scaled by the row's power of two, two fp16 planes; returns the inverse of the scale
        auto split_tile = [&](const Cur& c, const Piece& pc, int buf) {
            typedef unsigned v2u __attribute__((ext_vector_type(2)));
            v4f z[X];
            // the pass kind is wave-uniform: a real branch per kind (the empty asm keeps the compiler from flattening it into selects + the products of every kind)
            if (c.xb == 0) {
                asm volatile("" ::: "memory");
#pragma unroll
                for (int x = 0; x < X; ++x) z[x] = pc.hv[x] * pc.d;
            } else if (c.xb & 1) {
                asm volatile("" ::: "memory");
#pragma unroll
                for (int x = 0; x < X; ++x) z[x] = pc.sv[x];
            } else {
                asm volatile("" ::: "memory");
#pragma unroll
                for (int x = 0; x < X; ++x) z[x] = pc.hv[x] * pc.sv[x];
            }
            float m = fmaxf(abs_max_of(z[0], z[1], 0.f), abs_max_of(z[2], z[3], 0.f));
            m = row_lanes_max<8>(m);                                     // the row's eight threads are eight consecutive lanes
            float inv;
            const float sc = scale_up_for(m, inv);
#pragma unroll
            for (int x = 0; x < X; ++x) {
                unsigned h0, l0, h1, l1;
                split_pair_h2(z[x][0] * sc, z[x][1] * sc, h0, l0);
                split_pair_h2(z[x][2] * sc, z[x][3] * sc, h1, l1);
                const int off = row * ZRB + (((4 * x + (o >> 1)) ^ (row & 15)) << 4) + 8 * (o & 1);
                *reinterpret_cast<v2u*>(&zplanes[buf][0][0][0] + off) = v2u{h0, h1};
                *reinterpret_cast<v2u*>(&zplanes[buf][0][0][0] + ZPL + off) = v2u{l0, l1};
            }
            return inv;
        };
        v4f acc[G][XO];                                                  // partial sums of this thread's row piece of every tile of the open group
        // the partial sums of the phase before: unscaled and added to the tile's accumulators; after the last pass the row is finished and stored (no global loads in here).
        // In two halves of four 16-byte pieces: one set of temporaries live at a time.
        auto finish = [&](const Cur& c, int buf, float xinv, float d, float sc) {
            const float (*pp)[PS] = part[buf];
            const int64_t v = c.row0 + static_cast<int64_t>(c.t) * TE + row;
#pragma unroll
            for (int hx = 0; hx < XO; hx += 4) {
                v4f val[4];
#pragma unroll
                for (int x = 0; x < 4; ++x)
                    val[x] = *reinterpret_cast<const v4f*>(&pp[row][4 * o + CSTR * (hx + x)]) * (*reinterpret_cast<const v4f*>(&swinv[c.type][4 * o + CSTR * (hx + x)]) * xinv);
                if (c.p == 0) {
#pragma unroll
                    for (int x = 0; x < 4; ++x) val[x] += *reinterpret_cast<const v4f*>(&sbias[4 * o + CSTR * (hx + x)]) * d;
                }
#pragma unroll
                for (int t = 0; t < G; ++t) {
                    if (c.t == t) {
                        asm volatile("" ::: "memory");
#pragma unroll
                        for (int x = 0; x < 4; ++x) {
                            if (c.p != 0) val[x] += acc[t][hx + x];
                            acc[t][hx + x] = val[x];
                        }
                    }
                }
                if (c.p == PASSES - 1 && v < c.r_end) {
#pragma unroll
                    for (int x = 0; x < 4; ++x) *reinterpret_cast<v4f*>(out + v * ld_out + 4 * o + CSTR * (hx + x)) = val[x] * sc;
                }
            }
        };
        Cur cprev, cnext, cnext2;                                        // phases s - 1, s + 1, s + 2
        cprev.gi = g_begin;
        open_group(cprev);
        cnext = cprev;
        advance(cnext);
        cnext2 = cnext;
        advance(cnext2);
        Piece pc0, pc1;                                                  // values of phase m in pc<m & 1>
        load_piece(cprev, pc0);
        load_piece(cnext, pc1);
        float inv_prev = 1.f, inv_cur = split_tile(cprev, pc0, 0), inv_next = 1.f;     // inverse row scales of phases s - 1, s, s + 1
        __syncthreads();
        const float* const sc_src = scale != nullptr ? scale : deg;
        // phase s: images of phase s + 1 (`use`); partial sums of phase s - 1 into the accumulators (its row finished after the last pass); request: values of phase s + 2 (`fill`)
        auto phase = [&](int s, const Piece& use, Piece& fill) {
            // the finished phase's degree and output scale: requested FIRST and unconditionally (older than this phase's row requests: the wait for them in finish() leaves the
            // row requests in flight), consumed after the split
            const int64_t vp = std::min(cprev.row0 + static_cast<int64_t>(cprev.t) * TE + row, cprev.r_end - 1);
            const float d = deg[vp];
            float sc = sc_src[vp];
            load_piece(cnext2, fill);
            if (s + 1 < n_phases) inv_next = split_tile(cnext, use, (s + 1) & 1);
            if (scale == nullptr) sc = 1.f;
            if (s >= 1) {
                finish(cprev, (s - 1) & 1, inv_prev, d, sc);
                advance(cprev);
            }
            inv_prev = inv_cur;
            inv_cur = inv_next;
            cnext = cnext2;
            advance(cnext2);
            __syncthreads();
        };
        int s = 0;
#pragma clang loop unroll(disable)
        for (; s + 1 <= n_phases; s += 2) {
            phase(s, pc1, pc0);
            phase(s + 1, pc0, pc1);
        }
        if (s <= n_phases) phase(s, pc1, pc0);
        return;
    }

    // ---------------- matrix waves: wave m = output columns 64 m .. + 63, the pass's 128 contraction values
    v8h wreg[JT][KB][2];
    Cur c;
    c.gi = g_begin;
    open_group(c);
    __syncthreads();
    const int arow = lane & 15, kq = lane >> 4;
    auto plane_ptr = [&](const Cur& cw) { return wnx + static_cast<int64_t>(cw.type * PASSES + cw.p) * PASS_V4 + lane; };
    auto load_kb = [&](const v4u* wf, int kb) {                          // the planes of k-block kb of this wave's four column tiles
#pragma unroll
        for (int jt = 0; jt < JT; ++jt)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) wreg[jt][kb][pl] = __builtin_bit_cast(v8h, wf[(static_cast<int64_t>((wave * JT + jt) * KB + kb) * 2 + pl) * kWave]);
    };
    {
        const v4u* wf = plane_ptr(c);
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) load_kb(wf, kb);
    }
    // one tile: step = (k-block, row tile), twelve MFMAs on four accumulators, the fragment of the next step requested in front of them; `reload`: the last tile of a
    // (type, pass) - the next pass's planes requested k-block by k-block, each right behind the last MFMAs that read its registers
    auto tile = [&](bool reload, int s, const v4u* wf_next) {
        const unsigned char* zp = &zplanes[s & 1][0][0][0];
        int ar = arow, kqq = kq;                                         // opaque copies: the fragment offsets are re-derived per tile, not kept in registers across the loop
        asm volatile("" : "+v"(ar), "+v"(kqq));
        auto fragment = [&](int step, v8h (&f)[2]) {
            const int kb = step >> 1, rt = step & 1;
            const unsigned char* src = zp + (16 * rt + ar) * ZRB + (((4 * kb + kqq) ^ ar) << 4);
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) f[pl] = *reinterpret_cast<const v8h*>(src + pl * ZPL);
        };
        v4f acc[RT][JT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int jt = 0; jt < JT; ++jt) acc[rt][jt] = v4f{0.f, 0.f, 0.f, 0.f};
        v8h a[2], an[2];
        fragment(0, a);
#pragma unroll
        for (int step = 0; step < RT * KB; ++step) {
            const int kb = step >> 1, rt = step & 1;
            if (step + 1 < RT * KB) fragment(step + 1, an);
            IHG_PIN_ORDER();
#pragma unroll
            for (int term = 0; term < 3; ++term)
#pragma unroll
                for (int jt = 0; jt < JT; ++jt)
                    acc[rt][jt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wreg[jt][kb][kTermB2[term]], a[kTermA2[term]], acc[rt][jt], 0, 0, 0);
            IHG_PIN_ORDER();
            if (rt == RT - 1 && reload) load_kb(wf_next, kb);             // (a uniform branch around the requests: one code path, so that old and new planes share their registers)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) a[pl] = an[pl];
        }
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int jt = 0; jt < JT; ++jt) *reinterpret_cast<v4f*>(&part[s & 1][16 * rt + arow][64 * wave + 16 * jt + 4 * kq]) = acc[rt][jt];
    };
    for (int s = 0; s <= n_phases; ++s) {
        if (s < n_phases) {
            Cur cn = c;
            advance(cn);
            tile(s + 1 < n_phases && (cn.type != c.type || cn.p != c.p), s, plane_ptr(cn));
            c = cn;
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// The node-level forward for d = 64 (two fp16 terms per operand, like the grouped kernels): a workgroup owns the 64 output columns and all 448 values of the
// contraction index in ONE pass - matrix wave m: 16 columns x 448 values = 112 weight registers - over tiles of 16 node rows.  (The template still carries the pass /
// column-part geometry it served d = 128 and d = 256 with - d = 128: 592 us against 543 for four passes of 256 values, then the grouped kernel; d = 256: four launches x four
// column parts, every value split in each part and `out` read-modify-written four times: 60 ms at C5 against 38 for the grouped kernel above - only <64, 0> is instantiated.)
// A row's values go in scaled by ONE power of two (its 16 service threads agree on the largest magnitude), a weight row by one over the
// type's seven blocks (node_fwd_weight_scales_kernel); the partial sums leave through the two inverses in the service threads' epilogue.
// wnq[type][pass][part][m][kb < 16][plane < 2][lane][8 x fp16]: element i = plane of wsc[type][j] W[j][block(type, xb) d + c], j = 64 part + 16 m + (lane & 15), where
//   kk = 32 kb + 8 (lane >> 4) + i,  xb = pass (512 / d) + kk / d,  c = kk % d     (xb > 6, or the uqi block at order 2: zeros)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int node_xblock_weight(int type, int xb) {
    constexpr signed char kBlock[3][7] = {{0, 1, 3, 2, 5, 4, 6}, {1, 0, 3, 2, 4, 5, 6}, {2, 0, 5, 1, 4, 3, 6}};
    return kBlock[type][xb];
}

__global__ __launch_bounds__(kBlockThreads) void pack_planes_node_fwd_q_kernel(const float* __restrict__ w, int64_t ld_w, int d, int order, const float* __restrict__ wsc,
                                                                               v4u* __restrict__ wnq) {
    const int bpp = 512 / d, n_pass = (7 + bpp - 1) / bpp, parts = d / 64;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= 3 * n_pass * parts * 4 * 16 * kWave) return;
    const int lane = idx & 63, kb = (idx >> 6) & 15, m = (idx >> 10) & 3;
    int rest = idx >> 12;
    const int part = rest % parts;
    rest /= parts;
    const int pass = rest % n_pass, type = rest / n_pass;
    const int kk = 32 * kb + 8 * (lane >> 4), xb = pass * bpp + kk / d, c = kk % d;
    int b = xb < 7 ? node_xblock_weight(type, xb) : -1;
    if (b == 6 && order != 3) b = -1;
    v4u hi = v4u{0u, 0u, 0u, 0u}, lo = hi;
    if (b >= 0) {
        const int j = 64 * part + 16 * m + (lane & 15);
        const float sc = wsc[type * d + j];
        const float* src = w + static_cast<int64_t>(j) * ld_w + static_cast<int64_t>(b) * d + c;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            unsigned hh, ll;
            split_pair_h2(src[2 * i] * sc, src[2 * i + 1] * sc, hh, ll);
            hi[i] = hh;
            lo[i] = ll;
        }
    }
    wnq[(static_cast<int64_t>(idx >> 6) * 2 + 0) * kWave + lane] = hi;
    wnq[(static_cast<int64_t>(idx >> 6) * 2 + 1) * kWave + lane] = lo;
}

template <int D, int PASS, bool ACC, bool FINAL>
__global__ __launch_bounds__(kSplitThreads) void node_interact_fwd_q_kernel(const float* __restrict__ h, int64_t ld_h, const float* __restrict__ sums, int64_t ld_s,
                                                                            const float* __restrict__ deg, const float* __restrict__ scale, const float* __restrict__ bias,
                                                                            const v4u* __restrict__ wnq, const float* __restrict__ winv, RowTiles plan,
                                                                            float* __restrict__ out, int64_t ld_out) {
    constexpr int BPP = 512 / D, XB0 = PASS * BPP, NB = (7 - XB0) < BPP ? (7 - XB0) : BPP, NPASS = (7 + BPP - 1) / BPP, PARTS = D / 64;
    constexpr int TE = 16, KB = NB * D / 32, ZRB = (2 * NB * D + 255) / 256 * 256, ZPL = TE * ZRB, PS = 64 + 4, X = D / 64;
    constexpr bool NEED_A = XB0 <= 2 && XB0 + NB > 1, NEED_B = XB0 <= 4 && XB0 + NB > 3, NEED_AB = XB0 + NB > 5, NEED_DEG = XB0 == 0;
    static_assert(NB >= 1 && KB <= 16, "pass shape");
    __shared__ __attribute__((aligned(16))) unsigned char zplanes[2][2][TE][ZRB];
    __shared__ __attribute__((aligned(16))) float part[2][TE][PS];
    __shared__ __attribute__((aligned(16))) float swinv[3][64];          // inverse scales of this part's 64 weight rows, per node type
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bid = blockIdx.x;
    const int cpart = PARTS == 1 ? 0 : (bid >> 3) & (PARTS - 1);
    const int range = PARTS == 1 ? bid : (bid & 7) + 8 * (bid / (8 * PARTS)), n_ranges = static_cast<int>(gridDim.x) / PARTS;
    const int total_tiles = plan.tile_prefix[3];
    const int per = (total_tiles + n_ranges - 1) / n_ranges;
    const int t0 = range * per;
    const int n_my = std::max(0, std::min(per, total_tiles - t0));
    if (n_my == 0) return;
    const int coff = 64 * cpart;
    auto tile_rows = [&](int k, int64_t& r_base, int64_t& r_end) {      // (tiles past the range: the graph's last tile, read and dropped)
        const int tile_id = std::min(t0 + k, total_tiles - 1);
        const int type = tile_id >= plan.tile_prefix[2] ? 2 : (tile_id >= plan.tile_prefix[1] ? 1 : 0);
        r_base = plan.begin[type] + static_cast<int64_t>(tile_id - plan.tile_prefix[type]) * TE;
        r_end = plan.begin[type + 1];
        return type;
    };

    role_priority(wave >= 4);
    if (wave >= 4) {
        // ---------------- service waves: thread -> node row of the tile, source columns 4 o + 64 x .. (x < D / 64), output columns coff + 4 o ..
        const int st = tid - 256, row = st >> 4, o = st & 15;
        if (st < 3 * 64) swinv[st >> 6][st & 63] = winv[(st >> 6) * D + coff + (st & 63)];
        struct Piece { v4f hv[X], sa[X], sb[X], sab[X]; float d; };
        auto load_piece = [&](int k, Piece& pc) {
            int64_t r_base, r_end;
            tile_rows(k, r_base, r_end);
            const int64_t v = std::min(r_base + row, r_end - 1);
            const float* hp = h + v * ld_h + 4 * o;
            const float* sp = sums + v * ld_s + 4 * o;
            if (NEED_DEG) pc.d = deg[v];
#pragma unroll
            for (int x = 0; x < X; ++x) {
                pc.hv[x] = *reinterpret_cast<const v4f*>(hp + 64 * x);
                if (NEED_A) pc.sa[x] = *reinterpret_cast<const v4f*>(sp + 64 * x);
                if (NEED_B) pc.sb[x] = *reinterpret_cast<const v4f*>(sp + D + 64 * x);
                if (NEED_AB) pc.sab[x] = *reinterpret_cast<const v4f*>(sp + 2 * D + 64 * x);
            }
        };
        // what the epilogue of a tile adds or multiplies: requested a phase BEFORE its use, in front of that phase's row requests - the memory counter is in order,
        // a request issued and consumed inside one phase would make the phase wait for every row request in front of it
        struct First { v4f old; float d, sc; };
        auto load_first = [&](int k, First& f) {
            int64_t r_base, r_end;
            tile_rows(k, r_base, r_end);
            const int64_t v = std::min(r_base + row, r_end - 1);
            if (ACC) f.old = *reinterpret_cast<const v4f*>(out + v * ld_out + coff + 4 * o);
            else f.d = deg[v];
            if (FINAL) f.sc = scale != nullptr ? scale[v] : 1.f;
        };
        const v4f bias4 = (!ACC && bias != nullptr) ? *reinterpret_cast<const v4f*>(bias + coff + 4 * o) : v4f{0.f, 0.f, 0.f, 0.f};
        auto split_tile = [&](const Piece& pc, int buf, float& inv) {
            typedef unsigned v2u __attribute__((ext_vector_type(2)));
            v4f z[X][NB];
            float m = 0.f;
#pragma unroll
            for (int x = 0; x < X; ++x) {
#pragma unroll
                for (int b2 = 0; b2 < NB; ++b2) {
                    const int xb = XB0 + b2;
                    z[x][b2] = xb == 0 ? pc.hv[x] * pc.d : xb == 1 ? pc.sa[x] : xb == 2 ? pc.hv[x] * pc.sa[x] : xb == 3 ? pc.sb[x] : xb == 4 ? pc.hv[x] * pc.sb[x]
                               : xb == 5 ? pc.sab[x] : pc.hv[x] * pc.sab[x];
                    m = abs_max3(z[x][b2][2], z[x][b2][3], abs_max3(z[x][b2][0], z[x][b2][1], m));
                }
            }
            m = row_lanes_max<16>(m);
            const float sc = scale_up_for(m, inv);
#pragma unroll
            for (int x = 0; x < X; ++x) {
#pragma unroll
                for (int b2 = 0; b2 < NB; ++b2) {
                    unsigned h0, l0, h1, l1;
                    split_pair_h2(z[x][b2][0] * sc, z[x][b2][1] * sc, h0, l0);
                    split_pair_h2(z[x][b2][2] * sc, z[x][b2][3] * sc, h1, l1);
                    // columns D b2 + 64 x + 4 o ..: chunk (D / 8) b2 + 8 x + (o >> 1), half o & 1
                    const int off = row * ZRB + ((((D / 8) * b2 + 8 * x + (o >> 1)) ^ row) << 4) + 8 * (o & 1);
                    *reinterpret_cast<v2u*>(&zplanes[buf][0][0][0] + off) = v2u{h0, h1};
                    *reinterpret_cast<v2u*>(&zplanes[buf][0][0][0] + ZPL + off) = v2u{l0, l1};
                }
            }
        };
        auto epilogue = [&](int k, const First& f, float inv) {
            int64_t r_base, r_end;
            const int type = tile_rows(k, r_base, r_end);
            const int64_t v = r_base + row;
            v4f val = *reinterpret_cast<const v4f*>(&part[k & 1][row][4 * o]) * (*reinterpret_cast<const v4f*>(&swinv[type][4 * o]) * inv);
            if (ACC) val += f.old;
            else val += bias4 * f.d;
            if (FINAL) val *= f.sc;
            if (v < r_end) *reinterpret_cast<v4f*>(out + v * ld_out + coff + 4 * o) = val;
        };
        Piece pc0, pc1;
        First f0, f1;                                                    // of tile m in f<m & 1>
        float inv0 = 1.f, inv1 = 1.f;                                    // inverse row scale of tile m in inv<m & 1>
        load_first(0, f0);
        load_piece(0, pc0);
        load_piece(1, pc1);
        split_tile(pc0, 0, inv0);
        __syncthreads();
        // phase k: request of tile k + 2's rows and tile k's epilogue operands; epilogue of tile k - 1 (its scale's slot is then free); images of tile k + 1
        auto phase = [&](int k, const Piece& use, Piece& fill, First& f_req, const First& f_use, float& inv_slot) {
            if (k >= 1) load_first(k, f_req);
            load_piece(k + 2, fill);
            if (k >= 1) epilogue(k - 1, f_use, inv_slot);
            if (k + 1 < n_my) split_tile(use, (k + 1) & 1, inv_slot);
            __syncthreads();
        };
        int k = 0;
#pragma clang loop unroll(disable)
        for (; k + 1 <= n_my; k += 2) {
            phase(k, pc1, pc0, f0, f1, inv1);
            phase(k + 1, pc0, pc1, f1, f0, inv0);
        }
        if (k <= n_my) phase(k, pc1, pc0, f0, f1, inv1);
        return;
    }

    // ---------------- matrix waves: wave m = output columns coff + 16 m .. + 15, the pass's whole contraction index
    v8h wreg[KB][2];
    int cur_type = -1;
    __syncthreads();
    const int arow = lane & 15, kq = lane >> 4;
    for (int k = 0; k <= n_my; ++k) {
        if (k < n_my) {
            int64_t r_base, r_end;
            const int type = tile_rows(k, r_base, r_end);
            if (type != cur_type) {
                const v4u* wf = wnq + (static_cast<int64_t>((type * NPASS + PASS) * PARTS + cpart) * 4 + wave) * (16 * 2 * kWave) + lane;
#pragma unroll
                for (int kb = 0; kb < KB; ++kb)
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl) wreg[kb][pl] = __builtin_bit_cast(v8h, wf[(kb * 2 + pl) * kWave]);
                cur_type = type;
            }
            const unsigned char* zp = &zplanes[k & 1][0][0][0] + arow * ZRB;
            auto fragment = [&](int kb, v8h (&a)[2]) {
                const unsigned char* src = zp + (((4 * kb + kq) ^ arow) << 4);
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) a[pl] = *reinterpret_cast<const v8h*>(src + pl * ZPL);
            };
            v4f acc[2] = {v4f{0.f, 0.f, 0.f, 0.f}, v4f{0.f, 0.f, 0.f, 0.f}};       // one column tile per wave: two chains, alternate products
            v8h a[2], an[2];
            fragment(0, a);
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
                if (kb + 1 < KB) fragment(kb + 1, an);
                IHG_PIN_ORDER();
#pragma unroll
                for (int term = 0; term < 3; ++term)
                    acc[(kb + term) & 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wreg[kb][kTermB2[term]], a[kTermA2[term]], acc[(kb + term) & 1], 0, 0, 0);
                IHG_PIN_ORDER();
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) a[pl] = an[pl];
            }
            *reinterpret_cast<v4f*>(&part[k & 1][arow][16 * wave + 4 * kq]) = acc[0] + acc[1];
        }
        __syncthreads();
    }
}


// ------------------------------------------------------------------------------------------------
// Weight gradients of the interactive layer's product blocks at NODE level (d = 64 / 128 / 256; column parts as in the hyperedge form).  The layer's output row is
//     y[v] = scale[v] * ( ... + P_a (h S_a) + P_b (h S_b) + L_ab S_ab + W_uqi (h S_ab) )      (node_interact_fwd_kernel),
// so d W_block = sum over the nodes of (scale dy)[v] x X_block[v]^T with X = [h S_a | h S_b | S_ab | h S_ab] - N rows instead of E hyperedges, no
// member gathers.  The kernel is the hyperedge form's (interact_bwd_weight_split_ws_kernel: matrix wave b = block b with 8 x 4 accumulator
// tiles, transposed reads of row-major bf16 images, one slab per tile range) behind another front end: node rows of one type, their
// cotangent, feature and pair-sum rows as straight streams.  A tile range belongs to ONE node type (the assignment of X blocks to the
// blocks of w depends on the type); node_weight_reduce_kernel adds the ranges of each type into the w blocks that type's X blocks stand for.
// Arithmetic: TWO fp16 terms per operand, three MFMA products (the hyperedge form's kernel: three bf16 terms, six).  The contraction runs over the ROWS, so a row
// scale does not factor out of a column of the result - but d W = sum_v a_v x b_v^T is unchanged by a_v 2^s, b_v 2^-s for ANY per-row s: each row's two operands are
// scaled against each other (their largest magnitudes meet at 2^m, both inside fp16's range with 22 bits kept) under ONE power of two per workgroup that follows the
// largest la + lb of the rows seen so far; when a tile raises it, the resident accumulators are brought to the new scale first (a multiplication by a power of two:
// exact; it happens a few times per workgroup).  A row far below the largest loses low bits - but BOTH its operands are small, so its products are small squared: what
// it can add to any entry's error is bounded by 2^-25 of the largest row's product however many such rows there are (a single scale per operand without the balancing
// is not: 2^18 small rows x 2^-31 each).  A tile's exponents are published (tile_l) a phase before its split - from rows requested a phase before that: three sets of
// rows - so that every thread of both roles derives the same running scale without a second barrier per tile.  291 -> 225 us at C3 (same box).
// ------------------------------------------------------------------------------------------------
struct NodeRanges {
    int64_t begin[4];      // first row of every node type
    int range_prefix[4];   // tile ranges before type t (range_prefix[3] = ranges in all)
};

template <int D, int NBLK>
__global__ __launch_bounds__(kSplitThreads) void node_interact_weight_split_kernel(const float* __restrict__ h, int64_t ld_h, const float* __restrict__ sums, int64_t ld_s,
                                                                                   const float* __restrict__ dy, int64_t ld_dy, const float* __restrict__ dy_scale,
                                                                                   NodeRanges plan, float* __restrict__ slabs) {
    constexpr int TE = kSplitTE, PARTS = D == 64 ? 1 : (D == 128 ? 2 : 8), HC = D / PARTS, CT = HC / 16, JT = D / 16;
    constexpr int DRB = 2 * D < 256 ? 256 : 2 * D, ZRB = 8 * HC, DOCT = D / 64, ZX = HC / 32, DPL = TE * DRB, ZPL = TE * ZRB;
    constexpr int kTarget = 24;                                          // twice the exponent the largest balanced row is scaled to (2^12: headroom for the odd half-step)
    __shared__ __attribute__((aligned(16))) unsigned char dplanes[2][2][TE][DRB];
    __shared__ __attribute__((aligned(16))) unsigned char zplanes[2][2][TE][ZRB];
    __shared__ __attribute__((aligned(16))) int tile_l[4][4];            // per tile (mod 4) and service wave: the largest la + lb (biased exponents) of the wave's eight rows, -1: none
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bid = blockIdx.x;
    const int half = (bid >> 3) & (PARTS - 1), range = (bid & 7) + 8 * (bid / (8 * PARTS));      // `half`: this workgroup's column part
    const int type = range >= plan.range_prefix[2] ? 2 : (range >= plan.range_prefix[1] ? 1 : 0);
    const int64_t r_begin = plan.begin[type], r_end = plan.begin[type + 1];
    const int n_tiles = static_cast<int>((r_end - r_begin + TE - 1) / TE);
    const int n_ranges = plan.range_prefix[type + 1] - plan.range_prefix[type];
    const int per = n_ranges > 0 ? (n_tiles + n_ranges - 1) / n_ranges : 0;
    const int t0 = (range - plan.range_prefix[type]) * per;
    const int n_my = range < plan.range_prefix[3] ? std::max(0, std::min(per, n_tiles - t0)) : 0;
    // the running scale: l_max = the largest la + lb over the rows of the tiles so far (-1: none yet); both roles derive it from tile_l in the same way
    auto tile_max = [&](int k) {
        const int* t = tile_l[k & 3];
        return std::max(std::max(t[0], t[1]), std::max(t[2], t[3]));
    };

    role_priority(wave >= 4);
    if (wave >= 4) {
        // ---------------- split waves: thread -> node row of the tile, cotangent octets o and o + 8, columns HC half + 4 o + 32 x .. of h and the pair sums
        const int st = tid - 256, row = st >> 3, o = st & 7;
        struct Rows {
            v4f d[2 * DOCT], hv[ZX], sa[ZX], sb[ZX], sab[ZX];
            float sc, keep;                                              // the cotangent row's factor as loaded (dy_scale[v], or a stand-in word) and 1 / 0 (past the end): applied by publish()
            int ea, eb;                                                  // the row's exponents (publish), used again by the split
        };
        auto load_rows = [&](int k, Rows& r) {
            const int64_t first = r_begin + static_cast<int64_t>(std::min(t0 + k, n_tiles - 1)) * TE;   // tiles past the range: the type's last tile, dropped
            const bool live = t0 + k < n_tiles && first + row < r_end;
            const int64_t v = std::min(first + row, r_end - 1);
            const float* src = dy + v * ld_dy + 8 * o;
            // (every request unconditional - round 5, narrow.hip's comment on the in-order memory counter: a load under `dy_scale != nullptr ?` made the compiler drain the
            // queue in every phase.  Without scales the cotangent's first word is read in their place and replaced by 1; rows past the type's end take the factor 0.)
            r.sc = (dy_scale != nullptr ? dy_scale + v : src)[0];
            r.keep = live ? 1.f : 0.f;
#pragma unroll
            for (int x = 0; x < DOCT; ++x) {
                r.d[2 * x] = *reinterpret_cast<const v4f*>(src + 64 * x);
                r.d[2 * x + 1] = *reinterpret_cast<const v4f*>(src + 64 * x + 4);
            }
            const float* hp = h + v * ld_h + HC * half + 4 * o;
            const float* sp = sums + v * ld_s + HC * half + 4 * o;
#pragma unroll
            for (int x = 0; x < ZX; ++x) {
                r.hv[x] = *reinterpret_cast<const v4f*>(hp + 32 * x);
                // (the pair sums are read once here and not again: non-temporal, so that h and dy stay cached for the member-gradient kernel that follows - 2 % there)
                r.sa[x] = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(sp + 32 * x));
                r.sb[x] = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(sp + D + 32 * x));
                r.sab[x] = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(sp + 2 * D + 32 * x));
            }
        };
        auto product = [&](const Rows& r, int x, int b) { return b == 0 ? r.hv[x] * r.sa[x] : b == 1 ? r.hv[x] * r.sb[x] : b == 2 ? r.sab[x] : r.hv[x] * r.sab[x]; };
        // biased exponents of the row's largest cotangent magnitude (ea) and largest product magnitude (eb); -1 where the row is all zero on that side
        auto row_exponents = [&](const Rows& r, int& ea, int& eb) {
            float ma = 0.f, mb = 0.f;
#pragma unroll
            for (int x = 0; x < DOCT; ++x) ma = abs_max_of(r.d[2 * x], r.d[2 * x + 1], ma);
#pragma unroll
            for (int x = 0; x < ZX; ++x)
#pragma unroll
                for (int b = 0; b < NBLK; ++b) {
                    const v4f z = product(r, x, b);
                    mb = abs_max3(z[2], z[3], abs_max3(z[0], z[1], mb));
                }
            ma = row_lanes_max<8>(ma);
            mb = row_lanes_max<8>(mb);
            // (bit tests, not comparisons: a NaN or an infinity counts as the largest exponent and stays what it is through the scaling - it reaches the gradient)
            ea = (__float_as_uint(ma) & 0x7fffffffu) != 0u ? static_cast<int>((__float_as_uint(ma) >> 23) & 0xffu) : -1;
            eb = (__float_as_uint(mb) & 0x7fffffffu) != 0u ? static_cast<int>((__float_as_uint(mb) >> 23) & 0xffu) : -1;
        };
        // the wave's largest ea + eb (eight rows) -> tile_l[k & 3][wave]
        auto publish = [&](int k, Rows& r) {
#pragma unroll
            for (int x = 0; x < 2 * DOCT; ++x) r.d[x] *= (dy_scale != nullptr ? r.sc : 1.f) * r.keep;      // (here, a phase after the request - not where the rows are requested)
            row_exponents(r, r.ea, r.eb);
            const int ea = r.ea, eb = r.eb;
            int l = (ea >= 0 && eb >= 0) ? ea + eb : -1;
            l = std::max(l, __shfl_xor(l, 8));
            l = std::max(l, __shfl_xor(l, 16));
            l = std::max(l, __shfl_xor(l, 32));
            if (lane == 0) tile_l[k & 3][wave - 4] = l;
        };
        const int swz = tr_swizzle(row);
        int l_max = -1;
        // A row's two operands are scaled AGAINST each other by powers of two (a 2^sa, z 2^sz with sa + sz the same for every row of the workgroup: the products, and
        // so the gradient, carry one known factor) so that their largest magnitudes meet at 2^m, m = (la + lb - l_max + kTarget) / 2 <= 12: both fit fp16's range and keep
        // two fp16 terms = 22 bits; a row far below the largest loses low bits, but BOTH its operands are small - its products are small squared.
        auto split_tile = [&](int k, const Rows& r, int buf) {
            l_max = std::max(l_max, tile_max(k));
            const int ea = r.ea, eb = r.eb;
            int sa = -400, sz = -400;                                     // a row with an all-zero side contributes nothing: BOTH sides go in as zeros (an unscaled side could
            if (ea >= 0 && eb >= 0) {                                    // exceed fp16's range, and 0 x inf is a NaN in every gradient entry)
                const int m = (ea + eb - l_max + kTarget) >> 1;          // (unbiased exponent the row's maxima are brought to; ea + eb <= l_max)
                sa = m - (ea - 127);
                sz = (254 + kTarget - l_max) - sa;                       // sa + sz = 254 + kTarget - l_max for every row scaled under this l_max
            }
#pragma unroll
            for (int x = 0; x < DOCT; ++x) {
                v4u hi, lo;
#pragma unroll
                for (int pr = 0; pr < 4; ++pr) {
                    unsigned hh, ll;
                    split_pair_h2(__builtin_ldexpf(r.d[2 * x + (pr >> 1)][2 * (pr & 1)], sa), __builtin_ldexpf(r.d[2 * x + (pr >> 1)][2 * (pr & 1) + 1], sa), hh, ll);
                    hi[pr] = hh;
                    lo[pr] = ll;
                }
                const int off = row * DRB + 256 * ((o + 8 * x) >> 4) + ((((o + 8 * x) & 15) ^ swz) << 4);
                *reinterpret_cast<v4u*>(&dplanes[buf][0][0][0] + off) = hi;
                *reinterpret_cast<v4u*>(&dplanes[buf][0][0][0] + DPL + off) = lo;
            }
#pragma unroll
            for (int x = 0; x < ZX; ++x) {
                const int og = o + 8 * x;
#pragma unroll
                for (int b = 0; b < NBLK; ++b) {
                    const v4f z = product(r, x, b);
                    typedef unsigned v2u __attribute__((ext_vector_type(2)));
                    unsigned h0, l0, h1, l1;
                    split_pair_h2(__builtin_ldexpf(z[0], sz), __builtin_ldexpf(z[1], sz), h0, l0);
                    split_pair_h2(__builtin_ldexpf(z[2], sz), __builtin_ldexpf(z[3], sz), h1, l1);
                    const int byte = 2 * (b * HC + 4 * og);
                    const int off = row * ZRB + 256 * (byte >> 8) + ((((byte >> 4) & 15) ^ swz) << 4) + (byte & 8);
                    *reinterpret_cast<v2u*>(&zplanes[buf][0][0][0] + off) = v2u{h0, h1};
                    *reinterpret_cast<v2u*>(&zplanes[buf][0][0][0] + ZPL + off) = v2u{l0, l1};
                }
            }
        };
        if (n_my > 0) {
            Rows r0, r1, r2;                                             // rows of tile m in r<m % 3>
            load_rows(0, r0);
            load_rows(1, r1);
            load_rows(2, r2);
            publish(0, r0);
            publish(1, r1);
            __syncthreads();
            split_tile(0, r0, 0);
            __syncthreads();
            // phase k: images of tile k + 1 (`use`, whose exponents were published a phase ago); exponents of tile k + 2 (`arrive`, requested a phase ago: THREE sets of rows -
            // publishing what the phase itself requested would make every phase wait out a memory round trip); request of tile k + 3 (`fill`)
            auto phase = [&](int k, Rows& use, Rows& arrive, Rows& fill) {
                load_rows(k + 3, fill);
                if (k + 1 < n_my) split_tile(k + 1, use, (k + 1) & 1);
                publish(k + 2, arrive);
                __syncthreads();
            };
            int k = 0;
#pragma clang loop unroll(disable)
            for (; k + 2 < n_my; k += 3) {
                phase(k, r1, r2, r0);
                phase(k + 1, r2, r0, r1);
                phase(k + 2, r0, r1, r2);
            }
            if (k < n_my) phase(k, r1, r2, r0);
            if (k + 1 < n_my) phase(k + 1, r2, r0, r1);
        }
        return;
    }

    // ---------------- matrix waves: wave = X block, 8 x 4 accumulator tiles (all 128 cotangent columns x the block's 64 columns of the half)
    const int blk = wave;
    v4f acc[JT][CT];
#pragma unroll
    for (int jt = 0; jt < JT; ++jt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) acc[jt][ct] = v4f{0.f, 0.f, 0.f, 0.f};
    int l_max = -1;
    if (n_my > 0) {
        __syncthreads();
        __syncthreads();
        const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
        const int rlo = 8 * g + q, rhi = rlo + 4;
        auto a_addr = [&](int r, int jt) { return r * DRB + 256 * (jt >> 3) + (((2 * (jt & 7) + (pp >> 1)) ^ tr_swizzle(r)) << 4) + 8 * (pp & 1); };
        auto b_addr = [&](int r, int ct) {
            const int byte = 2 * (blk * HC + 16 * ct);
            return r * ZRB + 256 * (byte >> 8) + (((((byte >> 4) & 15) + (pp >> 1)) ^ tr_swizzle(r)) << 4) + 8 * (pp & 1);
        };
        auto fragment = [&](const unsigned char* lo, const unsigned char* hi) { return __builtin_bit_cast(v8h, read_tr_fragment(lo, hi)); };
        for (int k = 0; k < n_my; ++k) {
            // tile k was scaled under the running maximum INCLUDING its own rows: what the accumulators hold from the tiles before is brought to that scale first (exact:
            // a power of two; rare - the maximum grows a few times per workgroup)
            const int l_new = std::max(l_max, tile_max(k));
            if (l_new != l_max && l_max >= 0 && blk < NBLK) {
                const int shift = l_max - l_new;                         // (< 0)
#pragma unroll
                for (int jt = 0; jt < JT; ++jt)
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc[jt][ct][r] = __builtin_ldexpf(acc[jt][ct][r], shift);
            }
            l_max = l_new;
            const unsigned char* dp = &dplanes[k & 1][0][0][0];
            const unsigned char* zp = &zplanes[k & 1][0][0][0];
            if (blk < NBLK)
#pragma unroll
            for (int jh = 0; jh < JT / 4; ++jh) {
                v8h a[4][2];
#pragma unroll
                for (int jt = 0; jt < 4; ++jt)
#pragma unroll
                    for (int p = 0; p < 2; ++p) a[jt][p] = fragment(dp + p * DPL + a_addr(rlo, 4 * jh + jt), dp + p * DPL + a_addr(rhi, 4 * jh + jt));
                v8h b[2], bn[2];
#pragma unroll
                for (int p = 0; p < 2; ++p) b[p] = fragment(zp + p * ZPL + b_addr(rlo, 0), zp + p * ZPL + b_addr(rhi, 0));
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    if (ct + 1 < CT) {
#pragma unroll
                        for (int p = 0; p < 2; ++p) bn[p] = fragment(zp + p * ZPL + b_addr(rlo, ct + 1), zp + p * ZPL + b_addr(rhi, ct + 1));
                    }
#pragma unroll
                    for (int term = 0; term < 3; ++term)
#pragma unroll
                        for (int jt = 0; jt < 4; ++jt)
                            acc[4 * jh + jt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[jt][kTermA2[term]], b[kTermB2[term]], acc[4 * jh + jt][ct], 0, 0, 0);
#pragma unroll
                    for (int p = 0; p < 2; ++p) b[p] = bn[p];
                }
            }
            __syncthreads();
        }
    }
    if (blk >= NBLK) return;
    // the accumulators hold 2^(254 + kTarget - l_max) times the gradient (l_max < 0: no row of the range had a non-zero cotangent and product - they hold zeros)
    const int unscale = l_max >= 0 ? l_max - 254 - kTarget : 0;
    float* slab = slabs + static_cast<int64_t>(range) * D * NBLK * D;
    const int c = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int jt = 0; jt < JT; ++jt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                slab[static_cast<int64_t>(16 * jt + 4 * kq + r) * NBLK * D + blk * D + HC * half + 16 * ct + c] = __builtin_ldexpf(acc[jt][ct][r], unscale);
}

// dw[j][(3 + b) d + c] = sum over the node types of the sum over the type's slabs at the X block that stands for w block b there
// (b: 0 = uq, 1 = qi, 2 = iu, 3 = uqi; fixed order: users, queries, items, slabs in range order - bitwise reproducible)
__global__ __launch_bounds__(kBlockThreads) void node_weight_reduce_kernel(const float* __restrict__ slabs, NodeRanges plan, int d, int nblk, float* __restrict__ dw, int64_t ld_dw) {
    constexpr signed char kPos[3][4] = {{0, 2, 1, 3}, {0, 1, 2, 3}, {2, 1, 0, 3}};      // [type][w block] -> X block
    const int width = nblk * d;
    const int64_t total = static_cast<int64_t>(d) * width;
    for (int64_t base = static_cast<int64_t>(blockIdx.x) * kWave; base < total; base += static_cast<int64_t>(gridDim.x) * kWave) {
        const int64_t idx = base + (threadIdx.x & 63);
        const bool live = idx < total;
        const int j = static_cast<int>((live ? idx : 0) / width), col = static_cast<int>((live ? idx : 0) - static_cast<int64_t>(j) * width);
        const int b = col / d, c = col - b * d;
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int n = plan.range_prefix[t + 1] - plan.range_prefix[t];
            const int64_t at = static_cast<int64_t>(j) * width + kPos[t][b] * d + c;
            sum += slab_sum(slabs + static_cast<int64_t>(plan.range_prefix[t]) * total, n, total, at, live);
        }
        if ((threadIdx.x >> 6) == 0 && live) dw[static_cast<int64_t>(j) * ld_dw + 3 * static_cast<int64_t>(d) + col] = sum;
    }
}

// ------------------------------------------------------------------------------------------------
// Row GEMM (node-level linear maps, d = 128): out[v] = in[v] W_t^T (+ bias_t), rows grouped by node type.  A stream over [N, d] - 1 KB
// of traffic per row against 32 K multiply-adds - that the fp32 matrix pipe cannot feed at HBM speed and the bf16 pipe can.
// Eight waves, wave w owns output columns 16 w .. with the whole contraction index (its weight planes: 48 registers per node type,
// reloaded at the at most two type changes of a workgroup's tile sequence); tiles of 32 rows come in through registers two tiles
// ahead (each thread 8 values), are split inside the previous tile's matrix phase and laid down as three bf16 images like the
// member-gradient kernel's; the MFMA is issued as W x in^T, so a lane ends up with 4 consecutive columns of one row and stores
// them straight from the accumulator.  One barrier per tile, two workgroups per CU.
// pk[type][strip][kb][plane][lane][8]: element i = plane of Wt[k = 32 kb + 8 (lane >> 4) + i][c = 16 strip + (lane & 15)],
//                                       Wt[k][c] = W_t[c][k] (transpose == 0, out = in W^T) or W_t[k][c] (transpose == 1, out = in W)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlockThreads) void pack_planes_dense_kernel(const float* __restrict__ w, int64_t ld_w, int64_t type_stride, int n_types, int d,
                                                                          int transpose, v4u* __restrict__ pk) {
    const int kbs = d / 32, strips = d / 16;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_types * strips * kbs * kWave) return;
    const int lane = idx & 63, kb = (idx >> 6) % kbs, strip = ((idx >> 6) / kbs) % strips, type = (idx >> 6) / (kbs * strips);
    const int c = 16 * strip + (lane & 15), k0 = 32 * kb + 8 * (lane >> 4);
    const float* wt = w + type * type_stride;
    v4f x0, x1;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        x0[i] = transpose == 0 ? wt[static_cast<int64_t>(c) * ld_w + k0 + i] : wt[static_cast<int64_t>(k0 + i) * ld_w + c];
        x1[i] = transpose == 0 ? wt[static_cast<int64_t>(c) * ld_w + k0 + 4 + i] : wt[static_cast<int64_t>(k0 + 4 + i) * ld_w + c];
    }
    const Planes pl = split8(x0, x1);
#pragma unroll
    for (int p = 0; p < 3; ++p) pk[(static_cast<int64_t>(idx >> 6) * 3 + p) * kWave + lane] = pl.p[p];
}

// pk[type][strip][kb][plane < 2][lane][8 x fp16]: element i = plane of wsc[type][c] w(c, 32 kb + 8 (lane >> 4) + i), c = 16 strip + (lane & 15)
__global__ __launch_bounds__(kBlockThreads) void pack_planes_dense_h2_kernel(const float* __restrict__ w, int64_t ld_w, int64_t type_stride, int n_types, int d,
                                                                             int transpose, const float* __restrict__ wsc, v4u* __restrict__ pk) {
    const int kbs = d / 32, strips = d / 16;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_types * strips * kbs * kWave) return;
    const int lane = idx & 63, kb = (idx >> 6) % kbs, strip = ((idx >> 6) / kbs) % strips, type = (idx >> 6) / (kbs * strips);
    const int c = 16 * strip + (lane & 15), k0 = 32 * kb + 8 * (lane >> 4);
    const float* wt = w + type * type_stride;
    const float sc = wsc[type * d + c];
    v4u hi, lo;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float xa = transpose == 0 ? wt[static_cast<int64_t>(c) * ld_w + k0 + 2 * i] : wt[static_cast<int64_t>(k0 + 2 * i) * ld_w + c];
        const float xb = transpose == 0 ? wt[static_cast<int64_t>(c) * ld_w + k0 + 2 * i + 1] : wt[static_cast<int64_t>(k0 + 2 * i + 1) * ld_w + c];
        unsigned hh, ll;
        split_pair_h2(xa * sc, xb * sc, hh, ll);
        hi[i] = hh;
        lo[i] = ll;
    }
    pk[(static_cast<int64_t>(idx >> 6) * 2 + 0) * kWave + lane] = hi;
    pk[(static_cast<int64_t>(idx >> 6) * 2 + 1) * kWave + lane] = lo;
}

// D = 128: one workgroup covers all columns, two workgroups per CU.  D = 256: a workgroup covers a column HALF (64 weight registers per
// wave), the two halves of a tile sequence are two workgroups on one XCD (the second read of a row hits that L2), one workgroup per CU.
// Arithmetic: two fp16 terms per operand (node-level contraction above): a row is scaled by ONE power of two (its 16 staging threads agree on its largest magnitude with
// four shuffles), the weights by one per output column; the accumulators leave through the two inverse scales.  Half the MFMAs and two thirds of the weight / image
// registers of the three-bf16 form - the registers pay for a THIRD set of row pieces: a tile's rows are requested three tiles ahead and taken delivery of a whole phase
// after the request (two tiles of 16 KB in flight per workgroup instead of one: the kernel is a stream whose rate is bytes in flight over the memory round trip).
// ACC (out += ...) is a template parameter (round 5): as a run-time flag its two accumulator-shaped registers and the branch around their loads were in every instance - at
// D = 256 the four registers the kernel spilled (256 VGPRs + 20 B of scratch per lane; now no scratch).
template <int D, bool ACC>
__global__ __launch_bounds__(512, 2) void row_gemm_split_kernel(TypedRows in, int64_t ld_in, const v4u* __restrict__ pk,
                                                                             int64_t pk_type_stride, const float* __restrict__ winv, const float* __restrict__ bias, int bias_mask,
                                                                             int64_t bias_type_stride, RowTiles plan, TypedRowsOut out, int64_t ld_out) {
    constexpr int TE = 32, KB = D / 32, OCT = D / 128, HALVES = D / 128, RB = 2 * D, STEPS = 2 * KB;
    constexpr bool EARLY_IMAGE = D == 256 && ACC;                        // the accumulating instance at D = 256 writes a finished 16-byte piece of the next tile's images at once: the registers that would hold it to the end of the phase are the ones it lacked
    constexpr int NBUF = D == 128 ? 6 : 3;                               // sets of row pieces: a tile's rows are requested NBUF tiles ahead, NBUF - 2 tiles (16 KB each) in flight
    __shared__ __attribute__((aligned(16))) unsigned char planes[2][2][TE][RB];
    __shared__ float sinv[2][TE];                                        // inverse row scales of the tile whose images are in planes[.]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int total_tiles = plan.tile_prefix[3];
    const int bid = blockIdx.x;
    const int half = HALVES == 1 ? 0 : (bid >> 3) & 1;
    const int seq = HALVES == 1 ? bid : (bid & 7) + 8 * (bid >> 4);
    const int n_seq = gridDim.x / HALVES;
    const int n_my = seq < total_tiles ? (total_tiles - seq + n_seq - 1) / n_seq : 0;
    if (n_my == 0) return;
    auto tile_type = [&](int tile_id) { return tile_id >= plan.tile_prefix[2] ? 2 : (tile_id >= plan.tile_prefix[1] ? 1 : 0); };
    auto tile_rows = [&](int k, int64_t& r_base, int64_t& r_end) {      // (tiles past this workgroup's last: the last one again, read and dropped)
        const int tile_id = seq + std::min(k, n_my - 1) * n_seq;
        const int type = tile_type(tile_id);
        r_base = plan.begin[type] + static_cast<int64_t>(tile_id - plan.tile_prefix[type]) * TE;
        r_end = plan.begin[type + 1];
        return type;
    };
    const int row = tid >> 4, o = tid & 15;                              // staging role: row, octets o (and o + 16) of it
    const int chunk = (o ^ (row & 15)) << 4;
    auto load_rows = [&](int k, v4f (&dr)[2 * OCT]) {                    // unconditional: a branch around requests makes the compiler wait for all of them
        int64_t r_base, r_end;
        const int type = tile_rows(k, r_base, r_end);
        const int64_t v = std::min(r_base + row, r_end - 1);             // rows past the type's end re-read its last row (never stored)
        const float* src = typed_base(in, type) + v * ld_in + 8 * o;
#pragma unroll
        for (int x = 0; x < OCT; ++x) {
            dr[2 * x] = *reinterpret_cast<const v4f*>(src + 128 * x);
            dr[2 * x + 1] = *reinterpret_cast<const v4f*>(src + 128 * x + 4);
        }
    };
    // the row's power-of-two scale from this thread's piece and its 15 neighbours' (the row's staging threads are 16 consecutive lanes)
    auto row_scale = [&](const v4f (&dr)[2 * OCT], float& inv) {
        float m = 0.f;
#pragma unroll
        for (int j = 0; j < 2 * OCT; ++j) m = abs_max3(dr[j][2], dr[j][3], abs_max3(dr[j][0], dr[j][1], m));
        m = row_lanes_max<16>(m);
        return scale_up_for(m, inv);
    };
    v4f dbuf[NBUF][2 * OCT];                                             // row pieces of tile m in dbuf[m % NBUF] (indices are compile-time: the phases are unrolled NBUF at a time)
#pragma unroll
    for (int j = 0; j < NBUF; ++j) load_rows(j, dbuf[j]);
    {
        v4f (&d0)[2 * OCT] = dbuf[0];
        float inv;
        const float sc = row_scale(d0, inv);
#pragma unroll
        for (int x = 0; x < OCT; ++x) {
            v4u hi, lo;
#pragma unroll
            for (int pr = 0; pr < 4; ++pr) {
                unsigned hh, ll;
                split_pair_h2(d0[2 * x + (pr >> 1)][2 * (pr & 1)] * sc, d0[2 * x + (pr >> 1)][2 * (pr & 1) + 1] * sc, hh, ll);
                hi[pr] = hh;
                lo[pr] = ll;
            }
            *reinterpret_cast<v4u*>(&planes[0][0][row][chunk + 256 * x]) = hi;
            *reinterpret_cast<v4u*>(&planes[0][1][row][chunk + 256 * x]) = lo;
        }
        if (o == 0) sinv[0][row] = inv;
    }
    __syncthreads();

    v8h wreg[KB][2];
    v4f wiv = v4f{1.f, 1.f, 1.f, 1.f};                                    // inverse scales of this lane's four output columns
    v4f bv = v4f{0.f, 0.f, 0.f, 0.f};                                     // ... and their bias (per node type: loaded with the weights - a load issued and consumed inside a phase
    int cur_type = -1;                                                    //     would make the phase wait for every row request in front of it: the memory counter is in order)
    const int arow = lane & 15, kq = lane >> 4;
    const int c4 = 128 * half + 16 * wave + 4 * kq;
    // phase k: contraction of tile k; images of tile k + 1 (`use`); delivery of tile k + 2 (`arrive`, requested NBUF - 2 phases ago); request of tile k + NBUF (`fill`)
    auto phase = [&](int k, v4f (&use)[2 * OCT], v4f (&arrive)[2 * OCT], v4f (&fill)[2 * OCT]) {
        int64_t r_base, r_end;
        const int type = tile_rows(k, r_base, r_end);
        if (type != cur_type) {
            const v4u* wf = pk + type * pk_type_stride + static_cast<int64_t>(8 * half + wave) * (KB * 2) * kWave + lane;
#pragma unroll
            for (int kb = 0; kb < KB; ++kb)
#pragma unroll
                for (int p = 0; p < 2; ++p) wreg[kb][p] = __builtin_bit_cast(v8h, wf[(kb * 2 + p) * kWave]);
            wiv = *reinterpret_cast<const v4f*>(winv + type * (pk_type_stride == 0 ? 0 : D) + c4);
            bv = v4f{0.f, 0.f, 0.f, 0.f};
            if (bias != nullptr && ((bias_mask >> type) & 1)) bv = *reinterpret_cast<const v4f*>(bias + type * bias_type_stride + c4);
            cur_type = type;
        }
        // accumulate: `out` already holds another contribution to the same rows (out += ...): its rows of this tile are requested FIRST - older than the row requests
        // below, so the wait for them at the end of the phase leaves those in flight
        v4f gold[2] = {v4f{0.f, 0.f, 0.f, 0.f}, v4f{0.f, 0.f, 0.f, 0.f}};
        if (ACC) {
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                const int64_t v = std::min(r_base + 16 * rt + arow, r_end - 1);
                gold[rt] = *reinterpret_cast<const v4f*>(typed_base(out, type) + v * ld_out + c4);
            }
        }
        load_rows(k + NBUF, fill);
        float inv_next;
        const float sc_next = row_scale(use, inv_next);
        v4f acc[2] = {v4f{0.f, 0.f, 0.f, 0.f}, v4f{0.f, 0.f, 0.f, 0.f}};
        const unsigned char* pbase = &planes[k & 1][0][0][0];
        v4u sp[OCT][2];
#pragma unroll
        for (int step = 0; step < STEPS; ++step) {
            const int kb = step >> 1, rt = step & 1;
            v8h a[2];
            const unsigned char* src = pbase + (16 * rt + arow) * RB + (((4 * kb + kq) ^ arow) << 4);
#pragma unroll
            for (int p = 0; p < 2; ++p) a[p] = *reinterpret_cast<const v8h*>(src + p * (TE * RB));
            if (step < 4 * OCT) {   // two values of the next tile's row piece -> one dword of each plane
                const int x = step >> 2, pr = step & 3;
                unsigned hh, ll;
                split_pair_h2(use[2 * x + (pr >> 1)][2 * (pr & 1)] * sc_next, use[2 * x + (pr >> 1)][2 * (pr & 1) + 1] * sc_next, hh, ll);
                sp[x][0][pr] = hh;
                sp[x][1][pr] = ll;
                if (EARLY_IMAGE && pr == 3) {                            // (the other image buffer was last read a phase ago)
#pragma unroll
                    for (int p = 0; p < 2; ++p) *reinterpret_cast<v4u*>(&planes[(k + 1) & 1][p][row][chunk + 256 * x]) = sp[x][p];
                }
            }
#pragma unroll
            for (int term = 0; term < 3; ++term)
                acc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wreg[kb][kTermB2[term]], a[kTermA2[term]], acc[rt], 0, 0, 0);
        }
        const float iv0 = sinv[k & 1][arow], iv1 = sinv[k & 1][16 + arow];
        if (!EARLY_IMAGE) {
#pragma unroll
            for (int x = 0; x < OCT; ++x)
#pragma unroll
                for (int p = 0; p < 2; ++p) *reinterpret_cast<v4u*>(&planes[(k + 1) & 1][p][row][chunk + 256 * x]) = sp[x][p];     // (past the last tile: nobody reads it)
        }
        if (o == 0) sinv[(k + 1) & 1][row] = inv_next;
        // delivery of the rows requested a phase ago, before the stores (the memory counter is in order: a wait behind a store sits out the store's round trip)
        if (OCT == 1) asm volatile("" : "+v"(arrive[0]), "+v"(arrive[1]));
        else asm volatile("" : "+v"(arrive[0]), "+v"(arrive[1]), "+v"(arrive[2 * OCT - 2]), "+v"(arrive[2 * OCT - 1]));
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            const int64_t v = r_base + 16 * rt + arow;
            if (v < r_end) {
                v4f y = acc[rt] * (wiv * (rt == 0 ? iv0 : iv1)) + bv;
                if (ACC) y += gold[rt];
                *reinterpret_cast<v4f*>(typed_base(out, type) + v * ld_out + c4) = y;
            }
        }
        __syncthreads();
    };
    int k = 0;
    for (; k + NBUF <= n_my; k += NBUF) {
#pragma unroll
        for (int j = 0; j < NBUF; ++j) phase(k + j, dbuf[(j + 1) % NBUF], dbuf[(j + 2) % NBUF], dbuf[j]);
    }
#pragma unroll
    for (int j = 0; j < NBUF - 1; ++j)
        if (k + j < n_my) phase(k + j, dbuf[(j + 1) % NBUF], dbuf[(j + 2) % NBUF], dbuf[j]);
}

// ------------------------------------------------------------------------------------------------
// Weight gradient of the node-level linear maps: dW_t[i][j] = sum_{v of type t} dout[v][i] x[v][j], dbias_t[i] = sum_v dout[v][i].
// Same shape of problem as the interactive weight gradient (both operands are streams, the contraction runs over the rows), same
// machinery: row-major 16-bit images with the transposed-read swizzle, ds_read_b64_tr_b16 fragments, accumulators resident for the
// whole sweep, one slab per tile sequence (dense.hip's layout and reduction).  D = 128: a workgroup owns the whole 128 x 128 gradient
// (wave = 2 x 4 accumulator tiles); D = 256: a workgroup owns a column half (wave = 4 x 4 tiles), both halves of a tile sequence on
// one XCD.  The column sums of dout ride along in the staging threads' registers.  grid = (sequences x halves, 1, node types).
// Arithmetic: two fp16 terms per operand with a row's two operands scaled against each other under one running scale per workgroup - node_interact_weight_split_kernel's
// scheme (its comment has the argument); every wave is both roles here, so every thread keeps the two running maxima itself: l_split (all tiles up to the one being
// split) and l_acc (the scale the accumulators are at: a tile behind).
// ------------------------------------------------------------------------------------------------
// DX (d = 128): the input gradient dx = dout W_t of the same rows is formed here too - a ROW contraction, whose rows need their full relative accuracy each: it reads a
// second image of dout, every row scaled to 2^13 by itself (the balanced image of a small row is deliberately coarse), against weight planes scaled per output column
// (pack_planes_dense_h2_kernel); wave w takes output columns 16 w .. with the type's planes in 32 registers - and the separate row-GEMM pass over dout goes away.
// Round 5: every vector-memory request of the loop is unconditional (narrow.hip has the argument: the memory counter is in order and the compiler waits by count; a request
// under a branch makes the next wait vmcnt(0)).  ACC (dx += instead of dx =) is a template parameter - its reads of dx were loads under a run-time flag -, and a lane whose dx
// row lies past the type's end stores to a dump piece inside the workgroup's own slab (written for good only at the kernel's end) through a global-address-space pointer.
// (-DIHG_ABL_D_TRACE: clock stamps of wave 0 of one workgroup at marks inside its phases, as in split_arith.hip; tools/phase_trace.py --kernel linear)
#ifdef IHG_ABL_D_TRACE
__device__ unsigned long long g_dense_trace[64][6];
#define IHG_DTRACE(k, mark) \
    if (blockIdx.x == 40 && blockIdx.z == 0 && tid == 0 && (k) < 64) g_dense_trace[k][mark] = clock64();
#else
#define IHG_DTRACE(k, mark)
#endif
template <int D, bool DX, bool ACC>
__global__ __launch_bounds__(kSplitThreads) void dense_weight_grad_split_kernel(const float* __restrict__ dout, int64_t ld_dout, TypedRows x,
                                                                                int64_t ld_x, RowTiles plan, int single_weight, float* __restrict__ slabs,
                                                                                float* __restrict__ bias_slabs, const v4u* __restrict__ pk, int64_t pk_type_stride,
                                                                                const float* __restrict__ winv, TypedRowsOut dx, int64_t ld_dx) {
    static_assert(!DX || D == 128, "the fused input gradient holds a whole weight matrix per workgroup");
    static_assert(DX || !ACC, "accumulation is the fused input gradient's");
    constexpr int TE = 32, HALVES = D / 128, DOCT = D / 128, IT = D / 64, DRB = 2 * D;     // dout image rows: 2 D bytes, x image rows: 256 bytes (128 columns)
    constexpr int DPL = TE * DRB, XPL = TE * 256, kTarget = 24;
    __shared__ __attribute__((aligned(16))) unsigned char dplanes[2][2][TE][DRB];          // dout, balanced against x (dW)
    __shared__ __attribute__((aligned(16))) unsigned char xplanes[2][2][TE][256];
    __shared__ __attribute__((aligned(16))) unsigned char nplanes[DX ? 2 : 1][2][DX ? TE : 1][DX ? DRB : 16];      // dout, every row scaled by itself (dx)
    __shared__ float rown[2][TE];                                        // ... and the inverse of that scale
    __shared__ __attribute__((aligned(16))) int tile_l[4][8];            // per tile (mod 4) and wave: the largest ea + eb (biased exponents) of the wave's four rows, -1: none
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bid = blockIdx.x, type = blockIdx.z;
    const int half = HALVES == 1 ? 0 : (bid >> 3) & 1;
    const int seq = HALVES == 1 ? bid : (bid & 7) + 8 * (bid >> 4);
    const int n_seq = gridDim.x / HALVES;
    const int64_t r_begin = single_weight ? plan.begin[0] : plan.begin[type];
    const int64_t r_end = single_weight ? plan.begin[3] : plan.begin[type + 1];
    const int64_t n_tiles = (r_end - r_begin + TE - 1) / TE;
    const int n_my = seq < n_tiles ? static_cast<int>((n_tiles - seq + n_seq - 1) / n_seq) : 0;
    const int iq = wave & 3, jh = wave >> 2;
    // x and dx may be typed rows (TypedRows): with one weight for every node (single_weight) a tile sequence crosses the node types
    auto row_type = [&](int64_t v) { return single_weight ? (v >= plan.begin[2] ? 2 : (v >= plan.begin[1] ? 1 : 0)) : type; };
    auto tile_max = [&](int k) {
        const int* t = tile_l[k & 3];
        return std::max(std::max(std::max(t[0], t[1]), std::max(t[2], t[3])), std::max(std::max(t[4], t[5]), std::max(t[6], t[7])));
    };

    v4f acc[IT][4];
#pragma unroll
    for (int it = 0; it < IT; ++it)
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) acc[it][jt] = v4f{0.f, 0.f, 0.f, 0.f};
    v4f bsum[2 * DOCT];
#pragma unroll
    for (int i = 0; i < 2 * DOCT; ++i) bsum[i] = v4f{0.f, 0.f, 0.f, 0.f};
    int l_acc = -1, l_split = -1;                                        // running maxima of ea + eb: what the accumulators are scaled under / what the tile being split is

    const int row = tid >> 4, o = tid & 15;                              // staging role: row; dout columns 8 o .. (and 128 + 8 o ..), x columns 128 half + 8 o ..
    struct Rows {
        v4f d[2 * DOCT], x[2];
        int ea, eb;                                                      // biased exponents of the row's largest |dout| and |x| (-1: all zero), set by publish()
    };
    auto load_rows = [&](int k, Rows& r) {
        const int64_t v = r_begin + (static_cast<int64_t>(seq) + static_cast<int64_t>(k) * n_seq) * TE + row;
        const bool live = v < r_end;
        const int64_t vc = live ? v : r_end - 1;
        const float* ds = dout + vc * ld_dout + 8 * o;
        const float* xs = typed_base(x, row_type(vc)) + vc * ld_x + 128 * half + 8 * o;
        if (abl::d_no_loads) {
#pragma unroll
            for (int i = 0; i < 2 * DOCT; ++i) r.d[i] = v4f{1.f, 2.f, 3.f, 4.f} * static_cast<float>(vc + i);
            r.x[0] = r.x[1] = v4f{1.f, 2.f, 3.f, 4.f} * static_cast<float>(vc);
            return;
        }
#pragma unroll
        for (int i = 0; i < DOCT; ++i) {
            r.d[2 * i] = *reinterpret_cast<const v4f*>(ds + 128 * i);
            r.d[2 * i + 1] = *reinterpret_cast<const v4f*>(ds + 128 * i + 4);
        }
        r.x[0] = *reinterpret_cast<const v4f*>(xs);
        r.x[1] = *reinterpret_cast<const v4f*>(xs + 4);
        if (!live) {                                                     // rows past the type's end contribute nothing
#pragma unroll
            for (int i = 0; i < 2 * DOCT; ++i) r.d[i] = v4f{0.f, 0.f, 0.f, 0.f};
        }
    };
    // the row's exponents (its sixteen staging threads are sixteen consecutive lanes) and the wave's largest ea + eb (four rows) -> tile_l[k & 3][wave]
    auto publish = [&](int k, Rows& r) {
        float ma = 0.f;
#pragma unroll
        for (int i = 0; i < DOCT; ++i) ma = abs_max_of(r.d[2 * i], r.d[2 * i + 1], ma);
        float mb = abs_max_of(r.x[0], r.x[1], 0.f);
        ma = row_lanes_max<16>(ma);
        mb = row_lanes_max<16>(mb);
        // (bit tests, not comparisons: a NaN or an infinity counts as the largest exponent and stays what it is through the scaling)
        r.ea = (__float_as_uint(ma) & 0x7fffffffu) != 0u ? static_cast<int>((__float_as_uint(ma) >> 23) & 0xffu) : -1;
        r.eb = (__float_as_uint(mb) & 0x7fffffffu) != 0u ? static_cast<int>((__float_as_uint(mb) >> 23) & 0xffu) : -1;
        int l = (r.ea >= 0 && r.eb >= 0) ? r.ea + r.eb : -1;
        l = std::max(l, __shfl_xor(l, 16));
        l = std::max(l, __shfl_xor(l, 32));
        if (lane == 0) tile_l[k & 3][wave] = l;
    };
    const int swz = tr_swizzle(row);
    const int st_off = row * 256 + ((o ^ swz) << 4);                     // 16 bytes of a 256-byte segment of this thread's row
    // a staged tile -> images `buf`, one pair of values per slice: slices 0 .. 4 DOCT - 1 dout balanced (the column sums ride along), then 4 of x, then (DX) 4 DOCT of
    // dout scaled by its own row.  sa / sx / sn: this row's three exponents of two (set per tile by row_scales)
    v4u sph, spl;
    int sa = 0, sx = 0, sn = 0;
    auto row_scales = [&](const Rows& r) {
        sa = sx = -400;                                                  // a row with an all-zero side goes in as zeros on BOTH sides (an unscaled side could exceed fp16's range)
        if (r.ea >= 0 && r.eb >= 0) {
            const int m = (r.ea + r.eb - l_split + kTarget) >> 1;        // the row's two maxima are brought to 2^m, m <= 12
            sa = m - (r.ea - 127);
            sx = (254 + kTarget - l_split) - sa;                         // sa + sx is the same for every row split under this l_split
        }
        sn = r.ea >= 0 ? 13 - (r.ea - 127) : 0;
    };
    constexpr int SLICES = 4 * DOCT + 4 + (DX ? 4 * DOCT : 0), STEPS = IT * 4;
    auto split_slice = [&](int slice, const Rows& r, int buf, bool counted) {
        const int kind = slice < 4 * DOCT ? 0 : (slice < 4 * DOCT + 4 ? 1 : 2);
        const int s = kind == 0 ? slice : (kind == 1 ? slice - 4 * DOCT : slice - 4 * DOCT - 4), oct = s >> 2, pr = s & 3;
        const v4f v = kind == 1 ? r.x[pr >> 1] : r.d[2 * oct + (pr >> 1)];
        const int e = kind == 0 ? sa : (kind == 1 ? sx : sn);
        unsigned hh, ll;
        split_pair_h2(__builtin_ldexpf(v[2 * (pr & 1)], e), __builtin_ldexpf(v[2 * (pr & 1) + 1], e), hh, ll);
        sph[pr] = hh;
        spl[pr] = ll;
        if (kind == 0 && (pr & 1) == 1 && counted) bsum[2 * oct + (pr >> 1)] += v;
        if (pr == 3) {
            if (kind == 1) {
                *reinterpret_cast<v4u*>(&xplanes[buf][0][0][0] + st_off) = sph;
                *reinterpret_cast<v4u*>(&xplanes[buf][0][0][0] + XPL + st_off) = spl;
            } else {
                unsigned char* base = kind == 0 ? &dplanes[buf][0][0][0] : &nplanes[DX ? buf : 0][0][0][0];
                *reinterpret_cast<v4u*>(base + row * (DRB - 256) + 256 * oct + st_off) = sph;
                *reinterpret_cast<v4u*>(base + DPL + row * (DRB - 256) + 256 * oct + st_off) = spl;
            }
        }
    };
    auto note_row_scale = [&](int buf) {                                 // (DX) the inverse of the row's own scale, for the epilogue of its dx row
        if (DX && o == 0) rown[buf][row] = __builtin_ldexpf(1.f, -sn);
    };

    if (n_my > 0) {
        // rows of tile m in r<m & 1>: requested two tiles ahead, taken delivery of at the END of the requesting phase - where their exponents are published, a phase before the
        // split.  This kernel is not sensitive to that wait: a third set of rows with delivery a phase later measured the same in round 4
        // (15_ab_node_level_backward_three_row_sets.txt) and again in round 5 with every request of the loop unconditional and every wait an exact count (169 against 170 us at
        // C3, 256 registers) - it is not the rows' latency this kernel waits for.
        Rows r0, r1;
        load_rows(0, r0);
        load_rows(std::min(1, n_my - 1), r1);
        publish(0, r0);
        publish(1, r1);
        __syncthreads();
        l_split = tile_max(0);
        row_scales(r0);
        note_row_scale(0);
#pragma unroll
        for (int s2 = 0; s2 < SLICES; ++s2) split_slice(s2, r0, 0, true);
        __syncthreads();

        const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
        const int rlo = 8 * g + q, rhi = rlo + 4;
        v8h wdx[DX ? 4 : 1][2];                                          // DX: planes of wsc[c] W_t[32 kb + 8 (lane >> 4) + i][c], c = 16 wave + (lane & 15)
        v4f wiv = v4f{1.f, 1.f, 1.f, 1.f};                                // ... and the inverse scales of this lane's four output columns
        if (DX) {
            const v4u* wf = pk + (single_weight ? 0 : type) * pk_type_stride + static_cast<int64_t>(wave) * 8 * kWave + lane;
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int p = 0; p < 2; ++p) wdx[kb][p] = __builtin_bit_cast(v8h, wf[(kb * 2 + p) * kWave]);
            wiv = *reinterpret_cast<const v4f*>(winv + (single_weight ? 0 : type) * D + 16 * wave + 4 * (lane >> 4));
        }
        // dout columns (output rows i): tile IT iq + it -> byte 32 (IT iq + it) + 8 pp of a DRB-byte row, 256-byte segments swizzled separately
        auto a_addr = [&](int r, int it) {
            const int tile = IT * iq + it, seg = tile >> 3, ch = 2 * (tile & 7) + (pp >> 1);
            return r * DRB + 256 * seg + ((ch ^ tr_swizzle(r)) << 4) + 8 * (pp & 1);
        };
        auto b_addr = [&](int r, int jt) { return r * 256 + (((8 * jh + 2 * jt + (pp >> 1)) ^ tr_swizzle(r)) << 4) + 8 * (pp & 1); };
        auto fragment = [&](const unsigned char* lo, const unsigned char* hi) { return __builtin_bit_cast(v8h, read_tr_fragment(lo, hi)); };

        // phase k: contraction of tile k (images k & 1); images of tile k + 1 from `use`; request of tile k + 2 (`fill`), delivered and its exponents published at the end
        auto phase = [&](int k, Rows& use, Rows& fill) {
            const int BUF = k & 1;
            IHG_DTRACE(k, 0)
            // the accumulators to the scale tile k was split under (it includes tile k's own rows); then the running maximum moves on to tile k + 1 for the split
            if (l_split != l_acc) {
                if (l_acc >= 0) {
                    const int shift = l_acc - l_split;
#pragma unroll
                    for (int it = 0; it < IT; ++it)
#pragma unroll
                        for (int jt = 0; jt < 4; ++jt)
#pragma unroll
                            for (int r = 0; r < 4; ++r) acc[it][jt][r] = __builtin_ldexpf(acc[it][jt][r], shift);
                }
                l_acc = l_split;
            }
            if (k + 1 < n_my) l_split = std::max(l_split, tile_max(k + 1));
            row_scales(use);
            if (k + 1 < n_my) note_row_scale(BUF ^ 1);
            // dx_accumulate: dx already holds another contribution to the same gradient (the member gradients of the interactive step); its rows
            // of this tile seed the accumulators of the dx product at the end of the phase.  Requested BEFORE the row requests: the memory
            // counter is in order, so the wait for these leaves the younger row requests in flight
            v4f gold[2] = {v4f{0.f, 0.f, 0.f, 0.f}, v4f{0.f, 0.f, 0.f, 0.f}};
            if (DX && ACC) {
                const int64_t r_base = r_begin + (static_cast<int64_t>(seq) + static_cast<int64_t>(k) * n_seq) * TE;
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) {
                    const int64_t v = std::min(r_base + 16 * rt + (lane & 15), r_end - 1);          // (rows past the end: the last row, read and dropped)
                    gold[rt] = *reinterpret_cast<const v4f*>(typed_base(dx, row_type(v)) + v * ld_dx + 16 * wave + 4 * (lane >> 4));
                }
            }
            load_rows(std::min(k + 2, n_my - 1), fill);                  // (unconditional: a branch around requests makes the compiler wait for all of them; round 5, clock
            IHG_DTRACE(k, 1)                                             //  probe: the requests stand 785 cycles into the phase, behind the bookkeeping above - moved to its top: 150 -> 168 us)
            const unsigned char* dp = &dplanes[0][0][0][0] + BUF * (2 * DPL);
            const unsigned char* xp = &xplanes[0][0][0][0] + BUF * (2 * XPL);
            v8h a[IT][2];
#pragma unroll
            for (int it = 0; it < IT; ++it)
#pragma unroll
                for (int p = 0; p < 2; ++p) a[it][p] = fragment(dp + p * DPL + a_addr(rlo, it), dp + p * DPL + a_addr(rhi, it));
            v8h b[2], bn[2];
#pragma unroll
            for (int p = 0; p < 2; ++p) b[p] = fragment(xp + p * XPL + b_addr(rlo, 0), xp + p * XPL + b_addr(rhi, 0));
#pragma unroll
            for (int jt = 0; jt < 4; ++jt) {
                if (jt + 1 < 4) {
#pragma unroll
                    for (int p = 0; p < 2; ++p) bn[p] = fragment(xp + p * XPL + b_addr(rlo, jt + 1), xp + p * XPL + b_addr(rhi, jt + 1));
                }
#pragma unroll
                for (int it = 0; it < IT; ++it) {
                    const int step = jt * IT + it;                       // the next tile's split, spread over this tile's MFMA groups
#pragma unroll
                    for (int s2 = step * SLICES / STEPS; s2 < (step + 1) * SLICES / STEPS; ++s2)
                        if (!abl::d_no_split) split_slice(s2, use, BUF ^ 1, k + 1 < n_my);   // (past the last tile: stale rows, nobody reads those images)
                }
#pragma unroll
                for (int term = 0; term < 3; ++term)
#pragma unroll
                    for (int it = 0; it < IT; ++it)
                        if (!abl::d_no_dw) acc[it][jt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[it][kTermA2[term]], b[kTermB2[term]], acc[it][jt], 0, 0, 0);
#pragma unroll
                for (int p = 0; p < 2; ++p) b[p] = bn[p];
            }
            IHG_DTRACE(k, 2)
            v4f gx[2];
            if (DX) {                                                    // row reads of the row-scaled dout images: chunk 4 kb + (lane >> 4) of row 16 rt + (lane & 15)
                const unsigned char* np = &nplanes[0][0][0][0] + BUF * (2 * DPL);
                const int arow = lane & 15, kq = lane >> 4;
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) {
                    gx[rt] = v4f{0.f, 0.f, 0.f, 0.f};
                    const int r = 16 * rt + arow;
#pragma unroll
                    for (int kb = 0; kb < 4; ++kb) {
                        v8h d2[2];
#pragma unroll
                        for (int p = 0; p < 2; ++p) d2[p] = *reinterpret_cast<const v8h*>(np + p * DPL + r * DRB + (((4 * kb + kq) ^ tr_swizzle(r)) << 4));
#pragma unroll
                        for (int term = 0; term < 3; ++term)
                            if (!abl::d_no_dx) gx[rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wdx[kb][kTermB2[term]], d2[kTermA2[term]], gx[rt], 0, 0, 0);
                    }
                    gx[rt] = gx[rt] * (wiv * rown[BUF][r]) + gold[rt];
                }
            }
            IHG_DTRACE(k, 3)
            publish(k + 2, fill);                                        // (takes delivery of the requested rows)
            IHG_DTRACE(k, 4)
            if (DX) {                                                    // (after the delivery: the counter is in order)
                const int arow = lane & 15, kq = lane >> 4;
                const int64_t r_base = r_begin + (static_cast<int64_t>(seq) + static_cast<int64_t>(k) * n_seq) * TE;
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) {
                    const int64_t v = r_base + 16 * rt + arow;
                    const int64_t vc = std::min(v, r_end - 1);
                    uint64_t dst = v < r_end ? reinterpret_cast<uint64_t>(typed_base(dx, row_type(vc)) + vc * ld_dx + 16 * wave + 4 * kq)
                                             : reinterpret_cast<uint64_t>(slabs + (static_cast<int64_t>(type) * n_seq + seq) * D * D + 4 * tid);
                    asm("" : "+v"(dst));                                 // (opaque: left visible, the choice becomes two stores under complementary branches)
                    if (!abl::d_no_stores || gx[rt][0] == 1.2345e30f) *reinterpret_cast<__attribute__((address_space(1))) v4f*>(dst) = gx[rt];
                }
            }
            IHG_DTRACE(k, 5)
            __syncthreads();
        };
        int k = 0;
        for (; k < n_my; k += 2) {
            phase(k, r1, r0);
            if (k + 1 < n_my) phase(k + 1, r0, r1);
        }
    }
    // slab [type][sequence][i][j]; accumulator tile (it, jt): row i = 16 (IT iq + it) + 4 (lane >> 4) + r, column j = 128 half + 64 jh + 16 jt + (lane & 15).
    // The accumulators hold 2^(254 + kTarget - l_acc) times the gradient (l_acc < 0: zeros)
    const int unscale = l_acc >= 0 ? l_acc - 254 - kTarget : 0;
    float* slab = slabs + (static_cast<int64_t>(type) * n_seq + seq) * D * D;
    const int c = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int it = 0; it < IT; ++it)
#pragma unroll
        for (int jt = 0; jt < 4; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                slab[static_cast<int64_t>(16 * (IT * iq + it) + 4 * kq + r) * D + 128 * half + 64 * jh + 16 * jt + c] = __builtin_ldexpf(acc[it][jt][r], unscale);
    // column sums of dout: the 32 staging rows of a column meet in LDS (the images are free now) and are added in row order
    float* red = reinterpret_cast<float*>(&dplanes[0][0][0][0]);        // [32][D] floats = 16 KB .. 32 KB of the 32 KB .. 64 KB of images
    __syncthreads();
#pragma unroll
    for (int i = 0; i < DOCT; ++i) {
        *reinterpret_cast<v4f*>(red + row * D + 128 * i + 8 * o) = bsum[2 * i];
        *reinterpret_cast<v4f*>(red + row * D + 128 * i + 8 * o + 4) = bsum[2 * i + 1];
    }
    __syncthreads();
    if (half == 0 && tid < D) {
        float sum = 0.f;
#pragma unroll 8
        for (int r = 0; r < TE; ++r) sum += red[r * D + tid];
        bias_slabs[(static_cast<int64_t>(type) * n_seq + seq) * D + tid] = sum;
    }
}

}  // namespace

int64_t split_dense_plane_floats(int dim) { return dim == 128 || dim == 256 ? (3LL * 3 * dim * dim) / 2 : 0; }

bool split_row_gemm_ok(int dim, const float* out, int64_t ld_out, const float* bias, int64_t bias_type_stride) {
    return split_arith_enabled() && (dim == 128 || dim == 256) && aligned16(out) && ld_out % 4 == 0 && (bias == nullptr || (aligned16(bias) && bias_type_stride % 4 == 0));
}

void launch_row_gemm_split(int dim, TypedRows in, int64_t ld_in, const float* w, int64_t ld_w, int64_t w_type_stride, int transpose, const float* bias,
                           int bias_mask, int64_t bias_type_stride, const int64_t* type_begin, TypedRowsOut out, int64_t ld_out, void* planes, hipStream_t s, int accumulate) {
    const int n_types = w_type_stride == 0 ? 1 : 3;
    v4u* pk = static_cast<v4u*>(planes);
    const int items = n_types * (dim / 16) * (dim / 32) * kWave;
    // two fp16 planes of the scaled weights, followed by the output columns' scales and their inverses ([n_types][dim] floats each)
    float* wsc = reinterpret_cast<float*>(pk + static_cast<int64_t>(items) * 2);
    float* winv = wsc + 3 * dim;
    hipLaunchKernelGGL(dense_weight_scales_kernel, dim3(grid_for_waves(static_cast<int64_t>(n_types) * dim)), dim3(kBlockThreads), 0, s, w, ld_w, w_type_stride, n_types, dim, transpose,
                       wsc, winv);
    hipLaunchKernelGGL(pack_planes_dense_h2_kernel, dim3((items + kBlockThreads - 1) / kBlockThreads), dim3(kBlockThreads), 0, s, w, ld_w, w_type_stride, n_types, dim,
                       transpose, wsc, pk);
    RowTiles plan;
    int acc = 0;
    for (int t = 0; t < 4; ++t) plan.begin[t] = type_begin[t];
    for (int t = 0; t < 3; ++t) {
        plan.tile_prefix[t] = acc;
        acc += static_cast<int>((type_begin[t + 1] - type_begin[t] + 31) / 32);
    }
    plan.tile_prefix[3] = acc;
    if (acc == 0) return;
    const int64_t pk_type_stride = n_types == 1 ? int64_t{0} : static_cast<int64_t>(dim / 16) * (dim / 32) * 2 * kWave;
    const int n_seq = std::min((acc + 7) / 8 * 8, 256);                  // (d = 256) tile sequences: a multiple of 8, so that both halves of one land on one XCD
#define IHG_ROW_GEMM(D, ACC, GRID) \
    hipLaunchKernelGGL((row_gemm_split_kernel<D, ACC>), dim3(GRID), dim3(512), 0, s, in, ld_in, pk, pk_type_stride, winv, bias, bias_mask, bias_type_stride, plan, out, ld_out)
    if (dim == 128) {
        if (accumulate) IHG_ROW_GEMM(128, true, std::min(acc, 256));
        else IHG_ROW_GEMM(128, false, std::min(acc, 256));
    } else {
        if (accumulate) IHG_ROW_GEMM(256, true, 2 * n_seq);
        else IHG_ROW_GEMM(256, false, 2 * n_seq);
    }
#undef IHG_ROW_GEMM
}

// node-level forward of the interactive layer: node_interact_fwd_grouped_kernel (d = 128), node_interact_fwd_grouped256_kernel (d = 256), node_interact_fwd_q_kernel (d = 64)
static int64_t node_fwd_q_v4(int dim) {                                  // v4u of the q kernel's planes
    const int bpp = 512 / dim, n_pass = (7 + bpp - 1) / bpp, parts = dim / 64;
    return 3LL * n_pass * parts * 4 * 16 * 2 * kWave;
}

int64_t split_node_fwd_plane_floats(int dim) {
    if (dim != 64 && dim != 128 && dim != 256) return 0;
    return (dim == 128 ? 3LL * 4 * kNodePassV4 : node_fwd_q_v4(dim)) * 4 + 2 * 3 * dim;     // two fp16 planes per weight + the weight rows' scales and inverses
}

bool split_node_fwd_ok(int dim, int order, int64_t ld_h, int64_t ld_s, const float* out, int64_t ld_out, const float* bias) {
    return split_arith_enabled() && (dim == 64 || dim == 128 || dim == 256) && (order == 2 || order == 3) && ld_ok(ld_h) && ld_ok(ld_s) && ld_ok(ld_out) && aligned16(out) &&
           (bias == nullptr || aligned16(bias));
}

static RowTiles row_tiles(const int64_t* type_begin, int rows_per_tile) {
    RowTiles plan;
    int acc = 0;
    for (int t = 0; t < 4; ++t) plan.begin[t] = type_begin[t];
    for (int t = 0; t < 3; ++t) {
        plan.tile_prefix[t] = acc;
        acc += static_cast<int>((type_begin[t + 1] - type_begin[t] + rows_per_tile - 1) / rows_per_tile);
    }
    plan.tile_prefix[3] = acc;
    return plan;
}

void launch_node_fwd_split(int dim, int order, const float* h, int64_t ld_h, const float* sums, int64_t ld_s, const float* deg, const float* scale, const float* bias,
                           const float* w, int64_t ld_w, const int64_t* type_begin, float* out, int64_t ld_out, void* planes, hipStream_t s) {
    v4u* wnp = static_cast<v4u*>(planes);
    if (dim == 128) {
        // the planes (2 fp16 per weight) are followed by the weight rows' scales and their inverses ([3][128] floats each)
        float* wsc = reinterpret_cast<float*>(wnp + 3LL * 4 * kNodePassV4);
        float* winv = wsc + 3 * 128;
        hipLaunchKernelGGL(node_fwd_weight_scales_kernel, dim3(grid_for_waves(3 * 128)), dim3(kBlockThreads), 0, s, w, ld_w, 128, order, wsc, winv);
        hipLaunchKernelGGL(pack_planes_node_fwd_kernel, dim3((3 * 4 * 4 * 2 * 8 * kWave + kBlockThreads - 1) / kBlockThreads), dim3(kBlockThreads), 0, s, w, ld_w, order, wsc, wnp);
        const RowTiles plan = row_tiles(type_begin, 32);
        if (plan.tile_prefix[3] == 0) return;
        {
            NodeGroups groups;
            groups.tiles = plan;
            int acc = 0;
            for (int t = 0; t < 3; ++t) {
                groups.group_prefix[t] = acc;
                acc += (plan.tile_prefix[t + 1] - plan.tile_prefix[t] + kNodeGroupTiles - 1) / kNodeGroupTiles;
            }
            groups.group_prefix[3] = acc;
            const int grid = std::min(acc, 256);
            if (order == 3) hipLaunchKernelGGL(node_interact_fwd_grouped_kernel<3>, dim3(grid), dim3(kSplitThreads), 0, s, h, ld_h, sums, ld_s, deg, scale, bias, wnp, winv, groups, out, ld_out);
            else hipLaunchKernelGGL(node_interact_fwd_grouped_kernel<2>, dim3(grid), dim3(kSplitThreads), 0, s, h, ld_h, sums, ld_s, deg, scale, bias, wnp, winv, groups, out, ld_out);
            return;
        }
    }
    const int items = static_cast<int>(node_fwd_q_v4(dim) / 2);
    float* wsc = reinterpret_cast<float*>(wnp + node_fwd_q_v4(dim));
    float* winv = wsc + 3 * dim;
    hipLaunchKernelGGL(node_fwd_weight_scales_kernel, dim3(grid_for_waves(3 * dim)), dim3(kBlockThreads), 0, s, w, ld_w, dim, order, wsc, winv);
    if (dim == 256) {
        // one launch: fourteen (order 2: twelve) passes of 128 contraction values over groups of four 32-row tiles
        const int passes = order == 3 ? 14 : 12;
        const int pack_items = 3 * passes * 4 * 4 * 4 * kWave;
        if (order == 3) hipLaunchKernelGGL(pack_planes_node_fwd256_kernel<3>, dim3((pack_items + kBlockThreads - 1) / kBlockThreads), dim3(kBlockThreads), 0, s, w, ld_w, wsc, wnp);
        else hipLaunchKernelGGL(pack_planes_node_fwd256_kernel<2>, dim3((pack_items + kBlockThreads - 1) / kBlockThreads), dim3(kBlockThreads), 0, s, w, ld_w, wsc, wnp);
        const RowTiles tiles = row_tiles(type_begin, 32);
        if (tiles.tile_prefix[3] == 0) return;
        NodeGroups groups;
        groups.tiles = tiles;
        int acc = 0;
        for (int t = 0; t < 3; ++t) {
            groups.group_prefix[t] = acc;
            acc += (tiles.tile_prefix[t + 1] - tiles.tile_prefix[t] + kNodeGroupTiles256 - 1) / kNodeGroupTiles256;
        }
        groups.group_prefix[3] = acc;
        const int grid = std::min(acc, 256);
        if (order == 3) hipLaunchKernelGGL(node_interact_fwd_grouped256_kernel<3>, dim3(grid), dim3(kSplitThreads), 0, s, h, ld_h, sums, ld_s, deg, scale, bias, wnp, winv, groups, out, ld_out);
        else hipLaunchKernelGGL(node_interact_fwd_grouped256_kernel<2>, dim3(grid), dim3(kSplitThreads), 0, s, h, ld_h, sums, ld_s, deg, scale, bias, wnp, winv, groups, out, ld_out);
        return;
    }
    hipLaunchKernelGGL(pack_planes_node_fwd_q_kernel, dim3((items + kBlockThreads - 1) / kBlockThreads), dim3(kBlockThreads), 0, s, w, ld_w, dim, order, wsc, wnp);
    const RowTiles plan = row_tiles(type_begin, 16);
    if (plan.tile_prefix[3] == 0) return;
    const int n_ranges = std::min((plan.tile_prefix[3] + 7) / 8 * 8, 256);              // (d = 64: one column part, one pass)
    hipLaunchKernelGGL((node_interact_fwd_q_kernel<64, 0, false, true>), dim3(n_ranges), dim3(kSplitThreads), 0, s, h, ld_h, sums, ld_s, deg, scale, bias, wnp, winv, plan, out, ld_out);
}

// node-level weight gradients of the product blocks (d = 64 / 128 / 256): see node_interact_weight_split_kernel
static int node_weight_ranges(int dim) { return dim == 64 ? 256 : (dim == 128 ? 128 : 32); }

int64_t split_node_weight_slab_floats(int dim, int order) {
    return dim == 64 || dim == 128 || dim == 256 ? static_cast<int64_t>(node_weight_ranges(dim)) * dim * (order == 3 ? 4 : 3) * dim : 0;
}

bool split_node_weight_ok(int dim, int order, int64_t ld_h, int64_t ld_s, int64_t ld_dy, const float* dy) {
    return split_arith_enabled() && (dim == 64 || dim == 128 || dim == 256) && (order == 2 || order == 3) && ld_ok(ld_h) && ld_ok(ld_s) && ld_ok(ld_dy) && aligned16(dy);
}

void launch_node_weight_split(int dim, int order, const float* h, int64_t ld_h, const float* sums, int64_t ld_s, const float* dy, int64_t ld_dy, const float* dy_scale,
                              const int64_t* type_begin, float* slabs, float* dw, int64_t ld_dw, hipStream_t s) {
    NodeRanges plan;
    int64_t tiles[3], total = 0;
    for (int t = 0; t < 4; ++t) plan.begin[t] = type_begin[t];
    for (int t = 0; t < 3; ++t) {
        tiles[t] = (type_begin[t + 1] - type_begin[t] + kSplitTE - 1) / kSplitTE;
        total += tiles[t];
    }
    // tile ranges by type in proportion to the tiles, every non-empty type at least one, `ranges` in all at most
    const int ranges = node_weight_ranges(dim);
    int acc = 0;
    for (int t = 0; t < 3; ++t) {
        int n = tiles[t] == 0 ? 0 : static_cast<int>(std::max<int64_t>(1, tiles[t] * (ranges - 2) / std::max<int64_t>(total, 1)));
        n = static_cast<int>(std::min<int64_t>(n, tiles[t]));
        plan.range_prefix[t] = acc;
        acc += n;
    }
    plan.range_prefix[3] = acc;
    const int nblk = order == 3 ? 4 : 3;
#define IHG_NODE_WEIGHT(D)                                                                                                                                   \
    {                                                                                                                                                        \
        if (order == 3) hipLaunchKernelGGL((node_interact_weight_split_kernel<D, 4>), dim3(256), dim3(kSplitThreads), 0, s, h, ld_h, sums, ld_s, dy, ld_dy, dy_scale, plan, slabs); \
        else hipLaunchKernelGGL((node_interact_weight_split_kernel<D, 3>), dim3(256), dim3(kSplitThreads), 0, s, h, ld_h, sums, ld_s, dy, ld_dy, dy_scale, plan, slabs);           \
    }
    if (dim == 64) IHG_NODE_WEIGHT(64)
    else if (dim == 128) IHG_NODE_WEIGHT(128)
    else IHG_NODE_WEIGHT(256)
#undef IHG_NODE_WEIGHT
    const int total_w = dim * nblk * dim;
    hipLaunchKernelGGL(node_weight_reduce_kernel, dim3((total_w + kWave - 1) / kWave), dim3(kBlockThreads), 0, s, slabs, plan, dim, nblk, dw, ld_dw);
}

bool split_dense_weight_ok(int dim, const float* dout, int64_t ld_dout, const float* x, int64_t ld_x) {
    return split_arith_enabled() && (dim == 128 || dim == 256) && aligned16(dout) && aligned16(x) && ld_dout % 4 == 0 && ld_x % 4 == 0;
}

int launch_dense_weight_split(int dim, const float* dout, int64_t ld_dout, TypedRows x, int64_t ld_x, const int64_t* type_begin, int n_types, float* slabs,
                              float* bias_slabs, const float* w, int64_t ld_w, int64_t w_type_stride, const TypedRowsOut* dx_rows, int64_t ld_dx, void* planes, hipStream_t s,
                              int dx_accumulate) {
    const bool has_dx = dx_rows != nullptr;
    const TypedRowsOut dx = has_dx ? *dx_rows : typed_rows_out(nullptr);
    RowTiles plan;
    for (int t = 0; t < 4; ++t) plan.begin[t] = type_begin[t];
    for (int t = 0; t < 4; ++t) plan.tile_prefix[t] = 0;                 // (the kernel takes its tiles from the row ranges)
    if (dim == 128) {
        const int n_seq = 256;
        if (has_dx) {                                                    // fused input gradient: planes of W for out = in W
            // two fp16 planes per weight, then the output columns' scales and their inverses ([types][128] floats each)
            v4u* pk = static_cast<v4u*>(planes);
            float* wsc = reinterpret_cast<float*>(pk + static_cast<int64_t>(n_types) * 8 * 4 * 2 * kWave);
            float* winv = wsc + n_types * dim;
            const int items = n_types * 8 * 4 * kWave;
            hipLaunchKernelGGL(dense_weight_scales_kernel, dim3(grid_for_waves(static_cast<int64_t>(n_types) * dim)), dim3(kBlockThreads), 0, s, w, ld_w, w_type_stride, n_types, dim, 1,
                               wsc, winv);
            hipLaunchKernelGGL(pack_planes_dense_h2_kernel, dim3((items + kBlockThreads - 1) / kBlockThreads), dim3(kBlockThreads), 0, s, w, ld_w, w_type_stride, n_types,
                               dim, 1, wsc, pk);
            if (dx_accumulate)
                hipLaunchKernelGGL((dense_weight_grad_split_kernel<128, true, true>), dim3(n_seq, 1, n_types), dim3(kSplitThreads), 0, s, dout, ld_dout, x, ld_x, plan,
                                   n_types == 1 ? 1 : 0, slabs, bias_slabs, pk, n_types == 1 ? int64_t{0} : int64_t{8 * 4 * 2 * kWave}, winv, dx, ld_dx);
            else
                hipLaunchKernelGGL((dense_weight_grad_split_kernel<128, true, false>), dim3(n_seq, 1, n_types), dim3(kSplitThreads), 0, s, dout, ld_dout, x, ld_x, plan,
                                   n_types == 1 ? 1 : 0, slabs, bias_slabs, pk, n_types == 1 ? int64_t{0} : int64_t{8 * 4 * 2 * kWave}, winv, dx, ld_dx);
        } else {
            hipLaunchKernelGGL((dense_weight_grad_split_kernel<128, false, false>), dim3(n_seq, 1, n_types), dim3(kSplitThreads), 0, s, dout, ld_dout, x, ld_x, plan,
                               n_types == 1 ? 1 : 0, slabs, bias_slabs, static_cast<const v4u*>(nullptr), int64_t{0}, static_cast<const float*>(nullptr),
                               typed_rows_out(nullptr), int64_t{0});
        }
        return n_seq;
    }
    const int n_seq = 128;
    hipLaunchKernelGGL((dense_weight_grad_split_kernel<256, false, false>), dim3(2 * n_seq, 1, n_types), dim3(kSplitThreads), 0, s, dout, ld_dout, x, ld_x, plan,
                       n_types == 1 ? 1 : 0, slabs, bias_slabs, static_cast<const v4u*>(nullptr), int64_t{0}, static_cast<const float*>(nullptr), typed_rows_out(nullptr),
                       int64_t{0});
    return n_seq;
}

#ifdef IHG_ABL_D_TRACE
extern "C" int ihg_ablation_trace_dense(unsigned long long* out) {       // [64][6] clock stamps of the last node-level linear backward (tools/phase_trace.py --kernel linear)
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dense_trace), sizeof(g_dense_trace), 0, hipMemcpyDeviceToHost) == hipSuccess ? 0 : 1;
}
#endif
