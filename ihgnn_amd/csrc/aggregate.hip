// aggregate.hip - the two HBM-bound aggregation kernels: K5 node -> hyperedge gather-sum, K7 hyperedge -> node segment-sum
// (also: embedding-bag mean, scatter-add backward, two-hop first-order pass, weighted-CSR GCN propagation).
//
// Layout idea shared by both: a feature row of `dim` floats is owned by a GROUP of G = dim/4 lanes (16 B per lane, so one group
// instruction moves one whole row and one wave instruction moves 64/G rows = 1 KiB), every lane keeps its own 4 columns in
// registers for the whole reduction (no cross-lane adds), and the only cross-lane traffic is the index stream: indices are
// fetched once per wave with a single coalesced load and handed to their group with wavefront shuffles (ds_bpermute), so the
// dependent row gathers of several hyperedges are in flight together.
#include "common.hpp"
#include "split_common.hpp"

namespace {

// ================================================================================================
// K5  node -> hyperedge gather-sum
//   G lanes own one hyperedge row; a wave works on EPW = (64/G)*U consecutive hyperedges per iteration (U = 3 from d = 64 up):
//   one coalesced load brings their 3*EPW member ids (<= 64 ints), shuffles hand each group its ids, then
//   up to 3*U independent row gathers per lane are issued before the first add.  Group g takes the U CONSECUTIVE
//   hyperedges e0 + g U .. + U - 1: the layout numbers hyperedges by user, so neighbours usually share their user
//   row, which is then fetched once and reused from registers (a third of the gathers at ~10 hyperedges per user).
//   The [E, d] result is written once and not re-read by this kernel: non-temporal stores, so the stream does not
//   evict the node table from L2 / Infinity Cache (C3: 540 -> 460 us; C2: 160 -> 131 us).
// ================================================================================================
// consecutive hyperedges per lane group (3 U row gathers in flight per lane); A/B: tools/ab_aggregate.sh NAME -DIHG_K5_U32=n.  Round 6 measured U = 2 / 3 / 4 / 5 at
// d = 64 / 128 / 256 (profiles/r6/09_ab_k5_hyperedges_per_group.txt): THREE is the best everywhere - C2 (d = 64) 111 -> 90 us, C3 (d = 128) 418 -> 400, C4 684 -> 611,
// C5 x 0.2 (d = 256) 2,333 -> 2,156 - nine gathers in flight leave the registers for one more resident wave than twelve; five is slower than four.  The fp16-plane form
// (edge_gather_sum_planes256_kernel) stays at four: 11.3 ms against 12.0 (three) in the C5 step.
#ifndef IHG_K5_U16
#define IHG_K5_U16 3
#endif
#ifndef IHG_K5_U32
#define IHG_K5_U32 3
#endif
#ifndef IHG_K5_U64
#define IHG_K5_U64 3
#endif
#ifndef IHG_K5_UPLANES
#define IHG_K5_UPLANES 4
#endif
template <int VEC, int G, int U>
__global__ __launch_bounds__(kBlockThreads) void edge_gather_sum_kernel(
    const float* __restrict__ src, int64_t ld_src, const int32_t* __restrict__ i3,
    const float* __restrict__ node_scale, const float* __restrict__ bias, float alpha, const float* __restrict__ edge_scale,
    float* __restrict__ out, int64_t ld_out, int64_t n_edges, int dim_vec) {
    constexpr int GPW = kWave / G;
    constexpr int EPW = GPW * U;
    static_assert(EPW * 3 <= kWave, "member ids of one wave iteration must fit one coalesced load");
    const int lane = threadIdx.x & (kWave - 1);
    const int lig = lane & (G - 1);
    const int grp = lane / G;
    const int64_t n_ids = n_edges * 3;

    // the member ids of the NEXT iteration are requested before this iteration's rows: the id -> row dependency then costs one memory
    // round trip per iteration instead of two
    const int64_t stride = global_wave_count() * EPW;
    int64_t e0 = global_wave_id() * EPW;
    int next_id = (lane < EPW * 3 && e0 * 3 + lane < n_ids) ? i3[e0 * 3 + lane] : 0;
    for (; e0 < n_edges; e0 += stride) {
        const int64_t pos = e0 * 3 + lane;
        const bool have = lane < EPW * 3 && pos < n_ids;
        const int my_id = next_id;
        {
            const int64_t npos = pos + stride * 3;
            next_id = (lane < EPW * 3 && npos < n_ids) ? i3[npos] : 0;
        }
        const float my_scale = (node_scale != nullptr && have) ? node_scale[my_id] : 1.f;

        int ids[U][3];
        float sc[U][3];
#pragma unroll
        for (int t = 0; t < U; ++t) {
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                const int from = (grp * U + t) * 3 + m;
                ids[t][m] = __shfl(my_id, from);
                sc[t][m] = __shfl(my_scale, from);
            }
        }
        for (int c = lig; c < dim_vec; c += G) {
            Frag<VEC> rows[U][3];
#pragma unroll
            for (int t = 0; t < U; ++t) {
                const bool live = e0 + grp * U + t < n_edges;
#pragma unroll
                for (int m = 0; m < 3; ++m) {
                    if (m == 0 && t > 0 && ids[t][0] == ids[t - 1][0]) {
                        rows[t][0] = rows[t - 1][0];                    // same user as the previous hyperedge: row already here
                        continue;
                    }
                    rows[t][m] = live ? Frag<VEC>::load(src + static_cast<int64_t>(ids[t][m]) * ld_src + c * VEC)
                                      : Frag<VEC>::zero();
                }
            }
            Frag<VEC> b = bias != nullptr ? Frag<VEC>::load(bias + c * VEC) : Frag<VEC>::zero();
#pragma unroll
            for (int t = 0; t < U; ++t) {
                const int64_t e = e0 + grp * U + t;
                if (e >= n_edges) continue;
                Frag<VEC> acc = Frag<VEC>::zero();
                acc.add_scaled(rows[t][0], sc[t][0]);      // (u + q) + i, the order of a row-major SpMM row
                acc.add_scaled(rows[t][1], sc[t][1]);
                acc.add_scaled(rows[t][2], sc[t][2]);
                acc.mul(edge_scale != nullptr ? alpha * edge_scale[e] : alpha);      // edge_scale: the hyperedge's multiplicity (a layout with duplicate triples collapsed)
                acc.add(b);
                acc.store_stream(out + e * ld_out + c * VEC);
            }
        }
    }
}

// ================================================================================================
// K7  hyperedge -> node segment-sum (also: EmbeddingBag mean forward/backward, scatter-add backward)
//   G lanes own one output row and walk its id list in chunks of G ids: one coalesced id load per chunk,
//   shuffles broadcast each id inside the group, UNR (8 or 16) row gathers in flight per lane, adds in list order.
//   The chunk loop is made wave-uniform with a cross-group max so the shuffles always run converged.
// ================================================================================================
// STREAM: every source row is read exactly once by the launch (the member-gradient scatter): non-temporal loads, the rows do not displace what other
// kernels keep in the caches (C3: 488 -> 436 us; the same on rows that ARE re-read - the two-hop launches - costs 50 %: 740 -> 1,120 us)
#ifndef IHG_K7_UNR32
#define IHG_K7_UNR32 16
#endif
#ifndef IHG_K7_WAVES
#define IHG_K7_WAVES 1      // minimum resident waves per SIMD asked of the compiler for K7 / the pair sums (A/B: tools/ab_aggregate.sh NAME -DIHG_K7_WAVES=n -DIHG_PAIR_WAVES=n)
#endif
#ifndef IHG_PAIR_WAVES
#define IHG_PAIR_WAVES 1
#endif
#ifndef IHG_PAIR_UNR_W
#define IHG_PAIR_UNR_W 8      // the weighted pair sums (a multiplicity per pair: config C5) carry a weight per pair in flight as well: 8 ids, seven resident waves (13.06 -> 12.72 ms)
#endif
template <int VEC, int G, bool STREAM = false>
__device__ __forceinline__ Frag<VEC> accumulate_list(const float* __restrict__ src, int64_t ld_src,
                                                     const int32_t* __restrict__ ids, const float* __restrict__ src_scale,
                                                     const float* __restrict__ entry_scale, const uint8_t* __restrict__ src_mask,
                                                     int begin, int len, int wave_max_len, int lane, int col) {
    // row gathers in flight per lane: 16 at G = 32 (d = 128: 2-3 % on C3's launches over 8), 8 otherwise - at G = 64 (d = 256) eight leave the registers for six resident
    // waves instead of four and C5's two-hop launches go 9.19 / 9.43 -> 8.71 / 8.96 ms (round 6: profiles/r6/09_ab_gather_occupancy.txt; asking the compiler for more
    // waves than the loop's registers allow - launch bounds - spills and doubles the launch)
    constexpr int UNR = G < 8 ? G : (G == 32 ? IHG_K7_UNR32 : 8);
    const int lig = lane & (G - 1);
    const int group_base = lane & ~(G - 1);
    Frag<VEC> acc = Frag<VEC>::zero();
    for (int base = 0; base < wave_max_len; base += G) {
        const bool have = base + lig < len;
        int my_id = have ? ids[begin + base + lig] : -1;
        if (src_mask != nullptr && have && src_mask[my_id] == 0) my_id = -1;      // a source row known to be zero: not fetched
        const bool live = my_id >= 0;
        float my_w = (src_scale != nullptr && live) ? src_scale[my_id] : 1.f;
        if (entry_scale != nullptr && live) my_w *= entry_scale[begin + base + lig];
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
                row[k] = (id[k] >= 0 && col >= 0) ? (STREAM ? Frag<VEC>::load_stream(src + static_cast<int64_t>(id[k]) * ld_src + col * VEC)
                                                            : Frag<VEC>::load(src + static_cast<int64_t>(id[k]) * ld_src + col * VEC))
                                                  : Frag<VEC>::zero();
#pragma unroll
            for (int k = 0; k < UNR; ++k) acc.add_scaled(row[k], w[k]);
        }
    }
    return acc;
}

// The same sum for a launch with a source mask (the pull of the last layer's backward: a per cent or two of the ids are listed, yet two chunks of ids in
// three hold at least one, and the lists around a popular listed node hold dozens).  Only the listed ids of a chunk are visited, in list order (the unlisted
// ones add exact zeros in the dense form: the same sum, bit for bit), eight row requests in flight; no dense loop: half the registers, twice the resident
// waves.
template <int VEC, int G>
__device__ __forceinline__ Frag<VEC> accumulate_list_masked(const float* __restrict__ src, int64_t ld_src, const int32_t* __restrict__ ids,
                                                            const float* __restrict__ src_scale, const float* __restrict__ entry_scale,
                                                            const uint8_t* __restrict__ src_mask, int begin, int len, int wave_max_len, int lane, int col) {
    constexpr int FLY = 8;
    const int lig = lane & (G - 1);
    const int group_base = lane & ~(G - 1);
    Frag<VEC> acc = Frag<VEC>::zero();
    for (int base = 0; base < wave_max_len; base += G) {
        const bool have = base + lig < len;
        int my_id = have ? ids[begin + base + lig] : -1;
        if (have && src_mask[my_id] == 0) my_id = -1;
        const bool live = my_id >= 0;
        float my_w = (src_scale != nullptr && live) ? src_scale[my_id] : 1.f;
        if (entry_scale != nullptr && live) my_w *= entry_scale[begin + base + lig];
        uint64_t gm = __ballot(live) >> group_base;
        if (G < 64) gm &= (uint64_t{1} << G) - 1;
        while (__ballot(gm != 0) != 0) {                                     // wave-uniform: until every group has walked its listed ids
            Frag<VEC> row[FLY];
            float w[FLY];
#pragma unroll
            for (int t = 0; t < FLY; ++t) {
                const int pos = gm != 0 ? __builtin_ctzll(gm) : 0;
                const int idk = __shfl(my_id, group_base + pos);
                w[t] = __shfl(my_w, group_base + pos);
                row[t] = (gm != 0 && col >= 0) ? Frag<VEC>::load(src + static_cast<int64_t>(idk) * ld_src + col * VEC) : Frag<VEC>::zero();
                if (gm == 0) w[t] = 0.f;
                gm &= gm - 1;
            }
#pragma unroll
            for (int t = 0; t < FLY; ++t) acc.add_scaled(row[t], w[t]);
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
    mode &= 0xff;                                           // (IHG_SCALE_ACCUMULATE rides in the mode word)
    if (mode == IHG_SCALE_MULTIPLY) {
        acc.mul(out_scale[row]);
    } else if (mode == IHG_SCALE_DIVIDE) {
        const float s = out_scale[row];
        if (s != 0.f) acc.div(s);
    }
}

// Work list of one launch: first the fixed-length segments of the split (heavy) rows, then the light rows in `row_order`
// (decreasing length).  Unit u < n_segments writes partials[u]; unit u >= n_segments writes its output row.
template <int VEC, int G, bool MASKED = false, bool STREAM = false>
__global__ __launch_bounds__(kBlockThreads, IHG_K7_WAVES) void node_segment_sum_kernel(
    const float* __restrict__ src, int64_t ld_src, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ ids,
    const int32_t* __restrict__ row_order, const float* __restrict__ src_scale, const float* __restrict__ entry_scale,
    const float* __restrict__ out_scale, int mode,
    float* __restrict__ out, int64_t ld_out, int64_t n_rows, int dim, int dim_vec, int heavy_threshold,
    const int32_t* __restrict__ seg_begin, const int32_t* __restrict__ seg_end, int64_t n_segments, float* __restrict__ partials,
    const float* __restrict__ self_weight, const uint8_t* __restrict__ src_mask) {
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
            Frag<VEC> acc = MASKED ? accumulate_list_masked<VEC, G>(src, ld_src, ids, src_scale, entry_scale, src_mask, begin, len, wave_len, lane, col)
                                   : accumulate_list<VEC, G, STREAM>(src, ld_src, ids, src_scale, entry_scale, src_mask, begin, len, wave_len, lane, col);
            if (dst != nullptr && col >= 0) {
                if (scale_row >= 0) {
                    if (self_weight != nullptr && (src_mask == nullptr || src_mask[scale_row] != 0))   // square operators: the row's own source row, weighted (a masked row is zero: not fetched)
                        acc.add_scaled(Frag<VEC>::load(src + scale_row * ld_src + col * VEC),
                                       self_weight[scale_row] * (src_scale != nullptr ? src_scale[scale_row] : 1.f));
                    apply_out_scale<VEC>(acc, out_scale, mode, scale_row);
                    if (mode & IHG_SCALE_ACCUMULATE) acc.add(Frag<VEC>::load(dst + col * VEC));     // out += (hyperedge chunks of one scatter)
                }
                acc.store(dst + col * VEC);
            }
        }
    }
}

// ================================================================================================
// Pair sums: the gather half of the interactive layer in its NODE-LEVEL form ("the layer without hyperedge rows").
//   For node v, over its incident hyperedges e with their other two members (a_e, b_e) - the id pairs of the two-hop list -
//       S_a[v] = sum_e h[a_e]      S_b[v] = sum_e h[b_e]      S_ab[v] = sum_e h[a_e] * h[b_e]   (elementwise product)
//   written side by side as one row of 3 d floats.  Every product of member features that the interactive layer contracts has, for a fixed
//   node, that node's own feature as a constant factor: the sums of the hyperedge features over the node's hyperedges are linear maps of
//   (h[v], S_a, S_b, S_ab), so no [E, d] row is ever formed (ihg_node_interact_fwd applies the maps).  Same work list, lane layout and
//   split-row plan as K7; a segment of a split row must hold whole pairs (the layout keeps segment lengths even).
// ================================================================================================
#ifndef IHG_PAIR_UNR
#define IHG_PAIR_UNR 16
#endif
template <int G, bool WEIGHTED = false>
__global__ __launch_bounds__(kBlockThreads, IHG_PAIR_WAVES) void node_pair_sums_kernel(
    const float* __restrict__ h, int64_t ld_h, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ ids,
    const int32_t* __restrict__ row_order, float* __restrict__ out, int64_t ld_out, int64_t n_rows, int dim, int dim_vec,
    int heavy_threshold, const int32_t* __restrict__ seg_begin, const int32_t* __restrict__ seg_end, int64_t n_segments,
    float* __restrict__ partials, const float* __restrict__ pair_weight) {
    constexpr int GPW = kWave / G;
    constexpr int UNR = G < 8 ? G : (G >= 32 ? (WEIGHTED ? IHG_PAIR_UNR_W : IHG_PAIR_UNR) : 8);      // ids in flight per lane: UNR / 2 pairs
    const int lane = threadIdx.x & (kWave - 1);
    const int lig = lane & (G - 1);
    const int grp = lane / G;
    const int group_base = lane & ~(G - 1);
    const int64_t n_units = n_segments + n_rows;
    for (int64_t u0 = global_wave_id() * GPW; u0 < n_units; u0 += global_wave_count() * GPW) {
        const int64_t u = u0 + grp;
        int begin = 0, len = 0;
        float* dst = nullptr;
        if (u < n_segments) {
            begin = seg_begin[u];
            len = seg_end[u] - begin;
            dst = partials + u * 3 * dim;
        } else if (u < n_units) {
            int64_t r = u - n_segments;
            if (row_order != nullptr) r = row_order[r];
            begin = rowptr[r];
            len = rowptr[r + 1] - begin;
            if (heavy_threshold > 0 && len > heavy_threshold) len = 0;      // finished from the partials
            else dst = out + r * ld_out;
        }
        const int wave_len = wave_max_over_groups<G>(len);
        const int col_iters = (dim_vec + G - 1) / G;
        for (int ci = 0; ci < col_iters; ++ci) {
            const int c = ci * G + lig;
            const bool col_ok = c < dim_vec;
            Frag<4> sa = Frag<4>::zero(), sb = Frag<4>::zero(), sab = Frag<4>::zero();
            if (WEIGHTED) {
                // a layout with duplicate (user, query, item) triples collapsed: pair p of the list (ids 2 p, 2 p + 1; the lists start at even offsets) stands for
                // pair_weight[p] hyperedges - every sum takes the pair that many times
                for (int base = 0; base < wave_len; base += G) {
                    const bool have = base + lig < len;
                    const int my_id = have ? ids[begin + base + lig] : -1;
                    const float my_w = have ? pair_weight[(begin + base + lig) >> 1] : 0.f;
#pragma unroll 1
                    for (int j = 0; j < G; j += UNR) {
                        if (base + j >= wave_len) break;          // wave-uniform
                        int id[UNR];
                        float w[UNR / 2];
                        Frag<4> row[UNR];
#pragma unroll
                        for (int k = 0; k < UNR; ++k) id[k] = __shfl(my_id, group_base + j + k);
#pragma unroll
                        for (int k = 0; k < UNR; k += 2) w[k / 2] = __shfl(my_w, group_base + j + k);
#pragma unroll
                        for (int k = 0; k < UNR; ++k)
                            row[k] = (id[k] >= 0 && col_ok) ? Frag<4>::load(h + static_cast<int64_t>(id[k]) * ld_h + c * 4) : Frag<4>::zero();
#pragma unroll
                        for (int k = 0; k < UNR; k += 2) {
                            const float m = w[k / 2];
                            sa.add_scaled(row[k], m);
                            sb.add_scaled(row[k + 1], m);
                            sab.v.x += m * (row[k].v.x * row[k + 1].v.x);
                            sab.v.y += m * (row[k].v.y * row[k + 1].v.y);
                            sab.v.z += m * (row[k].v.z * row[k + 1].v.z);
                            sab.v.w += m * (row[k].v.w * row[k + 1].v.w);
                        }
                    }
                }
            } else
            for (int base = 0; base < wave_len; base += G) {
                const int my_id = base + lig < len ? ids[begin + base + lig] : -1;
#pragma unroll 1
                for (int j = 0; j < G; j += UNR) {
                    if (base + j >= wave_len) break;          // wave-uniform
                    int id[UNR];
                    Frag<4> row[UNR];
#pragma unroll
                    for (int k = 0; k < UNR; ++k) id[k] = __shfl(my_id, group_base + j + k);
#pragma unroll
                    for (int k = 0; k < UNR; ++k)
                        row[k] = (id[k] >= 0 && col_ok) ? Frag<4>::load(h + static_cast<int64_t>(id[k]) * ld_h + c * 4) : Frag<4>::zero();
#pragma unroll
                    for (int k = 0; k < UNR; k += 2) {
                        sa.add(row[k]);
                        sb.add(row[k + 1]);
                        sab.v.x += row[k].v.x * row[k + 1].v.x;
                        sab.v.y += row[k].v.y * row[k + 1].v.y;
                        sab.v.z += row[k].v.z * row[k + 1].v.z;
                        sab.v.w += row[k].v.w * row[k + 1].v.w;
                    }
                }
            }
            if (dst != nullptr && col_ok) {
                sa.store_stream(dst + c * 4);                     // (non-temporal: 576 MB at C3 that would only push the gathered table out of L2; 1-2 %)
                sb.store_stream(dst + dim + c * 4);
                sab.store_stream(dst + 2 * dim + c * 4);
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
    const float* __restrict__ src, int64_t ld_src, const float* __restrict__ src_scale, const float* __restrict__ self_weight,
    const uint8_t* __restrict__ src_mask = nullptr) {
    constexpr int GROUPS = kBlockThreads / G;
    __shared__ __attribute__((aligned(16))) float red[GROUPS][G * VEC];
    const int lig = threadIdx.x & (G - 1);
    const int grp = threadIdx.x / G;
    for (int64_t h = blockIdx.x; h < n_heavy; h += gridDim.x) {
        const int64_t row = heavy_rows[h];
        const int s_begin = heavy_segptr[h], s_end = heavy_segptr[h + 1];
        for (int c0 = blockIdx.y * G; c0 < dim_vec; c0 += gridDim.y * G) {     // (gridDim.y > 1: the column groups of a wide row - the 3 d-wide pair sums - on workgroups of their own)
            const int c = c0 + lig;
            Frag<VEC> acc = Frag<VEC>::zero();
            if (c < dim_vec) {
                // the longest row's chain of dependent round trips IS this kernel's duration (one workgroup per row): sixteen partials in flight per
                // lane group, added in index order
                constexpr int FLY = 16;
                int sgm = s_begin + grp;
                for (; sgm + (FLY - 1) * GROUPS < s_end; sgm += FLY * GROUPS) {
                    Frag<VEC> a[FLY];
#pragma unroll
                    for (int k = 0; k < FLY; ++k) a[k] = Frag<VEC>::load(partials + static_cast<int64_t>(sgm + k * GROUPS) * dim + c * VEC);
#pragma unroll
                    for (int k = 0; k < FLY; ++k) acc.add(a[k]);
                }
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
                if (self_weight != nullptr && (src_mask == nullptr || src_mask[row] != 0))
                    total.add_scaled(Frag<VEC>::load(src + row * ld_src + c * VEC), self_weight[row] * (src_scale != nullptr ? src_scale[row] : 1.f));
                apply_out_scale<VEC>(total, out_scale, mode, row);
                if (mode & IHG_SCALE_ACCUMULATE) total.add(Frag<VEC>::load(out + row * ld_out + c * VEC));
                total.store(out + row * ld_out + c * VEC);
            }
            __syncthreads();
        }
    }
}

// ------------------------------------------------------------------------------------------------
// K5 with its result laid down as the member-gradient kernel's operand (d = 256; round 5): the hyperedges' cotangents are read by exactly one consumer, the
// member-gradient kernel, whose eight column parts each scaled and took apart the same row again (a seventh of that kernel's time at config C5).  Here the row is
// formed as in edge_gather_sum_kernel (one wave per hyperedge row, four consecutive hyperedges per iteration, the shared user row fetched once), scaled by the power of
// two of its largest magnitude and written as the two fp16 terms of split_common.hpp - [e][plane][d] fp16, 4 d bytes per row like the fp32 row it replaces - with the
// inverse scale beside it.  Same sums, same split: the consumer's results are bit-identical to those from the fp32 rows.
// ------------------------------------------------------------------------------------------------
typedef unsigned v2u __attribute__((ext_vector_type(2)));
template <int U>
__global__ __launch_bounds__(kBlockThreads) void edge_gather_sum_planes256_kernel(const float* __restrict__ src, int64_t ld_src, const int32_t* __restrict__ i3,
                                                                                  const float* __restrict__ node_scale, const float* __restrict__ edge_scale,
                                                                                  v2u* __restrict__ planes, float* __restrict__ inv_out, int64_t n_edges) {
    constexpr int D = 256, EPW = U;
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t n_ids = n_edges * 3;
    const int64_t stride = global_wave_count() * EPW;
    int64_t e0 = global_wave_id() * EPW;
    int next_id = (lane < EPW * 3 && e0 * 3 + lane < n_ids) ? i3[e0 * 3 + lane] : 0;
    for (; e0 < n_edges; e0 += stride) {
        const int64_t pos = e0 * 3 + lane;
        const bool have = lane < EPW * 3 && pos < n_ids;
        const int my_id = next_id;
        {
            const int64_t npos = pos + stride * 3;
            next_id = (lane < EPW * 3 && npos < n_ids) ? i3[npos] : 0;
        }
        const float my_scale = (node_scale != nullptr && have) ? node_scale[my_id] : 1.f;
        int ids[U][3];
        float sc[U][3];
#pragma unroll
        for (int t = 0; t < U; ++t) {
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                ids[t][m] = __builtin_amdgcn_readlane(my_id, t * 3 + m);
                sc[t][m] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, my_scale), t * 3 + m));
            }
        }
        Frag<4> rows[U][3];
#pragma unroll
        for (int t = 0; t < U; ++t) {
            const bool live = e0 + t < n_edges;
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                if (m == 0 && t > 0 && ids[t][0] == ids[t - 1][0]) {
                    rows[t][0] = rows[t - 1][0];                        // same user as the previous hyperedge: row already here
                    continue;
                }
                rows[t][m] = live ? Frag<4>::load(src + static_cast<int64_t>(ids[t][m]) * ld_src + lane * 4) : Frag<4>::zero();
            }
        }
#pragma unroll
        for (int t = 0; t < U; ++t) {
            const int64_t e = e0 + t;
            if (e >= n_edges) continue;
            Frag<4> acc = Frag<4>::zero();
            acc.add_scaled(rows[t][0], sc[t][0]);          // (u + q) + i, as in edge_gather_sum_kernel
            acc.add_scaled(rows[t][1], sc[t][1]);
            acc.add_scaled(rows[t][2], sc[t][2]);
            if (edge_scale != nullptr) acc.mul(edge_scale[e]);      // the hyperedge's multiplicity (duplicate triples collapsed): the cotangent of all its copies
            // the row's largest magnitude: 16 lanes by DPP, the four 16-lane rows through scalar registers (unsigned order = float order for m >= 0)
            float m = abs_max3(acc.v.z, acc.v.w, abs_max3(acc.v.x, acc.v.y, 0.f));
            m = row_lanes_max<16>(m);
            const unsigned mu = __float_as_uint(m);
            const unsigned m0 = __builtin_amdgcn_readlane(mu, 0), m1 = __builtin_amdgcn_readlane(mu, 16), m2 = __builtin_amdgcn_readlane(mu, 32),
                           m3 = __builtin_amdgcn_readlane(mu, 48);
            const unsigned ma = m0 > m1 ? m0 : m1, mb = m2 > m3 ? m2 : m3;
            float inv;
            const float s = scale_up_for(__uint_as_float(ma > mb ? ma : mb), inv);
            unsigned h0, l0, h1, l1;
            split_pair_h2(acc.v.x * s, acc.v.y * s, h0, l0);
            split_pair_h2(acc.v.z * s, acc.v.w * s, h1, l1);
            v2u* dst = planes + e * (D / 2);                             // a row: D / 2 pairs of dwords (two planes of D / 4 each)
            __builtin_nontemporal_store(v2u{h0, h1}, dst + lane);
            __builtin_nontemporal_store(v2u{l0, l1}, dst + D / 4 + lane);
            if (lane == 0) inv_out[e] = inv;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Dispatch helpers
// ------------------------------------------------------------------------------------------------

// workgroups of a gather launch: one wave per unit of work up to IHG_AGG_MAX_BLOCKS workgroups, grid-stride beyond (A/B: tools/ab_aggregate.sh NAME -DIHG_AGG_MAX_BLOCKS=n).
// 256 CUs x 64: the launches' units differ in length (lists sorted longest first, split-row segments in front), and a grid of 2,048 persistent workgroups - what fits the
// chip at once, the library's cap elsewhere - ends with a long tail of the few workgroups whose stride met the long units; with eight times as many, shorter-lived
// workgroups the hardware scheduler evens that out.  Round 6, profiles/r6/09_ab_gather_grid.txt: step C3 7.53 -> 7.36 ms, C2 1.765 -> 1.717, C4 10.28 -> 10.04,
// C5 158.5 -> 152.2, HGCN layers at C3 5.45 -> 5.12 (two-hop 748 -> 725 us, pair sums 768 -> 733, masked pull 166 -> 148); 1,024 is 2 % slower than 2,048, 65,536 no better than 16,384.
#ifndef IHG_AGG_MAX_BLOCKS
#define IHG_AGG_MAX_BLOCKS (256 * 64)
#endif
inline int agg_grid(int64_t waves) {
    int64_t blocks = (waves + kWavesPerBlock - 1) / kWavesPerBlock;
    if (blocks < 1) blocks = 1;
    if (blocks > IHG_AGG_MAX_BLOCKS) blocks = IHG_AGG_MAX_BLOCKS;
    return static_cast<int>(blocks);
}

// Smallest power of two >= n, clamped to [4, 64].
inline int group_lanes(int n) {
    int g = 4;
    while (g < n && g < kWave) g <<= 1;
    return g;
}

template <int VEC>
int launch_edge_gather_sum(const float* src, int64_t ld_src, const int32_t* i3, const float* node_scale, const float* bias,
                           float alpha, const float* edge_scale, float* out, int64_t ld_out, int64_t n_edges, int dim, hipStream_t stream) {
    const int dim_vec = dim / VEC;
    const int g = group_lanes(dim_vec);
#define IHG_LAUNCH_K5(G, U)                                                                                         \
    {                                                                                                               \
        constexpr int EPW = (kWave / G) * U;                                                                        \
        const int grid = agg_grid((n_edges + EPW - 1) / EPW);                                                 \
        hipLaunchKernelGGL((edge_gather_sum_kernel<VEC, G, U>), dim3(grid), dim3(kBlockThreads), 0, stream, src,    \
                           ld_src, i3, node_scale, bias, alpha, edge_scale, out, ld_out, n_edges, dim_vec);         \
    }
    switch (g) {
        case 4: IHG_LAUNCH_K5(4, 1) break;
        case 8: IHG_LAUNCH_K5(8, 2) break;
        case 16: IHG_LAUNCH_K5(16, IHG_K5_U16) break;
        case 32: IHG_LAUNCH_K5(32, IHG_K5_U32) break;
        default: IHG_LAUNCH_K5(64, IHG_K5_U64) break;
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
    const uint8_t* src_mask;      // optional: source rows with a 0 here are known to be zero and are not fetched
};

template <int VEC, int G>
void launch_segment_sum_g(const float* src, int64_t ld_src, const int32_t* rowptr, const int32_t* ids, const int32_t* row_order,
                          const float* src_scale, const float* entry_scale, const float* out_scale, int mode, float* out, int64_t ld_out, int64_t n_rows, int dim,
                          int heavy_threshold, const HeavyPlan& hp, const float* self_weight, hipStream_t stream) {
    constexpr int GPW = kWave / G;
    const int dim_vec = dim / VEC;
    const int grid = agg_grid((n_rows + hp.n_segments + GPW - 1) / GPW);
    if (hp.src_mask != nullptr)                               // the masked pull: its own instance (see accumulate_list)
        hipLaunchKernelGGL((node_segment_sum_kernel<VEC, G, true>), dim3(grid), dim3(kBlockThreads), 0, stream, src, ld_src, rowptr, ids, row_order,
                           src_scale, entry_scale, out_scale, mode, out, ld_out, n_rows, dim, dim_vec, heavy_threshold, hp.seg_begin, hp.seg_end,
                           hp.n_segments, hp.partials, self_weight, hp.src_mask);
    else if (mode & IHG_SRC_READ_ONCE)
        hipLaunchKernelGGL((node_segment_sum_kernel<VEC, G, false, true>), dim3(grid), dim3(kBlockThreads), 0, stream, src, ld_src, rowptr, ids, row_order,
                           src_scale, entry_scale, out_scale, mode, out, ld_out, n_rows, dim, dim_vec, heavy_threshold, hp.seg_begin, hp.seg_end,
                           hp.n_segments, hp.partials, self_weight, hp.src_mask);
    else
        hipLaunchKernelGGL((node_segment_sum_kernel<VEC, G>), dim3(grid), dim3(kBlockThreads), 0, stream, src, ld_src, rowptr, ids, row_order,
                           src_scale, entry_scale, out_scale, mode, out, ld_out, n_rows, dim, dim_vec, heavy_threshold, hp.seg_begin, hp.seg_end,
                           hp.n_segments, hp.partials, self_weight, hp.src_mask);
    if (hp.n_heavy > 0)
        hipLaunchKernelGGL((heavy_finish_kernel<VEC, G>), dim3(static_cast<int>(std::min<int64_t>(hp.n_heavy, kMaxBlocks * 4))),
                           dim3(kBlockThreads), 0, stream, hp.partials, hp.heavy_rows, hp.heavy_segptr, hp.n_heavy, out_scale, mode, out,
                           ld_out, dim, dim_vec, src, ld_src, src_scale, self_weight, hp.src_mask);
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
    if (mode & ~(0xff | IHG_SCALE_ACCUMULATE | IHG_SRC_READ_ONCE)) return false;
    mode &= 0xff;
    if (mode == IHG_SCALE_NONE) return true;
    return (mode == IHG_SCALE_MULTIPLY || mode == IHG_SCALE_DIVIDE) && scale != nullptr;
}

}  // namespace

extern "C" {

int ihg_edge_gather_sum(const float* src, int64_t ld_src, const int32_t* i3, const float* node_scale, const float* bias,
                        float alpha, const float* edge_scale, float* out, int64_t ld_out, int64_t n_edges, int32_t dim, ihg_stream_t stream) {
    if (n_edges < 0 || dim <= 0 || ld_src < dim || ld_out < dim) return fail(IHG_ERR_INVALID, "ihg_edge_gather_sum: bad size (E=%lld dim=%d ld_src=%lld ld_out=%lld)", (long long)n_edges, dim, (long long)ld_src, (long long)ld_out);
    if (n_edges == 0) return IHG_OK;
    if (src == nullptr || i3 == nullptr || out == nullptr) return fail(IHG_ERR_INVALID, "ihg_edge_gather_sum: null pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool wide = dim % 4 == 0 && ld_src % 4 == 0 && ld_out % 4 == 0 && aligned16(src) && aligned16(out) && (bias == nullptr || aligned16(bias));
    return wide ? launch_edge_gather_sum<4>(src, ld_src, i3, node_scale, bias, alpha, edge_scale, out, ld_out, n_edges, dim, s)
                : launch_edge_gather_sum<1>(src, ld_src, i3, node_scale, bias, alpha, edge_scale, out, ld_out, n_edges, dim, s);
}

int32_t ihg_edge_gather_sum_planes_supported(int32_t dim, int64_t ld_src) { return dim == 256 && ld_src >= dim && ld_src % 4 == 0 ? 1 : 0; }

int ihg_edge_gather_sum_planes(const float* src, int64_t ld_src, const int32_t* i3, const float* node_scale, const float* edge_scale, void* planes, float* inv_scale,
                               int64_t n_edges, int32_t dim, ihg_stream_t stream) {
    if (!ihg_edge_gather_sum_planes_supported(dim, ld_src)) return fail(IHG_ERR_INVALID, "ihg_edge_gather_sum_planes: shape not supported (ask ihg_edge_gather_sum_planes_supported)");
    if (n_edges < 0) return fail(IHG_ERR_INVALID, "ihg_edge_gather_sum_planes: bad size");
    if (n_edges == 0) return IHG_OK;
    if (src == nullptr || i3 == nullptr || planes == nullptr || inv_scale == nullptr || !aligned16(src) || !aligned16(planes))
        return fail(IHG_ERR_INVALID, "ihg_edge_gather_sum_planes: null or unaligned pointer");
    constexpr int U = IHG_K5_UPLANES;
    const int grid = agg_grid((n_edges + U - 1) / U);
    hipLaunchKernelGGL((edge_gather_sum_planes256_kernel<U>), dim3(grid), dim3(kBlockThreads), 0, static_cast<hipStream_t>(stream), src, ld_src, i3, node_scale,
                       edge_scale, static_cast<v2u*>(planes), inv_scale, n_edges);
    return check_launch("ihg_edge_gather_sum_planes");
}

int ihg_node_segment_sum(const float* src, int64_t ld_src, const int32_t* rowptr, const int32_t* ids, const int32_t* row_order,
                         const float* src_scale, const float* entry_scale, const float* out_scale, int32_t out_scale_mode, float* out, int64_t ld_out,
                         int64_t n_rows, int32_t dim, int32_t heavy_threshold, const int32_t* seg_begin, const int32_t* seg_end,
                         int64_t n_segments, const int32_t* heavy_rows, const int32_t* heavy_segptr, int64_t n_heavy, float* partials,
                         const float* self_weight, const uint8_t* src_mask, ihg_stream_t stream) {
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
    const HeavyPlan hp{seg_begin, seg_end, n_segments, heavy_rows, heavy_segptr, n_heavy, partials, src_mask};
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool wide = dim % 4 == 0 && ld_src % 4 == 0 && ld_out % 4 == 0 && aligned16(src) && aligned16(out) && (n_heavy == 0 || aligned16(partials));
    return wide ? launch_segment_sum<4>(src, ld_src, rowptr, ids, row_order, src_scale, entry_scale, out_scale, out_scale_mode, out, ld_out, n_rows, dim, heavy_threshold, hp, self_weight, s)
                : launch_segment_sum<1>(src, ld_src, rowptr, ids, row_order, src_scale, entry_scale, out_scale, out_scale_mode, out, ld_out, n_rows, dim, heavy_threshold, hp, self_weight, s);
}

int ihg_node_pair_sums(const float* h, int64_t ld_h, const int32_t* pair_ptr, const int32_t* pair_ids, const int32_t* row_order, float* out,
                       int64_t ld_out, int64_t n_rows, int32_t dim, int32_t heavy_threshold, const int32_t* seg_begin, const int32_t* seg_end,
                       int64_t n_segments, const int32_t* heavy_rows, const int32_t* heavy_segptr, int64_t n_heavy, float* partials,
                       const float* pair_weight, ihg_stream_t stream) {
    if (n_rows < 0 || dim <= 0 || dim % 4 != 0 || ld_h < dim || ld_h % 4 != 0 || ld_out < 3 * static_cast<int64_t>(dim) || ld_out % 4 != 0 || n_segments < 0 || n_heavy < 0)
        return fail(IHG_ERR_INVALID, "ihg_node_pair_sums: bad size (rows=%lld dim=%d ld_h=%lld ld_out=%lld)", (long long)n_rows, dim, (long long)ld_h, (long long)ld_out);
    if (n_rows == 0) return IHG_OK;
    if (h == nullptr || pair_ptr == nullptr || pair_ids == nullptr || out == nullptr || !aligned16(h) || !aligned16(out))
        return fail(IHG_ERR_INVALID, "ihg_node_pair_sums: null or unaligned pointer");
    if (n_heavy > 0 && (heavy_threshold <= 0 || seg_begin == nullptr || seg_end == nullptr || heavy_rows == nullptr || heavy_segptr == nullptr || partials == nullptr || !aligned16(partials)))
        return fail(IHG_ERR_INVALID, "ihg_node_pair_sums: incomplete split-row plan");
    if (n_heavy == 0) {
        n_segments = 0;
        heavy_threshold = 0;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int dim_vec = dim / 4;
#define IHG_PAIRS(G)                                                                                                                        \
    {                                                                                                                                       \
        constexpr int GPW = kWave / G;                                                                                                      \
        const int grid = agg_grid((n_rows + n_segments + GPW - 1) / GPW);                                                             \
        if (pair_weight != nullptr)                                                                                                         \
            hipLaunchKernelGGL((node_pair_sums_kernel<G, true>), dim3(grid), dim3(kBlockThreads), 0, s, h, ld_h, pair_ptr, pair_ids, row_order, out, \
                               ld_out, n_rows, dim, dim_vec, heavy_threshold, seg_begin, seg_end, n_segments, partials, pair_weight);       \
        else                                                                                                                                \
            hipLaunchKernelGGL((node_pair_sums_kernel<G>), dim3(grid), dim3(kBlockThreads), 0, s, h, ld_h, pair_ptr, pair_ids, row_order, out, \
                               ld_out, n_rows, dim, dim_vec, heavy_threshold, seg_begin, seg_end, n_segments, partials, pair_weight);       \
        if (n_heavy > 0)                                                                                                                    \
            hipLaunchKernelGGL((heavy_finish_kernel<4, G>), dim3(static_cast<int>(std::min<int64_t>(n_heavy, kMaxBlocks * 4)), (3 * dim_vec + G - 1) / G), \
                               dim3(kBlockThreads), 0, s, partials, heavy_rows, heavy_segptr, n_heavy, nullptr, IHG_SCALE_NONE, out, ld_out, \
                               3 * dim, 3 * dim_vec, nullptr, 0, nullptr, nullptr);                                                         \
    }
    switch (group_lanes(dim_vec)) {
        case 4: IHG_PAIRS(4) break;
        case 8: IHG_PAIRS(8) break;
        case 16: IHG_PAIRS(16) break;
        case 32: IHG_PAIRS(32) break;
        default: IHG_PAIRS(64) break;
    }
#undef IHG_PAIRS
    return check_launch("ihg_node_pair_sums");
}

int ihg_bag_mean_fwd(const float* table, int64_t ld_table, const int32_t* bag_ptr, const int32_t* words, const float* bag_len,
                     float* out, int64_t ld_out, int64_t n_bags, int32_t dim, ihg_stream_t stream) {
    if (bag_len == nullptr && n_bags > 0) return fail(IHG_ERR_INVALID, "ihg_bag_mean_fwd: null bag_len");
    return ihg_node_segment_sum(table, ld_table, bag_ptr, words, nullptr, nullptr, nullptr, bag_len, IHG_SCALE_DIVIDE, out, ld_out, n_bags, dim, 0, nullptr, nullptr, 0, nullptr, nullptr, 0, nullptr, nullptr, nullptr, stream);
}

int ihg_bag_mean_bwd(const float* dout, int64_t ld_dout, const int32_t* word_ptr, const int32_t* word_bags, const float* inv_len,
                     float* dtable, int64_t ld_dtable, int64_t n_table_rows, int32_t dim, ihg_stream_t stream) {
    if (inv_len == nullptr && n_table_rows > 0) return fail(IHG_ERR_INVALID, "ihg_bag_mean_bwd: null inv_len");
    return ihg_node_segment_sum(dout, ld_dout, word_ptr, word_bags, nullptr, inv_len, nullptr, nullptr, IHG_SCALE_NONE, dtable, ld_dtable, n_table_rows, dim, 0, nullptr, nullptr, 0, nullptr, nullptr, 0, nullptr, nullptr, nullptr, stream);
}
}  // extern "C"
