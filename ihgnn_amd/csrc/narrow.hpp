// narrow.hpp - internal interface of narrow.hip: the interactive layer at the NARROW width d = 32 (the reference's default, Helpers/GlobalSettings.py:30 `embedding_size = 32`)
// in the forms rounds 3 - 4 built for d = 64 / 128 / 256: the layer's forward and the product blocks' weight gradients at node level (no [E, d] rows) and the gathering,
// user-reduced member-gradient kernel.  At this width every one of them is bound by its row traffic, not by the matrix pipe (a hyperedge's contraction is 4 x 32 x 32
// multiply-adds beside ~ 1 KB of gathered rows), so the arithmetic is plain fp32 on v_mfma_f32_16x16x4_f32 - exact products, no operand split, the same kernels under
// IHG_INTERACT_ARITH=f32 - and the design is the memory side: one wave = one tile of 16 rows, whole 64-byte row halves per load instruction, weights resident in registers,
// rows requested a tile ahead.  Not part of the C ABI: interact.hip dispatches here through the *_ok() predicates.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "common.hpp"

#ifndef IHG_INTERNAL
#define IHG_INTERNAL __attribute__((visibility("hidden")))
#endif

constexpr int kNarrowDim = 32;
constexpr int kNarrowMemberRanges = 2048;      // contiguous tile ranges of the member-gradient kernel (one per wave: 256 CUs x 8 waves); two boundary entries each
constexpr int kNarrowWeightRanges = 512;       // row ranges (= slabs) of the node-level weight-gradient kernel

// node-level forward: out = scale * (sum over a node's hyperedges of their features) from h and the pair sums; packed: narrow_node_fwd_floats() floats of workspace
IHG_INTERNAL int64_t narrow_node_fwd_floats();
IHG_INTERNAL bool narrow_node_fwd_ok(int dim, int order, int64_t ld_h, int64_t ld_s, const float* out, int64_t ld_out, const float* bias);
IHG_INTERNAL void launch_node_fwd_narrow(int order, const float* h, int64_t ld_h, const float* sums, int64_t ld_s, const float* deg, const float* scale, const float* bias,
                                         const float* w, int64_t ld_w, const int64_t* type_begin, float* out, int64_t ld_out, float* packed, hipStream_t s);

// node-level weight gradients of the product blocks: dw[:, 3 d ..]; slabs: narrow_node_weight_floats(order) floats of workspace
IHG_INTERNAL int64_t narrow_node_weight_floats(int order);
IHG_INTERNAL bool narrow_node_weight_ok(int dim, int order, int64_t ld_h, int64_t ld_s, int64_t ld_dy, const float* dy);
IHG_INTERNAL void launch_node_weight_narrow(int order, const float* h, int64_t ld_h, const float* sums, int64_t ld_s, const float* dy, int64_t ld_dy, const float* dy_scale,
                                            const int64_t* type_begin, float* slabs, float* dw, int64_t ld_dw, hipStream_t s);

// member gradients, user slot reduced on chip (g2 is [E, 2, d]; hyperedges numbered by user).  dy_scale / dout_store as in launch_members_split: with gather != 0 `dsrc` is
// the node-level cotangent [N, d] and the kernel forms the hyperedges' cotangents itself (dout_store != nullptr: and leaves them there).  packed: narrow_members_floats(order)
// floats of workspace; the boundary table holds 2 * kNarrowMemberRanges entries; *n_boundary_entries = entries written.  The boundary runs are added up by the launch itself.
IHG_INTERNAL int64_t narrow_members_floats(int order);
IHG_INTERNAL bool narrow_members_ok(int dim, int order, const float* g2, int64_t ld_h, int64_t ld_d, const float* dsrc);
IHG_INTERNAL void launch_members_narrow(int order, int gather, const float* h, int64_t ld_h, const int32_t* i3, const float* w, int64_t ld_w, float* packed, const float* dsrc,
                                        int64_t ld_d, const float* dy_scale, float* dout_store, int64_t ld_store, float* g2, int64_t n_edges, float* dh_user, int64_t ld_dh,
                                        float* bnd_val, int32_t* bnd_user, int* n_boundary_entries, hipStream_t s);

// node-level linear maps (ihg_node_linear_*) at dim 32 and 64: forward / input gradient as one stream over the rows; weight + bias gradient (and, dx != nullptr, the input gradient
// of the same rows) in one pass, slabs in dense.hip's layout (returns the slabs per type); pk: 3 dim^2 floats of workspace for the packed weights
IHG_INTERNAL bool narrow_linear_ok(int dim, int64_t ld_a, int64_t ld_b);
IHG_INTERNAL void launch_row_gemm_narrow(int dim, TypedRows in, int64_t ld_in, const float* w, int64_t ld_w, int64_t w_type_stride, int transpose, const float* bias, int bias_mask,
                                         int64_t bias_type_stride, const int64_t* type_begin, TypedRowsOut out, int64_t ld_out, int accumulate, float* pk, hipStream_t s);
IHG_INTERNAL int launch_dense_weight_narrow(int dim, const float* dout, int64_t ld_dout, TypedRows x, int64_t ld_x, const int64_t* type_begin, int n_types, float* slabs, float* bias_slabs,
                                            const float* w, int64_t ld_w, int64_t w_type_stride, const TypedRowsOut* dx, int64_t ld_dx, int dx_accumulate, float* pk, hipStream_t s);
