// split.hpp - internal interface of split_arith.hip (contractions per hyperedge) and split_node.hip (node-level contractions and linear maps): the interactive
// step's contractions (orders 2 and 3) and the node-level linear maps with every fp32 operand taken apart into 16-bit terms and the partial products accumulated
// in fp32 on the 16-bit matrix pipe.  Two schemes (split_common.hpp): TWO fp16 terms under a per-row / per-column power of two, round to nearest, three
// v_mfma_f32_16x16x32_f16 products per multiply (what is left out is <= 3 x 2^-22 of a product) - the default path: node-level contraction and its weight
// gradients, member gradients, node-level linear maps; and THREE bf16 terms, an exact split by truncation (x = hi + mid + lo), six v_mfma_f32_16x16x32_bf16
// products (<= 2^-21 left out) - the hyperedge form's forward and weight-gradient kernels.  Not part of the C ABI: interact.hip and dense.hip choose between
// these and their fp32-MFMA kernels through the *_ok() predicates.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "common.hpp"

#define IHG_INTERNAL __attribute__((visibility("hidden")))

// floats of workspace the weight planes of one direction take (sized for three bf16 per weight; the two-fp16 kernels use two thirds of it)
IHG_INTERNAL int64_t split_plane_floats(int dim, int order);
IHG_INTERNAL bool split_arith_enabled();                       // IHG_INTERACT_ARITH != "f32"
IHG_INTERNAL bool split_members_ok(int dim, int order, const float* g, int64_t ld_h, int64_t ld_dout, const float* dout, bool user_reduced);

// member gradients; dh_user == nullptr: g is [E, 3, d], else the user-reduced form (g is [E, 2, d], boundary table as in interact.hip).
// dout_store != nullptr (dim 128, user-reduced form): `dout` is a node-level cotangent [N, d]; the kernel forms dout[e] = sum over the members
// of dy_scale[m] dout[m] (dy_scale == nullptr: 1) itself and leaves it in dout_store [E, ld_store]
IHG_INTERNAL void launch_members_split(int dim, int order, const float* h, int64_t ld_h, const int32_t* i3, const float* w, int64_t ld_w, void* planes, const float* dout,
                                       int64_t ld_dout, float* g, int64_t n_edges, float* dh_user, int64_t ld_dh, float* bnd_val, int32_t* bnd_user,
                                       int* n_boundary_entries, hipStream_t s, const float* dy_scale = nullptr, float* dout_store = nullptr, int64_t ld_store = 0,
                                       const float* inv_src = nullptr);     // inv_src != nullptr (dim 256): `dout` rows are fp16 planes with these inverse scales (ihg_edge_gather_sum_planes)

// forward (first-order rows p required); planes: split_plane_floats(dim, order) floats of workspace
IHG_INTERNAL bool split_fwd_ok(int dim, int order, const float* p, int64_t ld_p, const float* out, int64_t ld_out, int64_t ld_h);
IHG_INTERNAL void launch_fwd_split(int dim, int order, const float* h, int64_t ld_h, const float* p, int64_t ld_p, const int32_t* i3, const float* w, int64_t ld_w, void* planes, float* out,
                                   int64_t ld_out, int64_t n_edges, hipStream_t s);

// weight gradients into slabs [range][j][b d + c] (interact.hip's slab layout); returns the number of slabs written
IHG_INTERNAL bool split_weight_ok(int dim, int order, int64_t ld_h, int64_t ld_dout, const float* dout);
IHG_INTERNAL int launch_weight_split(int dim, int order, const float* h, int64_t ld_h, const int32_t* i3, const float* dout, int64_t ld_dout, float* slabs, int64_t n_edges, hipStream_t s);

// node-level row GEMM (d = 128, 256): out = in W_t^T (transpose == 0) or in W_t (transpose == 1), rows grouped by node type
IHG_INTERNAL int64_t split_dense_plane_floats(int dim);
IHG_INTERNAL bool split_row_gemm_ok(int dim, const float* out, int64_t ld_out, const float* bias, int64_t bias_type_stride);
IHG_INTERNAL void launch_row_gemm_split(int dim, TypedRows in, int64_t ld_in, const float* w, int64_t ld_w, int64_t w_type_stride, int transpose, const float* bias,
                                        int bias_mask, int64_t bias_type_stride, const int64_t* type_begin, TypedRowsOut out, int64_t ld_out, void* planes, hipStream_t s,
                                        int accumulate = 0);      // accumulate: out += (out holds another contribution to the same rows)

// weight / bias gradient of the node-level linear maps into dense.hip's slabs ([type][slab][d][d], [type][slab][d]); returns the slabs per type
IHG_INTERNAL bool split_dense_weight_ok(int dim, const float* dout, int64_t ld_dout, const float* x, int64_t ld_x);
IHG_INTERNAL int launch_dense_weight_split(int dim, const float* dout, int64_t ld_dout, TypedRows x, int64_t ld_x, const int64_t* type_begin, int n_types,
                                           float* slabs, float* bias_slabs, const float* w, int64_t ld_w, int64_t w_type_stride, const TypedRowsOut* dx, int64_t ld_dx,
                                           void* planes, hipStream_t s, int dx_accumulate = 0);   // dx != nullptr (dim 128, 16-byte aligned rows): dx (+)= dout W_t of the same rows, fused

// node-level form of the interactive layer's forward (d = 64 / 128 / 256): out = scale * (sum over the node's hyperedges of their features) from h and the
// pair sums of ihg_node_pair_sums; planes: split_node_fwd_plane_floats(dim) floats of workspace
IHG_INTERNAL int64_t split_node_fwd_plane_floats(int dim);
IHG_INTERNAL bool split_node_fwd_ok(int dim, int order, int64_t ld_h, int64_t ld_s, const float* out, int64_t ld_out, const float* bias);
IHG_INTERNAL void launch_node_fwd_split(int dim, int order, const float* h, int64_t ld_h, const float* sums, int64_t ld_s, const float* deg, const float* scale, const float* bias,
                                        const float* w, int64_t ld_w, const int64_t* type_begin, float* out, int64_t ld_out, void* planes, hipStream_t s);

// node-level weight gradients of the product blocks (d = 64 / 128 / 256): dw[:, 3 d ..] from h, the pair sums and the node-level cotangent; slabs:
// split_node_weight_slab_floats(dim, order) floats of workspace
IHG_INTERNAL int64_t split_node_weight_slab_floats(int dim, int order);
IHG_INTERNAL bool split_node_weight_ok(int dim, int order, int64_t ld_h, int64_t ld_s, int64_t ld_dy, const float* dy);
IHG_INTERNAL void launch_node_weight_split(int dim, int order, const float* h, int64_t ld_h, const float* sums, int64_t ld_s, const float* dy, int64_t ld_dy, const float* dy_scale,
                                           const int64_t* type_begin, float* slabs, float* dw, int64_t ld_dw, hipStream_t s);
