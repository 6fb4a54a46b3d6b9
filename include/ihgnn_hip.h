/*
 * ihgnn_hip.h - C ABI of libihgnn_hip.so: the MI355X (gfx950) hypergraph message-passing path of IHGNN.
 *
 * This is the drop-in boundary.  The reference reaches this path through PyTorch
 * modules plus one third-party operator (torch_sparse.matmul); each entry point below names the
 * reference interface it replaces (paths relative to the CDboyOne/IHGNN checkout).
 *
 * Conventions for every device entry point:
 *   - plain pointers and sizes only; all device buffers are caller-owned, row-major fp32 / int32;
 *   - `ld_*` is the row stride in ELEMENTS (lets a layer read/write a column slice of the [N, D] feature
 *     matrix in place);
 *   - `stream` is a hipStream_t passed as void*; kernels are enqueued on it and the call returns at once:
 *     no allocation, no synchronisation, no host<->device copies (graph-capturable);
 *   - return value: 0 on success, negative IHG_ERR_* otherwise; ihg_last_error_string() describes the
 *     last failure on the calling thread.
 *   - Rows whose base address and stride are 16-byte aligned with dim % 4 == 0 take the 16-B/lane path;
 *     anything else is still computed (4-B/lane path), never rejected.
 */
#ifndef IHGNN_HIP_H
#define IHGNN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IHG_OK                0
#define IHG_ERR_INVALID      -1   /* bad argument (null pointer, negative size, unsupported order, ...) */
#define IHG_ERR_LAUNCH       -2   /* HIP runtime reported an error at launch */
#define IHG_ERR_WORKSPACE    -3   /* caller-provided workspace too small */

#define IHG_SCALE_NONE        0
#define IHG_SCALE_MULTIPLY    1   /* out[r] = scale[r] * sum   (Dv^-1, Dv^-1/2) */
#define IHG_SCALE_DIVIDE      2   /* out[r] = sum / scale[r]   (EmbeddingBag 'mean': scale = bag length) */
#define IHG_SRC_READ_ONCE     0x400 /* OR-ed into out_scale_mode of ihg_node_segment_sum: every source row is read exactly once by this launch (a scatter of per-member rows) - non-temporal loads */
#define IHG_SCALE_ACCUMULATE  0x100 /* OR-ed into out_scale_mode of ihg_node_segment_sum: out[r] += (scaled) sum - the hyperedge chunks of one scatter add up in `out` */

typedef void* ihg_stream_t;       /* hipStream_t */

/* Version of this ABI; bumped on any signature change. */
int32_t ihg_abi_version(void);

/* 1 when the library was built with an ablation switch of csrc/ablate.hpp (tools/ab_variant.sh ... -DIHG_ABL_*: a kernel with one class of its work removed, wrong
 * results on purpose, for timing only); the product build returns 0 and the binding refuses anything else unless IHG_ALLOW_ABLATION_BUILD=1. */
int32_t ihg_ablation_build(void);

/* Text of the last error raised on this thread ("" if none). */
const char* ihg_last_error_string(void);

/* ------------------------------------------------------------------------------------------------
 * HOST: hypergraph incidence layout.   Replaces PpsHyperGraph.from_interactions
 * (Helpers/Graph.py:94-134) and torch_sparse's SparseTensor...coalesce() (Models/GnnLayers.py:190).
 *
 * triples[e] = (user, query, item), 0-based per type, in file order (one hyperedge per positive
 * interaction; duplicates stay distinct).  Outputs, all host buffers sized by the caller:
 *   i3       [E,3]  global node ids  u, q+U, i+U+Q                     (Graph.py:110-111,117,129)
 *   rowptr   [N+1]  node-major CSR offsets; edge_ids [3E] hyperedge ids, ascending inside a node
 *                   (= the coalesced COO order of Graph.py:123-128)
 *   degree   [N]    incident hyperedges, 0 replaced by 1e-8             (Graph.py:112,120)
 * Returns IHG_ERR_INVALID if a triple is out of range.
 */
int ihg_build_csr(const int64_t* triples, int64_t n_edges,
                  int64_t n_users, int64_t n_queries, int64_t n_items,
                  int32_t* i3, int32_t* rowptr, int32_t* edge_ids, float* degree);

/* HOST: search-log CSV ingestion.  Replaces SearchLogCollection.read + SearchLog.parse + the positive / negative split
 * of GraphDataset.__init__ (Helpers/SearchLogCollection.py:25-32, Helpers/SearchLog.py:63-71, Dataset.py:195-213) for the
 * quantities the graph needs.  File: one header line, then rows `user,query,search_time,items,pages,positions,
 * interactions,times` whose list fields are space-separated; item k of a row is a positive interaction when
 * interactions[k] > 0.  Pass pos == neg == NULL to count (n_logs / n_pos / n_neg), then call again with buffers of
 * [n_pos,3] / [n_neg,3] int64 to receive the (user, query, item) triples in file order.  neg may stay NULL.
 * pos_log (may be NULL; [n_pos] int64): the 0-based row of each positive - rows with several positives are the
 * variable-arity hyperedges of ihg_build_log_hypergraph.
 */
int ihg_parse_search_logs(const char* path, int64_t* n_logs, int64_t* n_pos, int64_t* n_neg,
                          int64_t* pos, int64_t pos_capacity, int64_t* neg, int64_t neg_capacity, int64_t* pos_log);

/* HOST: `graph_info.txt` (one line "users queries items vocabulary", Dataset.py:143-147) -> counts[4]. */
int ihg_read_graph_info(const char* path, int64_t* counts);

/* HOST: `queries_multihot.txt` (line r = space-separated 0-based word ids of query r, Dataset.py:165-176; an empty line is a
 * query without words).  Two-pass like the CSV parser: offsets == words == NULL counts; then offsets[n_queries] = start of each
 * query's words in words[n_words] (the nn.EmbeddingBag input / offsets of Dataset.py:161-186, before the +1 padding shift).
 */
int ihg_read_query_bags(const char* path, int64_t* n_queries, int64_t* n_words, int64_t* offsets, int64_t offsets_capacity,
                        int64_t* words, int64_t words_capacity);

/* HOST: per-search-log hypergraph.  Replaces PpsLogHyperGraph.from_search_logs (Helpers/Graph.py:138-189): one hyperedge of
 * VARIABLE arity per search log with >= 1 positive, members [user, query + U, positive items + U + Q].  Input: the positives
 * and their row numbers as ihg_parse_search_logs returns them (positives of one log consecutive).  Output, caller-sized
 * (n_edges <= n_pos, nnz <= 3 n_pos): the incidence in BOTH orientations as CSR with values -
 *   edge_ptr [n_edges+1], edge_nodes / edge_vals [nnz]   edge-major (members ascending; a repeated item is one entry of value 2,
 *                                                        as coalesce() sums duplicates, Graph.py:178-184)
 *   node_ptr [N+1], node_edges / node_vals [nnz]         node-major, hyperedge ids ascending (= the coalesced COO order)
 *   edge_degree [n_edges] = len(members) with repeats (Graph.py:167), node_degree [N] = hyperedges containing the node, 0 -> 1e-8
 *   (Graph.py:166,171).
 */
int ihg_build_log_hypergraph(const int64_t* pos, const int64_t* pos_log, int64_t n_pos, int64_t n_users, int64_t n_queries, int64_t n_items,
                             int32_t* edge_ptr, int32_t* edge_nodes, float* edge_vals, float* edge_degree, int32_t* node_ptr,
                             int32_t* node_edges, float* node_vals, float* node_degree, int64_t* n_edges_out, int64_t* nnz_out);

/* HOST: pairwise graph of the GCN baseline.  Replaces Pps2DGraph.from_interactions (Helpers/Graph.py:19-81) and the
 * coalesce() of its adjacency.  completeness: 0 = uqi (u-q, q-i, i-u per interaction), 1 = uq, 2 = ui, 3 = qi
 * (Graph.py:40-63; every pair is stored in both directions).  Output: CSR of the symmetric [N x N] adjacency with
 * duplicate pairs SUMMED into `vals`, `degree` = entries per node before coalescing (+1 with self loops), 0 -> 1e-8
 * without self loops (Graph.py:68-69).  `cols`/`vals` must hold `capacity` >= 6E (+N) entries; *nnz_out = entries used.
 */
int ihg_build_pair_csr(const int64_t* triples, int64_t n_edges, int64_t n_users, int64_t n_queries, int64_t n_items,
                       int32_t completeness, int32_t self_loops,
                       int32_t* rowptr, int32_t* cols, float* vals, float* degree, int64_t capacity, int64_t* nnz_out);

/* HOST: merge the repeated ids of every row of a CSR: row r of (out_ptr, out_ids, out_counts) holds the DISTINCT ids of row r of (ptr, ids) in ascending
 * order with their multiplicities as floats (exact: a count) - or, with `weights` (one float per input id), the SUM of the weights of each distinct id.
 * Applied to the two-hop list of the hypergraph (node -> the other two members of each of
 * its hyperedges, Helpers/Graph.py:107-118 composed with itself as Models/GnnLayers.py:233-234 does through two SpMMs) it gives H H^T - diag(deg) as a
 * weighted CSR: a (user, query) pair that co-occurs in k hyperedges is ONE entry of weight k instead of k gathers (duplicate hyperedges stay distinct
 * hyperedges: the multiplicities carry them).  out_ids / out_counts hold ptr[n_rows] entries; *nnz_out = entries used.  out_ids == NULL: count only
 * (out_ptr and *nnz_out are written).
 */
int ihg_merge_id_lists(const int32_t* ptr, const int32_t* ids, const float* weights, int64_t n_rows, int32_t* out_ptr, int32_t* out_ids, float* out_counts,
                       int64_t* nnz_out);

/* HOST: the DISTINCT (user, query, item) triples of `triples` [n_edges, 3] (0-based per type, as ihg_build_csr takes them), ascending by (user, query, item), with
 * how often each occurs.  The reference makes one hyperedge per positive interaction, duplicates included (Helpers/Graph.py:107-118); every quantity the
 * path computes from duplicates - vertex degrees, H Ef, H^T of a cotangent - is the distinct hyperedge's value times its multiplicity, so a layout may keep each triple
 * once and carry count_out as a per-hyperedge weight (IncidenceLayout.edge_weight): same results, a fraction of the rows on search logs that repeat.
 *   unique_out [n_edges, 3] (capacity), count_out [n_edges], file_to_unique [n_edges] (optional: index of the distinct triple of every input row);
 *   *n_unique_out = distinct triples.  unique_out == NULL: count only.
 */
int ihg_unique_triples(const int64_t* triples, int64_t n_edges, int64_t n_users, int64_t n_queries, int64_t n_items,
                       int64_t* unique_out, float* count_out, int32_t* file_to_unique, int64_t* n_unique_out);

/* HOST: invert any CSR (row -> sorted list of column ids) into its transpose.  Used for the
 * EmbeddingBag backward (word -> bags containing it).  Replaces the autograd-generated
 * _embedding_bag_dense_backward of Models/EmbeddingLayers.py:79.
 *   ptr [n_rows+1], ids [nnz]  ->  t_ptr [n_cols+1], t_rows [nnz] (ascending row inside a column)
 */
int ihg_transpose_csr(const int32_t* ptr, const int32_t* ids, int64_t n_rows, int64_t n_cols,
                      int32_t* t_ptr, int32_t* t_rows);

/* ------------------------------------------------------------------------------------------------
 * DEVICE: node -> hyperedge gather-sum (K5, first-order / HGCN form).
 *   out[e,:] = alpha * ( s(i3[e,0])*src[i3[e,0],:] + s(i3[e,1])*src[i3[e,1],:] + s(i3[e,2])*src[i3[e,2],:] ) + bias
 * with s(v) = node_scale[v] (or 1 if node_scale == NULL) and bias optional [dim].
 * Replaces: node_features[I3] gather + Linear of FeatureInteractor order 1 (Models/CommonLayers.py:60-66,
 * after hoisting the Linear to node level), thsp.matmul(incidence_t, x) * De^-1 of HGCNLayer
 * (Models/GnnLayers.py:148-149), and the autograd backward of thsp.matmul(incidence, .) (GnnLayers.py:233).
 * edge_scale (optional, [n_edges]): the sum of hyperedge e is multiplied by alpha * edge_scale[e] - a layout that keeps each DISTINCT (user, query, item) triple once
 * passes the triple's multiplicity here where the result is the cotangent of all its copies (the backward of the hyperedge -> node pass; duplicates are distinct
 * hyperedges in the reference, Helpers/Graph.py:107-118, and m identical rows of H sum to m times one of them).
 */
int ihg_edge_gather_sum(const float* src, int64_t ld_src, const int32_t* i3,
                        const float* node_scale, const float* bias, float alpha, const float* edge_scale,
                        float* out, int64_t ld_out, int64_t n_edges, int32_t dim, ihg_stream_t stream);

/* The same sum (alpha = 1, no bias) written as the member-gradient kernel's operand instead of fp32 rows (the backward of thsp.matmul(incidence, .), GnnLayers.py:233, whose
 * result only ihg_interact_bwd_user_reduced_planes reads): row e of `planes` is [2][dim] fp16 - the row scaled by the power of two that brings its largest magnitude
 * to [2^13, 2^14), as hi = fp16(x) and lo = fp16(x - hi) - and inv_scale[e] the inverse of that power.  4 dim bytes per row, like the fp32 row.  dim 256
 * (ihg_edge_gather_sum_planes_supported).  edge_scale: as in ihg_edge_gather_sum (applied before the row is scaled and split).
 */
int32_t ihg_edge_gather_sum_planes_supported(int32_t dim, int64_t ld_src);
int ihg_edge_gather_sum_planes(const float* src, int64_t ld_src, const int32_t* i3, const float* node_scale, const float* edge_scale,
                               void* planes, float* inv_scale, int64_t n_edges, int32_t dim, ihg_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * DEVICE: hyperedge -> node segment-sum (K7 + K8).
 *   out[r,:] = scale_op( sum_{k in [rowptr[r], rowptr[r+1])} w(ids[k]) * src[ids[k],:] , out_scale[r] )
 * with w(ids[k]) = src_scale[ids[k]] (or 1), times entry_scale[k] when given (values of a weighted CSR: the pairwise
 * adjacency of GCNLayer, Models/GnnLayers.py:37-41).  Rows longer than `heavy_threshold` entries go through the split-row plan below
 * (pass n_heavy = 0 to sum every row with a single lane group).
 * `self_weight` (optional, [n_rows], only for square operators whose ids index the same table as the output rows) adds
 * self_weight[r] * w(r) * src[r,:] to row r before the output scale: the diagonal of a two-hop (node -> hyperedge -> node)
 * operator, whose off-diagonal part is then listed without the row's own id.
 * `row_order` (optional, [n_rows]) is the order in which rows are handed to lane groups - a permutation sorted by
 * decreasing row length keeps the groups of one wave equally busy; NULL = natural order.  Results do not depend on it.
 * It may also be a SUBSET of the rows (n_rows = its length; rowptr is still indexed by row id): only those rows, and the
 * rows of the split-row plan, are written - the last layer of a training step, whose output is read at the batch rows only.
 * `src_mask` (optional, one byte per source row): rows with a 0 are known to be all-zero and are not fetched (the gradient
 * of that last layer's output is zero outside the batch rows).
 * Replaces: thsp.matmul(self.incidence, edge_features) and Dv^-1 * / Dv^-1/2 * (Models/GnnLayers.py:151-152,
 * 233-234), nn.EmbeddingBag(mode='mean') (Models/EmbeddingLayers.py:79, via ihg_bag_mean_fwd), and the
 * index_put(accumulate) backward of the three row gathers (Models/CommonLayers.py:70-72).
 */
int ihg_node_segment_sum(const float* src, int64_t ld_src, const int32_t* rowptr, const int32_t* ids,
                         const int32_t* row_order,
                         const float* src_scale, const float* entry_scale,
                         const float* out_scale, int32_t out_scale_mode,
                         float* out, int64_t ld_out, int64_t n_rows, int32_t dim,
                         int32_t heavy_threshold,
                         const int32_t* seg_begin, const int32_t* seg_end, int64_t n_segments,
                         const int32_t* heavy_rows, const int32_t* heavy_segptr, int64_t n_heavy,
                         float* partials, const float* self_weight, const uint8_t* src_mask, ihg_stream_t stream);
/* Split rows (skewed / power-law degree distributions): the host cuts every row longer than `heavy_threshold` into
 * segments [seg_begin, seg_end) of the `ids` array.  The SAME launch sums every segment with its own lane group into
 * partials[s,:] (workspace, n_segments x dim floats) next to the light rows, then a second small kernel adds each heavy
 * row's partials in a fixed order (bitwise reproducible, no float atomics) and applies out_scale.
 *   heavy_rows [n_heavy] row ids; heavy_segptr [n_heavy+1] offsets into the segment arrays.  n_heavy == 0: no plan.
 */

/* ------------------------------------------------------------------------------------------------
 * DEVICE: query embedding bag, mode='mean' (K2).  Replaces nn.EmbeddingBag(mode='mean') forward/backward,
 * Models/EmbeddingLayers.py:79,100-104.  `words` are row ids into `table` (already +1 shifted, Dataset.py:168).
 *   fwd: out[q,:]  = ( sum_{k in bag q} table[words[k],:] ) / len(q)          (0 for an empty bag)
 *   bwd: dtable[w,:] = sum_{entries k with words[k]==w} dout[bag(k),:] * inv_len[bag(k)]
 *        given the transposed CSR (word_ptr [n_table_rows+1], word_bags [nnz]) from ihg_transpose_csr.
 */
int ihg_bag_mean_fwd(const float* table, int64_t ld_table, const int32_t* bag_ptr, const int32_t* words,
                     const float* bag_len, float* out, int64_t ld_out, int64_t n_bags, int32_t dim,
                     ihg_stream_t stream);
int ihg_bag_mean_bwd(const float* dout, int64_t ld_dout, const int32_t* word_ptr, const int32_t* word_bags,
                     const float* inv_len, float* dtable, int64_t ld_dtable, int64_t n_table_rows, int32_t dim,
                     ihg_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * DEVICE: interactive (order 2 / 3) node -> hyperedge step (K5 + K6 fused).
 * Replaces FeatureInteractor.forward orders 2 and 3 (Models/CommonLayers.py:70-85): the [E, k*d]
 * concatenation is never materialised.
 *
 *   fwd: out[e,:] = p[u,:] + p[q,:] + p[i,:]
 *                 + W_uq (h[u]*h[q]) + W_qi (h[q]*h[i]) + W_iu (h[i]*h[u]) [+ W_uqi (h[u]*h[q]*h[i])]
 *        (u,q,i) = i3[e,:];  w is the reference's aggregation.weight, row-major [dim, k*dim] with
 *        k = 6 (order 2) or 7 (order 3); only its product blocks (columns >= 3*dim) are read here.  The
 *        first-order blocks and the bias are hoisted to node level by the caller:
 *        p[v,:] = W_type(v) h[v,:] (+ bias on user rows), N*d*d flops instead of E*3*d*d.
 *   bwd: given dout [E, dim]:
 *        g[e,s,:]  (s = 0,1,2 for u,q,i; [E,3,dim] contiguous) = d loss / d h[member s of e] through the
 *                   product terms only; the caller finishes with ihg_node_segment_sum per node type.
 *        dw[:, 3*dim:]  = gradient of the product blocks of w (first-order columns are left untouched).
 *        `workspace` must hold ihg_interact_bwd_workspace_bytes(...) bytes (per-workgroup partial dW slabs,
 *        reduced in a fixed order - bitwise reproducible).
 *   Both directions take a caller-owned, 16-byte aligned device `workspace` of ihg_interact_{fwd,bwd}_workspace_bytes
 *   bytes: the product blocks of w re-packed into MFMA fragment order (+ the dW slabs for bwd).  For dim in
 *   {32, 64, 128, 256} with 16-byte aligned rows the contraction runs on the matrix cores in exact fp32
 *   (v_mfma_f32_32x32x2_f32 / 16x16x4_f32); every other shape takes a scalar kernel with the same results.
 *   Orders 2 and 3 at dim 64 / 128 / 256 (order 2 forward: 64 / 128) and the dim-128 / 256 node-level maps of ihg_node_linear_* multiply through three exact bf16
 *   terms per fp32 operand instead - six v_mfma_f32_16x16x32_bf16 products per multiply, fp32 accumulation, an error at or
 *   below the fp32-MFMA kernels' (csrc/split_arith.hip).  The environment variable IHG_INTERACT_ARITH=f32, read at every
 *   call, keeps those shapes on the fp32-MFMA kernels.
 * dw == NULL (here and in ihg_interact_bwd_user_reduced): member gradients only - the caller takes the product blocks' weight gradients from
 * ihg_node_interact_bwd_weight.
 */
int64_t ihg_interact_fwd_workspace_bytes(int64_t n_edges, int32_t dim, int32_t order);

int ihg_interact_fwd(const float* h, int64_t ld_h, const float* p, int64_t ld_p, const int32_t* i3,
                     const float* w, int64_t ld_w, int32_t order,
                     float* out, int64_t ld_out, void* workspace, int64_t workspace_bytes,
                     int64_t n_edges, int32_t dim, ihg_stream_t stream);

int64_t ihg_interact_bwd_workspace_bytes(int64_t n_edges, int32_t dim, int32_t order);

int ihg_interact_bwd(const float* h, int64_t ld_h, const int32_t* i3,
                     const float* w, int64_t ld_w, int32_t order,
                     const float* dout, int64_t ld_dout,
                     float* g, float* dw, int64_t ld_dw,
                     void* workspace, int64_t workspace_bytes,
                     int64_t n_edges, int32_t dim, ihg_stream_t stream);

/* The same backward with the USER slot of the member gradients reduced on chip (hyperedges must be numbered so that i3[:,0] is
 * non-decreasing, i.e. sorted by user, as ihgnn_amd's layout numbers them): g2 is [n_edges, 2, dim] (query slot, item slot) and
 * rows [0, n_users) of dh receive, for every user with hyperedges, the sum of its user-slot gradients in hyperedge order - exactly
 * what ihg_node_segment_sum over the user's incidence list would have produced from the [n_edges, 3, dim] buffer.  Rows of users
 * without hyperedges are not written (pre-zero them).  A third less written here and a third less read by the K7 pass over g2.
 * Available where ihg_interact_bwd_user_reduced_supported says so (dim 32, 128; dim 64 and 256 with the split arithmetic on); workspace as
 * for ihg_interact_bwd.  A caller that produces g2 in hyperedge chunks cuts them where the user changes: every call writes the rows
 * of the users whose hyperedges it was given.
 */
int32_t ihg_interact_bwd_user_reduced_supported(int32_t dim, int32_t order, int64_t ld_h);
int ihg_interact_bwd_user_reduced(const float* h, int64_t ld_h, const int32_t* i3, const float* w, int64_t ld_w, int32_t order,
                                  const float* dout, int64_t ld_dout, float* g2, float* dh, int64_t ld_dh, float* dw, int64_t ld_dw,
                                  void* workspace, int64_t workspace_bytes, int64_t n_edges, int32_t dim, ihg_stream_t stream);

/* The user-reduced member gradients (no weight gradients: dw as NULL above) from cotangent rows that were written ALREADY SCALED AND TAKEN APART by
 * ihg_edge_gather_sum_planes: planes_rows is [n_edges][2][dim] fp16 (4 dim bytes per row, like the fp32 row it stands for), inv_scale [n_edges] the rows' inverse
 * powers of two.  The kernel's eight column parts (dim 256) then copy the row's pieces instead of each finding its maximum, scaling and splitting it again; results
 * are bit-identical to ihg_interact_bwd_user_reduced on the fp32 rows.  Reference: the same lines as ihg_interact_bwd (Models/CommonLayers.py:70-85, autograd's backward).
 */
int32_t ihg_interact_bwd_user_reduced_planes_supported(int32_t dim, int32_t order, int64_t ld_h);
int ihg_interact_bwd_user_reduced_planes(const float* h, int64_t ld_h, const int32_t* i3, const float* w, int64_t ld_w, int32_t order,
                                         const void* planes_rows, const float* inv_scale, float* g2, float* dh, int64_t ld_dh,
                                         void* workspace, int64_t workspace_bytes, int64_t n_edges, int32_t dim, ihg_stream_t stream);

/* The user-reduced backward fused with the transpose of the hyperedge -> node pass that follows the interactive step in an IHGNN layer
 * (Models/GnnLayers.py:229-236: Y = Dv^-1 H Ef): the caller has the NODE-level cotangent dy [n_nodes, dim], and the hyperedge
 * cotangent dout[e] = sum over the three members m of dy_scale[m] * dy[m] (dy_scale NULL: 1) - what ihg_edge_gather_sum would
 * have produced - is formed inside the member-gradient kernel from three gathered rows; it is left in `dout` [n_edges, dim]
 * (written, not read) for the weight gradients here and for the caller's first-order scatter.  Everything else as
 * ihg_interact_bwd_user_reduced.  Available where ihg_interact_bwd_gathered_supported says so (dim 32; dim 64 and 128 with the split arithmetic on).
 * dw == NULL: member gradients and dout only (the weight gradients come from ihg_node_interact_bwd_weight); then dout == NULL as well: the
 * hyperedges' cotangents are not stored at all (the caller takes the first-order gradient from dy by the two-hop operator).
 */
int32_t ihg_interact_bwd_gathered_supported(int32_t dim, int32_t order, int64_t ld_h, int64_t ld_dy);
int ihg_interact_bwd_gathered(const float* h, int64_t ld_h, const int32_t* i3, const float* w, int64_t ld_w, int32_t order,
                              const float* dy, int64_t ld_dy, const float* dy_scale, float* dout, int64_t ld_dout,
                              float* g2, float* dh, int64_t ld_dh, float* dw, int64_t ld_dw,
                              void* workspace, int64_t workspace_bytes, int64_t n_edges, int32_t dim, ihg_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * The interactive layer in its NODE-LEVEL form: out = out_scale * H FeatureInteractor(h) without any [n_edges, dim] tensor.
 * Replaces, for the layer as a whole, the reference's FeatureInteractor.forward + torch.sparse.mm(norm_factor * H, .)
 * (CommonLayers.py:70-85, GnnLayers.py:229-236): for a fixed node every product term of a hyperedge feature has the node's own feature
 * as a constant factor, so the sum over the node's hyperedges is a linear map of deg(v) h[v], three sums over the OTHER two members
 * (a_e, b_e) of its hyperedges and their products with h[v].  Two calls:
 *
 * ihg_node_pair_sums: sums[v] = [ sum_e h[a_e] | sum_e h[b_e] | sum_e h[a_e] * h[b_e] ]  (3 dim floats per node; * elementwise).
 *   pair_ptr / pair_ids: CSR over the nodes with 2 ids per incident hyperedge - (query, item) for a user, (user, item) for a query,
 *   (user, query) for an item; the split-row plan is ihg_node_segment_sum's, with segments of even length.  Any dim % 4 == 0.
 *   pair_weight (optional, one float per PAIR, i.e. per two ids): the pair enters all three sums that many times - the multiplicity of its hyperedge in a
 *   layout that keeps each distinct (user, query, item) triple once (duplicates are distinct hyperedges in the reference, Helpers/Graph.py:107-118).
 *
 * ihg_node_interact_fwd: out[v] = out_scale[v] * ( degree[v] (A_t h[v] + bias) + the typed blocks of w applied to the sums and their
 *   products with h[v] ), w = [A_u | A_q | A_i | W_uq | W_qi | W_iu (| W_uqi)] as in ihg_interact_fwd, rows grouped by type_begin[4]
 *   (host array).  degree: the node's hyperedge count as float (0 for an isolated node); out_scale / bias may be NULL.  Available where
 *   ihg_node_interact_fwd_supported says so (dim 64 / 128 / 256, split arithmetic on: IHG_INTERACT_ARITH != f32); the hyperedge form (ihg_interact_fwd +
 *   ihg_node_segment_sum) is the path everywhere else and gives the same rows to the tolerance of DESIGN.md section 5.
 */
int ihg_node_pair_sums(const float* h, int64_t ld_h, const int32_t* pair_ptr, const int32_t* pair_ids, const int32_t* row_order, float* sums,
                       int64_t ld_sums, int64_t n_rows, int32_t dim, int32_t heavy_threshold, const int32_t* seg_begin, const int32_t* seg_end,
                       int64_t n_segments, const int32_t* heavy_rows, const int32_t* heavy_segptr, int64_t n_heavy, float* partials,
                       const float* pair_weight, ihg_stream_t stream);
int32_t ihg_node_interact_fwd_supported(int32_t dim, int32_t order, int64_t ld_h, int64_t ld_sums, int64_t ld_out);
int64_t ihg_node_interact_fwd_workspace_bytes(int32_t dim);
int ihg_node_interact_fwd(const float* h, int64_t ld_h, const float* sums, int64_t ld_sums, const float* degree, const float* out_scale, const float* bias,
                          const float* w, int64_t ld_w, int32_t order, const int64_t* type_begin, float* out, int64_t ld_out, void* workspace,
                          int64_t workspace_bytes, int32_t dim, ihg_stream_t stream);

/* ihg_node_interact_bwd_weight: dw[:, 3 dim ..] (the product blocks W_uq, W_qi, W_iu (, W_uqi)) of the layer above from node-level data only:
 *   d W_block = sum over the nodes of (dy_scale[v] dy[v]) x X_block[v]^T, X = h * sums_a | h * sums_b | sums_ab | h * sums_ab assigned to the blocks of w by
 *   node type - N rows instead of n_edges hyperedges, no gathers.  dy: the layer output's cotangent [N, dim]; dy_scale (NULL: 1): the out_scale of the
 *   forward.  The first-order blocks dw[:, :3 dim] are not written (ihg_node_linear_bwd_weight has them).  With this, ihg_interact_bwd_gathered is called
 *   with dw == NULL (member gradients only).  Available where ihg_node_interact_bwd_weight_supported says so (dim 64 / 128 / 256, split arithmetic on: IHG_INTERACT_ARITH != f32).
 */
int32_t ihg_node_interact_bwd_weight_supported(int32_t dim, int32_t order, int64_t ld_h, int64_t ld_sums, int64_t ld_dy);
int64_t ihg_node_interact_bwd_weight_workspace_bytes(int32_t dim, int32_t order);
int ihg_node_interact_bwd_weight(const float* h, int64_t ld_h, const float* sums, int64_t ld_sums, const float* dy, int64_t ld_dy, const float* dy_scale, int32_t order,
                                 const int64_t* type_begin, float* dw, int64_t ld_dw, void* workspace, int64_t workspace_bytes, int32_t dim, ihg_stream_t stream);


/* ------------------------------------------------------------------------------------------------
 * DEVICE: node-level dense transforms (K4 and the hoisted first-order blocks of K6).
 * Replaces nn.Linear(d, d) `feature_transform` (Models/GnnLayers.py:145, 224) and the u / q / i column blocks of
 * `aggregation` (Models/CommonLayers.py:43, 55, 64-66, 81-85) once they are applied per node instead of per hyperedge,
 * plus the autograd backward of both (torch issues a [d x N] x [N x d] rocBLAS GEMM there that has no split-K).
 *
 * Nodes are typed by contiguous row ranges: type t = rows [type_begin[t], type_begin[t+1]), t = 0,1,2 (host int64[4]).
 * W_t[c][j] = w[c * ld_w + t * w_type_stride + j]; w_type_stride == 0 means one weight for every row.
 *   fwd        out[v,:] = x[v,:] * W_t^T  (+ bias_t if bit t of bias_type_mask is set; bias_t = bias + t * bias_type_stride,
 *              stride 0 = one bias vector shared by the masked types)
 *   bwd_input  dx[v,:]  = dout[v,:] * W_t
 *   bwd_weight dW_t     = sum_{v in t} dout[v,:]^T x[v,:]      (one dW over all rows if dw_type_stride == 0)
 *              dbias[c] = sum over rows of the masked types of dout[v,c]     (dbias may be NULL); with dbias_type_stride != 0
 *              (typed weights only) every type gets its own sum at dbias + t * dbias_type_stride, 0 for unmasked types.
 *              With dx != NULL the call also produces bwd_input's result (w typed exactly like dw: block t at column
 *              t * dw_type_stride); at dim 64 / 128 both come from one pass over dout, otherwise it runs the bwd_input launch itself.
 *              dx_accumulate != 0: dx += instead of dx = (a second contribution to the same gradient: the member gradients of
 *              the interactive step land in dx first, Models/CommonLayers.py:70-85) - where ihg_node_linear_bwd_accumulates
 *              says so (dim 64; dim 128 on the split-arithmetic kernels)
 * Any dim > 0: 32, 64, 128, 256 with 16-byte aligned rows run on the matrix cores, every other shape on the any-width
 * kernels (row-slab partials + a fixed-order sum for the weight gradient).  `workspace`: ihg_node_linear_workspace_bytes(dim)
 * bytes, 16-byte aligned.
 */
int64_t ihg_node_linear_workspace_bytes(int32_t dim);
int32_t ihg_node_linear_bwd_accumulates(int32_t dim, int64_t ld_dout, int64_t ld_x, int64_t ld_dx);
int ihg_node_linear_fwd(const float* x, int64_t ld_x, const float* w, int64_t ld_w, int64_t w_type_stride,
                        const float* bias, int32_t bias_type_mask, int64_t bias_type_stride, const int64_t* type_begin,
                        float* out, int64_t ld_out, void* workspace, int64_t workspace_bytes,
                        int32_t dim, ihg_stream_t stream);
int ihg_node_linear_bwd_input(const float* dout, int64_t ld_dout, const float* w, int64_t ld_w, int64_t w_type_stride,
                              const int64_t* type_begin, float* dx, int64_t ld_dx,
                              void* workspace, int64_t workspace_bytes, int32_t dim, ihg_stream_t stream);
int ihg_node_linear_bwd_weight(const float* dout, int64_t ld_dout, const float* x, int64_t ld_x,
                               const int64_t* type_begin, float* dw, int64_t ld_dw, int64_t dw_type_stride,
                               float* dbias, int32_t bias_type_mask, int64_t dbias_type_stride,
                               const float* w, int64_t ld_w, float* dx, int64_t ld_dx, int32_t dx_accumulate,
                               void* workspace, int64_t workspace_bytes, int32_t dim, ihg_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * DEVICE: batch tail - HEM scoring head over the layer outputs.
 * Replaces torch.cat(gnn_outputs, 1) + the three row gathers of RawGnn.forward (Models/RawGnn.py:122-131) and
 * HemPredictionLayer.forward (Models/PredictionLayers.py:30-43, dot-product branch) for a training batch, and their
 * autograd backward up to (not including) the scatter of duplicate rows.
 *   layers   HOST array of n_layers (<= 8) device pointers, layer l = [N, dim] with row stride ld
 *   rows     device int64 [3*batch]: global node rows of the batch's users, then queries, then items
 *   items    device int64 [batch]: 0-based item ids (index into bias [I])
 *   fwd      scores[r] = sum_l <X_l[item_r], lambda*X_l[query_r] + (1-lambda)*X_l[user_r]> + bias[items[r]]
 *   bwd      rowgrad [3*batch, ld_rowgrad >= n_layers*dim]: gradient contribution of batch row r to its user / query / item
 *            row (blocks 0 / 1 / 2) for the upstream gradient grad_scale * dscores[batch].  If ld_rowgrad > n_layers*dim,
 *            column n_layers*dim carries d bias: dscores[r] on the item row, 0 on the other two.
 * ihg_bce_with_logits: nn.BCEWithLogitsLoss() (mean) and d loss / d scores in one launch (Main.py:191, TrainTestHelper.py:132).
 * ihg_batch_scatter_add: dense[rows[k], c] += rowgrad[k, c] for k < n_rows <= 16384, duplicates summed in batch order
 *            without atomics or a sort (one wave per batch row; the first occurrence of a destination adds all later ones).
 *            Column c lands at dense[(c / block_width) * block_stride + row * ld_dense + c % block_width] - with
 *            block_width = dim, block_stride = N * dim every layer's gradient is its own contiguous [N, dim] matrix; if `tail`
 *            is given, the last column goes to tail[row - tail_row_offset] instead (d bias).  Replaces the
 *            index_put_(accumulate=True) that autograd issues for the three row gathers (RawGnn.py:128-131).
 */
int ihg_hem_score_fwd(const float* const* layers, int32_t n_layers, int64_t ld, int32_t dim,
                      const int64_t* rows, const int64_t* items, const float* bias, float lambda_muq,
                      float* scores, int64_t batch, ihg_stream_t stream);
int ihg_hem_score_bwd(const float* const* layers, int32_t n_layers, int64_t ld, int32_t dim,
                      const int64_t* rows, const float* dscores, float grad_scale, float lambda_muq,
                      float* rowgrad, int64_t ld_rowgrad, int64_t batch, ihg_stream_t stream);
int ihg_bce_with_logits(const float* scores, const float* labels, int64_t n, float* loss, float* dscores, ihg_stream_t stream);
int64_t ihg_batch_scatter_workspace_bytes(int64_t n_rows);      /* 0, or -1 if n_rows > ihg_batch_scatter_max_rows() (caller scatters in row chunks) */
int32_t ihg_batch_scatter_max_rows(void);                        /* 32768: rows of one ihg_batch_scatter_add / ihg_batch_combine launch (its id list lives in LDS: <= 16384 rows in 64 KiB,
                                                                    beyond that - the union of the ranks' batches under the cotangent exchange - a 128 KiB instance) */
int ihg_batch_scatter_add(const float* rowgrad, int64_t ld_rowgrad, int32_t width, const int64_t* rows, int64_t n_rows,
                          float* dense, int64_t ld_dense, int32_t block_width, int64_t block_stride,
                          float* tail, int64_t tail_row_offset, int64_t tail_rows, ihg_stream_t stream);
/* The same combination without a destination, for gradients that are added into several matrices at different times (one
 * per layer output, as the backward reaches it): ihg_batch_combine sums the rows of equal destination INTO the first of them,
 * in place, and sets leader[k] = 1 on those first occurrences (0 elsewhere); ihg_batch_rows_add then does
 * dense[rows[k], 0:width] += src[k, 0:width] over the leader rows (src = any column window of the combined rowgrad), or, with
 * `tail`, tail[rows[k] - tail_row_offset] += src[k, 0].  Leaders have distinct destinations: no atomics, fixed order.
 * disjoint_block_rows (0 = n_rows): consecutive blocks of that many batch rows are known not to share destinations - the user,
 * query and item thirds of a batch - so equal destinations are only looked for inside a block. */
int ihg_batch_combine(float* rowgrad, int64_t ld_rowgrad, int32_t width, const int64_t* rows, int64_t n_rows,
                      int64_t disjoint_block_rows, int32_t* leader, ihg_stream_t stream);
int ihg_batch_rows_add(const float* src, int64_t ld_src, int32_t width, const int64_t* rows, const int32_t* leader, int64_t n_rows,
                       float* dense, int64_t ld_dense, float* tail, int64_t tail_row_offset, int64_t tail_rows, ihg_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * DEVICE: composition of the two linear maps of a first-order layer - `feature_transform` (W [d,d], b; Models/GnnLayers.py:224)
 * followed by the first-order blocks of `aggregation` (A [d,3d] = A_u | A_q | A_i, bias c; Models/CommonLayers.py:60-66) with
 * nothing non-linear between them: node type t sees x (A_t W)^T + (A_t b + [t == user] c), which is what
 * ihg_node_linear_fwd then applies in one pass (typed weights, per-type bias).
 *   fwd   w_eff[i][t d + j] = sum_k A[i][t d + k] W[k][j]      b_eff[t][i] = sum_k A[i][t d + k] b[k] + (t == 0 ? c[i] : 0)
 *   bwd   gradients of both outputs back to A (da [d,3d]), c (dc, may be NULL), W (dw [d,d]) and b (db)
 * Tiny (3 d^3 multiply-adds), one thread per output element, sums in index order.
 */
int ihg_compose_first_order_fwd(const float* a, int64_t ld_a, const float* c, const float* w, int64_t ld_w, const float* b,
                                float* w_eff, int64_t ld_w_eff, float* b_eff, int32_t dim, ihg_stream_t stream);
int ihg_compose_first_order_bwd(const float* a, int64_t ld_a, const float* w, int64_t ld_w, const float* b,
                                const float* dw_eff, int64_t ld_dw_eff, const float* db_eff,
                                float* da, int64_t ld_da, float* dc, float* dw, int64_t ld_dw, float* db,
                                int32_t dim, ihg_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * DEVICE: the optimiser step of the training loop (Main.py:192 `torch.optim.Adam(model.parameters(), lr, weight_decay=...)`,
 * stepped in TrainTestHelper.py:139-143), for all parameters in one launch per 24 tensors.  Same update as torch.optim.Adam
 * (amsgrad off, maximize off): g += wd * p; m += (g - m)(1 - b1); v = b2 v + (1 - b2) g^2;
 * p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps), t = `step` >= 1 (the caller counts).  `tensors` is a HOST array.
 */
typedef struct ihg_adam_tensor {
    float* param;
    const float* grad;
    float* exp_avg;
    float* exp_avg_sq;
    int64_t count;
} ihg_adam_tensor;
int ihg_adam_step(const ihg_adam_tensor* tensors, int32_t n_tensors, float lr, float beta1, float beta2, float eps,
                  float weight_decay, int64_t step, ihg_stream_t stream);
/* The same update with the two step-dependent scalars read from DEVICE memory when the kernel runs: step_scalars[0] = lr / (1 - beta1^t),
 * step_scalars[1] = sqrt(1 - beta2^t).  For a launch recorded in a hipGraph (ihgnn_amd/captured_step.py): the host refreshes the two floats
 * before every replay, the recorded launch stays the same. */
int ihg_adam_step_device_scalars(const ihg_adam_tensor* tensors, int32_t n_tensors, float beta1, float beta2, float eps, float weight_decay,
                                 const float* step_scalars, ihg_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * DEVICE: negative sampling of a training batch.  Replaces `random.sample(range(item_count), k)` per positive in
 * GraphDataset.__getitem__ (Dataset.py:107-109): out[row][0..k) = k DISTINCT item ids, uniform over [0, n_items), the positive
 * itself not excluded (as in the reference).  A pure function of (seed, counter, row): the caller advances `counter` per batch.
 * Not the reference's Mersenne-Twister stream - same distribution, different draws.  k <= 16.
 */
int ihg_sample_negatives(uint64_t seed, uint64_t counter, int64_t n_rows, int64_t n_items, int32_t k, int64_t* out, ihg_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * DEVICE: evaluation scoring with a running top-k.  Replaces, for `n_pairs` search logs at once, the per-log
 * sequence RawGnn.forward(u * ones(I), q * ones(I), None) (Models/RawGnn.py:124-137) -> HemPredictionLayer.forward
 * (Models/PredictionLayers.py:35-43) -> torch.sort(scores, descending=True)[:10] (Helpers/Metrics.py:60-61):
 *   score[c][i] = sum_k features[item_row0 + i][k] * (lambda * features[query_row0 + queries[c]][k] + (1 - lambda) * features[users[c]][k])
 *                 + item_bias[i]
 *   top_items[c][0..k) = the k items of highest score, best first; equal scores in ascending item order (a stable descending
 *   sort; the reference's sort is unstable on ties); top_scores[c][0..k) their scores.  With fewer than k items the tail is -1.
 * `features` is the cached [N, dim] propagation output (any row stride >= dim, dim <= ihg_score_topk_max_dim() = 1264: the mixed rows of 32 pairs
 * stay in LDS); k <= 10.
 * The [n_pairs, n_items] score matrix is never stored: 32 x 32 matrix-core tiles are reduced to per-lane top-k lists in registers.  Arithmetic: every row is
 * multiplied by a power of two and taken apart into two fp16 terms (hi + lo, 22 significand bits); a product is three v_mfma_f32_32x32x16_f16 partial products
 * accumulated in fp32 (relative error <= 3 x 2^-22 per product, as accurate as the fp32 matrix instructions on these sums - tests/test_gpu_parity.py); the item rows are
 * split once per call into the workspace.
 * Workspace: ihg_score_topk_workspace_bytes(n_pairs, n_items, dim) bytes (the split item rows + partial lists).
 */
int32_t ihg_score_topk_max_dim(void);
int64_t ihg_score_topk_workspace_bytes(int64_t n_pairs, int64_t n_items, int32_t dim);
int ihg_score_topk(const float* features, int64_t ld, int32_t dim, int64_t query_row0, int64_t item_row0, int64_t n_items,
                   const float* item_bias, const int64_t* users, const int64_t* queries, float lambda_muq, int64_t n_pairs,
                   int32_t k, float* top_scores, int32_t* top_items, void* workspace, int64_t workspace_bytes, ihg_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * DEVICE: the input features X0 WITHOUT assembling them.  The reference concatenates rows 1.. of the user table, the queries' bag means and
 * rows 1.. of the item table into one [N, d] matrix every step (Models/RawGnn.py:112-113, Models/EmbeddingLayers.py:70-79) and autograd cuts the
 * gradient apart again: four [N, d]-sized copies per training step.  The *_typed entry points take, instead of one matrix, the address of the FIRST ROW
 * of every node type (a HOST array of 3 device pointers: users, queries, items; common row stride) - the layer-0 transform, its weight / input gradient,
 * the scoring head's layer-0 rows and the scatter of the batch rows' gradients read and write the tables in place.
 *   ihg_node_linear_fwd_typed          ihg_node_linear_fwd with x given as typed rows
 *   ihg_node_linear_bwd_weight_typed   ihg_node_linear_bwd_weight with x (and the optional input gradient dx) as typed rows; bit t of
 *                                      zero_row_before_mask: the row in front of dx_rows[t] (the padding row 0 of an embedding table's gradient) is zeroed
 *   ihg_hem_score_fwd_typed0 / ihg_hem_score_bwd_typed0   ihg_hem_score_fwd / _bwd with layer 0 given as typed rows (row stride ld0; layer0_rows NULL: every layer a
 *                                      plain matrix); _bwd multiplies the upstream gradient by *grad_scale_device (a DEVICE scalar, NULL: 1) as well - d loss of a
 *                                      backward pass, without a host read or a separate multiply launch.  rows_upper (optional, [3 batch]): the rows of the layers ABOVE
 *                                      layer 0 where those are numbered differently from layer 0's - a layout that leaves the isolated nodes (no hyperedge: every layer
 *                                      output is zero, Graph.py:120 / App. B 2) out of its own numbering while layer 0 is read from the embedding tables by the reference's
 *                                      node id; a negative entry = isolated: its rows above layer 0 count as zero (ihg_batch_rows_add / _put skip negative rows)
 *   ihg_batch_rows_put                 ihg_batch_rows_add into typed rows; assign != 0: dense[rows[k]] = src[k] on the leader rows (a gradient that is
 *                                      zero elsewhere and read at these rows only: no fill of the matrix)
 * Available where ihg_node_linear_typed_supported says so (dim 128 / 256 on the split-arithmetic kernels, row strides % 4 == 0, 16-byte aligned rows).
 */
int32_t ihg_node_linear_typed_supported(int32_t dim, int64_t ld_x, int64_t ld_out);
int ihg_node_linear_fwd_typed(const float* const* x_rows, int64_t ld_x, const float* w, int64_t ld_w, int64_t w_type_stride,
                              const float* bias, int32_t bias_type_mask, int64_t bias_type_stride, const int64_t* type_begin,
                              float* out, int64_t ld_out, void* workspace, int64_t workspace_bytes, int32_t dim, ihg_stream_t stream);
int ihg_node_linear_bwd_weight_typed(const float* dout, int64_t ld_dout, const float* const* x_rows, int64_t ld_x, const int64_t* type_begin,
                                     float* dw, int64_t ld_dw, int64_t dw_type_stride, float* dbias, int32_t bias_type_mask, int64_t dbias_type_stride,
                                     const float* w, int64_t ld_w, float* const* dx_rows, int64_t ld_dx, int32_t zero_row_before_mask,
                                     void* workspace, int64_t workspace_bytes, int32_t dim, ihg_stream_t stream);
int ihg_hem_score_fwd_typed0(const float* const* layers, int32_t n_layers, int64_t ld, int32_t dim, const float* const* layer0_rows, int64_t ld0,
                             const int64_t* type_begin, const int64_t* rows, const int64_t* rows_upper, const int64_t* items, const float* bias, float lambda_muq,
                             float* scores, int64_t batch, ihg_stream_t stream);
int ihg_hem_score_bwd_typed0(const float* const* layers, int32_t n_layers, int64_t ld, int32_t dim, const float* const* layer0_rows, int64_t ld0,
                             const int64_t* type_begin, const int64_t* rows, const int64_t* rows_upper, const float* dscores, const float* grad_scale_device,
                             float grad_scale, float lambda_muq, float* rowgrad, int64_t ld_rowgrad, int64_t batch, ihg_stream_t stream);
int ihg_batch_rows_put(const float* src, int64_t ld_src, int32_t width, const int64_t* rows, const int32_t* leader, int64_t n_rows,
                       float* const* dense_rows, int64_t ld_dense, const int64_t* type_begin, int32_t assign, ihg_stream_t stream);

/* Small device-side helpers that keep a training step free of framework launches (torch fills / index ops of a few microseconds each):
 *   ihg_zero_floats      p[0 .. n) = 0
 *   ihg_mark_rows        mask[rows[k]] = value, k < n (rows as int64 OR int32: pass the other NULL) - the row mask of a sparse cotangent, set before
 *                        the pull that reads it and cleared after (ihg_node_segment_sum's src_mask)
 *   ihg_batch_node_rows  rows[0 .. 3 b) = users | queries + query_row0 | items + item_row0 (Models/RawGnn.py:128-131); rows32 (may be NULL): the same as int32
 *   ihg_zero_rows        base[rows[k], 0 .. dim) = 0, k < n
 */
int ihg_zero_floats(float* p, int64_t n, ihg_stream_t stream);
int ihg_mark_rows(const int64_t* rows64, const int32_t* rows32, int64_t n, uint8_t* mask, int32_t value, ihg_stream_t stream);
int ihg_batch_node_rows(const int64_t* users, const int64_t* queries, const int64_t* items, int64_t batch, int64_t query_row0, int64_t item_row0,
                        int64_t* rows, int32_t* rows32, ihg_stream_t stream);
int ihg_zero_rows(float* base, int64_t ld, int32_t dim, const int64_t* rows, int64_t n, ihg_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* IHGNN_HIP_H */
