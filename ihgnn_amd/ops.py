"""Differentiable operators of the hypergraph path, each a thin ``torch.autograd.Function`` over the C ABI.

PyTorch is plumbing here (device memory, current stream, autograd tape); every forward and backward below
is one or two launches of a hand-written gfx950 kernel in libihgnn_hip.so.  Tensors must live on the GPU:
there is no CPU implementation to fall back to.

Operator               forward kernel              backward kernel(s)
---------------------  --------------------------  ------------------------------------------------
edge_gather_sum  (K5)  ihg_edge_gather_sum         ihg_node_segment_sum   (its transpose)
node_segment_sum (K7)  ihg_node_segment_sum         ihg_edge_gather_sum    (its transpose)
bag_mean         (K2)  ihg_bag_mean_fwd            ihg_bag_mean_bwd
interact     (K5+K6)   ihg_interact_fwd            ihg_interact_bwd + 4x ihg_node_segment_sum
"""
from __future__ import annotations

import ctypes
import os as _os
from typing import Optional, Union

import torch
from torch import Tensor

from . import _lib
from . import profiler
from .layout import Csr, CsrRows, IncidenceLayout

_void = ctypes.c_void_p


def _ptr(t: Optional[Tensor]):
    return None if t is None else _void(t.data_ptr())


def _stream():
    return _void(torch.cuda.current_stream().cuda_stream)


def _rows(t: Tensor, name: str) -> Tensor:
    """A 2-D fp32 GPU tensor whose rows are contiguous (any row stride); copies only if it must."""
    if not t.is_cuda:
        raise _lib.IhgnnHipError(f'{name} must be a GPU tensor: ihgnn_amd has no CPU path (got device {t.device})')
    if t.dtype != torch.float32 or t.dim() != 2:
        raise TypeError(f'{name} must be a 2-D float32 tensor, got {tuple(t.shape)} {t.dtype}')
    if t.shape[1] > 1 and t.stride(1) != 1:
        t = t.contiguous()
    if t.shape[0] > 1 and t.stride(0) < t.shape[1]:
        t = t.contiguous()
    return t


def _ld(t: Tensor) -> int:
    return int(t.stride(0)) if t.shape[0] > 1 else int(t.shape[1])


def _workspace(n_bytes: int, device: torch.device) -> Tensor:
    """Scratch for one launch (torch's caching allocator hands back the same block step after step)."""
    return torch.empty(max(n_bytes, 16) // 4, dtype=torch.float32, device=device)


# ---------------------------------------------------------------------------------------------
# raw launches (no autograd)
# ---------------------------------------------------------------------------------------------
def edge_gather_sum_raw(src: Tensor, i3: Tensor, node_scale: Optional[Tensor] = None, bias: Optional[Tensor] = None,
                        alpha: float = 1.0, out: Optional[Tensor] = None, edge_scale: Optional[Tensor] = None) -> Tensor:
    """``edge_scale`` (``[E]``): every hyperedge's sum times its own factor - ``layout.edge_weight`` where the result is the cotangent of all copies of a hyperedge."""
    lib = _lib.load()
    src = _rows(src, 'src')
    n_edges, dim = int(i3.shape[0]), int(src.shape[1])
    if out is None:
        out = torch.empty(n_edges, dim, dtype=torch.float32, device=src.device)
    with profiler.kernel('edge_gather_sum', n_edges, dim):
        _lib.check(lib.ihg_edge_gather_sum(_ptr(src), _ld(src), _ptr(i3), _ptr(node_scale), _ptr(bias), float(alpha), _ptr(edge_scale),
                                           _ptr(out), _ld(out), n_edges, dim, _stream()), 'ihg_edge_gather_sum')
    return out


def node_segment_sum_raw(src: Tensor, csr: Union[Csr, CsrRows], src_scale: Optional[Tensor] = None,
                         out_scale: Optional[Tensor] = None, mode: int = _lib.SCALE_NONE,
                         out: Optional[Tensor] = None, entry_scale: Optional[Tensor] = None,
                         self_weight: Optional[Tensor] = None, rows: Optional[Tensor] = None, src_mask: Optional[Tensor] = None,
                         role: Optional[str] = None, accumulate: bool = False, read_once: bool = False) -> Tensor:
    """``role`` names the launch for the profiler (one kernel, several jobs with different byte counts: ``bench.py`` reports each).
    ``rows`` (int32, device): only these output rows are needed.  The split rows of the plan are always computed; of the
    others only the listed ones are, and the rest of ``out`` is left unwritten.  ``src_mask`` (uint8 per source row): rows with a 0
    are all-zero and are not fetched.  ``accumulate``: ``out +=`` instead of ``out =`` (``out`` required; the hyperedge chunks of one scatter; with ``rows``: the listed rows and the plan's split rows).
    ``read_once``: every source row is read exactly once by this launch (non-temporal loads)."""
    lib = _lib.load()
    src = _rows(src, 'src')
    dim = int(src.shape[1])
    if accumulate:
        if out is None:
            raise ValueError('accumulate needs an existing `out`')
        mode = mode | _lib.SCALE_ACCUMULATE
    if read_once:
        mode = mode | _lib.SRC_READ_ONCE
    if out is None:
        out = torch.empty(csr.n_rows, dim, dtype=torch.float32, device=src.device)
    heavy = csr.n_heavy > 0
    if rows is not None and rows.dtype != torch.int32:
        raise TypeError('rows must be an int32 tensor')
    order, n_light = (rows, int(rows.shape[0])) if rows is not None else (csr.row_order, csr.n_rows)
    with profiler.kernel((role or 'node_segment_sum') + ('' if rows is None else '_rows'), n_light, dim):
        _lib.check(lib.ihg_node_segment_sum(
            _ptr(src), _ld(src), _ptr(csr.ptr), _ptr(csr.ids), _ptr(order), _ptr(src_scale), _ptr(entry_scale), _ptr(out_scale), mode,
            _ptr(out), _ld(out), n_light, dim, csr.heavy_threshold if heavy else 0,
            _ptr(csr.seg_begin) if heavy else None, _ptr(csr.seg_end) if heavy else None, csr.n_segments if heavy else 0,
            _ptr(csr.heavy_rows) if heavy else None, _ptr(csr.heavy_segptr) if heavy else None, csr.n_heavy,
            _ptr(csr.partials(dim)) if heavy else None, _ptr(self_weight), _ptr(src_mask), _stream()), 'ihg_node_segment_sum')
    return out


def node_pair_sums_raw(h: Tensor, layout: IncidenceLayout, out: Optional[Tensor] = None) -> Tensor:
    """``[N, 3 d]``: per node the sums over its incident hyperedges' OTHER two members ``(a, b)`` of ``h[a]``, ``h[b]`` and ``h[a] * h[b]``
    (``ihg_node_pair_sums`` over ``layout.hop2_csr``) - the gather half of the interactive layer's node-level form."""
    lib = _lib.load()
    h = _rows(h, 'h')
    dim = int(h.shape[1])
    csr = layout.hop2_csr
    if out is None:
        out = torch.empty(layout.node_count, 3 * dim, dtype=torch.float32, device=h.device)
    heavy = csr.n_heavy > 0
    with profiler.kernel('node_pair_sums', layout.node_count, dim):
        _lib.check(lib.ihg_node_pair_sums(
            _ptr(h), _ld(h), _ptr(csr.ptr), _ptr(csr.ids), _ptr(csr.row_order), _ptr(out), _ld(out), csr.n_rows, dim,
            csr.heavy_threshold if heavy else 0, _ptr(csr.seg_begin) if heavy else None, _ptr(csr.seg_end) if heavy else None,
            csr.n_segments if heavy else 0, _ptr(csr.heavy_rows) if heavy else None, _ptr(csr.heavy_segptr) if heavy else None, csr.n_heavy,
            _ptr(csr.partials(3 * dim)) if heavy else None, _ptr(layout.pair_weight), _stream()), 'ihg_node_pair_sums')
    return out


def _check_out(out: Optional[Tensor], *inputs: Tensor) -> Optional[Tensor]:
    """``out=``: a caller-owned ``[rows, d]`` destination (any row stride - a column slice of the ``[N, d (L + 1)]`` feature matrix,
    ``RawGnn.propagate``).  Only outside autograd: a recorded op must own its output."""
    if out is not None and torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in inputs):
        raise ValueError('out= is for inference (torch.no_grad()): a differentiable op allocates its own output')
    return out


# ---------------------------------------------------------------------------------------------
# K5 / K7 as a transposed pair
# ---------------------------------------------------------------------------------------------
class _EdgeGatherSum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src: Tensor, layout: IncidenceLayout, node_scale: Optional[Tensor], alpha: float) -> Tensor:
        ctx.layout, ctx.node_scale, ctx.alpha = layout, node_scale, alpha
        return edge_gather_sum_raw(src, layout.i3, node_scale, None, alpha)

    @staticmethod
    def backward(ctx, grad_out: Tensor):
        scale = ctx.node_scale
        if ctx.alpha != 1.0:
            scale = (scale * ctx.alpha) if scale is not None else torch.full(
                (ctx.layout.node_count,), ctx.alpha, dtype=torch.float32, device=grad_out.device)
        mode = _lib.SCALE_NONE if scale is None else _lib.SCALE_MULTIPLY
        # (grad_out is the cotangent of the layout's rows - one per DISTINCT hyperedge under edge_weight -: each row is added once, whatever its multiplicity)
        return node_segment_sum_raw(grad_out, ctx.layout.node_csr, None, scale, mode, role='k7.edges_to_nodes_bwd_of_k5'), None, None, None


class _NodeSegmentSum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src: Tensor, layout: IncidenceLayout, out_scale: Optional[Tensor], rows: Optional[Tensor], out: Optional[Tensor]) -> Tensor:
        ctx.layout, ctx.out_scale = layout, out_scale
        mode = _lib.SCALE_NONE if out_scale is None else _lib.SCALE_MULTIPLY
        # layout.edge_weight (duplicate triples collapsed): a row stands for m_e hyperedges of the reference's incidence and enters the sum m_e times
        return node_segment_sum_raw(src, layout.node_csr, layout.edge_weight, out_scale, mode, rows=rows, role='k7.edges_to_nodes', out=_check_out(out, src))

    @staticmethod
    def backward(ctx, grad_out: Tensor):
        return edge_gather_sum_raw(grad_out, ctx.layout.i3, ctx.out_scale, None, 1.0, edge_scale=ctx.layout.edge_weight), None, None, None, None


def edge_gather_sum(src: Tensor, layout: IncidenceLayout, node_scale: Optional[Tensor] = None, alpha: float = 1.0) -> Tensor:
    """node -> hyperedge: ``out[e] = alpha * sum_{v in e} node_scale[v] * src[v]``  (``[N,d] -> [E,d]``)."""
    return _EdgeGatherSum.apply(src, layout, node_scale, float(alpha))


def node_segment_sum(src: Tensor, layout: IncidenceLayout, out_scale: Optional[Tensor] = None, rows: Optional[Tensor] = None,
                     out: Optional[Tensor] = None) -> Tensor:
    """hyperedge -> node: ``out[v] = out_scale[v] * sum_{e containing v} src[e]``  (``[E,d] -> [N,d]``; over ALL hyperedges of the reference's incidence: a layout
    that keeps a repeated triple once weights its row by the multiplicity).

    ``rows`` (int32): the caller reads only these rows of the result (the batch rows of the last layer's output in a training
    step); rows outside the list and outside the split-row plan are left UNWRITTEN.  The gradient must then be zero outside
    ``rows`` too - which it is when only those rows were read."""
    return _NodeSegmentSum.apply(src, layout, out_scale, rows, out)


# The first-order layers' gather launches walk the two-hop list either as it is (two ids per incidence) or with the repeated (destination, source) entries of a row merged
# into one weighted entry (layout.two_hop_merged).  The LAYOUT decides (IncidenceLayout.two_hop_merged_default: merged when >= 25 % of the entries are repeats, or when it
# carries hyperedge multiplicities): measured in round 5 on the C2 - C4 stand-ins, 8.6 - 17.9 % fewer gathers bought NOT A MICROSECOND (the repeats were cache hits:
# profiles/r5/01_ab_two_hop_merged.txt) - at C5 79 % of the entries are repeats, its 10 GB node table is far beyond the caches, and the three launches go from 22.5 / 14.4 /
# 23.9 ms to 13.0 / 6.4 / 12.8 (profiles/r6/01_ab_two_hop_C5_*.json: 326.8 -> 297.7 ms per step).  IHG_TWO_HOP_MERGED=1 / 0 (or this attribute: True / False) overrides.
_two_hop_env = _os.environ.get('IHG_TWO_HOP_MERGED', 'auto')
TWO_HOP_MERGED = {'1': True, '0': False}.get(_two_hop_env)          # None: per layout


def two_hop_merged_for(layout: IncidenceLayout) -> bool:
    """Whether this layout's first-order launches walk the merged (weighted) two-hop list."""
    if layout.edge_weight is not None:
        return True                                           # (its plain list has one entry per DISTINCT hyperedge: not the operator without the weights)
    return layout.two_hop_merged_default if TWO_HOP_MERGED is None else bool(TWO_HOP_MERGED)


def _two_hop_list(layout: IncidenceLayout):
    """``(csr, entry weights or None)`` of the two-hop operator's off-diagonal part ``H H^T - diag(deg)``."""
    if two_hop_merged_for(layout):
        csr, weights, _ = layout.two_hop_merged()
        return csr, weights
    return layout.hop2_csr, None


def _two_hop_first_order_gradient(dy: Tensor, layout: IncidenceLayout, out_scale: Optional[Tensor]) -> Tensor:
    """``d P = H H^T (out_scale * dy)``: the first-order blocks' gradient of the interactive layer by the two-hop operator on the node-level cotangent."""
    csr, weights = _two_hop_list(layout)
    return node_segment_sum_raw(dy, csr, out_scale, None, _lib.SCALE_NONE, entry_scale=weights, self_weight=layout.self_weight, role='k7.two_hop_first_order_gradient')


class _TwoHop(torch.autograd.Function):
    """``out = Do * (H H^T) (Di * x)``: node -> hyperedge -> node in ONE pass over the node table (no ``[E,d]`` round trip).

    ``H H^T`` is symmetric, so the backward is the same launch with the two diagonal scalings swapped."""

    @staticmethod
    def forward(ctx, x: Tensor, layout: IncidenceLayout, in_scale: Optional[Tensor], out_scale: Optional[Tensor], rows: Optional[Tensor],
                cotangent_rows: Optional[Tensor], out: Optional[Tensor]) -> Tensor:
        ctx.layout, ctx.in_scale, ctx.out_scale = layout, in_scale, out_scale
        # rows outside which the cotangent is zero: the rows that were computed at all, or the caller's explicit promise
        ctx.cot_rows = rows if rows is not None else (cotangent_rows if SPARSE_LAST_COTANGENT else None)
        mode = _lib.SCALE_NONE if out_scale is None else _lib.SCALE_MULTIPLY
        csr, weights = _two_hop_list(layout)
        return node_segment_sum_raw(x, csr, in_scale, out_scale, mode, entry_scale=weights, self_weight=layout.self_weight, rows=rows, role='k7.two_hop',
                                    out=_check_out(out, x))

    @staticmethod
    def backward(ctx, grad_out: Tensor):
        lay = ctx.layout
        mode = _lib.SCALE_NONE if ctx.in_scale is None else _lib.SCALE_MULTIPLY
        mask = None
        if ctx.cot_rows is not None:
            # grad_out is zero outside these rows: the pull skips the gathers of the zero rows (two thirds of them - every neighbour that
            # is not a query - and the ones that would miss the cache); same gradient as the dense pull.  The byte-per-row mask is the layout's, all zero
            # between uses: the listed rows are set here and cleared after the pull (two launches over 3 B ids instead of a fill of N bytes + an index fill)
            lib = _lib.load()
            mask = lay.row_mask()
            rows = ctx.cot_rows
            r64, r32 = (_ptr(rows), None) if rows.dtype == torch.int64 else (None, _ptr(rows))
            _lib.check(lib.ihg_mark_rows(r64, r32, int(rows.shape[0]), _ptr(mask), 1, _stream()), 'ihg_mark_rows')
            if CHECK_SPARSE_COTANGENT and bool((grad_out[mask == 0] != 0).any()):
                _lib.check(lib.ihg_mark_rows(r64, r32, int(rows.shape[0]), _ptr(mask), 0, _stream()), 'ihg_mark_rows')
                raise RuntimeError('node_two_hop: the cotangent is not zero outside cotangent_rows / rows - the output has a consumer the caller did not declare')
        try:
            csr, weights = _two_hop_list(lay)
            out = node_segment_sum_raw(grad_out, csr, ctx.out_scale, ctx.in_scale, mode, entry_scale=weights, self_weight=lay.self_weight, src_mask=mask,
                                       role='k7.two_hop_bwd' if mask is None else 'k7.two_hop_bwd_masked')
            if mask is not None:
                _lib.check(lib.ihg_mark_rows(r64, r32, int(rows.shape[0]), _ptr(mask), 0, _stream()), 'ihg_mark_rows')
        except BaseException:
            # the mask belongs to the layout and must be all zero between uses: a pull that raised leaves it so (a stale bit would make the next
            # masked pull gather rows of a cotangent buffer that only has its batch rows written)
            if mask is not None:
                lay.drop_row_mask()
            raise
        return out, None, None, None, None, None, None


def node_two_hop(x: Tensor, layout: IncidenceLayout, in_scale: Optional[Tensor] = None, out_scale: Optional[Tensor] = None,
                 rows: Optional[Tensor] = None, cotangent_rows: Optional[Tensor] = None, out: Optional[Tensor] = None) -> Tensor:
    """``out[v] = out_scale[v] * sum_{e containing v} sum_{w in e} in_scale[w] * x[w]`` - the first-order
    node -> hyperedge -> node step (K5 followed by K7) without materialising the hyperedge features.  ``rows``: as in
    ``node_segment_sum``.  ``cotangent_rows`` (int32 / int64 node rows): the caller's PROMISE that the gradient with respect to the
    output is zero outside these rows - true of the last layer of a training step, whose output is read at the batch rows only - so
    the backward pulls just them.  The promise is carried by the op itself (not looked up by address); ``IHG_CHECK_SPARSE_COTANGENT=1``
    verifies it at every backward.  ``IHG_SPARSE_LAST_COTANGENT=0`` ignores it (dense pull, same gradient)."""
    return _TwoHop.apply(x, layout, in_scale, out_scale, rows, cotangent_rows, out)


class _PairSpmm(torch.autograd.Function):
    """``out = Ds (A (Ds x))`` for a symmetric weighted adjacency: its own transpose, so backward is the same launch."""

    @staticmethod
    def forward(ctx, x: Tensor, graph, out: Optional[Tensor]) -> Tensor:
        ctx.graph = graph
        return node_segment_sum_raw(x, graph.csr, graph.inv_sqrt_deg, graph.inv_sqrt_deg, _lib.SCALE_MULTIPLY, entry_scale=graph.values, role='k7.pair_graph',
                                    out=_check_out(out, x))

    @staticmethod
    def backward(ctx, grad_out: Tensor):
        g = ctx.graph
        return node_segment_sum_raw(grad_out, g.csr, g.inv_sqrt_deg, g.inv_sqrt_deg, _lib.SCALE_MULTIPLY, entry_scale=g.values, role='k7.pair_graph'), None, None


class _CsrSpmm(torch.autograd.Function):
    """``out = Do (A (Ds x))`` for a sparse ``A`` given as CSR over its rows together with the CSR of its transpose: the backward
    is the same launch over the transpose with the two scalings swapped."""

    @staticmethod
    def forward(ctx, x: Tensor, csr, csr_t, values, values_t, src_scale, out_scale, role: str, out: Optional[Tensor]) -> Tensor:
        ctx.args = (csr, csr_t, values, values_t, src_scale, out_scale, role)
        mode = _lib.SCALE_NONE if out_scale is None else _lib.SCALE_MULTIPLY
        return node_segment_sum_raw(x, csr, src_scale, out_scale, mode, entry_scale=values, role=role, out=_check_out(out, x))

    @staticmethod
    def backward(ctx, grad_out: Tensor):
        csr, csr_t, values, values_t, src_scale, out_scale, role = ctx.args
        mode = _lib.SCALE_NONE if src_scale is None else _lib.SCALE_MULTIPLY
        return (node_segment_sum_raw(grad_out, csr_t, out_scale, src_scale, mode, entry_scale=values_t, role=role + '_bwd'),) + (None,) * 8


def hyper_node_to_edge(x: Tensor, layout, src_scale: Optional[Tensor] = None, out_scale: Optional[Tensor] = None) -> Tensor:
    """node -> hyperedge over a general (variable-arity) incidence: ``out[e] = out_scale[e] * sum_{v in e} val(v,e) src_scale[v] x[v]``
    (``thsp.matmul(incidence_t, .)``, ``GnnLayers.py:148``) - ``layout`` is a :class:`ihgnn_amd.layout.LogHyperLayout`."""
    return _CsrSpmm.apply(x, layout.edge_csr, layout.node_csr, layout.edge_values, layout.node_values, src_scale, out_scale, 'k7.hyper_node_to_edge', None)


def hyper_edge_to_node(x: Tensor, layout, src_scale: Optional[Tensor] = None, out_scale: Optional[Tensor] = None, out: Optional[Tensor] = None) -> Tensor:
    """hyperedge -> node over a general incidence: ``out[v] = out_scale[v] * sum_{e containing v} val(v,e) src_scale[e] x[e]``
    (``thsp.matmul(incidence, .)``, ``GnnLayers.py:151``)."""
    return _CsrSpmm.apply(x, layout.node_csr, layout.edge_csr, layout.node_values, layout.edge_values, src_scale, out_scale, 'k7.hyper_edge_to_node', out)


def pair_spmm(x: Tensor, graph, out: Optional[Tensor] = None) -> Tensor:
    """GCN propagation ``D^-1/2 A D^-1/2 x`` over a :class:`ihgnn_amd.layout.PairLayout` (``GnnLayers.py:35-38``)."""
    return _PairSpmm.apply(x, graph, out)


# ---------------------------------------------------------------------------------------------
# K2 query embedding bag (mean)
# ---------------------------------------------------------------------------------------------
class BagLayout:
    """Query -> word-row lists (``nn.EmbeddingBag`` input/offsets of Dataset.py:161-186) and their transpose."""

    def __init__(self, bag_input, bag_offsets, table_rows: int, device: torch.device):
        import numpy as np
        words = np.ascontiguousarray(np.asarray(bag_input, dtype=np.int64).reshape(-1))
        offsets = np.asarray(bag_offsets, dtype=np.int64).reshape(-1)
        if words.size and (words.min() < 0 or words.max() >= table_rows):
            raise ValueError('bag word id outside the embedding table')
        ptr = np.append(offsets, words.shape[0]).astype(np.int32)
        self.n_bags, self.table_rows = int(offsets.shape[0]), int(table_rows)
        self.bags = Csr(ptr, words.astype(np.int32), device, heavy_threshold=0)
        self.words_of = self.bags.transpose(table_rows)
        lens = np.diff(ptr.astype(np.int64)).astype(np.float32)
        self.bag_len = torch.from_numpy(lens).to(device)
        inv = np.where(lens > 0, 1.0 / np.maximum(lens, 1), 0).astype(np.float32)
        self.inv_len = torch.from_numpy(inv).to(device)


class _BagMean(torch.autograd.Function):
    @staticmethod
    def forward(ctx, table: Tensor, bag: BagLayout) -> Tensor:
        lib = _lib.load()
        ctx.bag = bag
        table = _rows(table, 'table')
        dim = int(table.shape[1])
        out = torch.empty(bag.n_bags, dim, dtype=torch.float32, device=table.device)
        with profiler.kernel('bag_mean_fwd', bag.n_bags, dim):
            _lib.check(lib.ihg_bag_mean_fwd(_ptr(table), _ld(table), _ptr(bag.bags.ptr), _ptr(bag.bags.ids), _ptr(bag.bag_len),
                                            _ptr(out), _ld(out), bag.n_bags, dim, _stream()), 'ihg_bag_mean_fwd')
        return out

    @staticmethod
    def backward(ctx, grad_out: Tensor):
        lib = _lib.load()
        bag = ctx.bag
        grad_out = _rows(grad_out, 'grad_out')
        dim = int(grad_out.shape[1])
        dtable = torch.empty(bag.table_rows, dim, dtype=torch.float32, device=grad_out.device)
        with profiler.kernel('bag_mean_bwd', bag.table_rows, dim):
            _lib.check(lib.ihg_bag_mean_bwd(_ptr(grad_out), _ld(grad_out), _ptr(bag.words_of.ptr), _ptr(bag.words_of.ids),
                                            _ptr(bag.inv_len), _ptr(dtable), _ld(dtable), bag.table_rows, dim, _stream()),
                       'ihg_bag_mean_bwd')
        return dtable, None


def bag_mean(table: Tensor, bag: BagLayout) -> Tensor:
    """``nn.EmbeddingBag(mode='mean')`` over every query: ``[V+1,d] -> [Q,d]``."""
    return _BagMean.apply(table, bag)


class _EmbedAllNodes(torch.autograd.Function):
    """``X0 = [user table rows 1.. ; bag means of the query words ; item table rows 1..]`` assembled in one buffer (the bag
    kernel writes its rows in place), with a backward that hands each table its gradient without a full-size zero fill."""

    @staticmethod
    def forward(ctx, user_table: Tensor, item_table: Tensor, word_table: Tensor, bag: BagLayout, out: Optional[Tensor]) -> Tensor:
        lib = _lib.load()
        word_table = _rows(word_table, 'word table')
        u, q, i = int(user_table.shape[0]) - 1, bag.n_bags, int(item_table.shape[0]) - 1
        dim = int(word_table.shape[1])
        x = _check_out(out, user_table, item_table, word_table)
        if x is None:
            x = torch.empty(u + q + i, dim, dtype=torch.float32, device=word_table.device)
        x[:u].copy_(user_table[1:])
        x[u + q:].copy_(item_table[1:])
        rows = x[u:u + q]
        if q > 0:
            with profiler.kernel('bag_mean_fwd', q, dim):
                _lib.check(lib.ihg_bag_mean_fwd(_ptr(word_table), _ld(word_table), _ptr(bag.bags.ptr), _ptr(bag.bags.ids), _ptr(bag.bag_len),
                                                _ptr(rows), _ld(x), q, dim, _stream()), 'ihg_bag_mean_fwd')
        ctx.bag, ctx.counts = bag, (u, q, i)
        return x

    @staticmethod
    def backward(ctx, grad: Tensor):
        lib = _lib.load()
        bag, (u, q, i) = ctx.bag, ctx.counts
        grad = _rows(grad, 'grad')
        dim = int(grad.shape[1])
        d_user = torch.empty(u + 1, dim, dtype=torch.float32, device=grad.device)
        d_user[0].zero_()                                     # the padding row gets no gradient
        d_user[1:].copy_(grad[:u])
        d_item = torch.empty(i + 1, dim, dtype=torch.float32, device=grad.device)
        d_item[0].zero_()
        d_item[1:].copy_(grad[u + q:])
        d_word = torch.empty(bag.table_rows, dim, dtype=torch.float32, device=grad.device)
        g_q = grad[u:u + q]
        with profiler.kernel('bag_mean_bwd', bag.table_rows, dim):
            _lib.check(lib.ihg_bag_mean_bwd(_ptr(g_q), _ld(grad), _ptr(bag.words_of.ptr), _ptr(bag.words_of.ids),
                                            _ptr(bag.inv_len), _ptr(d_word), dim, bag.table_rows, dim, _stream()), 'ihg_bag_mean_bwd')
        return d_user, d_item, d_word, None, None


def embed_all_nodes(user_table: Tensor, item_table: Tensor, word_table: Tensor, bag: BagLayout, out: Optional[Tensor] = None) -> Tensor:
    """The full-graph input features ``EmbeddingLayer(None, None, None)`` concatenated (``RawGnn.py:112-113``): ``[U+Q+I, d]``
    from the ``[U+1, d]`` / ``[I+1, d]`` tables (row 0 = padding) and the ``[V+1, d]`` word table."""
    return _EmbedAllNodes.apply(user_table, item_table, word_table, bag, out)


# ---------------------------------------------------------------------------------------------
# K4: node-level dense transforms (feature_transform and the hoisted u / q / i blocks of the aggregation)
# ---------------------------------------------------------------------------------------------
def node_linear_supported(x: Tensor, w: Tensor) -> bool:
    """True when the HIP node-level transform kernels take this input: fp32 GPU rows of any width (d in {32,64,128,256} with
    16-byte aligned rows runs on the MFMA row-GEMM, everything else on the any-width kernels of the same library)."""
    d = int(x.shape[1])
    return (x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and w.dtype == torch.float32 and x.stride(-1) == 1 and w.stride(-1) == 1
            and int(w.shape[0]) == d and int(w.shape[1]) >= d)


def node_linear_tiled(x: Tensor, w: Tensor) -> bool:
    """The MFMA row-GEMM form applies (what the fused one-node interactive path builds on)."""
    d = int(x.shape[1])
    return (node_linear_supported(x, w) and d in (32, 64, 128, 256) and x.stride(0) % 4 == 0 and w.stride(0) % 4 == 0
            and x.data_ptr() % 16 == 0 and w.data_ptr() % 16 == 0)


def _type_begin(layout: IncidenceLayout):
    tb = getattr(layout, '_type_begin_c', None)
    if tb is None:
        u, q = layout.user_count, layout.query_count
        tb = (ctypes.c_int64 * 4)(0, u, u + q, layout.node_count)
        layout._type_begin_c = tb
    return tb


class _NodeLinear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x: Tensor, w: Tensor, bias: Optional[Tensor], layout: IncidenceLayout, typed: bool, bias_mask: int) -> Tensor:
        lib = _lib.load()
        x = _rows(x, 'x')
        dim = int(x.shape[1])
        out = torch.empty(x.shape[0], dim, dtype=torch.float32, device=x.device)
        ws = _workspace(int(lib.ihg_node_linear_workspace_bytes(dim)), x.device)
        stride = dim if typed else 0
        per_type_bias = bias is not None and bias.dim() == 2                       # [3, d]: one bias vector per node type
        if per_type_bias:
            if not typed or tuple(bias.shape) != (3, dim):
                raise ValueError(f'a per-type bias is [3, {dim}] and needs typed weights, got {tuple(bias.shape)}')
            bias = bias.contiguous()
        with profiler.kernel('node_linear_fwd', x.shape[0], dim):
            _lib.check(lib.ihg_node_linear_fwd(_ptr(x), _ld(x), _ptr(w), int(w.stride(0)), stride, _ptr(bias), bias_mask, dim if per_type_bias else 0,
                                               _type_begin(layout), _ptr(out), _ld(out), _ptr(ws), ws.numel() * 4, dim, _stream()),
                       'ihg_node_linear_fwd')
        ctx.save_for_backward(x, w)
        ctx.layout, ctx.typed, ctx.bias_mask, ctx.has_bias, ctx.per_type_bias = layout, typed, bias_mask, bias is not None, per_type_bias
        return out

    @staticmethod
    def backward(ctx, grad_out: Tensor):
        lib = _lib.load()
        x, w = ctx.saved_tensors
        g = _rows(grad_out, 'grad_out')
        if g.stride(0) % 4 or g.data_ptr() % 16:
            g = g.contiguous()
        dim = int(x.shape[1])
        stride = dim if ctx.typed else 0
        tb = _type_begin(ctx.layout)
        ws = _workspace(int(lib.ihg_node_linear_workspace_bytes(dim)), x.device)
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dw = torch.zeros_like(w) if w.shape[1] != dim * (3 if ctx.typed else 1) else torch.empty_like(w)   # product-block columns of a [d, k*d] weight stay 0
        dbias = None
        if ctx.has_bias:
            dbias = torch.empty((3, dim) if ctx.per_type_bias else (dim,), dtype=torch.float32, device=x.device)
        # one call: the weight / bias gradient, and the input gradient from the same pass over grad_out where the width allows
        with profiler.kernel('node_linear_bwd', x.shape[0], dim):
            _lib.check(lib.ihg_node_linear_bwd_weight(_ptr(g), _ld(g), _ptr(x), _ld(x), tb, _ptr(dw), int(dw.stride(0)), stride,
                                                      _ptr(dbias), ctx.bias_mask, dim if ctx.per_type_bias else 0,
                                                      _ptr(w), int(w.stride(0)), _ptr(dx), _ld(dx) if dx is not None else 0, 0,
                                                      _ptr(ws), ws.numel() * 4, dim, _stream()), 'ihg_node_linear_bwd_weight')
        return dx, dw, dbias, None, None, None


class _ComposeFirstOrder(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a: Tensor, c: Optional[Tensor], w: Tensor, b: Tensor):
        lib = _lib.load()
        a, w = _rows(a, 'aggregation weight'), _rows(w, 'transform weight')
        b = b.contiguous()
        c = c.contiguous() if c is not None else None
        dim = int(w.shape[0])
        w_eff = torch.empty(dim, 3 * dim, dtype=torch.float32, device=w.device)
        b_eff = torch.empty(3, dim, dtype=torch.float32, device=w.device)
        _lib.check(lib.ihg_compose_first_order_fwd(_ptr(a), _ld(a), _ptr(c), _ptr(w), _ld(w), _ptr(b), _ptr(w_eff), 3 * dim, _ptr(b_eff), dim, _stream()),
                   'ihg_compose_first_order_fwd')
        ctx.save_for_backward(a, w, b)
        ctx.has_c = c is not None
        return w_eff, b_eff

    @staticmethod
    def backward(ctx, dw_eff: Tensor, db_eff: Tensor):
        lib = _lib.load()
        a, w, b = ctx.saved_tensors
        dim = int(w.shape[0])
        dw_eff, db_eff = _rows(dw_eff, 'dw_eff'), db_eff.contiguous()
        da, dw, db = torch.empty_like(a), torch.empty_like(w), torch.empty_like(b)
        dc = torch.empty(dim, dtype=torch.float32, device=w.device) if ctx.has_c else None
        _lib.check(lib.ihg_compose_first_order_bwd(_ptr(a), _ld(a), _ptr(w), _ld(w), _ptr(b), _ptr(dw_eff), _ld(dw_eff), _ptr(db_eff), _ptr(da), _ld(da),
                                                   _ptr(dc), _ptr(dw), _ld(dw), _ptr(db), dim, _stream()), 'ihg_compose_first_order_bwd')
        return da, dc, dw, db


def compose_first_order(a: Tensor, c: Optional[Tensor], w: Tensor, b: Tensor):
    """``(W_eff [d, 3d], b_eff [3, d])`` of ``first_order(linear(x; w, b); a, c)``: ``W_eff = [A_u w | A_q w | A_i w]``,
    ``b_eff[t] = A_t b`` (+ ``c`` for users).  Differentiable in all four arguments."""
    if tuple(a.shape) != (w.shape[0], 3 * w.shape[0]) or w.shape[0] != w.shape[1]:
        raise ValueError(f'compose_first_order takes a [d, 3d] block weight and a [d, d] transform, got {tuple(a.shape)} and {tuple(w.shape)}')
    return _ComposeFirstOrder.apply(a, c, w, b)


class NodeTables:
    """The input features ``X0 = [user_table[1:] ; bag means of the queries ; item_table[1:]]`` (``RawGnn.py:112-113``) WITHOUT assembling them: the
    first node-level transform of the model reads the three row blocks where they are (``ihg_node_linear_fwd_typed``), its backward writes the tables'
    gradients in place (no ``[N, d]`` gradient that autograd then cuts apart), and the batch tail reads / scatters its layer-0 rows the same way.
    Stands in for the ``[N, d]`` tensor between ``EmbeddingLayer.node_tables()`` and the first layer; ``node_linear`` resolves it (bag means, autograd token)."""

    def __init__(self, user_table: Tensor, item_table: Tensor, word_table: Tensor, bag: 'BagLayout', holder: 'TailGradients'):
        self.user_table, self.item_table, self.word_table, self.bag, self.holder = user_table, item_table, word_table, bag, holder
        self.query_rows: Optional[Tensor] = None        # [Q, d] bag means, set by the first node_linear
        self.token: Optional[Tensor] = None             # carries the autograd edge from the batch tail to that op
        self.dim = int(user_table.shape[1])
        self.shape = (int(user_table.shape[0]) - 1 + bag.n_bags + int(item_table.shape[0]) - 1, self.dim)
        self.device = user_table.device

    def row_pointers(self):
        """HOST array of three device pointers: the first row of the users, the queries, the items (row 0 of either table is its padding row)."""
        step = self.dim * 4
        return (ctypes.c_void_p * 3)(self.user_table.data_ptr() + step, self.query_rows.data_ptr(), self.item_table.data_ptr() + step)

    def type_begin(self):
        """First row of every node type in the numbering the TABLES are addressed by (the public one: ``[0, U, U + Q, N]``) - the layout's own unless it leaves the isolated
        nodes out (``gather_active_nodes`` sets it then)."""
        tb = getattr(self, '_public_type_begin', None)
        return tb if tb is not None else _type_begin(self.layout)

    @staticmethod
    def supported(user_table: Tensor, item_table: Tensor, word_table: Tensor) -> bool:
        lib = _lib.load()
        d = int(user_table.shape[1])
        return (user_table.is_cuda and all(t.dtype == torch.float32 and t.is_contiguous() and t.data_ptr() % 16 == 0 for t in (user_table, item_table, word_table))
                and d % 4 == 0 and bool(lib.ihg_node_linear_typed_supported(d, d, d)))


class _LinearFromTables(torch.autograd.Function):
    """``node_linear(X0)`` with X0 given by its tables (``NodeTables``): bag means of the queries + the typed-rows row GEMM forward; backward: weight / bias gradient
    and the input gradient written straight into the tables' gradients (padding rows zeroed by the library), the batch tail's layer-0 row gradients added
    there (this op is also the tap of layer 0), then the bag-mean backward."""

    @staticmethod
    def forward(ctx, user_table: Tensor, item_table: Tensor, word_table: Tensor, w: Tensor, bias: Optional[Tensor], nodes: NodeTables, layout: IncidenceLayout,
                typed: bool, bias_mask: int):
        lib = _lib.load()
        bag, dim = nodes.bag, nodes.dim
        n = nodes.shape[0]
        query_rows = torch.empty(bag.n_bags, dim, dtype=torch.float32, device=word_table.device)
        if bag.n_bags > 0:
            with profiler.kernel('bag_mean_fwd', bag.n_bags, dim):
                _lib.check(lib.ihg_bag_mean_fwd(_ptr(word_table), dim, _ptr(bag.bags.ptr), _ptr(bag.bags.ids), _ptr(bag.bag_len), _ptr(query_rows), dim, bag.n_bags, dim,
                                                _stream()), 'ihg_bag_mean_fwd')
        nodes.query_rows, nodes.layout = query_rows, layout
        nodes.token = torch.empty(1, dtype=torch.float32, device=word_table.device)
        out = torch.empty(n, dim, dtype=torch.float32, device=word_table.device)
        ws = _workspace(int(lib.ihg_node_linear_workspace_bytes(dim)), word_table.device)
        per_type_bias = bias is not None and bias.dim() == 2
        if per_type_bias:
            if not typed or tuple(bias.shape) != (3, dim):
                raise ValueError(f'a per-type bias is [3, {dim}] and needs typed weights, got {tuple(bias.shape)}')
            bias = bias.contiguous()
        with profiler.kernel('node_linear_fwd', n, dim):
            _lib.check(lib.ihg_node_linear_fwd_typed(nodes.row_pointers(), dim, _ptr(w), int(w.stride(0)), dim if typed else 0, _ptr(bias), bias_mask, dim if per_type_bias else 0,
                                                     _type_begin(layout), _ptr(out), dim, _ptr(ws), ws.numel() * 4, dim, _stream()), 'ihg_node_linear_fwd_typed')
        ctx.save_for_backward(user_table, item_table, w, query_rows)
        ctx.nodes, ctx.layout, ctx.typed, ctx.bias_mask, ctx.has_bias, ctx.per_type_bias = nodes, layout, typed, bias_mask, bias is not None, per_type_bias
        ctx.mark_non_differentiable(query_rows)
        ctx.set_materialize_grads(False)                     # (the token's and the bag means' gradients are never read: no zero tensors made for them)
        return out, nodes.token, query_rows

    @staticmethod
    def backward(ctx, grad_out: Tensor, _grad_token, _grad_rows):
        lib = _lib.load()
        user_table, item_table, w, query_rows = ctx.saved_tensors
        nodes, layout = ctx.nodes, ctx.layout
        bag, dim = nodes.bag, nodes.dim
        if grad_out is None:
            raise RuntimeError('the output of the first node-level transform received no gradient')
        g = _rows(grad_out, 'grad_out')
        if g.stride(0) % 4 or g.data_ptr() % 16:
            g = g.contiguous()
        dev = g.device
        d_user, d_item = torch.empty_like(user_table), torch.empty_like(item_table)
        d_query = torch.empty_like(query_rows)
        dw = torch.zeros_like(w) if w.shape[1] != dim * (3 if ctx.typed else 1) else torch.empty_like(w)
        dbias = None
        if ctx.has_bias:
            dbias = torch.empty((3, dim) if ctx.per_type_bias else (dim,), dtype=torch.float32, device=dev)
        step = dim * 4
        x_rows = (ctypes.c_void_p * 3)(user_table.data_ptr() + step, query_rows.data_ptr(), item_table.data_ptr() + step)
        dx_rows = (ctypes.c_void_p * 3)(d_user.data_ptr() + step, d_query.data_ptr(), d_item.data_ptr() + step)
        ws = _workspace(int(lib.ihg_node_linear_workspace_bytes(dim)), dev)
        with profiler.kernel('node_linear_bwd', nodes.shape[0], dim):
            _lib.check(lib.ihg_node_linear_bwd_weight_typed(_ptr(g), _ld(g), x_rows, dim, _type_begin(layout), _ptr(dw), int(dw.stride(0)), dim if ctx.typed else 0,
                                                            _ptr(dbias), ctx.bias_mask, dim if ctx.per_type_bias else 0, _ptr(w), int(w.stride(0)), dx_rows, dim, 0b101,
                                                            _ptr(ws), ws.numel() * 4, dim, _stream()), 'ihg_node_linear_bwd_weight_typed')
        holder = nodes.holder
        if holder is not None and holder.rowgrad is not None:              # the batch tail's gradient of the layer-0 rows: this op is their tap
            holder.put_into_typed(dx_rows, dim, layout, 0, dim)
        d_word = torch.empty(bag.table_rows, dim, dtype=torch.float32, device=dev)
        with profiler.kernel('bag_mean_bwd', bag.table_rows, dim):
            _lib.check(lib.ihg_bag_mean_bwd(_ptr(d_query), dim, _ptr(bag.words_of.ptr), _ptr(bag.words_of.ids), _ptr(bag.inv_len), _ptr(d_word), dim, bag.table_rows, dim,
                                            _stream()), 'ihg_bag_mean_bwd')
        return d_user, d_item, d_word, dw, dbias, None, None, None, None


class _GatherActiveNodes(torch.autograd.Function):
    """``X0`` of the nodes that have hyperedges, ``[N', d]`` in the numbering of a layout that leaves the isolated nodes out (``IncidenceLayout.compact``), gathered from the
    embedding tables' rows and the query bag means - the public ``[N, d]`` matrix is never assembled.  Resolves the ``NodeTables`` for the batch tail as the first
    node-level transform does otherwise: the tail reads its layer-0 rows from the tables in place (public ids) and this op's backward adds their gradients to the tables'
    gradients - dense ``[U + 1, d]`` / ``[I + 1, d]`` / ``[V + 1, d]`` tensors whose rows are zero for nodes that are neither active nor in the batch."""

    @staticmethod
    def forward(ctx, user_table: Tensor, item_table: Tensor, word_table: Tensor, nodes: NodeTables, layout: IncidenceLayout):
        lib = _lib.load()
        bag, dim = nodes.bag, nodes.dim
        u_pub, q_pub = layout.public_user_count, layout.public_query_count
        query_rows = torch.empty(bag.n_bags, dim, dtype=torch.float32, device=word_table.device)
        if bag.n_bags > 0:
            with profiler.kernel('bag_mean_fwd', bag.n_bags, dim):
                _lib.check(lib.ihg_bag_mean_fwd(_ptr(word_table), dim, _ptr(bag.bags.ptr), _ptr(bag.bags.ids), _ptr(bag.bag_len), _ptr(query_rows), dim, bag.n_bags, dim,
                                                _stream()), 'ihg_bag_mean_fwd')
        nodes.query_rows, nodes.layout = query_rows, layout
        nodes._public_type_begin = (ctypes.c_int64 * 4)(0, u_pub, u_pub + q_pub, layout.public_node_count)
        nodes.token = torch.empty(1, dtype=torch.float32, device=word_table.device)
        picks = getattr(layout, '_active_table_rows', None)
        if picks is None:
            act = layout.active_nodes
            ua, qa = layout.user_count, layout.query_count
            picks = layout._active_table_rows = (act[:ua] + 1, act[ua:ua + qa] - u_pub, act[ua + qa:] - (u_pub + q_pub) + 1)      # table rows (row 0 = padding), query ids
        ua, qa = int(picks[0].shape[0]), int(picks[1].shape[0])
        x = torch.empty(layout.node_count, dim, dtype=torch.float32, device=word_table.device)
        torch.index_select(user_table, 0, picks[0], out=x[:ua])
        torch.index_select(query_rows, 0, picks[1], out=x[ua:ua + qa])
        torch.index_select(item_table, 0, picks[2], out=x[ua + qa:])
        ctx.save_for_backward(user_table, item_table)
        ctx.nodes, ctx.layout, ctx.picks = nodes, layout, picks
        ctx.mark_non_differentiable(query_rows)
        ctx.set_materialize_grads(False)
        return x, nodes.token, query_rows

    @staticmethod
    def backward(ctx, grad_x: Tensor, _grad_token, _grad_rows):
        lib = _lib.load()
        user_table, item_table = ctx.saved_tensors
        nodes, layout, picks = ctx.nodes, ctx.layout, ctx.picks
        bag, dim = nodes.bag, nodes.dim
        if grad_x is None:
            raise RuntimeError('the gathered input features received no gradient')
        g = _rows(grad_x, 'grad of the gathered input features')
        ua, qa = int(picks[0].shape[0]), int(picks[1].shape[0])
        d_user, d_item = torch.zeros_like(user_table), torch.zeros_like(item_table)
        d_query = torch.zeros(bag.n_bags, dim, dtype=torch.float32, device=g.device)
        d_user.index_copy_(0, picks[0], g[:ua])
        d_query.index_copy_(0, picks[1], g[ua:ua + qa])
        d_item.index_copy_(0, picks[2], g[ua + qa:])
        holder = nodes.holder
        if holder is not None and holder.rowgrad is not None:              # the batch tail's gradient of its layer-0 rows (public ids; isolated batch nodes included)
            step = dim * 4
            dx_rows = (ctypes.c_void_p * 3)(d_user.data_ptr() + step, d_query.data_ptr(), d_item.data_ptr() + step)
            holder.put_into_typed(dx_rows, dim, layout, 0, dim, type_begin=nodes._public_type_begin)
        d_word = torch.empty(bag.table_rows, dim, dtype=torch.float32, device=g.device)
        with profiler.kernel('bag_mean_bwd', bag.table_rows, dim):
            _lib.check(lib.ihg_bag_mean_bwd(_ptr(d_query), dim, _ptr(bag.words_of.ptr), _ptr(bag.words_of.ids), _ptr(bag.inv_len), _ptr(d_word), dim, bag.table_rows, dim,
                                            _stream()), 'ihg_bag_mean_bwd')
        return d_user, d_item, d_word, None, None


def gather_active_nodes(nodes: NodeTables, layout: IncidenceLayout) -> Tensor:
    """``[N', d]`` input features of the nodes of a compact layout straight from the embedding tables (``NodeTables``), which it resolves for the batch tail."""
    if not getattr(layout, 'compact', False):
        raise ValueError('gather_active_nodes is for a layout that leaves the isolated nodes out')
    if nodes.query_rows is not None:
        raise RuntimeError('NodeTables: the input features were already consumed (one per forward)')
    x, _token, _rows_q = _GatherActiveNodes.apply(nodes.user_table, nodes.item_table, nodes.word_table, nodes, layout)
    return x


def node_linear(x, w: Tensor, bias: Optional[Tensor], layout: IncidenceLayout, typed: bool = False, bias_mask: int = 0b111) -> Tensor:
    """``out[v] = x[v] @ W_type(v).T (+ bias)``.  ``typed=False``: one ``[d,d]`` weight for every node; ``typed=True``:
    ``w`` is ``[d, k*d]`` and node type t uses its column block ``w[:, t*d:(t+1)*d]``; a ``[d]`` bias is added to the types in
    ``bias_mask``, a ``[3, d]`` bias gives every (masked) type its own vector.  ``x`` may be a ``NodeTables`` (the input features given by their tables)."""
    if isinstance(x, NodeTables):
        if x.query_rows is not None:
            raise RuntimeError('NodeTables: the input features were already consumed by a node-level transform (one per forward)')
        out, _token, _rows_q = _LinearFromTables.apply(x.user_table, x.item_table, x.word_table, w, bias, x, layout, bool(typed), int(bias_mask))
        return out
    return _NodeLinear.apply(x, w, bias, layout, bool(typed), int(bias_mask))


# ---------------------------------------------------------------------------------------------
# Widths between the tiled ones: zero-padded to the next width the MFMA kernels take
# ---------------------------------------------------------------------------------------------
FAST_WIDTHS = (32, 64, 128, 256)
# IHG_PAD_WIDTHS=1 (default): an embedding width that is a multiple of 4 below 256 and not one of FAST_WIDTHS - the reference takes any --emb (Helpers/ArgsParser.py:94-95,
# Main.py:23): 48, 96, 160, 192, 224 ... - runs on the kernels of the NEXT fast width with zero columns appended: zero weight rows / columns and zero bias entries keep the
# padding columns exactly zero through every layer (a linear map of zeros, products with zeros), so the first d columns are the d-wide model's values - the same sums with
# exact zeros added - and the parameters keep their shapes.  0: the any-width kernels (one thread per output), which is what every width ran on before round 6.
PAD_WIDTHS = _os.environ.get('IHG_PAD_WIDTHS', '1') != '0'


def padded_width(dim: int) -> int:
    """The width the kernels run a ``dim``-wide model at: ``dim`` itself when it is tiled (or padding is off / does not apply), else the next of ``FAST_WIDTHS``."""
    dim = int(dim)
    if not PAD_WIDTHS or dim in FAST_WIDTHS or dim > FAST_WIDTHS[-1] or dim < 1:
        return dim
    return next(wd for wd in FAST_WIDTHS if wd >= dim)


def pad_columns(x: Tensor, width: int) -> Tensor:
    """``[n, d] -> [n, width]`` with zero columns appended (differentiable; ``x`` itself when it is already that wide)."""
    return x if int(x.shape[-1]) == width else torch.nn.functional.pad(x, (0, width - int(x.shape[-1])))


def pad_square(w: Tensor, width: int) -> Tensor:
    """A ``[d, d]`` weight as the top-left block of a zero ``[width, width]`` one."""
    d = int(w.shape[0])
    return w if d == width else torch.nn.functional.pad(w, (0, width - d, 0, width - d))


def pad_blocks(w: Tensor, width: int) -> Tensor:
    """A ``[d, k d]`` block weight (``aggregation.weight``: blocks u, q, i, uq, qi, iu, uqi side by side) as ``[width, k width]``: every ``[d, d]`` block in the top-left of its own
    zero ``[width, width]`` block."""
    d = int(w.shape[0])
    if d == width:
        return w
    k = int(w.shape[1]) // d
    return torch.nn.functional.pad(w.reshape(d, k, d), (0, width - d, 0, 0, 0, width - d)).reshape(width, k * width)


def pad_vector(b: Optional[Tensor], width: int) -> Optional[Tensor]:
    if b is None or int(b.shape[-1]) == width:
        return b
    return torch.nn.functional.pad(b, (0, width - int(b.shape[-1])))


# ---------------------------------------------------------------------------------------------
# K5+K6 interactive step (orders 2 and 3)
# ---------------------------------------------------------------------------------------------
# the [E, 3, d] member-gradient buffer of the interactive backward is produced in hyperedge chunks beyond this many bytes
MEMBER_BUFFER_LIMIT_BYTES = 48 << 30
# use ihg_interact_bwd_user_reduced (user slot summed on chip, [E, 2, d] member buffer) where the library offers it (tests set this attribute to compare with the
# [E, 3, d] form, which every shape without such a kernel runs anyway)
USER_REDUCED_BACKWARD = _os.environ.get('IHG_USER_REDUCED_BACKWARD', '1') != '0'
# the last layer of a training step is told which rows of its output are read (node_two_hop's cotangent_rows): its backward pulls only those
SPARSE_LAST_COTANGENT = _os.environ.get('IHG_SPARSE_LAST_COTANGENT', '1') != '0'
CHECK_SPARSE_COTANGENT = _os.environ.get('IHG_CHECK_SPARSE_COTANGENT', '0') == '1'
# the forward of the interactive layer in its node-level form (no [E, d] rows: ihg_node_pair_sums + ihg_node_interact_fwd) where the library has it;
# IHG_NODE_LEVEL_FORWARD=0: the hyperedge form (typed first-order GEMM -> ihg_interact_fwd -> K7) everywhere (A/B, tests)
NODE_LEVEL_FORWARD = _os.environ.get('IHG_NODE_LEVEL_FORWARD', '1') != '0'
# ... and the product blocks' weight gradients from node-level data (ihg_node_interact_bwd_weight) after such a forward; IHG_NODE_LEVEL_WEIGHT=0: the hyperedge kernel
NODE_LEVEL_WEIGHT = _os.environ.get('IHG_NODE_LEVEL_WEIGHT', '1') != '0'
# the first-order gradient d P = H H^T (scale * dy): a scatter of the stored [E, d] cotangents - or, for tables beyond this many bytes, the two-hop operator on dy
# (at C3 the scatter wins, 8.74 against 8.81 ms per step; at C5 the 51 GB table's scatter reads HBM at random and the two-hop form wins, 22 against 32 ms)
FIRST_ORDER_TWO_HOP_BYTES = int(_os.environ.get('IHG_FIRST_ORDER_TWO_HOP_BYTES', 8 << 30))     # where the member-gradient kernel does not form the hyperedges' cotangents itself: [E, d] tables larger than this take the two-hop form


def _node_level_forward_ok(h: Tensor, w: Tensor, bias: Optional[Tensor], out: Optional[Tensor], dim: int, order: int) -> bool:
    lib = _lib.load()
    ld_out = dim if out is None else _ld(out)
    return (bool(lib.ihg_node_interact_fwd_supported(dim, order, _ld(h), 3 * dim, ld_out)) and h.data_ptr() % 16 == 0 and w.data_ptr() % 16 == 0 and _ld(w) % 4 == 0
            and (bias is None or bias.data_ptr() % 16 == 0) and (out is None or out.data_ptr() % 16 == 0))


# the input features read from the embedding tables in place (NodeTables) in a training step through the fused batch tail; IHG_NODE_TABLES=0: X0 assembled by copies (A/B, tests)
NODE_TABLES = _os.environ.get('IHG_NODE_TABLES', '1') != '0'




def _user_reduced_ok(h: Tensor, w: Tensor, grad_out: Tensor, layout: IncidenceLayout, order: int) -> bool:
    """The interactive backward can run ``ihg_interact_bwd_user_reduced`` on these operands."""
    lib = _lib.load()
    return bool(USER_REDUCED_BACKWARD and layout.edge_count > 0 and getattr(layout, 'user_sorted', False)
                and lib.ihg_interact_bwd_user_reduced_supported(int(h.shape[1]), order, _ld(h)) and h.data_ptr() % 16 == 0 and w.data_ptr() % 16 == 0
                and _ld(w) % 4 == 0 and _ld(grad_out) % 4 == 0 and grad_out.data_ptr() % 16 == 0)


def _zero_isolated_users(dh: Tensor, layout: IncidenceLayout) -> None:
    """Users without hyperedges are not written by the user-reduced kernels: their rows of ``dh`` are zeroed here (an index fill of those
    rows; a fill of the whole user block is 118 MB at C3)."""
    idx = layout.users_without_hyperedges()
    if idx.numel():
        _lib.check(_lib.load().ihg_zero_rows(_ptr(dh), _ld(dh), int(dh.shape[1]), _ptr(idx), int(idx.numel()), _stream()), 'ihg_zero_rows')


# the hyperedges' cotangents of the interactive layer's backward written by K5 as the member-gradient kernel's operand (two fp16 planes per row + the row's inverse scale)
# where nothing else reads them (d = 256 beyond FIRST_ORDER_TWO_HOP_BYTES: config C5); IHG_COTANGENT_PLANES=0: fp32 rows, scaled and split by every column part of the kernel
COTANGENT_PLANES = _os.environ.get('IHG_COTANGENT_PLANES', '1') != '0'


def _interact_backward(h: Tensor, w: Tensor, grad_out: Tensor, layout: IncidenceLayout, order: int, dw: Optional[Tensor], inv_scale: Optional[Tensor] = None) -> Tensor:
    """Product-block weight gradient into ``dw`` (its columns from ``3 d`` on; ``None``: not wanted - the caller has it from the node-level
    kernel) and the member gradients scattered to nodes (returned).  One pass when the ``[E, 3, d]`` buffer fits ``MEMBER_BUFFER_LIMIT_BYTES``, otherwise hyperedge chunks, each with
    its own member lists (``IncidenceLayout.member_csr_chunks``): same sums, associated chunk by chunk.  ``inv_scale`` (``[E]``): ``grad_out`` holds fp16 planes
    (``ihg_edge_gather_sum_planes``), not fp32 rows; the caller has checked ``_cotangent_planes_ok``."""
    lib = _lib.load()
    n_edges, dim = layout.edge_count, int(h.shape[1])
    if inv_scale is not None and (dw is not None or not _user_reduced_ok(h, w, grad_out, layout, order)):
        raise RuntimeError('cotangent planes are read by the user-reduced member-gradient kernel only')
    if _user_reduced_ok(h, w, grad_out, layout, order):
        # hyperedges are numbered by user: the kernel sums the user slot on chip and writes dh[users] itself; only the query and item
        # slots go through the member buffer ([E, 2, d]) and the K7 pass.  Beyond MEMBER_BUFFER_LIMIT_BYTES the buffer is produced in hyperedge chunks
        # cut where the user changes: every launch writes the rows of its own users, the K7 passes after the first add onto the query and item rows.
        n_chunks = max(1, -(-(n_edges * 2 * dim * 4) // MEMBER_BUFFER_LIMIT_BYTES))
        if n_chunks == 1:
            csr_qi, qi_rows = layout.member_csr_qi()
            parts = [(0, n_edges, csr_qi, qi_rows)]
        else:
            parts = layout.member_csr_qi_chunks(n_chunks)
        dh = torch.empty(layout.node_count, dim, dtype=torch.float32, device=h.device)
        _zero_isolated_users(dh, layout)
        for index, (e0, e1, csr_qi, qi_rows) in enumerate(parts):
            n = e1 - e0
            g2 = torch.empty(n, 2 * dim, dtype=torch.float32, device=h.device)
            dw_part = dw if index == 0 or dw is None else torch.empty_like(dw)
            ws = _workspace(int(lib.ihg_interact_bwd_workspace_bytes(n, dim, order)), h.device)
            go = grad_out[e0:e1]
            with profiler.kernel('interact_bwd', n, dim):
                if inv_scale is not None:
                    _lib.check(lib.ihg_interact_bwd_user_reduced_planes(_ptr(h), _ld(h), _ptr(layout.i3[e0:e1]), _ptr(w), _ld(w), order, _ptr(go), _ptr(inv_scale[e0:e1]),
                                                                        _ptr(g2), _ptr(dh), dim, _ptr(ws), ws.numel() * 4, n, dim, _stream()),
                               'ihg_interact_bwd_user_reduced_planes')
                else:
                    _lib.check(lib.ihg_interact_bwd_user_reduced(_ptr(h), _ld(h), _ptr(layout.i3[e0:e1]), _ptr(w), _ld(w), order, _ptr(go), _ld(grad_out), _ptr(g2),
                                                                 _ptr(dh), dim, _ptr(dw_part), _ld(dw_part) if dw_part is not None else 0, _ptr(ws), ws.numel() * 4, n, dim,
                                                                 _stream()),
                               'ihg_interact_bwd_user_reduced')
            if index > 0 and dw is not None:
                dw[:, 3 * dim:].add_(dw_part[:, 3 * dim:])
            node_segment_sum_raw(g2.view(2 * n, dim), csr_qi, out=dh, rows=qi_rows, role='k7.member_gradients', accumulate=index > 0, read_once=True)
            del g2
        return dh
    n_chunks = max(1, -(-(n_edges * 3 * dim * 4) // MEMBER_BUFFER_LIMIT_BYTES))
    if n_chunks == 1:
        parts = [(0, n_edges, layout.member_csr)]
    else:
        parts = layout.member_csr_chunks(n_chunks)
    dh = None
    for index, (e0, e1, csr) in enumerate(parts):
        n = e1 - e0
        g = torch.empty(n, 3 * dim, dtype=torch.float32, device=h.device)
        dw_part = dw if index == 0 or dw is None else torch.empty_like(dw)
        ws = _workspace(int(lib.ihg_interact_bwd_workspace_bytes(n, dim, order)), h.device)
        go = grad_out[e0:e1]
        with profiler.kernel('interact_bwd', n, dim):
            _lib.check(lib.ihg_interact_bwd(_ptr(h), _ld(h), _ptr(layout.i3[e0:e1]), _ptr(w), _ld(w), order, _ptr(go), _ld(grad_out),
                                            _ptr(g), _ptr(dw_part), _ld(dw_part) if dw_part is not None else 0, _ptr(ws), ws.numel() * 4, n, dim, _stream()),
                       'ihg_interact_bwd')
        if index > 0 and dw is not None:
            dw[:, 3 * dim:].add_(dw_part[:, 3 * dim:])
        # (node v, hyperedge e) reads row 3 (e - e0) + type(v); the chunks after the first ADD onto dh inside the kernel (no [N, d] add pass)
        dh = node_segment_sum_raw(g.view(3 * n, dim), csr, role='k7.member_gradients', out=dh, accumulate=dh is not None, read_once=True)
        del g
    return dh


class _Interact(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h: Tensor, p: Tensor, w: Tensor, layout: IncidenceLayout, order: int) -> Tensor:
        lib = _lib.load()
        h, p = _rows(h, 'h'), _rows(p, 'p')
        w = _rows(w, 'w')
        dim = int(h.shape[1])
        out = torch.empty(layout.edge_count, dim, dtype=torch.float32, device=h.device)
        ws = _workspace(int(lib.ihg_interact_fwd_workspace_bytes(layout.edge_count, dim, order)), h.device)
        with profiler.kernel('interact_fwd', layout.edge_count, dim):
            _lib.check(lib.ihg_interact_fwd(_ptr(h), _ld(h), _ptr(p), _ld(p), _ptr(layout.i3), _ptr(w), _ld(w), order,
                                            _ptr(out), _ld(out), _ptr(ws), ws.numel() * 4, layout.edge_count, dim, _stream()),
                       'ihg_interact_fwd')
        ctx.save_for_backward(h, w)
        ctx.layout, ctx.order = layout, order
        return out

    @staticmethod
    def backward(ctx, grad_out: Tensor):
        lib = _lib.load()
        h, w = ctx.saved_tensors
        layout, order = ctx.layout, ctx.order
        grad_out = _rows(grad_out, 'grad_out')
        n_edges, dim = layout.edge_count, int(h.shape[1])
        dw = torch.zeros_like(w)
        dh = _interact_backward(h, w, grad_out, layout, order, dw)
        dp = node_segment_sum_raw(grad_out, layout.node_csr, role='k7.first_order_gradient')
        return dh, dp, dw, None, None


def _gathered_backward_ok(h: Tensor, w: Tensor, dy: Tensor, layout: IncidenceLayout, order: int) -> bool:
    lib = _lib.load()
    n_edges, dim = layout.edge_count, int(h.shape[1])
    # (the gathering kernel forms sum_m scale[m] dy[m] itself and has no per-hyperedge factor: a layout with multiplicities takes the K5 + member-kernel sequence)
    return (USER_REDUCED_BACKWARD and n_edges > 0 and getattr(layout, 'user_sorted', False) and layout.edge_weight is None
            and n_edges * 3 * dim * 4 <= MEMBER_BUFFER_LIMIT_BYTES
            and bool(lib.ihg_interact_bwd_gathered_supported(dim, order, _ld(h), _ld(dy)))
            and h.data_ptr() % 16 == 0 and w.data_ptr() % 16 == 0 and dy.data_ptr() % 16 == 0 and _ld(w) % 4 == 0)


class _InteractToNodes(torch.autograd.Function):
    """Interactive node -> hyperedge step and the hyperedge -> node pass behind it (``GnnLayers.py:229-236``) as ONE autograd node:
    ``y = out_scale * H interact(h, p, w)``.  Forward: the two kernels of the separate ops.  Backward: the hyperedge cotangent
    ``sum_m out_scale[m] dy[m]`` is not produced by a node -> hyperedge launch (K5) - the member-gradient kernel gathers the three
    ``dy`` rows of a hyperedge itself and leaves their sum for the weight gradients and the first-order scatter
    (``ihg_interact_bwd_gathered``); where that kernel does not apply, the separate ops' sequence."""

    @staticmethod
    def forward(ctx, h: Tensor, p: Tensor, w: Tensor, layout: IncidenceLayout, order: int, out_scale: Optional[Tensor], rows: Optional[Tensor],
                out: Optional[Tensor]) -> Tensor:
        lib = _lib.load()
        h, p, w = _rows(h, 'h'), _rows(p, 'p'), _rows(w, 'w')
        dim = int(h.shape[1])
        _check_out(out, h, p, w)
        edge = torch.empty(layout.edge_count, dim, dtype=torch.float32, device=h.device)
        ws = _workspace(int(lib.ihg_interact_fwd_workspace_bytes(layout.edge_count, dim, order)), h.device)
        with profiler.kernel('interact_fwd', layout.edge_count, dim):
            _lib.check(lib.ihg_interact_fwd(_ptr(h), _ld(h), _ptr(p), _ld(p), _ptr(layout.i3), _ptr(w), _ld(w), order,
                                            _ptr(edge), _ld(edge), _ptr(ws), ws.numel() * 4, layout.edge_count, dim, _stream()),
                       'ihg_interact_fwd')
        mode = _lib.SCALE_NONE if out_scale is None else _lib.SCALE_MULTIPLY
        y = node_segment_sum_raw(edge, layout.node_csr, layout.edge_weight, out_scale, mode, rows=rows, role='k7.edges_to_nodes', out=out)
        ctx.save_for_backward(h, w)
        ctx.layout, ctx.order, ctx.out_scale = layout, order, out_scale
        return y

    @staticmethod
    def backward(ctx, dy: Tensor):
        lib = _lib.load()
        h, w = ctx.saved_tensors
        layout, order, out_scale = ctx.layout, ctx.order, ctx.out_scale
        dy = _rows(dy, 'dy')
        n_edges, dim = layout.edge_count, int(h.shape[1])
        dw = torch.zeros_like(w)
        if _gathered_backward_ok(h, w, dy, layout, order):
            csr_qi, qi_rows = layout.member_csr_qi()
            dout = torch.empty(n_edges, dim, dtype=torch.float32, device=h.device)
            g2 = torch.empty(n_edges, 2 * dim, dtype=torch.float32, device=h.device)
            dh = torch.empty(layout.node_count, dim, dtype=torch.float32, device=h.device)
            _zero_isolated_users(dh, layout)
            ws = _workspace(int(lib.ihg_interact_bwd_workspace_bytes(n_edges, dim, order)), h.device)
            with profiler.kernel('interact_bwd', n_edges, dim):
                _lib.check(lib.ihg_interact_bwd_gathered(_ptr(h), _ld(h), _ptr(layout.i3), _ptr(w), _ld(w), order, _ptr(dy), _ld(dy), _ptr(out_scale),
                                                         _ptr(dout), dim, _ptr(g2), _ptr(dh), dim, _ptr(dw), _ld(dw), _ptr(ws), ws.numel() * 4,
                                                         n_edges, dim, _stream()), 'ihg_interact_bwd_gathered')
            node_segment_sum_raw(g2.view(2 * n_edges, dim), csr_qi, out=dh, rows=qi_rows, role='k7.member_gradients', read_once=True)
        else:
            dout = edge_gather_sum_raw(dy, layout.i3, out_scale, None, 1.0, edge_scale=layout.edge_weight)      # (the cotangent of ALL copies of a row: x m_e)
            dh = _interact_backward(h, w, dout, layout, order, dw)
        dp = node_segment_sum_raw(dout, layout.node_csr, role='k7.first_order_gradient')
        return dh, dp, dw, None, None, None, None, None


def interact_to_nodes(h: Tensor, p: Tensor, w: Tensor, layout: IncidenceLayout, order: int, out_scale: Optional[Tensor] = None,
                      rows: Optional[Tensor] = None, out: Optional[Tensor] = None) -> Tensor:
    """``node_segment_sum(interact(h, p, w, layout, order), layout, out_scale, rows)`` with a backward that has no node -> hyperedge launch."""
    if order not in (2, 3):
        raise ValueError('interact_to_nodes handles interaction orders 2 and 3')
    return _InteractToNodes.apply(h, p, w, layout, int(order), out_scale, rows, out)


class _InteractLayer(torch.autograd.Function):
    """The whole interactive layer behind ``feature_transform`` as ONE autograd node (``CommonLayers.py:70-85`` + ``GnnLayers.py:229-236``):
    hoisted first-order blocks (typed row GEMM) -> product blocks (``ihg_interact_fwd``) -> hyperedge -> node pass.  Backward: the member-
    gradient kernel forms the hyperedges' cotangents itself where it can (``ihg_interact_bwd_gathered``, else a K5 launch), the first-order
    path's contribution to ``d h`` is ADDED by the node-level weight-gradient kernel onto the member gradients' scatter result
    (``dx_accumulate``: no separate ``[N, d]`` add, no second ``[N, d]`` buffer), and both halves of ``d aggregation.weight`` land in one tensor."""

    @staticmethod
    def forward(ctx, h: Tensor, w: Tensor, bias: Optional[Tensor], layout: IncidenceLayout, order: int, out_scale: Optional[Tensor], rows: Optional[Tensor],
                out: Optional[Tensor]) -> Tensor:
        lib = _lib.load()
        h, w = _rows(h, 'h'), _rows(w, 'w')
        dim = int(h.shape[1])
        _check_out(out, h, w, bias)
        ctx.layout, ctx.order, ctx.out_scale, ctx.has_bias = layout, order, out_scale, bias is not None
        if NODE_LEVEL_FORWARD and rows is None and _node_level_forward_ok(h, w, bias, out, dim, order):
            # no hyperedge rows: sums over each node's other members, then a node-level contraction (ihg_node_interact_fwd)
            sums = node_pair_sums_raw(h, layout)
            ctx.save_for_backward(h, w, sums)                  # the pair sums: the backward's node-level weight gradients read them again
            y = out if out is not None else torch.empty(layout.node_count, dim, dtype=torch.float32, device=h.device)
            ws = _workspace(int(lib.ihg_node_interact_fwd_workspace_bytes(dim)), h.device)
            with profiler.kernel('node_interact_fwd', layout.node_count, dim):
                _lib.check(lib.ihg_node_interact_fwd(_ptr(h), _ld(h), _ptr(sums), _ld(sums), _ptr(layout.self_weight), _ptr(out_scale), _ptr(bias), _ptr(w), _ld(w),
                                                     order, _type_begin(layout), _ptr(y), _ld(y), _ptr(ws), ws.numel() * 4, dim, _stream()),
                           'ihg_node_interact_fwd')
            return y
        ctx.save_for_backward(h, w)
        p = torch.empty(h.shape[0], dim, dtype=torch.float32, device=h.device)
        ws = _workspace(int(lib.ihg_node_linear_workspace_bytes(dim)), h.device)
        with profiler.kernel('node_linear_fwd', h.shape[0], dim):
            _lib.check(lib.ihg_node_linear_fwd(_ptr(h), _ld(h), _ptr(w), int(w.stride(0)), dim, _ptr(bias), 0b001, 0, _type_begin(layout),
                                               _ptr(p), _ld(p), _ptr(ws), ws.numel() * 4, dim, _stream()), 'ihg_node_linear_fwd')
        edge = torch.empty(layout.edge_count, dim, dtype=torch.float32, device=h.device)
        ws2 = _workspace(int(lib.ihg_interact_fwd_workspace_bytes(layout.edge_count, dim, order)), h.device)
        with profiler.kernel('interact_fwd', layout.edge_count, dim):
            _lib.check(lib.ihg_interact_fwd(_ptr(h), _ld(h), _ptr(p), _ld(p), _ptr(layout.i3), _ptr(w), _ld(w), order,
                                            _ptr(edge), _ld(edge), _ptr(ws2), ws2.numel() * 4, layout.edge_count, dim, _stream()),
                       'ihg_interact_fwd')
        mode = _lib.SCALE_NONE if out_scale is None else _lib.SCALE_MULTIPLY
        return node_segment_sum_raw(edge, layout.node_csr, layout.edge_weight, out_scale, mode, rows=rows, role='k7.edges_to_nodes', out=out)

    @staticmethod
    def backward(ctx, dy: Tensor):
        lib = _lib.load()
        h, w = ctx.saved_tensors[:2]
        sums = ctx.saved_tensors[2] if len(ctx.saved_tensors) > 2 else None
        layout, order, out_scale = ctx.layout, ctx.order, ctx.out_scale
        dy = _rows(dy, 'dy')
        n_edges, dim = layout.edge_count, int(h.shape[1])
        dw = torch.empty_like(w)                               # product blocks from the interact kernels, first-order blocks from the row-GEMM pass
        gathered = _gathered_backward_ok(h, w, dy, layout, order)
        # the product blocks' weight gradients from node-level data (N rows, no gathers) where the forward left the pair sums
        node_weight = (sums is not None and NODE_LEVEL_WEIGHT and dy.data_ptr() % 16 == 0 and _ld(dw) % 4 == 0 and h.data_ptr() % 16 == 0
                       and bool(lib.ihg_node_interact_bwd_weight_supported(dim, order, _ld(h), _ld(sums), _ld(dy))))
        if node_weight:
            ws_w = _workspace(int(lib.ihg_node_interact_bwd_weight_workspace_bytes(dim, order)), h.device)
            with profiler.kernel('node_interact_bwd_weight', layout.node_count, dim):
                _lib.check(lib.ihg_node_interact_bwd_weight(_ptr(h), _ld(h), _ptr(sums), _ld(sums), _ptr(dy), _ld(dy), _ptr(out_scale), order, _type_begin(layout),
                                                            _ptr(dw), _ld(dw), _ptr(ws_w), ws_w.numel() * 4, dim, _stream()), 'ihg_node_interact_bwd_weight')
        del sums
        if gathered:
            csr_qi, qi_rows = layout.member_csr_qi()
            # with the weight gradients taken at node level nobody but the first-order scatter would read the hyperedges' cotangents: then that gradient is
            # the two-hop operator applied to the node-level cotangent and the [E, d] rows are not stored at all.  (Round 3 measured the opposite order - 8.81 against
            # 8.74 ms per C3 step; since the member-gradient kernel became bound by its memory traffic the 1.1 GB it no longer writes are worth more than the two-hop
            # launch costs over the scatter: C3 7.89 -> 7.78 ms, C2 1.92 -> 1.87, C4 10.59 -> 10.45; member kernel 1,426-1,467 -> 1,188 us, first-order gradient 672-681 -> 823-833)
            keep_dout = not node_weight
            dout = torch.empty(n_edges, dim, dtype=torch.float32, device=h.device) if keep_dout else None
            g2 = torch.empty(n_edges, 2 * dim, dtype=torch.float32, device=h.device)
            dh = torch.empty(layout.node_count, dim, dtype=torch.float32, device=h.device)
            _zero_isolated_users(dh, layout)
            ws = _workspace(int(lib.ihg_interact_bwd_workspace_bytes(n_edges, dim, order)), h.device)
            with profiler.kernel('interact_bwd', n_edges, dim):
                _lib.check(lib.ihg_interact_bwd_gathered(_ptr(h), _ld(h), _ptr(layout.i3), _ptr(w), _ld(w), order, _ptr(dy), _ld(dy), _ptr(out_scale),
                                                         _ptr(dout) if keep_dout else None, dim, _ptr(g2), _ptr(dh), dim, None if node_weight else _ptr(dw), _ld(dw), _ptr(ws), ws.numel() * 4,
                                                         n_edges, dim, _stream()), 'ihg_interact_bwd_gathered')
            node_segment_sum_raw(g2.view(2 * n_edges, dim), csr_qi, out=dh, rows=qi_rows, role='k7.member_gradients', read_once=True)
            del g2
            if keep_dout:
                dp = node_segment_sum_raw(dout, layout.node_csr, role='k7.first_order_gradient')
            else:
                dp = _two_hop_first_order_gradient(dy, layout, out_scale)
        else:
            two_hop_first = n_edges * dim * 4 > FIRST_ORDER_TWO_HOP_BYTES
            inv = None
            if (COTANGENT_PLANES and node_weight and two_hop_first and dy.data_ptr() % 16 == 0 and bool(lib.ihg_edge_gather_sum_planes_supported(dim, _ld(dy)))
                    and bool(lib.ihg_interact_bwd_user_reduced_planes_supported(dim, order, _ld(h)))):
                # only the member-gradient kernel reads the hyperedges' cotangents here: K5 writes them as that kernel's operand (same bytes per row), scaled and split once
                # instead of once per column part of the kernel
                dout = torch.empty(n_edges, dim, dtype=torch.float32, device=h.device)
                if _user_reduced_ok(h, w, dout, layout, order):
                    inv = torch.empty(n_edges, dtype=torch.float32, device=h.device)
                    with profiler.kernel('edge_gather_sum', n_edges, dim):
                        _lib.check(lib.ihg_edge_gather_sum_planes(_ptr(dy), _ld(dy), _ptr(layout.i3), _ptr(out_scale), _ptr(layout.edge_weight), _ptr(dout), _ptr(inv), n_edges,
                                                                  dim, _stream()), 'ihg_edge_gather_sum_planes')
                else:
                    del dout
            if inv is None:
                dout = edge_gather_sum_raw(dy, layout.i3, out_scale, None, 1.0, edge_scale=layout.edge_weight)      # (x m_e: the cotangent of all copies of the row)
            if two_hop_first:
                # a [E, d] table far beyond the caches (config C5: 51 GB): its scatter reads HBM at random, the two-hop operator on the node-level
                # cotangent (10 GB) gathers twice the rows and is still the shorter launch (22 against 32 ms)
                dp = _two_hop_first_order_gradient(dy, layout, out_scale)
            else:
                # the scatter of dout goes first: K5 has just written it, so most of its rows are still in the Infinity Cache for these random
                # reads; the interact kernels read it as a stream and do not care
                dp = node_segment_sum_raw(dout, layout.node_csr, role='k7.first_order_gradient')
            dh = _interact_backward(h, w, dout, layout, order, None if node_weight else dw, inv_scale=inv)
            del inv
        del dout
        dbias = torch.empty(dim, dtype=torch.float32, device=h.device) if ctx.has_bias else None
        ws2 = _workspace(int(lib.ihg_node_linear_workspace_bytes(dim)), h.device)
        accumulate = bool(lib.ihg_node_linear_bwd_accumulates(dim, _ld(dp), _ld(h), _ld(dh))) and h.data_ptr() % 16 == 0 and dh.data_ptr() % 16 == 0 and dp.data_ptr() % 16 == 0
        dx = dh if accumulate else torch.empty_like(dh)
        with profiler.kernel('node_linear_bwd', h.shape[0], dim):
            _lib.check(lib.ihg_node_linear_bwd_weight(_ptr(dp), _ld(dp), _ptr(h), _ld(h), _type_begin(layout), _ptr(dw), int(dw.stride(0)), dim,
                                                      _ptr(dbias), 0b001, 0, _ptr(w), int(w.stride(0)), _ptr(dx), _ld(dx), 1 if accumulate else 0,
                                                      _ptr(ws2), ws2.numel() * 4, dim, _stream()), 'ihg_node_linear_bwd_weight')
        if not accumulate:
            dh.add_(dx)
        return dh, dw, dbias, None, None, None, None, None


def interact_layer(h: Tensor, w: Tensor, bias: Optional[Tensor], layout: IncidenceLayout, order: int, out_scale: Optional[Tensor] = None,
                   rows: Optional[Tensor] = None, out: Optional[Tensor] = None) -> Tensor:
    """``out_scale * H FeatureInteractor(h)`` for interaction orders 2 / 3 with ``w = [A_u | A_q | A_i | W_uq | W_qi | W_iu (| W_uqi)]`` and
    ``bias`` = the aggregation's bias: ``node_segment_sum(interact(h, first_order(h), w), out_scale, rows)`` as one differentiable op."""
    if order not in (2, 3):
        raise ValueError('interact_layer handles interaction orders 2 and 3')
    dim, wide = int(h.shape[1]), padded_width(int(h.shape[1]))
    if wide != dim and int(w.shape[0]) == dim and h.is_cuda:
        # a width between the tiled ones: the same layer at the next tiled width on zero-padded operands (the padding columns of the result are exactly zero and are cut off)
        if out is not None:
            _check_out(out, h, w, bias)
            out.copy_(_InteractLayer.apply(pad_columns(h, wide), pad_blocks(w, wide), pad_vector(bias, wide), layout, int(order), out_scale, rows, None)[:, :dim])
            return out
        return _InteractLayer.apply(pad_columns(h, wide), pad_blocks(w, wide), pad_vector(bias, wide), layout, int(order), out_scale, rows, None)[:, :dim]
    return _InteractLayer.apply(h, w, bias, layout, int(order), out_scale, rows, out)


class _InteractFromNodes(torch.autograd.Function):
    """First-order blocks (typed row-GEMM) + product blocks (MFMA interact kernel) of ``FeatureInteractor`` as ONE autograd
    node, so that in the backward the first-order path's contribution to ``d h`` is accumulated by the kernel into the member
    gradients' scatter result (no separate ``[N, d]`` add) and both halves of ``d aggregation.weight`` land in one tensor."""

    @staticmethod
    def forward(ctx, h: Tensor, w: Tensor, bias: Optional[Tensor], layout: IncidenceLayout, order: int) -> Tensor:
        lib = _lib.load()
        h, w = _rows(h, 'h'), _rows(w, 'w')
        dim = int(h.shape[1])
        p = torch.empty_like(h)
        ws = _workspace(int(lib.ihg_node_linear_workspace_bytes(dim)), h.device)
        with profiler.kernel('node_linear_fwd', h.shape[0], dim):
            _lib.check(lib.ihg_node_linear_fwd(_ptr(h), _ld(h), _ptr(w), int(w.stride(0)), dim, _ptr(bias), 0b001, 0, _type_begin(layout),
                                               _ptr(p), _ld(p), _ptr(ws), ws.numel() * 4, dim, _stream()), 'ihg_node_linear_fwd')
        out = torch.empty(layout.edge_count, dim, dtype=torch.float32, device=h.device)
        ws2 = _workspace(int(lib.ihg_interact_fwd_workspace_bytes(layout.edge_count, dim, order)), h.device)
        with profiler.kernel('interact_fwd', layout.edge_count, dim):
            _lib.check(lib.ihg_interact_fwd(_ptr(h), _ld(h), _ptr(p), _ld(p), _ptr(layout.i3), _ptr(w), _ld(w), order,
                                            _ptr(out), _ld(out), _ptr(ws2), ws2.numel() * 4, layout.edge_count, dim, _stream()),
                       'ihg_interact_fwd')
        ctx.save_for_backward(h, w)
        ctx.layout, ctx.order, ctx.has_bias = layout, order, bias is not None
        return out

    @staticmethod
    def backward(ctx, grad_out: Tensor):
        lib = _lib.load()
        h, w = ctx.saved_tensors
        layout, order = ctx.layout, ctx.order
        grad_out = _rows(grad_out, 'grad_out')
        n_edges, dim = layout.edge_count, int(h.shape[1])
        dw = torch.empty_like(w)                               # product blocks from the interact kernels, first-order blocks from the row-GEMM pass
        # the scatter of grad_out goes first: K5 has just written it, so most of its 256-byte rows are still in the Infinity Cache
        # for these random reads; the interact kernels read it as a stream and do not care
        dp = node_segment_sum_raw(grad_out, layout.node_csr, role='k7.first_order_gradient')
        dh = _interact_backward(h, w, grad_out, layout, order, dw)
        dbias = torch.empty(dim, dtype=torch.float32, device=h.device) if ctx.has_bias else None
        ws2 = _workspace(int(lib.ihg_node_linear_workspace_bytes(dim)), h.device)
        with profiler.kernel('node_linear_bwd', h.shape[0], dim):
            _lib.check(lib.ihg_node_linear_bwd_weight(_ptr(dp), _ld(dp), _ptr(h), _ld(h), _type_begin(layout), _ptr(dw), int(dw.stride(0)), dim,
                                                      _ptr(dbias), 0b001, 0, _ptr(w), int(w.stride(0)), _ptr(dh), _ld(dh), 1,
                                                      _ptr(ws2), ws2.numel() * 4, dim, _stream()), 'ihg_node_linear_bwd_weight')
        return dh, dw, dbias, None, None


def interact_from_nodes_supported(h: Tensor, w: Tensor) -> bool:
    """The one-node form needs the fused row-GEMM backward (d = 64, aligned rows) and the tiled interact kernels."""
    return node_linear_tiled(h, w) and int(h.shape[1]) == 64 and h.is_contiguous()


def interact_from_nodes(h: Tensor, w: Tensor, bias: Optional[Tensor], layout: IncidenceLayout, order: int) -> Tensor:
    """``FeatureInteractor`` of order 2 / 3 from the node features: ``out[e] = sum_m (h[m] A_m^T) + c + sum_b W_b z_b[e]``
    with ``w = [A_u | A_q | A_i | W_uq | W_qi | W_iu (| W_uqi)]`` (``CommonLayers.py:70-85``)."""
    if order not in (2, 3):
        raise ValueError('interact_from_nodes handles interaction orders 2 and 3')
    return _InteractFromNodes.apply(h, w, bias, layout, int(order))


def interact(h: Tensor, p: Tensor, w: Tensor, layout: IncidenceLayout, order: int) -> Tensor:
    """Interactive node -> hyperedge step: first-order part from ``p`` (hoisted), products contracted with ``w``."""
    if order not in (2, 3):
        raise ValueError('interact handles interaction orders 2 and 3; order 1 is edge_gather_sum on the hoisted features')
    return _Interact.apply(h, p, w, layout, int(order))


# ---------------------------------------------------------------------------------------------
# Batch tail: HEM scores of a training batch straight from the layer outputs
# ---------------------------------------------------------------------------------------------
def _hem_row_gradients(layers, rows: Tensor, items: Tensor, bias: Tensor, lam: float, dscores: Tensor, grad_scale: float, tables=None, grad_scale_device=None, rows_upper=None):
    """Per-batch-row gradients of the tail: ``[3B, (L+1) d + 4]``, layer l in columns ``l d .. (l+1) d``, d bias in column ``(L+1) d``.  ``tables`` (a resolved
    ``NodeTables``): layer 0 is read from the embedding tables in place and ``layers`` are the outputs of the layers above it."""
    lib = _lib.load()
    batch = int(items.shape[0])
    dim = int(layers[0].shape[1]) if layers else tables.dim
    n_layers = len(layers) + (1 if tables is not None else 0)
    width = n_layers * dim
    rowgrad = torch.empty(3 * batch, width + 4, dtype=torch.float32, device=bias.device)
    if grad_scale_device is not None and (grad_scale_device.dtype != torch.float32 or grad_scale_device.numel() != 1):
        raise TypeError('grad_scale_device is a float32 device scalar')
    with profiler.kernel('hem_score_bwd', batch, dim):
        if tables is None:
            ptrs = (ctypes.c_void_p * n_layers)(*[x.data_ptr() for x in layers])
            _lib.check(lib.ihg_hem_score_bwd_typed0(ptrs, n_layers, _ld(layers[0]), dim, None, 0, None, _ptr(rows), _ptr(rows_upper), _ptr(dscores), _ptr(grad_scale_device),
                                                    float(grad_scale), float(lam), _ptr(rowgrad), width + 4, batch, _stream()), 'ihg_hem_score_bwd_typed0')
        else:
            ptrs = (ctypes.c_void_p * n_layers)(tables.query_rows.data_ptr(), *[x.data_ptr() for x in layers])
            _lib.check(lib.ihg_hem_score_bwd_typed0(ptrs, n_layers, _ld(layers[0]) if layers else dim, dim, tables.row_pointers(), dim, tables.type_begin(), _ptr(rows),
                                                    _ptr(rows_upper), _ptr(dscores), _ptr(grad_scale_device), float(grad_scale), float(lam), _ptr(rowgrad), width + 4, batch, _stream()),
                       'ihg_hem_score_bwd_typed0')
    return rowgrad


SCATTER_CHUNK_ROWS = 32768          # rows one ihg_batch_scatter_add / ihg_batch_combine launch takes (= ihg_batch_scatter_max_rows(): its id list lives in LDS; tests/test_abi.py)


def _scatter_rows(rowgrad: Tensor, col0: int, width: int, rows: Tensor, dense: Optional[Tensor], tail: Optional[Tensor] = None, tail_offset: int = 0):
    """``dense[rows[k]] += rowgrad[k, col0 : col0 + width]`` (duplicates combined in a fixed order; deterministic).  With
    ``tail``: the LAST of the ``width`` columns goes to ``tail[rows[k] - tail_offset]`` instead.  Batches beyond one launch's
    capacity go through in row chunks, in order (same sums, associated chunk by chunk)."""
    lib = _lib.load()
    n = int(rows.shape[0])
    target = dense if dense is not None else tail
    block = int(dense.shape[1]) if dense is not None else 1
    for lo in range(0, n, SCATTER_CHUNK_ROWS):
        hi = min(lo + SCATTER_CHUNK_ROWS, n)
        src = rowgrad[lo:hi, col0:]
        with profiler.kernel('batch_scatter_add', hi - lo, width):
            _lib.check(lib.ihg_batch_scatter_add(_ptr(src), int(rowgrad.stride(0)), width, _ptr(rows[lo:hi]), hi - lo, _ptr(target),
                                                 _ld(target) if dense is not None else 1, block, 0, _ptr(tail), int(tail_offset),
                                                 int(tail.shape[0]) if tail is not None else 0, _stream()), 'ihg_batch_scatter_add')


def _hem_backward(layers, rows: Tensor, items: Tensor, bias: Tensor, lam: float, dscores: Tensor, grad_scale: float, item_row_offset: int):
    """Backward of the batch tail without taps: per-row gradients (one kernel), then ONE deterministic scatter that lands every
    layer's gradient in its own contiguous ``[N, d]`` matrix and d bias beside them.  -> (d bias, layer gradients)."""
    lib = _lib.load()
    batch, dim, n_layers = int(items.shape[0]), int(layers[0].shape[1]), len(layers)
    width = n_layers * dim
    n_nodes = int(layers[0].shape[0])
    rowgrad = _hem_row_gradients(layers, rows, items, bias, lam, dscores, grad_scale)
    n_bias = int(bias.shape[0])
    flat = torch.zeros(n_layers * n_nodes * dim + n_bias, dtype=torch.float32, device=bias.device)
    dense = flat[:n_layers * n_nodes * dim].view(n_layers, n_nodes, dim)
    dbias = flat[n_layers * n_nodes * dim:]
    for lo in range(0, 3 * batch, SCATTER_CHUNK_ROWS):        # one launch lands every layer's block and the bias column; big batches in row chunks
        hi = min(lo + SCATTER_CHUNK_ROWS, 3 * batch)
        with profiler.kernel('batch_scatter_add', hi - lo, width + 1):
            _lib.check(lib.ihg_batch_scatter_add(_ptr(rowgrad[lo:hi]), width + 4, width + 1, _ptr(rows[lo:hi]), hi - lo, _ptr(dense), dim, dim, n_nodes * dim,
                                                 _ptr(dbias), int(item_row_offset), n_bias, _stream()), 'ihg_batch_scatter_add')
    return dbias, tuple(dense[l] for l in range(n_layers))


class TailGradients:
    """Side channel from the batch tail's backward to the taps on the layer outputs (one per training step).

    A layer output ``X_l`` feeds the next layer AND the batch tail; the tail's gradient for it has 3B non-zero rows out of N.
    Handing autograd a dense ``[N, d]`` tensor per layer costs a zero fill plus one full add per layer; instead the tail's
    backward leaves its ``[3B, .]`` row gradients here and every ``tap`` adds its own columns into the gradient that came
    down from the next layer, in place."""

    def __init__(self, exchange=None, grad_scale: float = 1.0):
        self.rows: Optional[Tensor] = None
        self.rowgrad: Optional[Tensor] = None      # [3B, .], rows of equal destination already summed into their first occurrence
        self.leader: Optional[Tensor] = None       # int32 [3B]: 1 on those first occurrences (None: rowgrad is not combined)
        # data parallel, cotangent exchange (ihgnn_amd.distributed.CotangentSync): ``exchange(rows, rowgrad) -> (rows, rowgrad)`` of ALL ranks' batches, called by the
        # batch tail's backward between its row-gradient kernel and the combine; ``grad_scale`` = 1 / world size (the average over ranks) applied by that kernel
        self.exchange = exchange
        self.grad_scale = float(grad_scale)
        # a layout that numbers its nodes without the isolated ones (IncidenceLayout.node_map): layer 0 is addressed by the public rows (``rows``), the layers above by
        # ``rows_upper`` = row_map[rows] (-1: an isolated node, whose rows above layer 0 are zero constants - the add / put kernels skip it)
        self.row_map: Optional[Tensor] = None
        self.rows_upper: Optional[Tensor] = None

    def _rows_of(self, upper: bool) -> Tensor:
        return self.rows_upper if (upper and self.rows_upper is not None) else self.rows

    def put_into_typed(self, dense_rows, ld_dense: int, layout, col0: int, width: int, type_begin=None) -> None:
        """``add_into`` for a destination whose node types start at their own addresses (``dense_rows``: host array of three device pointers; ``type_begin``: the first row
        of every type in the destination's numbering when that is not the layout's - the embedding tables under a layout without the isolated nodes)."""
        _put_rows(self, dense_rows, ld_dense, layout, col0, width, False, False, type_begin)

    def assign_into(self, dense: Tensor, layout, col0: int, width: int, upper: bool = False) -> None:
        """``dense[rows[k]] = rowgrad[k, col0 : col0 + width]`` on the (combined) batch rows, nothing else written: a gradient that is zero outside the batch rows
        and READ at those rows only (the last layer's cotangent under the sparse pull) needs no ``[N, d]`` zero fill."""
        base = dense.data_ptr()
        tb = _type_begin(layout)
        step = _ld(dense) * 4
        _put_rows(self, (ctypes.c_void_p * 3)(base + tb[0] * step, base + tb[1] * step, base + tb[2] * step), _ld(dense), layout, col0, width, True, upper)

    def add_into(self, dense: Optional[Tensor], col0: int, width: int, tail: Optional[Tensor] = None, tail_offset: int = 0, upper: bool = False) -> None:
        """``dense[rows[k]] += rowgrad[k, col0 : col0 + width]`` (or ``tail[rows[k] - tail_offset] += rowgrad[k, col0]``); ``upper``: the destination is a layer above layer 0
        (addressed by ``rows_upper`` where the layout's numbering differs from the public one)."""
        rows = self._rows_of(upper)
        if self.leader is None:
            if upper and self.rows_upper is not None:
                raise _lib.IhgnnHipError('a compact layout needs the combined row gradients (batches of at most 32,768 rows)')
            _scatter_rows(self.rowgrad, col0, width, rows, dense, tail, tail_offset)
            return
        lib = _lib.load()
        n = int(rows.shape[0])
        src = self.rowgrad[:, col0:]
        with profiler.kernel('batch_rows_add', n, width):
            _lib.check(lib.ihg_batch_rows_add(_ptr(src), int(self.rowgrad.stride(0)), width, _ptr(rows), _ptr(self.leader), n, _ptr(dense),
                                              _ld(dense) if dense is not None else 0, _ptr(tail), int(tail_offset),
                                              int(tail.shape[0]) if tail is not None else 0, _stream()), 'ihg_batch_rows_add')


def _put_rows(holder: 'TailGradients', dense_rows, ld_dense: int, layout, col0: int, width: int, assign: bool, upper: bool = False, type_begin=None) -> None:
    lib = _lib.load()
    if holder.leader is None:
        raise _lib.IhgnnHipError('typed / assigning row scatter needs the combined row gradients (batches of at most 32,768 rows)')
    rows = holder._rows_of(upper)
    n = int(rows.shape[0])
    src = holder.rowgrad[:, col0:]
    with profiler.kernel('batch_rows_add', n, width):
        _lib.check(lib.ihg_batch_rows_put(_ptr(src), int(holder.rowgrad.stride(0)), width, _ptr(rows), _ptr(holder.leader), n, dense_rows, ld_dense,
                                          type_begin if type_begin is not None else _type_begin(layout), 1 if assign else 0, _stream()), 'ihg_batch_rows_put')


class _Tap(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x: Tensor, holder: TailGradients, index: int, sparse_layout):
        ctx.holder, ctx.index, ctx.sparse_layout = holder, int(index), sparse_layout
        ctx.shape, ctx.device = tuple(x.shape), x.device
        ctx.set_materialize_grads(False)
        return x.view_as(x), x.view_as(x)

    @staticmethod
    def backward(ctx, g_next: Optional[Tensor], g_tail: Optional[Tensor]):
        holder, (n, dim) = ctx.holder, ctx.shape
        if holder.rowgrad is None:                           # the tail did not run a backward through this holder
            if g_next is None or g_tail is None:
                return (g_next if g_next is not None else g_tail), None, None, None
            return g_next + g_tail, None, None, None
        if g_next is None:
            # the last layer's output feeds the batch tail only: its cotangent is zero outside the batch rows (RawGnn tells that layer so)
            if ctx.sparse_layout is not None and holder.leader is not None:
                # ... and that layer's backward READS it at those rows only (the masked pull of node_two_hop): the rows are written, nothing is filled
                g = torch.empty(n, dim, dtype=torch.float32, device=ctx.device)
                if ctx.index > 0 and holder.rows_upper is not None:
                    # (a layout without the isolated nodes lists row 0 in place of every isolated batch node - RawGnn.propagate_layers -: the pull reads that row, so it
                    #  holds zeros unless it is a batch row itself, in which case the assignment below overwrites it)
                    _lib.check(_lib.load().ihg_zero_floats(_ptr(g), dim, _stream()), 'ihg_zero_floats')
                holder.assign_into(g, ctx.sparse_layout, ctx.index * dim, dim, upper=ctx.index > 0)
                return g, None, None, None
            g = torch.empty(n, dim, dtype=torch.float32, device=ctx.device)
            _lib.check(_lib.load().ihg_zero_floats(_ptr(g), n * dim, _stream()), 'ihg_zero_floats')
        else:
            g = g_next if (g_next.is_contiguous() and g_next.dtype == torch.float32) else g_next.contiguous().float()
        holder.add_into(g, ctx.index * dim, dim, upper=ctx.index > 0)
        return g, None, None, None


def tap(x: Tensor, holder: TailGradients, index: int, sparse_layout=None):
    """``(x for the next layer, x for the batch tail)``: same values; see ``TailGradients``.  ``sparse_layout`` (an ``IncidenceLayout``; only for an output that
    feeds NOTHING but the tail): the caller's promise that the backward of the op that produced ``x`` reads its cotangent at the batch rows only (the last layer
    under ``cotangent_rows`` / ``rows``) - the tap then writes those rows of an uninitialised ``[N, d]`` tensor instead of filling it with zeros first."""
    return _Tap.apply(x, holder, int(index), sparse_layout)


_ZERO_SCALARS = {}


def _zero_like_expanded(shape, device: torch.device) -> Tensor:
    """A zero tensor of ``shape`` that owns one element (stride 0): placeholder gradient, never read."""
    z = _ZERO_SCALARS.get(device)
    if z is None:
        z = _ZERO_SCALARS[device] = torch.zeros((), dtype=torch.float32, device=device)
    return z.expand(*shape)


def _same_layout(layers):
    layers = tuple(_rows(x, 'layer output') for x in layers)
    ld = _ld(layers[0])
    if any(_ld(x) != ld or x.shape[1] != layers[0].shape[1] for x in layers):
        layers = tuple(x.contiguous() for x in layers)
    return layers


class _HemScore(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rows: Tensor, items: Tensor, bias: Tensor, lam: float, item_row_offset: int, *layers: Tensor) -> Tensor:
        lib = _lib.load()
        layers = _same_layout(layers)
        batch, dim = int(items.shape[0]), int(layers[0].shape[1])
        ptrs = (ctypes.c_void_p * len(layers))(*[x.data_ptr() for x in layers])
        scores = torch.empty(batch, dtype=torch.float32, device=bias.device)
        with profiler.kernel('hem_score_fwd', batch, dim):
            _lib.check(lib.ihg_hem_score_fwd(ptrs, len(layers), _ld(layers[0]), dim, _ptr(rows), _ptr(items), _ptr(bias), float(lam), _ptr(scores),
                                             batch, _stream()), 'ihg_hem_score_fwd')
        ctx.save_for_backward(rows, items, bias, *layers)
        ctx.lam, ctx.offset = float(lam), int(item_row_offset)
        return scores

    @staticmethod
    def backward(ctx, dscores: Tensor):
        rows, items, bias, *layers = ctx.saved_tensors
        dbias, grads = _hem_backward(layers, rows, items, bias, ctx.lam, dscores.contiguous(), 1.0, ctx.offset)
        return (None, None, dbias, None, None) + grads


class _HemBceLoss(torch.autograd.Function):
    """Scores + mean BCE-with-logits in the forward; d loss / d scores is produced there too, so the backward starts at the
    per-row gradient kernel.  With a ``TailGradients`` holder the layer gradients leave through the taps (placeholders are
    returned here); without one they are returned dense."""

    @staticmethod
    def forward(ctx, rows: Tensor, items: Tensor, labels: Tensor, bias: Tensor, lam: float, item_row_offset: int, holder, tables, rows_upper, *layers: Tensor) -> Tensor:
        # tables (a resolved NodeTables): layer 0 is the embedding tables in place; layers[0] is then its token (the autograd edge), not a matrix
        lib = _lib.load()
        real = _same_layout(layers[1:] if tables is not None else layers)
        batch, dim = int(items.shape[0]), int(real[0].shape[1]) if real else tables.dim
        scores = torch.empty(batch, dtype=torch.float32, device=bias.device)
        dscores = torch.empty(batch, dtype=torch.float32, device=bias.device)
        loss = torch.empty((), dtype=torch.float32, device=bias.device)
        labels = labels.to(torch.float32).contiguous()
        with profiler.kernel('hem_score_fwd', batch, dim):
            if tables is None and rows_upper is None:
                ptrs = (ctypes.c_void_p * len(real))(*[x.data_ptr() for x in real])
                _lib.check(lib.ihg_hem_score_fwd(ptrs, len(real), _ld(real[0]), dim, _ptr(rows), _ptr(items), _ptr(bias), float(lam), _ptr(scores),
                                                 batch, _stream()), 'ihg_hem_score_fwd')
            elif tables is None:
                # a layout without the isolated nodes: layer 0 is a matrix in the public numbering (rows), the layers above are in the layout's (rows_upper; -1: zero row)
                ptrs = (ctypes.c_void_p * len(real))(*[x.data_ptr() for x in real])
                _lib.check(lib.ihg_hem_score_fwd_typed0(ptrs, len(real), _ld(real[0]), dim, None, 0, None, _ptr(rows), _ptr(rows_upper), _ptr(items), _ptr(bias), float(lam),
                                                        _ptr(scores), batch, _stream()), 'ihg_hem_score_fwd_typed0')
            else:
                ptrs = (ctypes.c_void_p * (len(real) + 1))(tables.query_rows.data_ptr(), *[x.data_ptr() for x in real])     # (slot 0 is replaced by the typed rows)
                _lib.check(lib.ihg_hem_score_fwd_typed0(ptrs, len(real) + 1, _ld(real[0]) if real else dim, dim, tables.row_pointers(), dim, tables.type_begin(),
                                                        _ptr(rows), _ptr(rows_upper), _ptr(items), _ptr(bias), float(lam), _ptr(scores), batch, _stream()), 'ihg_hem_score_fwd_typed0')
            _lib.check(lib.ihg_bce_with_logits(_ptr(scores), _ptr(labels), batch, _ptr(loss), _ptr(dscores), _stream()), 'ihg_bce_with_logits')
        ctx.save_for_backward(rows, items, bias, dscores, *real)
        ctx.rows_upper = rows_upper
        ctx.lam, ctx.offset, ctx.holder, ctx.tables = float(lam), int(item_row_offset), holder, tables
        ctx.token_shape = tuple(layers[0].shape) if tables is not None else None
        return loss

    @staticmethod
    def backward(ctx, grad_loss: Tensor):
        rows, items, bias, dscores, *layers = ctx.saved_tensors
        tables = ctx.tables
        if ctx.holder is None:
            if tables is not None or ctx.rows_upper is not None:
                raise _lib.IhgnnHipError('hem_bce_loss over NodeTables / a compact layout needs a TailGradients holder')
            dbias, grads = _hem_backward(layers, rows, items, bias, ctx.lam, dscores * grad_loss, 1.0, ctx.offset)
            return (None, None, None, dbias, None, None, None, None, None) + grads
        holder = ctx.holder
        rowgrad = _hem_row_gradients(layers, rows, items, bias, ctx.lam, dscores, holder.grad_scale, tables, grad_loss.contiguous(), ctx.rows_upper)     # d loss stays on the device: no host read, no multiply launch
        lib = _lib.load()
        if holder.exchange is not None:
            # every rank's propagation is the same function of the same parameters and its backward is linear in the cotangent of the layer outputs, which is non-zero on
            # the batch rows only: the ranks exchange THOSE rows (3B x (D + 1) floats each) and every rank runs the one propagation backward on the union - the averaged
            # gradient of all parameters without a dense all-reduce (RawGnn.py:122-142: the batch reads F at 3B rows).
            # A rank's own duplicates are summed BEFORE the exchange (a batch repeats every positive's user and query with its ten negatives: ~ 1,300 of 3,300 rows are
            # first occurrences) and marked in a spare column of the row gradients, which travels with them: the union's combine then skips the other rows (and the
            # zero rows that pad a shorter batch) - its duplicate search is quadratic in the rows that take part
            n_own = int(rows.shape[0])
            width_own = (len(layers) + (1 if tables is not None else 0)) * (int(layers[0].shape[1]) if layers else tables.dim)
            if lib.ihg_batch_scatter_workspace_bytes(n_own) >= 0:
                own = torch.empty(n_own, dtype=torch.int32, device=rows.device)
                with profiler.kernel('batch_combine', n_own, width_own + 1):
                    _lib.check(lib.ihg_batch_combine(_ptr(rowgrad), int(rowgrad.stride(0)), width_own + 1, _ptr(rows), n_own, n_own // 3, _ptr(own), _stream()), 'ihg_batch_combine')
                rowgrad[:, width_own + 1] = own.to(torch.float32)
            else:
                rowgrad[:, width_own + 1] = 1.0
            rows, rowgrad = holder.exchange(rows, rowgrad)
            rows = torch.where(rowgrad[:, width_own + 1] > 0, rows, torch.full_like(rows, -1))
        holder.rows, holder.rowgrad, holder.leader = rows, rowgrad, None
        # (the union's rows after an exchange; this batch's otherwise - through the layout's map where it numbers its nodes without the isolated ones)
        if holder.row_map is None:
            holder.rows_upper = None
        elif holder.exchange is None and ctx.rows_upper is not None:
            holder.rows_upper = ctx.rows_upper
        else:
            mapped = holder.row_map[rows.clamp_min(0)]
            holder.rows_upper = torch.where(rows < 0, torch.full_like(mapped, -1), mapped)
        lib = _lib.load()
        n_layers = len(layers) + (1 if tables is not None else 0)
        n, width = int(rows.shape[0]), n_layers * (int(layers[0].shape[1]) if layers else tables.dim)
        if lib.ihg_batch_scatter_workspace_bytes(n) >= 0:    # one pass sums duplicate destinations; the taps then add plain rows
            holder.leader = torch.empty(n, dtype=torch.int32, device=rows.device)
            with profiler.kernel('batch_combine', n, width + 1):
                _lib.check(lib.ihg_batch_combine(_ptr(rowgrad), int(rowgrad.stride(0)), width + 1, _ptr(rows), n, n // 3, _ptr(holder.leader), _stream()),
                           'ihg_batch_combine')
        dbias = torch.empty_like(bias)
        _lib.check(lib.ihg_zero_floats(_ptr(dbias), dbias.numel(), _stream()), 'ihg_zero_floats')
        holder.add_into(None, width, 1, dbias, ctx.offset)
        placeholders = tuple(_zero_like_expanded(x.shape, x.device) for x in layers)
        if tables is not None:
            placeholders = (_zero_like_expanded(ctx.token_shape, bias.device),) + placeholders
        return (None, None, None, dbias, None, None, None, None, None) + placeholders


def score_topk_max_width() -> int:
    """The widest feature row ``ihg_score_topk`` scores (1264): asked of the library, the one place that knows its LDS budget."""
    return int(_lib.load().ihg_score_topk_max_dim())


def score_topk_supported(features: Tensor) -> bool:
    """True when ``ihg_score_topk`` takes this feature matrix: float32 GPU rows of any width up to ``score_topk_max_width()`` (any row stride)."""
    return (features.is_cuda and features.dtype == torch.float32 and features.dim() == 2 and features.stride(1) == 1 and 0 < features.shape[1] <= score_topk_max_width()
            and (features.shape[0] <= 1 or features.stride(0) >= features.shape[1]))


def score_topk(features: Tensor, users: Tensor, queries: Tensor, query_row0: int, item_row0: int, item_bias: Tensor, lam: float, k: int = 10):
    """Evaluation scoring: for every (user, query) pair the ``k`` best items over ALL items and their HEM scores,
    ``(top_items [C, k] int32, top_scores [C, k])``, best first, ties in ascending item order; the ``[C, I]`` score matrix is
    never materialised.  ``features`` = the cached ``[N, D]`` propagation output (any width up to ``score_topk_max_width()``); items are its rows from ``item_row0`` on."""
    lib = _lib.load()
    if not score_topk_supported(features):
        raise _lib.IhgnnHipError(f'ihg_score_topk needs a float32 GPU feature matrix of width <= {score_topk_max_width()}, got {tuple(features.shape)} {features.dtype} on {features.device}')
    n_pairs = int(users.shape[0])
    n_items = int(features.shape[0]) - int(item_row0)
    dim = int(features.shape[1])
    users = users.to(device=features.device, dtype=torch.int64).contiguous()
    queries = queries.to(device=features.device, dtype=torch.int64).contiguous()
    bias = item_bias.detach().to(torch.float32).contiguous()
    top_scores = torch.empty(n_pairs, k, dtype=torch.float32, device=features.device)
    top_items = torch.empty(n_pairs, k, dtype=torch.int32, device=features.device)
    ws = _workspace(int(lib.ihg_score_topk_workspace_bytes(n_pairs, n_items, dim)), features.device)
    with profiler.kernel('score_topk', n_pairs, dim):
        _lib.check(lib.ihg_score_topk(_ptr(features), _ld(features), dim, int(query_row0), int(item_row0), n_items, _ptr(bias),
                                      _ptr(users), _ptr(queries), float(lam), n_pairs, int(k), _ptr(top_scores), _ptr(top_items), _ptr(ws),
                                      ws.numel() * 4, _stream()), 'ihg_score_topk')
    return top_items, top_scores


def batch_node_rows(users: Tensor, queries: Tensor, items: Tensor, query_row0: int, item_row0: int) -> Tensor:
    """``torch.cat([users, queries + query_row0, items + item_row0])`` (``RawGnn.py:128-131``): the global node rows of a batch, int64 ``[3 B]``, one launch."""
    lib = _lib.load()
    if not users.is_cuda:
        return torch.cat([users, queries + query_row0, items + item_row0])
    u, q, i = (t.to(torch.int64).contiguous() for t in (users, queries, items))
    b = int(u.shape[0])
    rows = torch.empty(3 * b, dtype=torch.int64, device=u.device)
    rows32 = torch.empty(3 * b, dtype=torch.int32, device=u.device)
    _lib.check(lib.ihg_batch_node_rows(_ptr(u), _ptr(q), _ptr(i), b, int(query_row0), int(item_row0), _ptr(rows), _ptr(rows32), _stream()), 'ihg_batch_node_rows')
    rows.as_int32 = rows32                                   # the layers' row lists are int32: written by the same launch instead of a cast
    return rows


_UNIT = {}


def backward(loss: Tensor) -> None:
    """``loss.backward()`` with the root gradient taken from a cached ones scalar (autograd otherwise fills a fresh one every step: a framework launch)."""
    one = _UNIT.get(loss.device)
    if one is None:
        one = _UNIT[loss.device] = torch.ones((), dtype=loss.dtype, device=loss.device)
    loss.backward(one)


def hem_score(layers, rows: Tensor, items: Tensor, bias: Tensor, lam: float, item_row_offset: int) -> Tensor:
    """HEM scores of a batch: ``rows`` = global node rows of users, queries, items (``[3B]`` int64), ``items`` = 0-based item
    ids (``[B]``), ``layers`` = the ``[N,d]`` outputs ``X_0..X_L`` whose concatenation the reference scores on."""
    return _HemScore.apply(rows, items, bias, float(lam), int(item_row_offset), *layers)


def hem_bce_loss(layers, rows: Tensor, items: Tensor, labels: Tensor, bias: Tensor, lam: float, item_row_offset: int,
                 holder: Optional[TailGradients] = None, rows_upper: Optional[Tensor] = None) -> Tensor:
    """``nn.BCEWithLogitsLoss()(hem_score(...), labels)`` as one differentiable op (scalar).  ``holder``: the layers are the
    tail halves of ``tap`` outputs made with this holder, and their gradients travel through it (see ``TailGradients``).  ``rows_upper`` (with ``holder.row_map``): the
    layers above layer 0 are numbered by the layout's own node ids (a layout without the isolated nodes): their rows of the batch, -1 = isolated = zero.  ``layers[0]`` may be a
    ``NodeTables`` (resolved by the first layer's transform): the head reads its layer-0 rows from the embedding tables in place."""
    tables = None
    layers = list(layers)
    if layers and isinstance(layers[0], NodeTables):
        tables = layers[0]
        if tables.query_rows is None or tables.token is None:
            raise RuntimeError('hem_bce_loss: the NodeTables were not consumed by a node-level transform (a model without layers?)')
        layers[0] = tables.token
    if rows_upper is not None and holder is not None and holder.row_map is None:
        raise ValueError('rows_upper comes with holder.row_map (the layout\'s public -> own node map)')
    return _HemBceLoss.apply(rows, items, labels, bias, float(lam), int(item_row_offset), holder, tables, rows_upper, *layers)
