"""ctypes binding of libihgnn_hip.so (C ABI: include/ihgnn_hip.h).

There is exactly one implementation of the hot path - the HIP library.  If it cannot be loaded the
first call raises; nothing in this package falls back to PyTorch or CPU code.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_float, c_int32, c_int64, c_void_p
from typing import Optional

# torch ships its own HIP runtime (torch/lib/libamdhip64.so, SONAME libamdhip64.so.7).  It must be in the process
# BEFORE libihgnn_hip.so is opened, so that the library's DT_NEEDED entry resolves to that same runtime: kernels are
# then enqueued on torch's streams, against torch's allocations, by one runtime.  Two runtimes in one process see
# "no ROCm-capable device".
import torch  # noqa: F401  (side effect: loads the HIP runtime)

_HERE = os.path.dirname(os.path.abspath(__file__))
# IHGNN_HIP_LIBRARY points at another build of the same ABI (A/B timing of kernel variants); default: the in-tree library
LIB_PATH = os.environ.get('IHGNN_HIP_LIBRARY') or os.path.join(_HERE, 'csrc', 'libihgnn_hip.so')

ABI_VERSION = 33

OK, ERR_INVALID, ERR_LAUNCH, ERR_WORKSPACE = 0, -1, -2, -3
SCALE_NONE, SCALE_MULTIPLY, SCALE_DIVIDE = 0, 1, 2
SCALE_ACCUMULATE = 0x100      # OR-ed into the mode of ihg_node_segment_sum: out += instead of out =
SRC_READ_ONCE = 0x400         # OR-ed into the mode of ihg_node_segment_sum: every source row of this launch is read exactly once (non-temporal loads)

_i64p, _i32p, _f32p = POINTER(c_int64), POINTER(c_int32), POINTER(c_float)

# name -> (restype, argtypes); mirrors include/ihgnn_hip.h line by line.  Device pointers travel as
# c_void_p (tensor.data_ptr()); host-side builders take real typed pointers.
SIGNATURES = {
    'ihg_abi_version': (c_int32, []),
    'ihg_ablation_build': (c_int32, []),
    'ihg_last_error_string': (c_char_p, []),
    'ihg_build_csr': (ctypes.c_int, [_i64p, c_int64, c_int64, c_int64, c_int64, _i32p, _i32p, _i32p, _f32p]),
    'ihg_parse_search_logs': (ctypes.c_int, [c_char_p, _i64p, _i64p, _i64p, _i64p, c_int64, _i64p, c_int64, _i64p]),
    'ihg_read_graph_info': (ctypes.c_int, [c_char_p, _i64p]),
    'ihg_read_query_bags': (ctypes.c_int, [c_char_p, _i64p, _i64p, _i64p, c_int64, _i64p, c_int64]),
    'ihg_build_log_hypergraph': (ctypes.c_int, [_i64p, _i64p, c_int64, c_int64, c_int64, c_int64, _i32p, _i32p, _f32p, _f32p, _i32p, _i32p, _f32p, _f32p,
                                                _i64p, _i64p]),
    'ihg_build_pair_csr': (ctypes.c_int, [_i64p, c_int64, c_int64, c_int64, c_int64, c_int32, c_int32, _i32p, _i32p, _f32p, _f32p,
                                          c_int64, _i64p]),
    'ihg_transpose_csr': (ctypes.c_int, [_i32p, _i32p, c_int64, c_int64, _i32p, _i32p]),
    'ihg_merge_id_lists': (ctypes.c_int, [_i32p, _i32p, _f32p, c_int64, _i32p, _i32p, _f32p, _i64p]),
    'ihg_unique_triples': (ctypes.c_int, [_i64p, c_int64, c_int64, c_int64, c_int64, _i64p, _f32p, _i32p, _i64p]),
    'ihg_edge_gather_sum': (ctypes.c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_float, c_void_p,
                                           c_void_p, c_int64, c_int64, c_int32, c_void_p]),
    'ihg_edge_gather_sum_planes_supported': (c_int32, [c_int32, c_int64]),
    'ihg_edge_gather_sum_planes': (ctypes.c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_void_p]),
    'ihg_node_segment_sum': (ctypes.c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32,
                                            c_void_p, c_int64, c_int64, c_int32, c_int32,
                                            c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    'ihg_bag_mean_fwd': (ctypes.c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
                                        c_int64, c_int32, c_void_p]),
    'ihg_bag_mean_bwd': (ctypes.c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
                                        c_int64, c_int32, c_void_p]),
    'ihg_interact_fwd_workspace_bytes': (c_int64, [c_int64, c_int32, c_int32]),
    'ihg_interact_fwd': (ctypes.c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int32,
                                        c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int32, c_void_p]),
    'ihg_interact_bwd_workspace_bytes': (c_int64, [c_int64, c_int32, c_int32]),
    'ihg_interact_bwd': (ctypes.c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int32, c_void_p, c_int64,
                                        c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int32, c_void_p]),
    'ihg_interact_bwd_user_reduced_supported': (c_int32, [c_int32, c_int32, c_int64]),
    'ihg_interact_bwd_user_reduced': (ctypes.c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int32, c_void_p, c_int64, c_void_p, c_void_p, c_int64,
                                                     c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int32, c_void_p]),
    'ihg_interact_bwd_user_reduced_planes_supported': (c_int32, [c_int32, c_int32, c_int64]),
    'ihg_interact_bwd_user_reduced_planes': (ctypes.c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
                                                            c_void_p, c_int64, c_int64, c_int32, c_void_p]),
    'ihg_interact_bwd_gathered_supported': (c_int32, [c_int32, c_int32, c_int64, c_int64]),
    'ihg_interact_bwd_gathered': (ctypes.c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int32, c_void_p, c_int64, c_void_p, c_void_p, c_int64,
                                                 c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int32, c_void_p]),
    'ihg_node_pair_sums': (ctypes.c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int32, c_int32,
                                          c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p]),
    'ihg_node_interact_fwd_supported': (c_int32, [c_int32, c_int32, c_int64, c_int64, c_int64]),
    'ihg_node_interact_fwd_workspace_bytes': (c_int64, [c_int32]),
    'ihg_node_interact_fwd': (ctypes.c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32, _i64p,
                                             c_void_p, c_int64, c_void_p, c_int64, c_int32, c_void_p]),
    'ihg_node_interact_bwd_weight_supported': (c_int32, [c_int32, c_int32, c_int64, c_int64, c_int64]),
    'ihg_node_interact_bwd_weight_workspace_bytes': (c_int64, [c_int32, c_int32]),
    'ihg_node_interact_bwd_weight': (ctypes.c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int32, _i64p,
                                                    c_void_p, c_int64, c_void_p, c_int64, c_int32, c_void_p]),
    'ihg_node_linear_workspace_bytes': (c_int64, [c_int32]),
    'ihg_node_linear_bwd_accumulates': (c_int32, [c_int32, c_int64, c_int64, c_int64]),
    'ihg_node_linear_fwd': (ctypes.c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_int32, c_int64, _i64p,
                                           c_void_p, c_int64, c_void_p, c_int64, c_int32, c_void_p]),
    'ihg_node_linear_bwd_input': (ctypes.c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int64, _i64p, c_void_p, c_int64,
                                                 c_void_p, c_int64, c_int32, c_void_p]),
    'ihg_node_linear_bwd_weight': (ctypes.c_int, [c_void_p, c_int64, c_void_p, c_int64, _i64p, c_void_p, c_int64, c_int64,
                                                  c_void_p, c_int32, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int32,
                                                  c_void_p, c_int64, c_int32, c_void_p]),
    'ihg_hem_score_fwd': (ctypes.c_int, [POINTER(c_void_p), c_int32, c_int64, c_int32, c_void_p, c_void_p, c_void_p, c_float,
                                         c_void_p, c_int64, c_void_p]),
    'ihg_hem_score_bwd': (ctypes.c_int, [POINTER(c_void_p), c_int32, c_int64, c_int32, c_void_p, c_void_p, c_float, c_float,
                                         c_void_p, c_int64, c_int64, c_void_p]),
    'ihg_bce_with_logits': (ctypes.c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p]),
    'ihg_batch_scatter_workspace_bytes': (c_int64, [c_int64]),
    'ihg_batch_scatter_max_rows': (c_int32, []),
    'ihg_batch_scatter_add': (ctypes.c_int, [c_void_p, c_int64, c_int32, c_void_p, c_int64, c_void_p, c_int64, c_int32, c_int64,
                                             c_void_p, c_int64, c_int64, c_void_p]),
    'ihg_compose_first_order_fwd': (ctypes.c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_int32, c_void_p]),
    'ihg_compose_first_order_bwd': (ctypes.c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p,
                                                   c_void_p, c_int64, c_void_p, c_int32, c_void_p]),
    'ihg_adam_step': (ctypes.c_int, [c_void_p, c_int32, c_float, c_float, c_float, c_float, c_float, c_int64, c_void_p]),
    'ihg_adam_step_device_scalars': (ctypes.c_int, [c_void_p, c_int32, c_float, c_float, c_float, c_float, c_void_p, c_void_p]),
    'ihg_batch_combine': (ctypes.c_int, [c_void_p, c_int64, c_int32, c_void_p, c_int64, c_int64, c_void_p, c_void_p]),
    'ihg_sample_negatives': (ctypes.c_int, [ctypes.c_uint64, ctypes.c_uint64, c_int64, c_int64, c_int32, c_void_p, c_void_p]),
    'ihg_score_topk_max_dim': (c_int32, []),
    'ihg_score_topk_workspace_bytes': (c_int64, [c_int64, c_int64, c_int32]),
    'ihg_score_topk': (ctypes.c_int, [c_void_p, c_int64, c_int32, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_float, c_int64,
                                      c_int32, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    'ihg_batch_rows_add': (ctypes.c_int, [c_void_p, c_int64, c_int32, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int64,
                                          c_void_p]),
    # typed rows: the first row of every node type at its own address (host arrays of three device pointers)
    'ihg_node_linear_typed_supported': (c_int32, [c_int32, c_int64, c_int64]),
    'ihg_node_linear_fwd_typed': (ctypes.c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_int32, c_int64, _i64p, c_void_p, c_int64,
                                                 c_void_p, c_int64, c_int32, c_void_p]),
    'ihg_node_linear_bwd_weight_typed': (ctypes.c_int, [c_void_p, c_int64, c_void_p, c_int64, _i64p, c_void_p, c_int64, c_int64, c_void_p, c_int32, c_int64,
                                                        c_void_p, c_int64, c_void_p, c_int64, c_int32, c_void_p, c_int64, c_int32, c_void_p]),
    'ihg_hem_score_fwd_typed0': (ctypes.c_int, [c_void_p, c_int32, c_int64, c_int32, c_void_p, c_int64, _i64p, c_void_p, c_void_p, c_void_p, c_void_p, c_float,
                                                c_void_p, c_int64, c_void_p]),
    'ihg_hem_score_bwd_typed0': (ctypes.c_int, [c_void_p, c_int32, c_int64, c_int32, c_void_p, c_int64, _i64p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_float,
                                                c_void_p, c_int64, c_int64, c_void_p]),
    'ihg_batch_rows_put': (ctypes.c_int, [c_void_p, c_int64, c_int32, c_void_p, c_void_p, c_int64, c_void_p, c_int64, _i64p, c_int32, c_void_p]),
    'ihg_zero_floats': (ctypes.c_int, [c_void_p, c_int64, c_void_p]),
    'ihg_mark_rows': (ctypes.c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int32, c_void_p]),
    'ihg_batch_node_rows': (ctypes.c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_void_p]),
    'ihg_zero_rows': (ctypes.c_int, [c_void_p, c_int64, c_int32, c_void_p, c_int64, c_void_p]),
}

_lib: Optional[ctypes.CDLL] = None


class IhgnnHipError(RuntimeError):
    """A libihgnn_hip entry point returned a non-zero status."""


def load() -> ctypes.CDLL:
    """Bind the library (once).  Raises if it is missing, stale in ABI, or lacks a declared symbol."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise IhgnnHipError(
            f'{LIB_PATH} is missing: build it with `python -c "import __graft_entry__ as g; g.build()"` '
            f'(hipcc --offload-arch=gfx950).  ihgnn_amd has no CPU or PyTorch fallback for this path.')
    lib = ctypes.CDLL(LIB_PATH)
    for name, (restype, argtypes) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as exc:
            raise IhgnnHipError(f'{LIB_PATH} does not export {name} (declared in include/ihgnn_hip.h)') from exc
        fn.restype, fn.argtypes = restype, argtypes
    got = lib.ihg_abi_version()
    if got != ABI_VERSION:
        raise IhgnnHipError(f'{LIB_PATH} has ABI version {got}, this package needs {ABI_VERSION}: rebuild it')
    if lib.ihg_ablation_build() and os.environ.get('IHG_ALLOW_ABLATION_BUILD') != '1':
        raise IhgnnHipError(f'{LIB_PATH} is an ablation build (csrc/ablate.hpp: a kernel with part of its work removed, wrong results on purpose); '
                            'only the timing tools load one (IHG_ALLOW_ABLATION_BUILD=1)')
    _lib = lib
    return lib


def last_error() -> str:
    return load().ihg_last_error_string().decode('utf-8', 'replace')


def check(status: int, what: str) -> None:
    if status != OK:
        raise IhgnnHipError(f'{what} failed with status {status}: {last_error()}')
