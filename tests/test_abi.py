"""The C-ABI shared library: loads, exports exactly what include/ihgnn_hip.h declares, rejects bad arguments with
an error string, and the product refuses to run without it or on CPU tensors.  No GPU kernel is launched here."""
import ctypes
import os
import re

import pytest
import torch

from conftest import REPO
from ihgnn_amd import _lib


def declared_symbols():
    text = open(os.path.join(REPO, 'include', 'ihgnn_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(ihg_[a-z0-9_]+)\s*\(', text)))


def test_header_and_binding_agree():
    assert declared_symbols() == sorted(_lib.SIGNATURES)


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(lib, name), f'{name} declared in include/ihgnn_hip.h but not exported'
    assert _lib.load().ihg_abi_version() == _lib.ABI_VERSION
    assert lib.ihg_ablation_build() == 0, 'the shipped library was built with an ablation switch (csrc/ablate.hpp)'


def test_widest_scored_row_is_one_number_everywhere():
    """The evaluation kernel's width limit comes from the library's LDS budget; the model's constructor check and the header quote the same number."""
    from ihgnn_amd.Models import RawGnn
    lib = _lib.load()
    assert lib.ihg_score_topk_max_dim() == RawGnn.MAX_SCORED_WIDTH
    assert f'ihg_score_topk_max_dim() = {RawGnn.MAX_SCORED_WIDTH}' in open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'include', 'ihgnn_hip.h')).read()
    ws = ctypes.c_void_p(16)
    # one past the limit is refused with a message that names the limit (argument checks come before any launch: runs without a GPU)
    rc = lib.ihg_score_topk(ws, 2000, RawGnn.MAX_SCORED_WIDTH + 1, 0, 0, 10, ws, ws, ws, 0.5, 1, 10, ws, ws, ws, 1 << 40, None)
    assert rc != 0 and str(RawGnn.MAX_SCORED_WIDTH) in _lib.last_error()


def test_bad_arguments_return_codes_and_messages():
    lib = _lib.load()
    assert lib.ihg_edge_gather_sum(None, 4, None, None, None, 1.0, None, None, 4, 5, 4, None) == _lib.ERR_INVALID
    assert 'null pointer' in _lib.last_error()
    assert lib.ihg_edge_gather_sum(None, 2, None, None, None, 1.0, None, None, 4, 5, 4, None) == _lib.ERR_INVALID   # ld < dim
    assert lib.ihg_node_segment_sum(None, 4, None, None, None, None, None, None, 7, None, 4, 3, 4, 0, None, None, 0, None, None, 0, None, None, None, None) == _lib.ERR_INVALID
    assert 'out_scale_mode' in _lib.last_error()
    assert lib.ihg_interact_fwd(None, 4, None, 4, None, None, 28, 1, None, 4, None, 0, 3, 4, None) == _lib.ERR_INVALID
    assert 'order' in _lib.last_error()
    assert lib.ihg_interact_fwd(None, 4, None, 4, None, None, 8, 3, None, 4, None, 0, 3, 4, None) == _lib.ERR_INVALID   # ld_w < 7*dim
    # fp32-packed product blocks (3 or 4 of them) + at d = 64 / 128 / 256 the three bf16 planes of four block slots (1.5 x 4 d^2)
    assert lib.ihg_interact_fwd_workspace_bytes(1000, 64, 2) == (3 * 64 * 64 + 6 * 64 * 64) * 4 and lib.ihg_interact_fwd_workspace_bytes(1000, 12, 3) == 0
    assert lib.ihg_interact_fwd_workspace_bytes(1000, 64, 3) == (4 * 64 * 64 + 6 * 64 * 64) * 4 and lib.ihg_interact_fwd_workspace_bytes(1000, 32, 3) == 4 * 32 * 32 * 4
    assert lib.ihg_interact_bwd_workspace_bytes(1000, 64, 3) > lib.ihg_interact_fwd_workspace_bytes(1000, 64, 3)
    # empty problems are fine and launch nothing
    assert lib.ihg_edge_gather_sum(None, 4, None, None, None, 1.0, None, None, 4, 0, 4, None) == _lib.OK
    assert lib.ihg_node_segment_sum(None, 4, None, None, None, None, None, None, 0, None, 4, 0, 4, 0, None, None, 0, None, None, 0, None, None, None, None) == _lib.OK
    with pytest.raises(_lib.IhgnnHipError, match='status -1'):
        _lib.check(_lib.ERR_INVALID, 'probe')


def test_missing_library_is_a_hard_error(monkeypatch):
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', '/nonexistent/libihgnn_hip.so')
    with pytest.raises(_lib.IhgnnHipError, match='no CPU or PyTorch fallback'):
        _lib.load()


def test_ops_refuse_cpu_tensors():
    from ihgnn_amd import ops
    from ihgnn_amd.layout import IncidenceLayout
    import numpy as np
    lay = IncidenceLayout(np.array([[0, 0, 0], [1, 0, 1]]), 2, 1, 2, torch.device('cpu'))
    with pytest.raises(_lib.IhgnnHipError, match='no CPU path'):
        ops.edge_gather_sum(torch.zeros(5, 4), lay)
    with pytest.raises(_lib.IhgnnHipError, match='no CPU path'):
        ops.node_segment_sum(torch.zeros(2, 4), lay)


def test_product_never_imports_the_oracle():
    for root, _, files in os.walk(os.path.join(REPO, 'ihgnn_amd')):
        for fn in files:
            if fn.endswith(('.py', '.hip', '.cpp', '.h')):
                assert 'oracle' not in open(os.path.join(root, fn)).read(), f'{fn} mentions the oracle'


def test_header_is_plain_c():
    """include/ihgnn_hip.h is the contract for non-Python callers: it must compile as C99 and as C++ on its own."""
    import shutil
    import subprocess
    header = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'include', 'ihgnn_hip.h')
    for compiler, lang in (('gcc', ['-x', 'c', '-std=c99']), ('g++', ['-x', 'c++'])):
        if shutil.which(compiler) is None:
            pytest.skip(f'{compiler} not installed')
        proc = subprocess.run([compiler, '-fsyntax-only', '-Wall', '-Werror'] + lang + [header], capture_output=True, text=True)
        assert proc.returncode == 0, proc.stderr
