import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    # The C-ABI library is a build artefact (git-ignored): make sure it exists and is current before any test binds it.
    from ihgnn_amd import build as hip_build
    try:
        hip_build.build()
    except RuntimeError as exc:                      # no hipcc here: tests that need the library will say so themselves
        print(f'[conftest] could not build libihgnn_hip.so: {exc}')


def pytest_collection_modifyitems(config, items):
    """GPU tests are skipped, loudly, where no GPU exists - they never fall back to a CPU path."""
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='no GPU in this container')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope='session')
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name))
    return load


def f8_case(tag, fixture='f8_wide_models.npz'):
    """Wide-model fixture F8 (``tests/golden/f8_wide_models.npz``; F10, the training curves at those widths, keeps its weights the same way):
    ``(cfg, state dict as numpy, fixture)`` with the weights regenerated from the seed the generator used (``tests/golden/seeded_weights.py``)."""
    if GOLDEN not in sys.path:
        sys.path.insert(0, GOLDEN)
    from seeded_weights import seeded_state
    z = np.load(os.path.join(GOLDEN, fixture))
    L, order, d, seed = (int(v) for v in z[f'{tag}.cfg'])
    shapes = [(str(k), tuple(int(x) for x in str(s).split(';'))) for k, s in zip(z[f'{tag}.keys'], z[f'{tag}.shapes'])]
    return (L, order, d), seeded_state(shapes, seed), z


F10_TAGS = ('d128_l3_o3', 'd64_l2_o3', 'd32_l2_o3', 'd128_l3_o2', 'd256_l2_o3')


def f10_case(tag):
    """Training-curve fixture F10 (``tests/golden/f10_training.npz`` + ``f10_workload.npz``): ``(cfg, initial state dict, fixture, workload)``."""
    cfg, sd, z = f8_case(tag, 'f10_training.npz')
    return cfg, sd, z, np.load(os.path.join(GOLDEN, 'f10_workload.npz'))


def state_digest(tensors):
    """[[sum, sum of squares]] per tensor in float64 - what F10 keeps of the trained weights."""
    return np.array([[float(np.asarray(v, np.float64).sum()), float((np.asarray(v, np.float64) ** 2).sum())] for v in tensors], np.float64)


def f8_gradient_error(z, tag, name, grad):
    """max relative deviation of a parameter gradient from what F8 keeps of it (full, or every 4th row + row/col sums)."""
    g = np.asarray(grad, np.float64)
    pre = f'{tag}.grad.{name}.'

    def rel(a, b):
        return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))
    if pre + 'full' in z.files:
        return rel(g, z[pre + 'full'].astype(np.float64))
    return max(rel(g[::4], z[pre + 'rows4'].astype(np.float64)), rel(g.sum(1), z[pre + 'rowsum']), rel(g.sum(0), z[pre + 'colsum']))
