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
