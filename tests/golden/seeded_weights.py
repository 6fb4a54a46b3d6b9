"""Model weights as a pure function of (state-dict key order, shapes, seed) through numpy's PCG64 stream.

The wide-model fixtures (F8: d = 128 / 256) would otherwise carry megabytes of random weights; instead the fixture
generator loads THESE into the reference model before running it and the tests regenerate the same arrays
(``numpy.random.default_rng`` streams are stable across platforms and numpy versions by policy).  Scales follow the
reference's initialisers (uniform, bound 1/sqrt(fan_in) for matrices; N(0,1) item bias), so activations have
trained-model magnitudes.
"""
import numpy as np


def seeded_state(shapes, seed):
    """``{key: float32 array}`` for ``shapes = [(key, shape), ...]`` in that order."""
    rng = np.random.default_rng(seed)
    out = {}
    for key, shape in shapes:
        shape = tuple(int(s) for s in shape)
        if key.endswith('items_bias'):
            a = rng.standard_normal(shape)
        elif len(shape) == 2:
            a = rng.uniform(-1.0, 1.0, shape) / np.sqrt(shape[1])
            if 'embedding' in key:
                a *= 4.0                                  # embedding rows of O(1/sqrt(d)) per element x 4: products stay well above fp32 noise
        else:
            a = rng.uniform(-0.1, 0.1, shape)
        out[key] = a.astype(np.float32)
    return out


def big_gradient_digest(g, limit=20_000):
    """What a fixture keeps of a parameter gradient: all of it up to ``limit`` elements, otherwise every 4th row plus
    float64 row and column sums (the interact-kernel tests compare the full matrix with the oracle)."""
    g = np.asarray(g)
    if g.size <= limit or g.ndim != 2:
        return {'full': g.astype(np.float32)}
    return {'rows4': g[::4].astype(np.float32), 'rowsum': g.astype(np.float64).sum(1), 'colsum': g.astype(np.float64).sum(0)}
