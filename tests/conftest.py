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


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + '.npz'))


def tape_from_golden(g, prefix='tape'):
    kinds = g['%s_kinds' % prefix]
    return [(str(k), g['%s_%02d' % (prefix, i)]) for i, k in enumerate(kinds)]


@pytest.fixture(scope='session')
def gen_labels():
    return np.array([0, 14, 15, 16, 2, 3, 4, 5, 7, 8, 10, 11, 12, 13, 17, 18, 26, 28, 31], dtype=np.int32)


def regen_weights(names, shapes, seed):
    """the seeded weight recipe of tests/golden/gen/keras_layers_shim._weight, replayed in the order of `names`
    (used for goldens whose 13 M weights are not stored; the golden holds per-tensor checksums)"""
    rng = np.random.default_rng(int(seed))
    out = {}
    for nm, shp in zip(names, shapes):
        nm = str(nm)
        shape = tuple(int(s) for s in shp if s > 0)
        kind = nm.split('/')[-1]
        if kind == 'kernel':
            rf = int(np.prod(shape[:-2]))
            lim = np.sqrt(6.0 / (rf * shape[-2] + rf * shape[-1]))
            a = rng.uniform(-lim, lim, shape)
        elif kind == 'bias':
            a = rng.normal(0, .05, shape)
        elif kind == 'gamma':
            a = rng.uniform(.5, 1.5, shape)
        elif kind == 'beta':
            a = rng.normal(0, .1, shape)
        elif kind == 'moving_mean':
            a = rng.normal(0, .2, shape)
        elif kind == 'moving_variance':
            a = rng.uniform(.5, 2., shape)
        else:
            raise KeyError(nm)
        out[nm] = a.astype(np.float32)
    return out


def golden_weights(g, prefix):
    """all arrays stored under '<prefix><layer>/<weight>' of a golden file"""
    return {k[len(prefix):]: g[k] for k in g.files if k.startswith(prefix)}
