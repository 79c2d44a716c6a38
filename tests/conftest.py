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
