import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


class Golden:
    """Lazy access to one tests/golden/*.npz produced by tests/golden/gen_golden.py."""

    def __init__(self, name):
        self.z = np.load(os.path.join(GOLDEN, name))
        self.meta = json.loads(str(self.z['meta']))

    def arr(self, prefix, i, key):
        return self.z['%s%03d_%s' % (prefix, i, key)]


_cache = {}


def load_golden(name):
    if name not in _cache:
        _cache[name] = Golden(name)
    return _cache[name]


@pytest.fixture(scope='session')
def golden_uniform():
    return load_golden('uniform.npz')


@pytest.fixture(scope='session')
def golden_nonuniform():
    return load_golden('nonuniform.npz')


@pytest.fixture(scope='session')
def golden_ste():
    return load_golden('ste.npz')


@pytest.fixture(scope='session')
def golden_nonuniform_options():
    return load_golden('nonuniform_options.npz')


@pytest.fixture(scope='session')
def golden_nonfinite():
    return load_golden('nonfinite.npz')


@pytest.fixture(scope='session')
def golden_misc():
    return load_golden('misc.npz')


@pytest.fixture(scope='session')
def golden_big():
    with open(os.path.join(GOLDEN, 'big_checksums.json')) as f:
        return json.load(f)
