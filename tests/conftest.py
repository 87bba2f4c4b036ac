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
    """Committed vectors produced by the reference operator (tests/golden/generate.py)."""

    def __init__(self):
        self.small = np.load(os.path.join(GOLDEN, 'golden_small.npz'))
        self.large = np.load(os.path.join(GOLDEN, 'golden_large.npz'))

    def small_names(self):
        return [str(n) for n in self.small['names']]

    def large_names(self):
        return [str(n) for n in self.large['names']]

    def small_case(self, name):
        g = self.small
        return (g[name + '/observation'], g[name + '/batch_frames'], g[name + '/transition'],
                g[name + '/initial'], g[name + '/indices'])

    def large_case(self, name):
        from torbi_amd import synth
        g = self.large
        B, T, S, seed = (int(x) for x in g[name + '/shape'])
        obs, trans, init = synth.problem(B, T, S, seed=seed)
        return obs, g[name + '/batch_frames'], trans, init, g[name + '/indices']


@pytest.fixture(scope='session')
def golden():
    return Golden()


def _names(kind):
    g = np.load(os.path.join(GOLDEN, f'golden_{kind}.npz'))
    return [str(n) for n in g['names']]


SMALL_NAMES = _names('small')
LARGE_NAMES = _names('large')
