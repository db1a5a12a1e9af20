"""A slice of the soak run inside the suite the driver sees: the seeded sweeps of
test_gpu_random_sweep.test_random_scan[auto], tests/test_gpu_dirty.test_dirty_random and the lag-search sweep
test_gpu_xcorr.test_lag_search_random_shapes are drawn again under three seed bases that follow the TREE -- a digest of
detex_amd/csrc/*.hip|*.h (the GPU box has no .git: the kernel sources are what a commit changes that matters here), so
every change of the kernels meets draws no earlier run has seen, while a given tree stays reproducible.  The bases are
printed; DETEX_TEST_SEED_BASE=<base> python -m pytest -m gpu <file> reproduces one."""
import glob
import hashlib
import os
import sys

import pytest

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)


def tree_bases(k=3):
    h = hashlib.sha1()
    for f in sorted(glob.glob(os.path.join(os.path.dirname(HERE), 'detex_amd', 'csrc', '*.h*'))):
        with open(f, 'rb') as fh:
            h.update(os.path.basename(f).encode() + b'\0' + fh.read())
    d = h.hexdigest()
    # (bases 1 .. 20 000: the soak scripts of round 5 used 1 .. 400)
    return d[:12], [1 + int(d[8 * i:8 * i + 8], 16) % 20000 for i in range(k)]


@pytest.fixture(scope='module')
def ctx():
    from detex_amd.hip import HipContext
    c = HipContext(0)
    yield c
    c.close()


def test_soak_slice_scan(ctx, monkeypatch):
    import test_gpu_random_sweep as sweep
    import test_gpu_dirty as dirty
    digest, bases = tree_bases()
    print('soak slice: source digest %s, seed bases %s' % (digest, bases))
    for base in bases:
        monkeypatch.setattr(sweep, 'SOAK', 100000 * base)
        monkeypatch.setattr(dirty, 'SOAK', 100000 * base)
        for seed in range(34):
            sweep.test_random_scan(ctx, seed, 'auto')
        for seed in range(4):
            dirty.test_dirty_random(ctx, seed)


def test_soak_slice_lag_search(ctx, monkeypatch):
    import test_gpu_xcorr as xc
    digest, bases = tree_bases()
    print('soak slice: source digest %s, seed bases %s' % (digest, bases))
    for base in bases:
        monkeypatch.setattr(xc, 'SOAK', 100000 * base)
        for seed in (0, 3, 6, 9, 11, 14):
            xc.test_lag_search_random_shapes(ctx, monkeypatch, seed)
