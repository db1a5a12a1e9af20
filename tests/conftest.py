import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN


@pytest.fixture(scope='session', autouse=True)
def _scratch_cwd(tmp_path_factory):
    """createCluster / SubSpace.detex write clust.pkl / SubSpace.db into the working directory by
    default, like the reference (construct.py:31-32, subspace.py:1747): keep them out of the repo."""
    old = os.getcwd()
    os.chdir(str(tmp_path_factory.mktemp('cwd')))
    yield
    os.chdir(old)
