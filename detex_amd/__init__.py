"""detex_amd -- MI355X-native hot path of Detex (d-chambers/Detex v1.0.9).

The package keeps the reference's API names for the path it accelerates
(createCluster, createSubSpace, SubSpace.SVD / SubSpace.detex) and runs the
all-pairs cross-correlation and the subspace detection statistic in
hand-written HIP kernels for gfx950 behind a C ABI (include/detex_hip.h).
"""
__version__ = '0.1.0'

from ._lib import DetexHipError  # noqa: F401
from .construct import createCluster, createSubSpace, fast_normcorr  # noqa: F401,E402
from .getdata import ContinuousFetcher, EventFetcher, multiplex  # noqa: F401,E402
from .subspace import Cluster, ClusterStream, SubSpace, loadClusters, loadSubSpace  # noqa: F401,E402
from .results import detResults, SSResults  # noqa: F401,E402
