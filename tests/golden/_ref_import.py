"""Import the read-only reference (Detex v1.0.9, pure Python 2.7 era) under
Python 3.10 / pandas 2 / numpy 2 by stubbing the absent GUI / ObsPy modules and
restoring the handful of removed pandas/numpy/scipy aliases its hot path calls.

Used ONLY by tests/golden/make_golden.py, in the build container, to generate
golden input/output vectors.  Nothing here (and nothing under /root/reference)
is needed, read or importable on the GPU box.
"""
import builtins
import sys
import types
import warnings

import numpy as np
import pandas as pd
import scipy
import scipy.fftpack  # noqa: F401
import scipy.linalg  # noqa: F401
import scipy.stats  # noqa: F401

REFERENCE_ROOT = '/root/reference'


def _mk(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class _Dummy(object):
    """Permissive stand-in for ObsPy / Qt classes never touched on the hot path."""

    def __init__(self, *a, **k):
        pass

    def __getattr__(self, k):
        return _Dummy()

    def __call__(self, *a, **k):
        return _Dummy()


def import_reference():
    if 'detex' in sys.modules:
        return sys.modules['detex']
    warnings.simplefilter('ignore')
    _mk('obspy', UTCDateTime=_Dummy, Stream=_Dummy, Trace=_Dummy, read=_Dummy(),
        read_inventory=_Dummy())
    for sub in ['obspy.clients', 'obspy.clients.fdsn', 'obspy.clients.neic',
                'obspy.clients.earthworm', 'obspy.core', 'obspy.core.event',
                'obspy.core.inventory', 'obspy.signal', 'obspy.signal.trigger',
                'obspy.station', 'obspy.station.inventory']:
        m = _mk(sub, Client=_Dummy, UTCDateTime=_Dummy, event=_Dummy(),
                classic_sta_lta=_Dummy(), Inventory=_Dummy, Stream=_Dummy,
                Trace=_Dummy)
        parent, _, leaf = sub.rpartition('.')
        setattr(sys.modules[parent], leaf, m)

    class QMainWindow(object):
        pass

    class QDialog(object):
        pass

    g = _mk('PyQt4.QtGui', QMainWindow=QMainWindow, QDialog=QDialog,
            QApplication=_Dummy)
    c = _mk('PyQt4.QtCore')
    _mk('PyQt4', QtGui=g, QtCore=c)
    _mk('simplekml', Kml=_Dummy)
    _mk('matplotlib.backends.backend_qt4agg', FigureCanvasQTAgg=_Dummy)
    # pandas 0.17 rolling_* (ddof=1 for std/var), removed in pandas >= 0.23
    pd.rolling_mean = lambda x, n, center=False: pd.Series(x).rolling(
        int(n), center=center).mean().values
    pd.rolling_std = lambda x, n: pd.Series(x).rolling(int(n)).std().values
    pd.rolling_var = lambda x, n: pd.Series(x).rolling(int(n)).var().values
    pd.DataFrame.iteritems = pd.DataFrame.items     # removed in pandas 2.0 (util.py:925)
    scipy.real = np.real
    scipy.dot = np.dot
    np.NAN = np.NaN = np.nan
    np.float = float
    np.lib.pad = np.pad          # numpy < 2 alias used at construct.py:444
    builtins.xrange = range
    sys.path.insert(0, REFERENCE_ROOT)
    import detex  # noqa: E402
    _py2_container_semantics(detex)
    return detex


class ListDict(dict):
    """Python 2 dict: keys() / values() return lists (sortable, indexable)."""

    def keys(self):
        return list(dict.keys(self))

    def values(self):
        return list(dict.values(self))


def _py2_container_semantics(detex):
    """Test-side shims that let the reference's basis-selection and alignment code run
    UNMODIFIED under Python 3 / pandas 2 (the reference files are not touched):

    * SubSpace._getFracEnergy / _getUsedBasis (subspace.py:968-1013) call
      ``svdDict.keys().sort()``: their dict argument is handed over as a ListDict;
    * _getDelays (construct.py:725) stores a list in one DataFrame cell with
      ``df.loc[row, col] = [...]``, which pandas 0.17 accepted: a scalar/scalar ``.loc``
      assignment of a list is routed to ``.at``."""
    SS = detex.subspace.SubSpace
    fe, ub = SS._getFracEnergy, SS._getUsedBasis
    SS._getFracEnergy = lambda self, ind, row, svdDict, U: fe(self, ind, row, ListDict(svdDict), U)
    SS._getUsedBasis = (lambda self, ind, row, svdDict, frac, crit, val:
                        ub(self, ind, row, ListDict(svdDict), frac, crit, val))
    from pandas.core.indexing import _LocIndexer
    orig = _LocIndexer.__setitem__

    def setitem(self, key, value):
        if (isinstance(value, list) and isinstance(key, tuple) and len(key) == 2 and
                all(np.isscalar(k) for k in key)):
            self.obj.at[key[0], key[1]] = value
            return
        return orig(self, key, value)

    _LocIndexer.__setitem__ = setitem
