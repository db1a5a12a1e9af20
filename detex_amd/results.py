"""Association of the detections of ``SubSpace.detex`` across stations into
events -- the consumer of the SQLite tables the scan writes
(detex/results.py:22-166, 325-513).  Host-side pandas logic,
restated from the reference's behaviour:

  1. per station, detections of one event by several subspaces / singles are
     reduced to the one with the highest detection statistic (_deleteDetDups,
     results.py:371-400);
  2. detections whose predicted-origin intervals [MSTAMPmin, MSTAMPmax] overlap
     within ``associateBuffer`` are grouped, groups seen on fewer than
     ``requiredNumStations`` stations are dropped unless "exceptional"
     (_associateDetections, results.py:403-458);
  3. a group that overlaps a template's origin time is an auto-detection
     (Autos), anything else a new detection (Dets) (_createAutoTable /
     _createDetTable, results.py:476-503);
  4. optionally the events are matched against a catalogue (_verifyEvents,
     results.py:231-293).

Times are epoch seconds (floats) or ISO strings; ObsPy is not needed.

Divergence: the reference calls ``_buildSQL(PfKey, trigCon, trigParameter,
starttime, stations, endtime, table)`` (results.py:379-380) against the signature
``(..., stations, starttime, endtime, ...)`` (results.py:319-320), so its
``starttime`` / ``endtime`` never limit the selection (and crash together with a
station list).  Here they do what the reference's docstring says.  File
writers for location programs (hypoDD, hypoInverse), waveform export and the KML
output of SSResults are out of scope.
"""
import datetime
import numbers
import os
import sqlite3

import numpy as np
import pandas as pd

from ._lib import DetexHipError

DET_COLS = ['Event', 'DSav', 'DSmax', 'NumStations', 'DS_STALTA', 'MSTAMPmin', 'MSTAMPmax', 'Mag',
            'ProEnMag', 'Verified', 'Dets']
_FOREVER = 4500 * 3600 * 24 * 365.25            # results.py:326-328: "no time limits"


def _error(msg, exc=DetexHipError):
    raise exc(msg)


def to_timestamp(t):
    """Epoch seconds of a float / int, an ISO-8601 string or a datetime
    (what obspy.UTCDateTime(t).timestamp gives for those)."""
    if isinstance(t, numbers.Real):
        return float(t)
    if isinstance(t, datetime.datetime):
        if t.tzinfo is None:
            t = t.replace(tzinfo=datetime.timezone.utc)
        return t.timestamp()
    s = str(t).strip().rstrip('Z').replace(' ', 'T')
    for fmt in ('%Y-%m-%dT%H:%M:%S.%f', '%Y-%m-%dT%H:%M:%S', '%Y-%m-%dT%H-%M-%S', '%Y-%m-%d'):
        try:
            d = datetime.datetime.strptime(s, fmt)
            return d.replace(tzinfo=datetime.timezone.utc).timestamp()
        except ValueError:
            continue
    try:
        return float(s)
    except ValueError:
        _error('cannot read the time %r' % (t,), ValueError)


def event_name(stamp):
    """results.py:478-479: str(UTCDateTime(stamp)) with ':' -> '-' and the
    fractional seconds cut: '2014-05-13T17-01-38'."""
    # UTCDateTime.__str__ prints microseconds of the ROUNDED time stamp
    d = datetime.datetime.fromtimestamp(round(float(stamp), 6), datetime.timezone.utc)
    return d.strftime('%Y-%m-%dT%H-%M-%S')


def _load_table(ssDB, table, where=None, params=()):
    """detex.util.loadSQLite: a table (or a selection of it) as a DataFrame,
    None if the table does not exist or nothing matches."""
    con = sqlite3.connect(ssDB)
    try:
        have = pd.read_sql('SELECT name FROM sqlite_master WHERE type="table" AND name=?', con,
                           params=(table,))
        if not len(have):
            return None
        sql = 'SELECT * FROM %s' % table
        if where:
            sql += ' WHERE ' + where
        df = pd.read_sql(sql, con, params=params)
    finally:
        con.close()
    return df if len(df) else None


def read_template_key(templateKey):
    """detex.util.readKey(..., 'template') for a CSV path or a DataFrame with
    at least NAME and TIME (an EventFetcher is accepted too: its event names and
    origin times).  Adds STMP = epoch seconds of TIME (results.py:418)."""
    if hasattr(templateKey, 'events') and hasattr(templateKey, 'stations'):
        seen = {}
        for sta in templateKey.stations():
            for name, rec in templateKey.getTemData(sta):
                seen.setdefault(name, rec)
        df = pd.DataFrame([dict(NAME=k, TIME=v['origintime'], LAT=np.nan, LON=np.nan,
                                DEPTH=np.nan, MAG=v['magnitude']) for k, v in sorted(seen.items())])
    elif isinstance(templateKey, pd.DataFrame):
        df = templateKey.copy()
    elif isinstance(templateKey, str):
        if not os.path.exists(templateKey):
            _error('%s does not exists' % templateKey, IOError)          # results.py:585-588
        df = pd.read_csv(templateKey)
    else:
        _error('Data type of templateKey not understood')
    if not {'NAME', 'TIME'}.issubset(df.columns):
        _error('Required columns not in template key: NAME, TIME')
    df = df.reset_index(drop=True)
    df['STMP'] = [to_timestamp(t) for t in df.TIME]
    return df


# ---------------------------------------------------------------- Pf thresholds

def approximate_threshold(beta_a, beta_b, target, numintervals=1000, numloops=3):
    """results.py:206-228: nested grid search on the survival function, used when
    beta.isf returns a value close to 1 (scipy issue 4677)."""
    import scipy.stats
    lo, hi = 0.0, 1.0
    for _ in range(numloops):
        xs = np.linspace(lo, hi, numintervals)
        pfs = scipy.stats.beta.sf(xs, beta_a, beta_b)
        k = int(np.abs(pfs - target).argmin())
        if k == 0 or k == numintervals - 1:
            _error('Grind search failing, set threshold manually', ValueError)
        best_x, best_pf = float(xs[k]), float(pfs[k])
        lo, hi = xs[k - 1], xs[k + 1]
    return best_x, best_pf


def make_pf_key(info, Pf):
    """results.py:169-203: DS threshold of every (station, detector) for a
    probability of false detection Pf from its fitted beta distribution."""
    import scipy.stats
    if not Pf or not isinstance(info, pd.DataFrame):
        return None
    rows = []
    for _, r in info.iterrows():
        th = scipy.stats.beta.isf(Pf, r.beta1, r.beta2, 0, 1)
        if th > .94:
            th, _pf = approximate_threshold(r.beta1, r.beta2, Pf, 1000, 3)
        rows.append(dict(Sta=r.Sta, Name=r.Name, DS=float(th), betadist=[r.beta1, r.beta2, 0, 1]))
    return pd.DataFrame(rows, columns=['Sta', 'Name', 'DS', 'betadist'])


# ---------------------------------------------------------------- step 1

def load_detections(ssDB, table, trigCon=0, trigParameter=0, starttime=None, endtime=None,
                    stations=None, PfKey=None):
    """The selections of results.py:325-368 (_buildSQL) as one DataFrame.
    Both time limits must be given for either to apply (results.py:326-331)."""
    if not starttime or not endtime:
        t0, t1 = 0.0, _FOREVER
    else:
        t0, t1 = to_timestamp(starttime), to_timestamp(endtime)
    parts = []
    if isinstance(PfKey, pd.DataFrame):
        if isinstance(stations, (list, tuple)):
            PfKey = PfKey[PfKey.Sta.isin(stations)]
        for _, r in PfKey.iterrows():
            tab = 'sg_df' if 'SG' in r.Name else 'ss_df'      # results.py:346-348
            parts.append(_load_table(ssDB, tab, 'Sta=? AND Name=? AND DS>=? AND MSTAMPmin>? AND '
                                     'MSTAMPmin<?', (r.Sta, r.Name, float(r.DS), t0, t1)))
    else:
        cond = 'DS' if trigCon == 0 else 'DS_STALTA'
        where = '%s >= ? AND MSTAMPmin>=? AND MSTAMPmin<=?' % cond
        if isinstance(stations, (list, tuple)):
            for sta in stations:
                parts.append(_load_table(ssDB, table, 'Sta=? AND ' + where,
                                         (sta, float(trigParameter), t0, t1)))
        else:
            parts.append(_load_table(ssDB, table, where, (float(trigParameter), t0, t1)))
    parts = [p for p in parts if isinstance(p, pd.DataFrame)]
    if not parts:
        return None
    return pd.concat(parts, ignore_index=True)


def delete_det_dups(df, associateBuffer):
    """results.py:388-400: per station, runs of detections whose origin intervals
    touch within ``associateBuffer`` keep only the member with the highest DS."""
    if df is None or not len(df):
        return None
    df = df.sort_values(by=['Sta', 'MSTAMPmin'], kind='stable').reset_index(drop=True)
    new_run = ((df.MSTAMPmin - associateBuffer) > df.MSTAMPmax.shift()) | (df.Sta != df.Sta.shift())
    df['Gnum'] = new_run.cumsum()
    df = df.sort_values(by=['Gnum', 'DS'], kind='stable')
    df = df.drop_duplicates(subset='Gnum', keep='last').reset_index(drop=True)
    return df


# ---------------------------------------------------------------- steps 2, 3

def _median_mag(vals):
    vals = np.asarray(vals, dtype=np.float64)
    return float(np.nanmedian(vals)) if np.any(~np.isnan(vals)) else np.nan


def _row_of(g, event):
    return [event, g.DS.mean(), g.DS.max(), len(g), g.DS_STALTA.mean(), g.MSTAMPmin.min(),
            g.MSTAMPmax.max(), _median_mag(g.Mag), _median_mag(g.ProEnMag), False, g]


def _auto_event(g, temkey, associateBuffer):
    """results.py:487-494: the template a group coincides with: for every member
    the first template whose origin lies in its [MSTAMPmin, MSTAMPmax] widened by
    the buffer; the LAST member with a match names the event."""
    event = None
    stmp = temkey.STMP.values
    for _, r in g.iterrows():
        hit = np.nonzero((stmp + associateBuffer > r.MSTAMPmin) &
                         (stmp - associateBuffer < r.MSTAMPmax))[0]
        if len(hit):
            event = temkey.NAME.iloc[hit[0]]
    return event


def _is_exceptional(g, exth):
    if isinstance(exth, float):                                   # results.py:437-439
        return bool(g.DS.max() >= exth)
    if isinstance(exth, dict):                                    # results.py:461-465
        lim = np.array([exth.get(s, 100) for s in g.Sta], dtype=np.float64)
        return bool(np.any((g.DS.values >= lim) & (g.DS.values <= 1.01)))
    return False


def associate_detections(df, requiredNumStations, associateBuffer, temkey,
                         exceptionalThreshold=None):
    """results.py:403-458 (associateReq == 0) -> (Dets, Autos)."""
    df = df.sort_values(by='MSTAMPmin', kind='stable').reset_index(drop=True)
    gid = (df.MSTAMPmin - associateBuffer > df.MSTAMPmax.shift()).cumsum()
    dets, autos = [], []
    for _, g in df.groupby(gid):
        ok = g.Sta.nunique() >= requiredNumStations
        if not ok:
            ok = _is_exceptional(g, exceptionalThreshold)
        if not ok:
            continue
        if g.Sta.nunique() < len(g):       # one detector per station: the highest DS
            g = (g.sort_values(by='DS', kind='stable').drop_duplicates(subset='Sta', keep='last')
                 .sort_values('MSTAMPmin', kind='stable'))
        event = _auto_event(g, temkey, associateBuffer)
        if event is not None:
            autos.append(_row_of(g, event))
        else:
            mid = np.mean([g.MSTAMPmin.mean(), g.MSTAMPmax.mean()])
            dets.append(_row_of(g, event_name(mid)))
    return (pd.DataFrame(dets, columns=DET_COLS), pd.DataFrame(autos, columns=DET_COLS))


# ---------------------------------------------------------------- step 4

def read_veri_file(veriFile):
    """results.py:296-316: CSV, pickled DataFrame, SQLite table 'verify' or a
    DataFrame with TIME, LAT, LON, MAG, DEPTH, NAME."""
    if isinstance(veriFile, pd.DataFrame):
        df = veriFile.copy()
    else:
        df = None
        for reader in (pd.read_csv, pd.read_pickle, lambda p: _load_table(p, 'verify')):
            try:
                df = reader(veriFile)
                if isinstance(df, pd.DataFrame):
                    break
            except Exception:
                df = None
        if df is None:
            _error('%s could not be read, it must either be csv, pickled dataframe or sqlite '
                   'database' % veriFile)
    need = ['TIME', 'LAT', 'LON', 'MAG', 'DEPTH', 'NAME']
    if not set(need).issubset(df.columns):
        _error('%s does not have the required columns, it needs TIME,LAT,LON,MAG,DEPTH,NAME'
               % (veriFile if isinstance(veriFile, str) else 'veriFile'))
    return df


def verify_events(Dets, Autos, veriFile, veriBuffer=1, includeAllVeriColumns=True):
    """results.py:231-293: every catalogue event claims the not-yet-verified
    detection (else auto-detection) with the highest DSav whose origin interval,
    widened by veriBuffer / 2 each side, contains its time.  Marks ``Verified``
    in Dets / Autos and returns the table of verified events."""
    if veriFile is None:
        return None
    if isinstance(veriFile, str) and (not veriFile or not os.path.exists(veriFile)):
        return None                                               # results.py:235-238 (warning)
    ver = read_veri_file(veriFile)
    ver['STMP'] = [to_timestamp(t) for t in ver.TIME]
    std = ['TIME', 'LAT', 'LON', 'MAG', 'ProEnMag', 'DEPTH', 'NAME']
    extra = [c for c in ver.columns if c not in std]
    out = []
    for _, v in ver.iterrows():
        for table in (Dets, Autos):
            free = ~table.Verified.astype(bool)
            cand = table[(table.MSTAMPmin - veriBuffer / 2.0 < v.STMP) &
                         (table.MSTAMPmax + veriBuffer / 2.0 > v.STMP) & free]
            if not len(cand):
                continue
            best = cand[cand.DSav == cand.DSav.max()].iloc[[0]].copy()
            table.loc[best.index[0], 'Verified'] = True
            if includeAllVeriColumns:
                for col in extra:
                    if col not in best.columns:
                        best[col] = v[col]
            best['VerMag'], best['VerLat'], best['VerLon'] = v.MAG, v.LAT, v.LON
            best['VerDepth'], best['VerName'] = v.DEPTH, v.NAME
            out.append(best)
            break
    if not out:
        return pd.DataFrame()
    return pd.concat(out, ignore_index=True).drop('Verified', axis=1)


# ---------------------------------------------------------------- entry point

class SSResults(object):
    """results.py:611-626: Dets (new events), Autos (re-detected templates), Vers
    (catalogue matches), the detector info and filter tables of the scan."""

    def __init__(self, Dets, Autos, Vers, ss_info, ss_filt, temkey, stakey=None, fetcher=None):
        self.Autos = Autos
        self.Dets = Dets
        self.NumVerified = len(Vers) if isinstance(Vers, pd.DataFrame) else 'N/A'
        self.Vers = Vers
        self.info = ss_info
        self.filt = ss_filt
        self.StationKey = stakey
        self.TemplateKey = temkey
        self.fetcher = fetcher

    def __repr__(self):
        return ('SSResults instance with %d autodections and %d new detections, %s are verified'
                % (len(self.Autos), len(self.Dets), self.NumVerified))     # results.py:1023-1027


def _check_inputs(trigCon, trigParameter, associateReq, associateBuffer, requiredNumStations):
    """results.py:536-576."""
    if not isinstance(trigCon, int) or trigCon not in (0, 1):
        _error('trigcon must be an int, either 0 or 1')
    real = isinstance(trigParameter, numbers.Real)
    if trigCon == 0 and (not real or trigParameter > 1 or trigParameter < 0):
        _error('When trigCon==0 trigParameter is the required detection statistic and therefore '
               'must be between 0 and 1')
    if trigCon == 1 and (not real or (trigParameter < 1 and trigParameter != 0)):
        _error('When trigCon==1 trigParameter is the STA/LTA of the detection statistic vector '
               'and therefore must be greater than 1')
    if not isinstance(associateReq, int) or associateReq < 0:
        _error('AssociateReq must be an integer 0 or greater')
    if not isinstance(associateBuffer, numbers.Real) or associateBuffer < 0:
        _error('associateBuffer must be a real number greater than 0')
    if not isinstance(requiredNumStations, int) or requiredNumStations < 1:
        _error('requiredNumStations must be an integer greater than 0')


def detResults(trigCon=0, trigParameter=0, associateReq=0, ss_associateBuffer=1,
               sg_associateBuffer=2.5, requiredNumStations=4, veriBuffer=1, ssDB='SubSpace.db',
               templateKey='TemplateKey.csv', stationKey=None, veriFile=None,
               includeAllVeriColumns=True, reduceDets=True, Pf=False, stations=None,
               starttime=None, endtime=None, fetch=None, exceptionalThreshold=None):
    """results.py:22-166 with the reference's arguments.  ``templateKey`` may be a
    CSV path, a DataFrame (NAME, TIME, ...) or the EventFetcher the cluster was
    built from; ``stationKey`` / ``fetch`` are carried along untouched."""
    if not os.path.exists(ssDB):
        _error('%s does not exists' % ssDB, IOError)
    _check_inputs(trigCon, trigParameter, associateReq, ss_associateBuffer, requiredNumStations)
    if associateReq != 0:
        _error('associateReq values other than 0 not yet supported')        # results.py:116-118
    temkey = read_template_key(templateKey)
    ss_info = _load_table(ssDB, 'ss_info')
    if isinstance(ss_info, pd.DataFrame):
        ss_info['NumEvents'] = [len(str(e).split(',')) for e in ss_info.Events]
    sg_info = _load_table(ssDB, 'sg_info')
    if isinstance(sg_info, pd.DataFrame):
        sg_info['NumEvents'] = 1
    filt = _load_table(ssDB, 'filt_params')
    ss_pf, sg_pf = make_pf_key(ss_info, Pf), make_pf_key(sg_info, Pf)
    if reduceDets:
        ssdf = delete_det_dups(load_detections(ssDB, 'ss_df', trigCon, trigParameter, starttime,
                                               endtime, stations, ss_pf), ss_associateBuffer)
        sgdf = delete_det_dups(load_detections(ssDB, 'sg_df', trigCon, trigParameter, starttime,
                                               endtime, stations, sg_pf), sg_associateBuffer)
    else:
        if Pf:
            _error('When using the Pf parameter reduceDets must be True')
        ssdf, sgdf = _load_table(ssDB, 'ss_df'), _load_table(ssDB, 'sg_df')
    if ssdf is None and sgdf is None:
        _error('No detections found that meet given criteria')
    df = pd.concat([d for d in (ssdf, sgdf) if d is not None], ignore_index=True)
    if isinstance(stations, (list, tuple)):
        df = df[df.Sta.isin(stations)]
    Dets, Autos = associate_detections(df, requiredNumStations, ss_associateBuffer, temkey,
                                       exceptionalThreshold)
    Vers = verify_events(Dets, Autos, veriFile, veriBuffer, includeAllVeriColumns)
    return SSResults(Dets, Autos, Vers, ss_info, filt, temkey, stationKey, fetch)
