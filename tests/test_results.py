"""detex_amd.results (association of detections across stations) against golden
vectors produced by the reference's own detex/results.py functions
(tests/golden/make_golden.py::gen_assoc)."""
import json
import os
import sqlite3

import numpy as np
import pandas as pd
import pytest

from detex_amd import results as R

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, 'golden', 'assoc.json')) as _f:
    GOLD = json.load(_f)


def _db(tmp_path, case):
    db = str(tmp_path / 'ss.db')
    con = sqlite3.connect(db)
    for k in ('ss_df', 'sg_df'):
        pd.DataFrame(case[k], columns=GOLD['cols']).to_sql(k, con, index=False)
    pd.DataFrame([[1, 10, 2, 1]], columns=['FREQMIN', 'FREQMAX', 'CORNERS', 'ZEROPHASE']).to_sql(
        'filt_params', con, index=False)
    con.close()
    return db


def _same(a, b):
    if isinstance(b, float) and np.isnan(b):
        return isinstance(a, float) and np.isnan(a)
    if isinstance(b, float):
        return abs(a - b) <= 1e-9 * max(1.0, abs(b))
    return a == b


def _check_table(got, want):
    assert len(got) == len(want)
    for (_, g), w in zip(got.iterrows(), want):
        for c in R.DET_COLS[:-1]:
            assert _same(g[c].item() if hasattr(g[c], 'item') else g[c], w[c]), (c, g[c], w[c])
        mem = [[m.Sta, m.Name, float(m.STMP)] for _, m in g.Dets.iterrows()]
        assert mem == w['members']


@pytest.mark.parametrize('k', range(len(GOLD['cases'])))
def test_association_matches_reference(tmp_path, k):
    case = GOLD['cases'][k]
    db = _db(tmp_path, case)
    ss = R.delete_det_dups(R.load_detections(db, 'ss_df'), 1)
    sg = R.delete_det_dups(R.load_detections(db, 'sg_df'), 2.5)
    df = pd.concat([ss, sg], ignore_index=True)
    assert [[r.Sta, r.Name, float(r.STMP)] for _, r in df.iterrows()] == case['reduced']
    temkey = R.read_template_key(pd.DataFrame(case['temkey']))
    exc = case['exceptionalThreshold']
    Dets, Autos = R.associate_detections(df, case['requiredNumStations'], 1, temkey, exc)
    Vers = R.verify_events(Dets, Autos, pd.DataFrame(case['veri']), 1, True)
    _check_table(Dets, case['Dets'])          # incl. the Verified flags set by verify_events
    _check_table(Autos, case['Autos'])
    got = [[r.Event, r.VerName, float(r.VerMag), r.EXTRA] for _, r in Vers.iterrows()] \
        if len(Vers) else []
    assert got == case['Vers']


def test_detResults_entry_point(tmp_path):
    case = GOLD['cases'][2]
    db = _db(tmp_path, case)
    res = R.detResults(ssDB=db, templateKey=pd.DataFrame(case['temkey']),
                       requiredNumStations=case['requiredNumStations'],
                       exceptionalThreshold=case['exceptionalThreshold'],
                       veriFile=pd.DataFrame(case['veri']))
    _check_table(res.Dets, case['Dets'])
    _check_table(res.Autos, case['Autos'])
    assert res.NumVerified == len(case['Vers'])
    assert 'autodections' in repr(res)
    # station filter and DS floor only remove detections
    few = R.detResults(ssDB=db, templateKey=pd.DataFrame(case['temkey']), requiredNumStations=1,
                       trigParameter=0.8, stations=['TA.M17A'])
    mem = pd.concat(list(few.Dets.Dets) + list(few.Autos.Dets))
    assert set(mem.Sta) == {'TA.M17A'} and mem.DS.min() >= 0.8
    # time limits (both needed) are honoured -- the reference never applies them
    t = sorted(r[5] for r in case['ss_df'])
    part = R.detResults(ssDB=db, templateKey=pd.DataFrame(case['temkey']), requiredNumStations=1,
                        starttime=t[3], endtime=t[-3])
    mem = pd.concat(list(part.Dets.Dets) + list(part.Autos.Dets))
    assert mem.MSTAMPmin.min() >= t[3] and mem.MSTAMPmin.max() <= t[-3]


def test_detResults_argument_checks(tmp_path):
    case = GOLD['cases'][0]
    db = _db(tmp_path, case)
    tk = pd.DataFrame(case['temkey'])
    with pytest.raises(IOError):
        R.detResults(ssDB=str(tmp_path / 'missing.db'), templateKey=tk)
    for kw in (dict(trigCon=2), dict(trigParameter=1.5), dict(trigCon=1, trigParameter=.5),
               dict(associateReq=1), dict(requiredNumStations=0), dict(ss_associateBuffer=-1),
               dict(reduceDets=False, Pf=1e-8)):
        with pytest.raises(Exception):
            R.detResults(ssDB=db, templateKey=tk, **kw)
    with pytest.raises(Exception, match='No detections'):
        R.detResults(ssDB=db, templateKey=tk, stations=['XX.NONE'])


def test_pf_key_and_threshold_search(tmp_path):
    import scipy.stats
    info = pd.DataFrame([dict(Sta='A', Name='SS0', beta1=2.0, beta2=60.0),
                         dict(Sta='A', Name='SS1', beta1=1.5, beta2=400.0)])
    key = R.make_pf_key(info, 1e-8)
    for (_, k), (_, r) in zip(key.iterrows(), info.iterrows()):
        assert abs(scipy.stats.beta.sf(k.DS, r.beta1, r.beta2) / 1e-8 - 1) < 1e-6
    x, pf = R.approximate_threshold(2.0, 60.0, 1e-8)
    assert abs(x - scipy.stats.beta.isf(1e-8, 2.0, 60.0)) < 1e-5
    assert R.make_pf_key(info, False) is None


def test_time_helpers():
    assert R.to_timestamp('2014-05-13T17:01:38.250000Z') == R.to_timestamp(1400000498.25)
    assert R.event_name(1400000498.25) == '2014-05-13T17-01-38'
    assert R.to_timestamp('2014-05-13') == 1399939200.0
