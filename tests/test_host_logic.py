"""CPU tests of the host-side logic around the kernels (no GPU): event loading,
dissimilarity + linkage + dendrogram cut, alignment through the dendrogram, SVD
basis selection, chunk geometry, STA/LTA at a trigger, beta MLE.  The GPU seam
(_makeDFcclags -> dtx_xcorr_allpairs) is stood in for by the oracle through a
fake context -- test infrastructure only."""
import os
import sys

import numpy as np
import pytest
import torch
from scipy.cluster.hierarchy import fcluster

from detex_amd import (ContinuousFetcher, EventFetcher, createCluster, createSubSpace)
from detex_amd import construct, detect, fas
from oracle import detex_oracle as orc

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


from stub_ctx import OracleCtx  # noqa: E402


def _fetcher_from_golden(tag):
    z = np.load(os.path.join(G, 'cclags.npz'))
    ev, Nc, fam = z[tag + '_events'], int(z[tag + '_Nc']), z[tag + '_fam']
    f = EventFetcher(100.0, ['C%d' % c for c in range(Nc)])
    for i, e in enumerate(ev):
        f.add('NET.STA', 'ev%02d' % i, e.reshape(-1, Nc).T, starttime=1000.0 + i,
              origintime=995.0 + i, magnitude=1.0 + 0.1 * i)
    return f, z, fam


@pytest.mark.parametrize('tag', ['a', 'b'])
def test_createCluster_matches_reference_linkage(tag):
    f, z, fam = _fetcher_from_golden(tag)
    cl = createCluster(CCreq=0.5, fetch_arg=f, ctx=OracleCtx())
    row = cl.trdf.iloc[0]
    assert list(cl.trdf.columns) == ['Station', 'Link', 'CCs', 'Lags', 'Subsamp', 'Events', 'Stats']
    assert np.allclose(row.CCs.values.astype(float), z[tag + '_cc'], equal_nan=True, atol=1e-9)
    assert np.array_equal(np.nan_to_num(row.Lags.values.astype(float), nan=-1),
                          np.nan_to_num(z[tag + '_lag'], nan=-1))
    assert np.allclose(row.Link, z[tag + '_link'], atol=1e-9)
    assert list(row.CCs.index) == list(range(len(fam) - 1))
    assert list(row.CCs.columns) == list(range(1, len(fam)))
    # dendrogram cut == scipy fcluster at the same height, families recovered
    c = cl['NET.STA']
    T = fcluster(row.Link, 1 - 0.5, criterion='distance')
    groups = {}
    for i, t in enumerate(T):
        groups.setdefault(t, []).append('ev%02d' % i)
    want = sorted(sorted(g) for g in groups.values() if len(g) > 1)
    assert sorted(sorted(g) for g in c.clusts) == want
    assert sorted(c.singles) == sorted(g[0] for g in groups.values() if len(g) == 1)
    cl.updateReqCC(0.999999)
    assert c.clusts == [] and len(c.singles) == len(fam)
    cl.updateReqCC(0.0)
    assert len(c.clusts) == 1 and c.singles == []
    with pytest.raises(Exception):
        cl.updateReqCC(1.5)


def _shifted_family(rs, Nc, L, shifts, snr=20.0):
    pad = max(shifts) + 5
    wav = [np.convolve(rs.standard_normal(L + pad + 8), np.hanning(9), 'valid') *
           np.exp(-np.arange(L + pad) / (0.25 * L)) for _ in range(Nc)]
    evs = []
    for sh in shifts:
        evs.append(np.array([w[pad - sh: pad - sh + L] + rs.standard_normal(L) * np.std(w) / snr
                             for w in wav]))
    return evs


def test_alignment_through_dendrogram():
    """After createSubSpace the aligned events correlate best at zero lag and the
    recovered delays equal the planted shifts (Harris 2006 App. B logic of
    construct.py:710-760)."""
    rs = np.random.RandomState(4)
    Nc, L = 3, 300
    shifts = [0, 7, 3, 12, 5]
    evs = _shifted_family(rs, Nc, L, shifts)
    other = _shifted_family(rs, Nc, L, [0, 4])
    f = EventFetcher(50.0, ['E', 'N', 'Z'])
    for i, e in enumerate(evs + other):
        f.add('NET.STA', 'ev%d' % i, e, starttime=100.0 * i, origintime=100.0 * i - 2.0)
    cl = createCluster(CCreq=0.6, fetch_arg=f, ctx=OracleCtx())
    assert sorted(map(sorted, cl['STA'].clusts)) == [['ev%d' % i for i in range(5)], ['ev5', 'ev6']]
    ss = createSubSpace(clust=cl, conDatFetcher=None)
    df = ss.subspaces['NET.STA']
    row = df[[len(e) == 5 for e in df.Events]].iloc[0]
    ali = row.AlignedTD
    lens = set(len(v) for v in ali.values())
    assert len(lens) == 1 and lens.pop() == Nc * (L - (max(shifts) - min(shifts)))
    ref = ali['ev0']
    for k in ['ev1', 'ev2', 'ev3', 'ev4']:
        cc, lag, _, _ = orc.ccx2_fft(ref, ali[k], Nc)
        assert lag == 0 and cc > 0.9, (k, cc, lag)
    # start times moved by the trimmed delay: event shifted by s samples starts s/sr later
    st = {k: row.Stats[k]['starttime'] - 100.0 * int(k[2:]) for k in ali}
    d = np.array([st['ev%d' % i] for i in range(5)]) * 50.0
    assert np.allclose(d - d[0], -(np.array(shifts) - shifts[0]), atol=1e-9) or \
        np.allclose(d - d[0], (np.array(shifts) - shifts[0]), atol=1e-9)
    assert len(row.Offsets) == 3 and row.Offsets[0] <= row.Offsets[1] <= row.Offsets[2]
    # singles: none here; minEvents filter
    ss3 = createSubSpace(clust=cl, minEvents=3)
    assert len(ss3.subspaces['NET.STA']) == 1


def test_svd_basis_selection_and_thresholds():
    rs = np.random.RandomState(8)
    Nc, L = 3, 300
    evs = _shifted_family(rs, Nc, L, [0, 2, 4, 1, 3, 5], snr=6.0)
    f = EventFetcher(50.0, ['E', 'N', 'Z'])
    for i, e in enumerate(evs):
        f.add('NET.STA', 'ev%d' % i, e * rs.uniform(.5, 2))
    cl = createCluster(CCreq=0.4, fetch_arg=f, ctx=OracleCtx())
    ss = createSubSpace(clust=cl)
    ss.attachPickTimes(startSeconds=0.2, defaultDuration=3.0)
    row = ss.subspaces['NET.STA'].iloc[0]
    assert row.SampleTrims['Starttime'] == 10 * Nc and row.SampleTrims['Endtime'] == 160 * Nc
    ss.SVD(selectCriteria=3, selectValue=0.8, useSingles=False)
    row = ss.subspaces['NET.STA'].iloc[0]
    U, s = row.SVD['U'], row.SVD['s']
    keys = sorted(row.Events)
    arr = orc.trim_groups([row.AlignedTD[k] for k in keys], 10 * Nc, 160 * Nc)
    Uo, so = orc.svd_basis(arr)
    assert np.allclose(s, so) and np.allclose(np.abs(U.T @ Uo), np.eye(len(s)), atol=1e-8)
    fr, av, mn = orc.frac_energy(Uo, [row.AlignedTD[k][10 * Nc:160 * Nc] for k in row.Events])
    nb = orc.used_basis_count(av, 3, 0.8)
    assert row.NumBasis == nb and len(row.UsedSVDKeys) == nb
    assert np.isclose(row.Threshold, orc.threshold_criteria3(mn, nb, 0.8))
    ss.SVD(selectCriteria=4, selectValue=1, threshold=0.25, useSingles=False)
    row = ss.subspaces['NET.STA'].iloc[0]
    assert row.NumBasis == 2 and row.Threshold == 0.25
    names, kinds, Ub, off, offsets, thr = ss._detector_bank('NET.STA')
    assert names == ['SS0'] and Ub.shape == (2, 150 * Nc) and list(off) == [0, 2]
    with pytest.raises(ValueError):
        ss.SVD(selectCriteria=2, selectValue=1.5)
    with pytest.raises(Exception):
        ss.detex(multiprocess=True)


def _subspace_from_golden(z, k):
    """A one-subspace SubSpace of the product holding the fixture's aligned events."""
    import pandas as pd
    from detex_amd.subspace import SubSpace
    ev, t, ht = z[k + '_events'], z[k + '_trims'], bool(z[k + '_has_trims'])
    names = ['ev%02d' % i for i in range(len(ev))]
    row = dict(Name='SS0', Station='NET.STA', Events=names, AlignedTD=dict(zip(names, ev)), SVD=None,
               UsedSVDKeys=None, FracEnergy=None, SVDdefined=False,
               SampleTrims={'Starttime': int(t[0]), 'Endtime': int(t[1])} if ht else {},
               Threshold=np.nan, FAS=None, NumBasis=0, Offsets=[0, 0, 0],
               Stats={n: dict(magnitude=1.0) for n in names}, numEvents=len(names))
    return SubSpace({}, {'NET.STA': pd.DataFrame([row], dtype=object)}, None, 'double', 1e-12, None)


def test_SubSpace_SVD_matches_reference_golden():
    """SubSpace.SVD of the product against the reference's own SubSpace.SVD
    (tests/golden/svd.npz, subspace.py:786-1054)."""
    z = np.load(os.path.join(G, 'svd.npz'))
    for nm in z['names']:
        k = nm.split(':')[0]
        ss = _subspace_from_golden(z, k)
        thr = float(z[k + '_thr_in'])
        ss.SVD(selectCriteria=int(z[k + '_crit']),
               selectValue=float(z[k + '_val']) if int(z[k + '_crit']) != 4 else int(z[k + '_val']),
               threshold=thr if thr > 0 else None, normalize=bool(z[k + '_normalize']),
               useSingles=False)
        row = ss.subspaces['NET.STA'].iloc[0]
        U, s = row.SVD['U'], row.SVD['s']
        assert np.allclose(s, z[k + '_s'], rtol=1e-12), nm
        sg = np.sign(np.sum(U * z[k + '_U'], axis=0))
        assert np.allclose(U * sg, z[k + '_U'], atol=1e-9), nm
        assert int(row.NumBasis) == int(z[k + '_numbasis']), nm
        assert np.allclose(row.UsedSVDKeys, z[k + '_used'], rtol=1e-12), nm
        assert np.allclose(row.FracEnergy['Average'], z[k + '_average'], atol=1e-12), nm
        assert np.allclose(row.FracEnergy['Minimum'], z[k + '_minimum'], atol=1e-12), nm
        for i, e in enumerate(row.Events):
            assert np.allclose(row.FracEnergy[e], z[k + '_frac'][i], atol=1e-12), nm
        assert np.isclose(float(row.Threshold), float(z[k + '_threshold']), rtol=1e-12), nm
        # the stacked detector the scan uploads: the selected vectors, as rows
        names, kinds, Ub, off, _, th = ss._detector_bank('NET.STA')
        nb = int(z[k + '_numbasis'])
        assert Ub.shape[0] == nb and list(off) == [0, nb]
        assert np.allclose(Ub * sg[:nb, None], z[k + '_U'][:, :nb].T, atol=1e-9), nm


def test_getDelays_alignTD_match_reference_golden():
    """construct._getDelays / _alignTD of the product against the reference's own
    (tests/golden/delays.npz, construct.py:710-760, 486-504), and the same through
    createCluster -> createSubSpace on the fixture's events."""
    z = np.load(os.path.join(G, 'delays.npz'))
    for nm in z['names']:
        k = nm.split(':')[0]
        ev, Nc = z[k + '_events'], int(z[k + '_Nc'])
        d = construct._getDelays(z[k + '_link'], z[k + '_cx'], z[k + '_lags_in'], len(ev))
        assert np.array_equal(d, z[k + '_delays']), nm
        names = ['ev%02d' % i for i in range(len(ev))]
        ali = construct._alignTD(d - d.min(), names, dict(zip(names, ev)))
        assert np.array_equal(np.array([ali[n] for n in names]), z[k + '_aligned']), nm
        if 'nfam=1' not in nm:
            continue
        # one family = one cluster: the whole construction path reproduces the alignment
        f = EventFetcher(100.0, ['C%d' % c for c in range(Nc)])
        for n_, e in zip(names, ev):
            f.add('NET.STA', n_, e.reshape(-1, Nc).T, starttime=50.0, origintime=48.0)
        cl = createCluster(CCreq=0.05, fetch_arg=f, ctx=OracleCtx())
        assert len(cl['STA'].clusts) == 1 and len(cl['STA'].clusts[0]) == len(ev)
        ss = createSubSpace(clust=cl)
        row = ss.subspaces['NET.STA'].iloc[0]
        assert np.array_equal(np.array([row.AlignedTD[n_] for n_ in names]), z[k + '_aligned']), nm
        sd = z[k + '_sample_delays']
        st = np.array([row.Stats[n_]['starttime'] for n_ in names])
        assert np.allclose(st, 50.0 + sd / (100.0 * Nc), atol=1e-12)        # construct.py:346-366


def test_chunk_geometry():
    """getdata.py:509-518, 869-886: starts on multiples of conDatDuration, each
    chunk conDatDuration + conBuff long, the last trimmed to the data."""
    c = ContinuousFetcher(100.0, ['E', 'N', 'Z'])
    T = int(2.5 * 3600 * 100)
    c.add('NET.STA', np.zeros((3, T), np.float32), starttime=7200.0)
    b = c.chunk_bounds('NET.STA')
    assert [x[0] for x in b] == [0, 360000, 720000]
    assert [x[1] for x in b] == [372000, 372000, T - 720000]
    assert [x[2] for x in b] == [7200.0, 10800.0, 14400.0]
    assert detect._chunk_batches(b) == [(0, 2, 360000), (2, 3, T - 720000)]
    chunks = list(c.getConData('NET.STA'))
    assert chunks[0][0].shape == (3, 372000)
    b2 = c.chunk_bounds('NET.STA', utcStart=7200.0 + 1800.0, utcEnd=7200.0 + 5400.0)
    assert [x[2] for x in b2][0] == 7200.0 and b2[-1][0] + b2[-1][1] == 540000


def test_stalta_at_matches_oracle():
    rs = np.random.RandomState(0)
    C = np.abs(rs.standard_normal(900)) * 0.05
    fetch = lambda lo, hi: np.abs(C[lo:hi])  # noqa: E731
    for LTA, STA in [(100, 0), (100, 10), (99, 7), (900, 0)]:
        want = orc.get_stalta(C, LTA, STA)
        for i in [0, 1, 48, 49, 50, 51, 450, 849, 850, 851, 899]:
            got = detect.stalta_at(fetch, i, len(C), LTA, STA)
            assert np.isclose(got, want[i], rtol=1e-12, equal_nan=True), (LTA, STA, i)
    with pytest.raises(IndexError):
        detect.stalta_at(fetch, 5, len(C), 901, 0)


def test_beta_mle_matches_scipy():
    import scipy.stats
    rs = np.random.RandomState(1)
    x = rs.beta(2.5, 60.0, size=20000)
    p, nnlf = fas.beta_mle(len(x), x.sum(), (x * x).sum(), np.log(x).sum(), np.log1p(-x).sum())
    ref = scipy.stats.beta.fit(x, floc=0, fscale=1)
    assert np.allclose(p, ref, rtol=1e-8)
    assert np.isclose(nnlf, scipy.stats.beta.nnlf(ref, x), rtol=1e-10)
    thr = scipy.stats.beta.isf(1e-12, p[0], p[1], 0, 1)
    assert 0 < thr < 1


def test_classic_sta_lta_restatement():
    rs = np.random.RandomState(2)
    a = rs.standard_normal(2000)
    a[1200:1230] *= 15
    cft = fas.classic_sta_lta(a, 10, 100)
    assert np.all(cft[:99] == 0) and cft.argmax() in range(1200, 1245) and cft.max() > 7.5
    i = 700
    assert np.isclose(cft[i], np.mean(a[i - 9:i + 1] ** 2) / np.mean(a[i - 99:i + 1] ** 2))
    assert fas._checkSTALTA(np.stack([a, a, a]), ['HHE', 'HHN', 'HHZ'], 1.0, 10, 100, 7.5) is False
    assert fas._checkSTALTA(np.stack([a, a, a]), ['HHE', 'HHN', 'HHZ'], 1.0, 10, 100, None) is True


def test_est_mag_matches_reference():
    """Oracle (literal, with UtU) against the reference's _estMag, and the product's host half
    (detect.mags_from_stats: the scalar formulas applied to a dtx_trigger_stats row) fed with
    the oracle's restatement of that row; the device half is tests/test_gpu_pipeline.py::
    test_trigger_stats_kernel."""
    z = np.load(os.path.join(G, 'estmag.npz'))
    MPcon, U, ewf, mags, Nc = z['MPcon'], z['U'], z['ewf'], z['mags'], int(z['Nc'])
    n = U.shape[1]
    for tag, trig in zip(['early', 'late'], z['trigs']):
        for kind, issub in (('ss', True), ('sg', False)):
            ref = z['%s_%s' % (kind, tag)]
            m = mags if issub else mags[:1]
            e = ewf if issub else ewf[:1]
            o = orc.est_mag(int(trig), Nc, MPcon, m, U, e, issubspace=issub)
            assert np.allclose(o, ref, rtol=1e-10), (kind, tag)
            row = orc.trigger_stats_row(int(trig), Nc, MPcon, U, e)
            c = detect.mag_constants(e, m, U, issub)
            g = detect.mags_from_stats(c, issub, n, row)
            assert np.allclose(g, ref, rtol=2e-6), (kind, tag)     # events held in float32
    c = detect.mag_constants(ewf, np.array([-99., -99, -99, -99]), U, True)
    g = detect.mags_from_stats(c, True, n, orc.trigger_stats_row(40, Nc, MPcon, U, ewf))
    assert np.isnan(g[0]) and np.isnan(g[1]) and g[2] > 0


def small_workflow(ctx):
    """Ten events in two families + a loner, 400 s of data in 120 s + 20 s chunks: small
    enough for the CPU oracle, the same calls as the GPU pipeline tests."""
    rs = np.random.RandomState(1)
    SR, NC, L = 20.0, 3, 80
    env = np.exp(-np.arange(L + 20) / (0.3 * L))
    k = np.hanning(7)
    fams = [[np.convolve(rs.standard_normal(L + 26), k / k.sum(), 'valid') * env for _ in range(NC)]
            for _ in range(2)]
    ef = EventFetcher(SR, ['HHE', 'HHN', 'HHZ'])
    n = 0
    for wav in fams:
        for _ in range(4):
            sh = int(rs.randint(0, 15))
            data = np.array([rs.uniform(.5, 2) * w[sh:sh + L] + rs.standard_normal(L) * np.std(w) / 8
                             for w in wav])
            ef.add('TA.M17A', 'ev%02d' % n, data, starttime=5000.0 + 50 * n, origintime=4990.0 + 50 * n,
                   magnitude=1.0 + 0.25 * n)
            n += 1
    ef.add('TA.M17A', 'ev%02d' % n, rs.standard_normal((NC, L)), starttime=9000., origintime=8990.,
           magnitude=2.0)
    T = int(400 * SR)
    con = rs.standard_normal((NC, T)) * 0.05
    plants = [(600, 0, 1.0), (2300, 1, 0.7), (2500, 0, 0.5), (4100, 1, 0.9), (7000, 0, 1.2)]
    for (t, f, a) in plants:
        for c in range(NC):
            con[c, t:t + L] += a * fams[f][c][5:5 + L]
    cf = ContinuousFetcher(SR, ['HHE', 'HHN', 'HHZ'], conDatDuration=120, conBuff=20)
    cf.add('TA.M17A', con.astype(np.float32), starttime=1.2e6)
    cl = createCluster(CCreq=0.5, fetch_arg=ef, ctx=ctx)
    ss = createSubSpace(Pf=1e-10, clust=cl, conDatFetcher=cf)
    ss.attachPickTimes(startSeconds=0.0, defaultDuration=1.5)
    ss.SVD(selectCriteria=3, selectValue=0.75, threshold=None, useSingles=False)
    ss.setSinglesThresholds(threshold=0.4)
    return ss, cf, plants, (SR, NC, L)


def multi_station_workflow(ctx, nsta=3, pin=False):
    """Three stations x two days of 1 Hz data in 12 h + 120 s chunks (two chunks = one day = one
    sharding unit, six units in all): the shape of BASELINE configs[3] at a size the CPU oracle
    scans in seconds."""
    rs = np.random.RandomState(7)
    SR, NC, L = 1.0, 3, 80
    env = np.exp(-np.arange(L + 20) / (0.3 * L))
    k = np.hanning(7)
    stas = ['TA.S%02d' % i for i in range(nsta)]
    ef = EventFetcher(SR, ['HHE', 'HHN', 'HHZ'])
    cf = ContinuousFetcher(SR, ['HHE', 'HHN', 'HHZ'], conDatDuration=43200, conBuff=120)
    T = 2 * 86400 + 120
    for si, sta in enumerate(stas):
        fams = [[np.convolve(rs.standard_normal(L + 26), k / k.sum(), 'valid') * env for _ in range(NC)]
                for _ in range(2)]
        n = 0
        for wav in fams:
            for _ in range(4):
                sh = int(rs.randint(0, 15))
                data = np.array([rs.uniform(.5, 2) * w[sh:sh + L] + rs.standard_normal(L) * np.std(w) / 8
                                 for w in wav])
                ef.add(sta, 'ev%02d' % n, data, starttime=5000.0 + 500 * n, origintime=4990.0 + 500 * n,
                       magnitude=1.0 + 0.25 * n)
                n += 1
        con = rs.standard_normal((NC, T)) * 0.05
        for j in range(14):
            t = int(rs.randint(100, T - 200))
            f, a = int(rs.randint(2)), rs.uniform(.5, 1.2)
            for c in range(NC):
                con[c, t:t + L] += a * fams[f][c][5:5 + L]
        cf.add(sta, con.astype(np.float32), starttime=86400.0 * 10)
    if pin:
        cf.pin()
    cl = createCluster(CCreq=0.5, fetch_arg=ef, ctx=ctx)
    ss = createSubSpace(Pf=1e-10, clust=cl, conDatFetcher=cf)
    ss.attachPickTimes(startSeconds=0.0, defaultDuration=30)
    ss.SVD(selectCriteria=3, selectValue=0.75, threshold=None, useSingles=False)
    return ss, cf, stas, (SR, NC, L)


def oracle_scan(ss, cf, sta, SR, NC, lta_s=5, sta_s=0):
    """The reference's per-chunk loop (detect.py:137-218) with the CPU oracle."""
    from detex_amd.getdata import multiplex
    names, kinds, U, off, offsets, thr = ss._detector_bank(sta)
    want = []
    hist = {nm: np.zeros(400, dtype=np.int64) for nm in names}
    for chunk, t0 in cf.getConData(sta):
        x = multiplex(list(chunk)).astype(np.float64)
        if len(x) <= U.shape[1] or (len(x) - U.shape[1]) // NC + 1 < 10:
            continue
        for d, nm in enumerate(names):
            ds = orc.mpxds_fft(x, U[off[d]:off[d + 1]].astype(np.float32), NC)
            ds, maxds = orc.scrub_ds(ds)
            ds = ds.astype(np.float32).astype(np.float64)
            hist[nm] += orc.ds_histogram(ds)
            if orc.eval_trig_con(ds.max(), np.float32(thr[nm])):
                sl = orc.get_stalta(ds, lta_s * SR, sta_s * SR)
                for t in orc.pick_triggers(ds, np.float32(thr[nm]), SR, t0, offsets[nm], stalta=sl):
                    if t['DS'] <= 1.05:
                        mg = orc.est_mag(t['index'], NC, x, ss._bank_mags[nm], U[off[d]:off[d + 1]],
                                         ss._bank_ewf[nm], kinds[d] == 'ss')
                        want.append((nm, t['STMP'], t['DS'], t['DS_STALTA'], t['MSTAMPmin'],
                                     t['MSTAMPmax'], mg))
    want.sort(key=lambda r: (r[0], r[1]))
    return want, hist


def check_detex(res, want, hist, ss, sta):
    import pandas as pd
    got = pd.concat([res['ss_df'], res['sg_df']], ignore_index=True)
    got = got.sort_values(['Name', 'STMP']).reset_index(drop=True)
    assert len(got) == len(want) and len(want) >= 3
    for (_, g), w in zip(got.iterrows(), want):
        assert g.Name == w[0] and g.STMP == w[1]
        assert abs(g.DS - w[2]) <= 1e-6
        assert abs(g.DS_STALTA - w[3]) <= 1e-6 * max(1, abs(w[3]))
        assert abs(g.MSTAMPmin - w[4]) < 1e-6 and abs(g.MSTAMPmax - w[5]) < 1e-6
        assert np.allclose([g.ProEnMag, g.Mag, g.SNR], list(w[6]), rtol=2e-5, equal_nan=True)
    allh = dict(ss.histSubSpaces[sta])
    allh.update(ss.histSingles[sta])
    for nm, h in hist.items():
        assert np.array_equal(allh[nm], h), nm


def test_detex_host_logic_on_cpu(tmp_path):
    """SubSpace.detex / scan_station end to end with the HIP context replaced by the oracle
    stand-in: batching of the chunks, sparse trigger picking, STA/LTA window placement,
    magnitude formulas, histogram bookkeeping, SQLite tables -- against the reference's
    per-chunk loop restated with the oracle."""
    import sqlite3
    ctx = OracleCtx()
    ss, cf, plants, (SR, NC, L) = small_workflow(ctx)
    db = str(tmp_path / 'SubSpace.db')
    res = ss.detex(subspaceDB=db, useSingles=True, ctx=ctx)            # estimateMags defaults to True
    want, hist = oracle_scan(ss, cf, 'TA.M17A', SR, NC)
    check_detex(res, want, hist, ss, 'TA.M17A')
    scans = [c for c in ctx.calls if c[0] == 'scan']
    assert [c[1] for c in scans] == [4]                                # 3 full chunks + the short tail: one ragged launch
    # ... and as one launch per chunk length (DETEX_AMD_RAGGED=0): the same tables
    os.environ['DETEX_AMD_RAGGED'] = '0'
    try:
        ctx.calls.clear()
        res_u = ss.detex(subspaceDB=None, useSingles=True, ctx=ctx)
    finally:
        del os.environ['DETEX_AMD_RAGGED']
    assert [c[1] for c in ctx.calls if c[0] == 'scan'] == [3, 1]
    check_detex(res_u, want, hist, ss, 'TA.M17A')
    found = sum(bool((np.abs(res['ss_df'].STMP - (1.2e6 + (t + 5) / SR)) < 1.0).any())
                for (t, f, a) in plants)
    assert found >= 3
    con = sqlite3.connect(db)
    n_db = con.execute('select count(*) from ss_df').fetchone()[0]
    con.close()
    assert n_db == len(res['ss_df'])
    # STA > 0, and fillZeros (STA/LTA forced to 0)
    res2 = ss.detex(subspaceDB=None, useSingles=True, ctx=ctx, triggerSTATime=0.5, estimateMags=False)
    want2, _ = oracle_scan(ss, cf, 'TA.M17A', SR, NC, sta_s=0.5)
    got2 = res2['ss_df'].sort_values(['Name', 'STMP'])
    w2 = [w for w in want2 if w[0].startswith('SS')]
    assert np.allclose(got2.DS_STALTA.values, [w[3] for w in w2], rtol=1e-6)
    assert got2.Mag.isna().all() and got2.SNR.isna().all()
    res3 = ss.detex(subspaceDB=None, ctx=ctx, fillZeros=True, estimateMags=False)
    assert (res3['ss_df'].DS_STALTA == 0).all() and len(res3['ss_df']) == len(got2)


def test_candidate_overflow_and_failing_batch(caplog, monkeypatch):
    """More candidates than the list holds: listed again at the counted size (no exception);
    a batch whose scan fails is logged and skipped, the others are kept (detect.py:160-174)."""
    import logging
    from detex_amd import DetexHipError
    ctx = OracleCtx()
    ss, cf, plants, (SR, NC, L) = small_workflow(ctx)
    ref = ss.detex(subspaceDB=None, ctx=ctx)

    monkeypatch.setattr(detect, 'CAND_CAP', 3)
    with caplog.at_level(logging.WARNING, logger='detex_amd.detect'):
        got = ss.detex(subspaceDB=None, ctx=OracleCtx())
    monkeypatch.setattr(detect, 'CAND_CAP', 1 << 18)
    assert got['ss_df'].equals(ref['ss_df'])
    assert any('above threshold' in r.message for r in caplog.records)

    class FailsOnce(OracleCtx):
        def __init__(self, which):
            OracleCtx.__init__(self)
            self.which, self.seen = which, 0

        def subspace_scan(self, x, **kw):
            self.seen += 1
            if self.seen == self.which:
                raise DetexHipError('dtx_subspace_scan failed (-2): injected fault')
            return OracleCtx.subspace_scan(self, x, **kw)
    # a failing ragged launch is repeated one chunk length at a time: nothing is lost
    caplog.clear()
    with caplog.at_level(logging.WARNING, logger='detex_amd.detect'):
        got = ss.detex(subspaceDB=None, ctx=FailsOnce(1))
    assert any('ragged launch failed' in r.message for r in caplog.records)
    assert got['ss_df'].equals(ref['ss_df'])
    monkeypatch.setenv('DETEX_AMD_RAGGED', '0')                          # one launch per chunk length
    for which, keep in ((2, lambda t: t < 1.2e6 + 360), (1, lambda t: t >= 1.2e6 + 360)):
        caplog.clear()
        with caplog.at_level(logging.WARNING, logger='detex_amd.detect'):
            got = ss.detex(subspaceDB=None, ctx=FailsOnce(which))
        assert any('failing to run detector' in r.message for r in caplog.records)
        # batch 1 = the three full chunks (0 .. 380 s), batch 2 = the short tail from 360 s:
        # the other batch's detections are all there
        left = ref['ss_df'][[keep(t) for t in ref['ss_df'].STMP]].reset_index(drop=True)
        assert len(got['ss_df']) == len(left)
        assert np.allclose(got['ss_df'].STMP.values, left.STMP.values)
    assert len(ref['ss_df']) >= 3


def test_validateClusters_drops_a_misaligned_event():
    """subspace.py:737-783 with the oracle stand-in: an event whose aligned waveform was
    replaced by an unrelated one no longer reaches ccReq against any later event and is
    removed; the well-aligned ones stay."""
    ctx = OracleCtx()
    ss, cf, plants, _ = small_workflow(ctx)
    df = ss.subspaces['TA.M17A']
    row = df.iloc[0]
    events = list(row.Events)
    assert len(events) >= 4
    rs = np.random.RandomState(3)
    row.AlignedTD[events[0]] = rs.standard_normal(len(row.AlignedTD[events[0]]))
    # the reference's loop (subspace.py:745-771) with the oracle, on a copy
    st = row.SampleTrims
    wf = {e: np.asarray(row.AlignedTD[e], dtype=np.float32).astype(np.float64)[st['Starttime']:st['Endtime']]
          for e in events}
    ccreq = ss.clusters['TA.M17A'].ccReq
    keep = [e for i, e in enumerate(events)
            if i == len(events) - 1 or
            max(orc.fast_normcorr(wf[e], wf[e2])[0] for e2 in events[i + 1:]) >= ccreq]
    ss.validateClusters(ctx=ctx)
    assert list(ss.subspaces['TA.M17A'].iloc[0].Events) == keep
    assert events[0] not in keep and events[0] not in ss.subspaces['TA.M17A'].iloc[0].AlignedTD
    assert len(keep) >= 3


def test_mags_from_stats_many_equals_row_by_row():
    """The vectorised magnitude formulas (all triggers of a detector at once) against the scalar
    ones the golden estmag vectors pin, on random trigger rows incl. unusable magnitudes and NaNs."""
    from detex_amd import detect
    rs = np.random.RandomState(4)
    n, ne = 600, 5
    ewf = rs.standard_normal((ne, n))
    U = np.linalg.qr(rs.standard_normal((n, 3)))[0].T
    for issub in (True, False):
        for mags in ([1.0, 2.0, -20.0, 0.5, 3.0], [-99.0] * 5, [np.nan, 1, 1, 1, 1]):
            c = detect.mag_constants(ewf, mags, U, issub)
            O = rs.standard_normal((9, 6 + ne)) * 3
            O[:, 3] = np.abs(O[:, 3]) + .1
            O[:, 4] = np.abs(O[:, 4]) + .1
            O[:, 5] = np.abs(O[:, 5]) + .1
            O[2, 5] = 0.0                                # zero noise -> inf SNR like the reference
            O[4, 6:] = np.nan
            got = detect.mags_from_stats_many(c, issub, n, O)
            for r in range(len(O)):
                want = detect.mags_from_stats(c, issub, n, O[r])
                for g, w in zip(got, want):
                    assert (np.isnan(g[r]) and np.isnan(w)) or g[r] == w or abs(g[r] - w) <= 1e-12 * abs(w), (issub, mags, r)


def test_detex_sets_the_collector_aside_and_puts_it_back():
    """SubSpace.detex runs with the process's existing objects frozen out of the garbage collector and releases
    them afterwards, also when the call raises; a caller who runs without the collector, or who has frozen objects
    of their own, is left alone."""
    import gc
    from detex_amd import SubSpace
    from detex_amd.subspace import _collector_set_aside
    assert getattr(SubSpace.detex, '__wrapped__', None) is not None      # (the product call is the wrapped one)
    seen = {}

    @_collector_set_aside
    def body(fail=False):
        seen['frozen'], seen['enabled'] = gc.get_freeze_count(), gc.isenabled()
        if fail:
            raise ValueError('x')
        return 7
    assert gc.isenabled() and gc.get_freeze_count() == 0
    assert body() == 7 and seen['frozen'] > 0 and seen['enabled'] and gc.get_freeze_count() == 0
    with pytest.raises(ValueError):
        body(fail=True)
    assert gc.get_freeze_count() == 0 and gc.isenabled()
    gc.disable()
    try:
        assert body() == 7 and seen['frozen'] == 0 and not gc.isenabled()
    finally:
        gc.enable()
    gc.freeze()
    try:
        mine = gc.get_freeze_count()
        assert body() == 7 and gc.get_freeze_count() == mine > 0
    finally:
        gc.unfreeze()


def test_unfiltered_chunks_are_handed_over_as_views_when_equally_spaced():
    """StationScan._filtered: a group of equally spaced chunks goes to the filter as ONE strided view of the
    record (overlapping chunks, no packed copy); unequal spacing or a record with a sample stride falls back to
    the packed copy.  Same values either way."""
    import torch
    from detex_amd import detect

    class _Ctx(object):
        def preprocess(self, planes, filt=None, sampling_rate=None, out=None):
            self.seen = planes
            return planes

    class _Clusters(object):
        filt = [1, 10, 2, True]

    class _SS(object):
        clusters = _Clusters()
    rec = torch.arange(3 * 5000, dtype=torch.float32).reshape(3, 5000)[:, 7:]          # (a view with a storage offset)
    job = detect.StationScan.__new__(detect.StationScan)
    job.ss = _SS()
    ctx = _Ctx()
    for starts, shares in (([100, 1000, 1900, 2800], True), ([100, 1000, 1950, 2800], False), ([40], True)):
        job.st = {'bounds': [(a, 1200, 0.0, k) for k, a in enumerate(starts)]}
        got = job._filtered(rec, 0, len(starts), 1, 100.0, 100.0, ctx=ctx)
        want = torch.stack([rec[:, a:a + 1200] for a in starts])
        assert torch.equal(got, want)
        assert (ctx.seen.untyped_storage().data_ptr() == rec.untyped_storage().data_ptr()) == shares
    job.st = {'bounds': [(100, 500, 0.0, 0), (700, 500, 0.0, 1)]}
    got = job._filtered(rec[:, ::2], 0, 2, 1, 100.0, 100.0, ctx=ctx)
    assert torch.equal(got, torch.stack([rec[:, ::2][:, a:a + 500] for a in (100, 700)]))
    assert ctx.seen.untyped_storage().data_ptr() != rec.untyped_storage().data_ptr()


def test_bank_wide_magnitudes_equal_per_detector_ones():
    """_MagBank.estimate_all (every trigger of a batch in one call, constants padded to the longest
    event list) against mags_from_stats_many detector by detector: subspaces and singles mixed,
    detectors with fewer events than the padding (their unused columns hold NaN, as the device leaves
    them unwritten), unusable magnitudes."""
    from detex_amd import detect
    rs = np.random.RandomState(11)
    n = 400
    names = ['SS0', 'SS1', 'SG0', 'SS2', 'SG1']
    kinds = ['ss', 'ss', 'sg', 'ss', 'sg']
    nev = [5, 2, 1, 7, 1]
    mags = {'SS0': [1.0, 2.0, -20.0, 0.5, 3.0], 'SS1': [-99.0, -99.0], 'SG0': [1.5], 'SS2': list(rs.rand(7) * 3),
            'SG1': [np.nan]}
    ewf = {nm: rs.standard_normal((k, n)) for nm, k in zip(names, nev)}
    ks = [3, 2, 1, 4, 1]
    doff = np.concatenate([[0], np.cumsum(ks)])
    U = np.concatenate([np.linalg.qr(rs.standard_normal((n, k)))[0].T for k in ks])

    class _SS(object):
        _bank_aux = {'X': (ewf, mags)}
    bank = detect._MagBank(_SS(), 'X', names, kinds, U, doff, 'cpu')
    assert bank.emax == 7
    k = 40
    dets = rs.randint(0, len(names), k)
    O = rs.standard_normal((k, 6 + bank.emax)) * 3
    O[:, 3:6] = np.abs(O[:, 3:6]) + .1
    O[3, 5] = 0.0
    for r in range(k):
        O[r, 6 + nev[dets[r]]:] = np.nan
    O[7, 6:] = np.nan
    got = bank.estimate_all(dets, n, O)
    for d in range(len(names)):
        sel = np.nonzero(dets == d)[0]
        want = bank.estimate_many(d, kinds[d] == 'ss', n, O[sel])
        for g, w in zip(got, want):
            np.testing.assert_allclose(g[sel], w, rtol=1e-12, atol=0, equal_nan=True)


def test_grouped_trigger_picking_equals_group_by_group():
    """pick_triggers_grouped (the single-peak groups answered with array operations, the rest by the loop)
    against pick_triggers_sparse group by group: peaks at the block's ends, groups wider than the zeroed
    range, two peaks in one group, ties for the maximum, a maximum not above threshold."""
    from detex_amd import detect
    rs = np.random.RandomState(5)
    sr, P = 10.0, 5000
    for trial in range(30):
        groups = []
        for g in range(rs.randint(1, 12)):
            kind = rs.randint(0, 6)
            c0 = [0, P - 8, rs.randint(300, P - 300), rs.randint(300, P - 300), rs.randint(300, P - 900), rs.randint(300, P - 300)][kind]
            width = [6, 8, 5, 150, 700, 199][kind]
            idx = np.unique(np.clip(c0 + rs.randint(0, width + 1, rs.randint(1, 9)), 0, P - 1))
            val = (0.5 + 0.5 * rs.rand(len(idx))).astype(np.float32)
            if rs.rand() < 0.3 and len(idx) > 1:
                val[-1] = val[0] = val.max()             # a tie: the first occurrence wins
            groups.append((idx, val))
        cp = np.concatenate([g[0] for g in groups])
        cv = np.concatenate([g[1] for g in groups])
        cnt = np.array([len(g[0]) for g in groups])
        last = np.cumsum(cnt)
        first = last - cnt
        thr = np.where(rs.rand(len(groups)) < 0.2, 0.99, 0.5)
        maxds = np.array([g[1].max() for g in groups], dtype=np.float64)
        maxds[rs.rand(len(groups)) < 0.15] = 0.5         # not strictly above threshold: no trigger
        Pg = np.full(len(groups), P)
        got = detect.pick_triggers_grouped(cp, cv, first, last, Pg, thr, maxds, sr)
        for g, (idx, val) in enumerate(groups):
            want = detect.pick_triggers_sparse(idx, val, P, float(thr[g]), sr, float(maxds[g]))
            assert got[g] == want, (trial, g)


def test_centered_lo_many_equals_scalar():
    from detex_amd import detect
    for P in (1, 2, 3, 5, 8, 40, 41):
        for n in (0, 1, 2, 3, 4, 5, 7, 40, 41, 60):
            i = np.arange(P)
            lo, bad = detect._centered_lo_many(i, n, np.full(P, P))
            for j in range(P):
                try:
                    want = detect._centered_lo(j, n, P)
                    assert not bad[j] and lo[j] == want, (P, n, j)
                except IndexError:
                    assert bad[j] and lo[j] == -1, (P, n, j)


def test_raw_counts_trimming_is_gated_like_applyFilter():
    """construct.py:1003-1023 on a record that arrives as int32 fragments (ContinuousFetcher.add_counts):
    a chunk with ONE trace per channel is never merged -- genuine zero counts in it are ordinary samples
    and it is only trimmed to the span all channels cover; only a chunk with several traces on some channel
    goes through _mergeChannels' longest-run rule (where a zero is a gap); a chunk a channel is missing
    from is dropped.  Against the oracle's chunk_span -> filter -> DS chain, on the CPU stand-in."""
    from detex_amd.getdata import multiplex
    rs = np.random.RandomState(33)
    sr, Nc, L = 20.0, 3, 160
    filt = [1, 8, 2, True]
    wav = [np.convolve(rs.standard_normal(L + 6), np.hanning(7), 'valid') * np.exp(-np.arange(L) / 50.0)
           for _ in range(Nc)]
    wavf = orc.apply_filter([w.astype(np.float32) for w in wav], filt, sr)
    ef = EventFetcher(sr, ['E', 'N', 'Z'])
    for i in range(4):
        ef.add('XX.STA', 'ev%d' % i, np.array([rs.uniform(.5, 2) * w + 0.02 * rs.standard_normal(L)
                                                for w in wavf]), starttime=100. * i)
    T = int(2700 * sr)                                   # chunks of 600 s + 60 s: 0, 12000, 24000, 36000, 48000
    con = rs.standard_normal((Nc, T)) * 300
    for t in (4000, 9000, 16000, 22000, 27000, 40000, 50000):
        for c in range(Nc):
            con[c, t:t + L] += 3000.0 * wav[c]
    cnt = np.rint(con).astype(np.int32)
    assert (cnt[:, :11800] == 0).sum() >= 3              # zero-mean counts: exact zeros are routine
    # chunk 0 (0..13199): one trace per channel, WITH zero counts; channel 1 ends early (at 11800).
    # chunk 1 (12000..25199): channel 1 starts late (at 13500), still one trace per channel.  chunk 2 (24000..37199): channel 2 in two traces with a hole ->
    # _mergeChannels.  chunk 3 (36000..49199): channel 0 has no data at all.  chunk 4 (48000..53999): whole.
    cnt[2, 30000:30100] = 5                              # (what is under the hole does not matter)
    seg = cnt[:, 24000:37200]
    seg[seg == 0] = 1                                    # in the merged chunk a zero would be a gap (construct.py:1047-1050)
    frags = [(0, 0, cnt[0, :36000]), (0, 49200, cnt[0, 49200:]),
             (1, 0, cnt[1, :11800]), (1, 13500, cnt[1, 13500:]),
             (2, 0, cnt[2, :30000]), (2, 30100, cnt[2, 30100:])]
    cf = ContinuousFetcher(sr, ['E', 'N', 'Z'], conDatDuration=600, conBuff=60)
    cf.add_counts('XX.STA', frags, T, starttime=6.0e5)
    ctx = OracleCtx()
    cl = createCluster(CCreq=0.5, fetch_arg=ef, filt=filt, ctx=ctx)
    ss = createSubSpace(clust=cl, conDatFetcher=cf)
    ss.attachPickTimes(0.0, 6.0)
    ss.SVD(selectCriteria=4, selectValue=0, threshold=0.35, useSingles=False)
    res = ss.detex(subspaceDB=None, ctx=ctx, estimateMags=False)
    names, kinds, U, off, offsets, thr = ss._detector_bank('XX.STA', useSingles=False)
    merged = orc.merge_fill(frags, Nc, T)
    spans, want = [], []
    for (s0, n, ts) in cf.chunk_bounds('XX.STA'):
        sp = orc.chunk_span(frags, Nc, s0, n)
        spans.append(sp)
        if sp is None:
            continue
        a, m = sp
        x = multiplex(orc.apply_filter([p.astype(np.float32) for p in merged[:, a:a + m]], filt, sr))
        if len(x) <= U.shape[1]:
            continue
        ds = orc.mpxds_fft(x, U.astype(np.float32), Nc)
        if orc.eval_trig_con(ds.max(), np.float32(thr['SS0'])):
            want += [(t['STMP'], t['DS']) for t in orc.pick_triggers(ds, np.float32(thr['SS0']), sr, ts + (a - s0) / sr)]
    assert spans[0] == (0, 11800)                        # zeros inside: untouched; common span of single traces
    assert spans[1] == (13500, 11700)
    assert spans[2] == (24000, 6000)                     # channel 0 ends at 36000: 24000..29999 beats 30100..35999
    assert spans[3] is None and spans[4] == (49200, T - 49200)
    # the device pass ran for the fractured chunk only
    assert [c for c in ctx.calls if c[0] == 'gapfree'] == [('gapfree', 13200)]
    got = res['ss_df'].sort_values('STMP')
    assert len(got) == len(want) >= 4
    for (_, g), w in zip(got.iterrows(), sorted(want)):
        assert g.STMP == w[0] and abs(g.DS - w[1]) <= 1e-5


def test_resolve_overlaps_rule():
    """Overlapping count fragments (re-sent telemetry packets) are resolved on the host as ObsPy's default merge
    resolves them (construct.py:1038 / 1070 st.merge(fill_value=0.0) -> Trace.__add__, method 0): an overlap whose
    samples are all identical is kept once, otherwise the whole overlap becomes the fill value 0.  ObsPy is not
    installable here (parity vs ObsPy unpinned): the fragment-list implementation (getdata.resolve_overlaps) is
    held to an independent dense restatement (oracle.merge_fill_overlaps) and to hand-written cases."""
    from detex_amd.getdata import resolve_overlaps, trace_extents
    a = np.arange(1, 21, dtype=np.int32)

    def dense(frs, T=30):
        out = np.zeros(T)
        for (_, t0, x) in frs:
            assert not out[t0:t0 + len(x)].any() or not np.any(x)      # non-overlapping
            out[t0:t0 + len(x)] = x
        return out
    # identical partial overlap: AAAAAAAA + ....AAAAAAAA -> one trace
    got = resolve_overlaps([(0, 0, a[:12]), (0, 8, a[8:20])])
    assert np.array_equal(dense(got)[:20], a) and got.traces == [(0, 0, 12), (0, 8, 12)]
    # differing partial overlap: the whole overlap (8..11) is zeroed, not just the sample that differs
    b = a[8:20].copy()
    b[1] += 5
    want = a.astype(float).copy()
    want[8:12] = 0
    assert np.array_equal(dense(resolve_overlaps([(0, 0, a[:12]), (0, 8, b)]))[:20], want)
    # contained trace: identical -> dropped; differing -> its extent zeroed
    assert np.array_equal(dense(resolve_overlaps([(0, 0, a), (0, 5, a[5:9])]))[:20], a)
    c = a[5:9].copy()
    c[0] = 99
    want = a.astype(float).copy()
    want[5:9] = 0
    assert np.array_equal(dense(resolve_overlaps([(0, 0, a), (0, 5, c)]))[:20], want)
    # a third packet identical to the ORIGINAL data over a zeroed overlap now differs from the running trace
    want[5:9] = 0
    assert np.array_equal(dense(resolve_overlaps([(0, 0, a), (0, 5, c), (0, 5, a[5:9])]))[:20], want)
    # no overlap: every trace comes back as it is -- abutting traces stay two traces (a seam is a seam)
    got = resolve_overlaps([(1, 10, a[10:]), (1, 0, a[:10]), (0, 3, a[:4])])
    assert [(c_, t0, len(x)) for c_, t0, x in got] == [(0, 3, 4), (1, 0, 10), (1, 10, 10)] and got.traces is None
    assert trace_extents(got) == [(0, 3, 4), (1, 0, 10), (1, 10, 10)]
    # random records against the dense restatement
    rs = np.random.RandomState(0)
    nover = 0
    for trial in range(400):
        Nc, T = int(rs.randint(1, 4)), int(rs.randint(30, 200))
        base = rs.randint(1, 50, size=(Nc, T)).astype(np.int32)
        frags = []
        for ch in range(Nc):
            for _ in range(int(rs.randint(1, 6))):
                lo = int(rs.randint(0, T - 1))
                hi = int(rs.randint(lo + 1, min(T, lo + 60) + 1))
                x = base[ch, lo:hi].copy()
                r = rs.rand()
                if r < 0.3:
                    x[rs.randint(len(x))] += 1
                elif r < 0.4:
                    x[:] = 0
                frags.append((ch, lo, x))
        got = resolve_overlaps(frags)
        planes = np.zeros((Nc, T))
        cover = np.zeros((Nc, T), int)
        for (ch, t0, x) in got:
            planes[ch, t0:t0 + len(x)] = x
            cover[ch, t0:t0 + len(x)] += 1
        assert cover.max() <= 1
        assert np.array_equal(planes, orc.merge_fill_overlaps(frags, Nc, T)), trial
        nover += got.traces is not None
    assert nover > 200


def test_pickled_cluster_stream_rebuilds_the_same_subspaces(tmp_path, monkeypatch):
    """ClusterStream.write() (subspace.py:261-267) with the lazily multiplexed MPtd of round 5: a cluster read back
    from its pickle gives createSubSpace the same aligned waveforms."""
    import pickle
    monkeypatch.chdir(tmp_path)
    ctx = OracleCtx()
    ss, cf, plants, _ = small_workflow(ctx)
    cl = ss.clusters
    cl.filename = 'clust.pkl'
    cl.write()
    with open('clust.pkl', 'rb') as f:
        cl2 = pickle.load(f)
    ss2 = createSubSpace(Pf=1e-10, clust=cl2, conDatFetcher=cf)
    for sta in ss.subspaces:
        for (_, r1), (_, r2) in zip(ss.subspaces[sta].iterrows(), ss2.subspaces[sta].iterrows()):
            assert r1.Events == r2.Events
            assert all(np.array_equal(r1.AlignedTD[k], r2.AlignedTD[k]) for k in r1.AlignedTD)
    for sta in ss.singles:
        for (_, r1), (_, r2) in zip(ss.singles[sta].iterrows(), ss2.singles[sta].iterrows()):
            assert all(np.array_equal(r1.MPtd[k], r2.MPtd[k]) for k in r1.MPtd)


def test_clip_fragments_keeps_data_and_trace_extents():
    """getdata.clip_fragments (the shard of a rank / the window of an event: parallel.rebase_fragments,
    construct._preprocess_events): resolved fragments clipped to spans and re-based give the dense record restricted
    to the spans, and the ORIGINAL traces' extents are clipped the same way (the 'fractured' decisions of a shard
    equal those of the whole record on the chunks it holds)."""
    from detex_amd.getdata import clip_fragments, resolve_overlaps, trace_extents
    rs = np.random.RandomState(3)
    for trial in range(200):
        Nc, T = int(rs.randint(1, 4)), int(rs.randint(60, 300))
        base = rs.randint(1, 50, size=(Nc, T)).astype(np.int32)
        frags = []
        for ch in range(Nc):
            for _ in range(int(rs.randint(1, 6))):
                lo = int(rs.randint(0, T - 1))
                hi = int(rs.randint(lo + 1, min(T, lo + 80) + 1))
                x = base[ch, lo:hi].copy()
                if rs.rand() < 0.3:
                    x[rs.randint(len(x))] += 1
                frags.append((ch, lo, x))
        res = resolve_overlaps(frags)
        dense = orc.merge_fill_overlaps(frags, Nc, T)
        # two disjoint spans laid end to end (parallel.plan_shard's runs)
        a0 = int(rs.randint(0, T // 2)); b0 = int(rs.randint(a0 + 1, T // 2 + 1))
        a1 = int(rs.randint(T // 2, T - 1)); b1 = int(rs.randint(a1 + 1, T + 1))
        spans = [(a0, b0, 0), (a1, b1, b0 - a0)]
        cl = clip_fragments(res, spans)
        got = np.zeros((Nc, (b0 - a0) + (b1 - a1)))
        for (c, t0, x) in cl:
            assert not got[c, t0:t0 + len(x)].any()
            got[c, t0:t0 + len(x)] = x
        want = np.concatenate([dense[:, a0:b0], dense[:, a1:b1]], axis=1)
        assert np.array_equal(got, want), trial
        # the traces' extents: every original trace clipped to every span
        ext = sorted((c, off + max(t0, a) - a, min(t0 + len(x), b) - max(t0, a))
                     for (c, t0, x) in frags for (a, b, off) in spans if min(t0 + len(x), b) > max(t0, a))
        if res.traces is not None:
            assert sorted(trace_extents(cl)) == ext, trial
        else:
            assert sorted(trace_extents(cl)) == sorted((c, t0, len(x)) for (c, t0, x) in cl)


def test_raw_counts_with_resent_packets():
    """A record with re-sent (overlapping) packets through ContinuousFetcher.add_counts -> SubSpace.detex on the
    CPU stand-in, against the oracle's chunk_span -> filter -> DS chain on the ORIGINAL traces:
    chunk 0: a duplicate packet, identical, no zero anywhere -> the merged chunk is scanned whole
             (_mergeChannels finds no zero: `st = st1`, construct.py:1065-1066);
    chunk 1: a packet that DIFFERS from the data it overlaps -> the overlap is zero-filled -> the longest
             stretch beside it is kept;
    chunk 2: a hole on one channel AND a duplicate packet inside the kept stretch -> the unmerged stream,
             trimmed (construct.py:1064), still has two traces there -> 'Stream is too fractured' -> dropped."""
    from detex_amd.getdata import multiplex
    rs = np.random.RandomState(34)
    sr, Nc, L = 20.0, 3, 160
    filt = [1, 8, 2, True]
    wav = [np.convolve(rs.standard_normal(L + 6), np.hanning(7), 'valid') * np.exp(-np.arange(L) / 50.0)
           for _ in range(Nc)]
    wavf = orc.apply_filter([w.astype(np.float32) for w in wav], filt, sr)
    ef = EventFetcher(sr, ['E', 'N', 'Z'])
    for i in range(4):
        ef.add('XX.STA', 'ev%d' % i, np.array([rs.uniform(.5, 2) * w + 0.02 * rs.standard_normal(L)
                                                for w in wavf]), starttime=100. * i)
    T = 36000                                            # chunks of 600 s (no buffer): 0, 12000, 24000
    con = rs.standard_normal((Nc, T)) * 300
    for t in (4000, 9000, 14000, 22000, 27000, 33000):
        for c in range(Nc):
            con[c, t:t + L] += 3000.0 * wav[c]
    cnt = np.rint(con).astype(np.int32)
    cnt[cnt == 0] = 1
    bad = cnt[1, 18000:18400].copy()
    bad[7] += 3
    frags = [(0, 0, cnt[0, :12000]), (0, 12000, cnt[0, 12000:24000]), (0, 24000, cnt[0, 24000:]),
             (1, 0, cnt[1, :12000]), (1, 12000, cnt[1, 12000:24000]), (1, 24000, cnt[1, 24000:]),
             (2, 0, cnt[2, :12000]), (2, 12000, cnt[2, 12000:24000]),
             (2, 24000, cnt[2, 24000:25000]), (2, 25100, cnt[2, 25100:]),          # chunk 2: a hole ...
             (0, 5000, cnt[0, 5000:5600]),                                         # chunk 0: re-sent, identical
             (1, 18000, bad),                                                      # chunk 1: re-sent, differs
             (1, 30000, cnt[1, 30000:30500])]                                      # chunk 2: ... and a duplicate
    cf = ContinuousFetcher(sr, ['E', 'N', 'Z'], conDatDuration=600, conBuff=0)
    cf.add_counts('XX.STA', frags, T, starttime=6.0e5)
    assert cf.counts['XX.STA'].traces is not None
    ctx = OracleCtx()
    cl = createCluster(CCreq=0.5, fetch_arg=ef, filt=filt, ctx=ctx)
    ss = createSubSpace(clust=cl, conDatFetcher=cf)
    ss.attachPickTimes(0.0, 6.0)
    ss.SVD(selectCriteria=4, selectValue=0, threshold=0.35, useSingles=False)
    res = ss.detex(subspaceDB=None, ctx=ctx, estimateMags=False)
    names, kinds, U, off, offsets, thr = ss._detector_bank('XX.STA', useSingles=False)
    merged = orc.merge_fill_overlaps(frags, Nc, T)
    spans, want = [], []
    for (s0, n, ts) in cf.chunk_bounds('XX.STA'):
        sp = orc.chunk_span(frags, Nc, s0, n)
        spans.append(sp)
        if sp is None:
            continue
        a, m = sp
        x = multiplex(orc.apply_filter([p.astype(np.float32) for p in merged[:, a:a + m]], filt, sr))
        ds = orc.mpxds_fft(x, U.astype(np.float32), Nc)
        if orc.eval_trig_con(ds.max(), np.float32(thr['SS0'])):
            want += [(t['STMP'], t['DS']) for t in orc.pick_triggers(ds, np.float32(thr['SS0']), sr, ts + (a - s0) / sr)]
    assert spans[:3] == [(0, 12000), (12000, 6000), None], spans     # (chunk 1: 12000..17999 beats 18400..23999)
    got = res['ss_df'].sort_values('STMP')
    assert len(got) == len(want) >= 3
    for (_, g), w in zip(got.iterrows(), sorted(want)):
        assert g.STMP == w[0] and abs(g.DS - w[1]) <= 1e-5
    # fillZeros: merged only (_mergeChannelsFill, construct.py:1069-1071) -- every chunk is scanned whole
    resz = ss.detex(subspaceDB=None, ctx=ctx, estimateMags=False, fillZeros=True)
    assert len(resz['ss_df']) >= 6


def raw_event_records(seed=5, sr=40.0, Nc=3, nev=7, pre=30.0, post=50.0):
    """Unprocessed records around template events (two families + a loner): drifting offsets, a slow trend
    and noise on top of band-limited wavelets; origin `pre` seconds into each record."""
    rs = np.random.RandomState(seed)
    T = int((pre + post) * sr)
    L = int(12 * sr)
    k = np.hanning(9)
    wavs = [[np.convolve(rs.standard_normal(L + 8), k / k.sum(), 'valid') * np.exp(-np.arange(L) / (0.3 * L)) * 400
             for _ in range(Nc)] for _ in range(2)]
    recs = []
    for e in range(nev):
        fam = e % 2 if e < nev - 1 else -1
        data = rs.standard_normal((Nc, T)) * 6.0 + 1000.0 + 0.05 * np.arange(T)[None, :] * rs.uniform(-1, 1, (Nc, 1))
        arr = int((pre + 2.0) * sr) + int(rs.randint(0, 12))
        if fam >= 0:
            a = rs.uniform(.6, 1.5)
            for c in range(Nc):
                data[c, arr:arr + L] += a * wavs[fam][c]
        else:
            data[:, arr:arr + L] += rs.standard_normal((Nc, L)) * 80
        t0 = 7200.0 * e + 100.0
        recs.append(dict(name='ev%02d' % e, data=np.round(data).astype(np.float32), starttime=t0, origintime=t0 + pre,
                         magnitude=1.0 + 0.1 * e))
    return recs, sr, Nc


def prefiltered_from_raw(recs, sr, Nc, filt, trim, dec):
    """The reference's order of work on the CPU oracle: ObsPy-trim the window, decimate, detrend, band-pass
    (getdata.py:441-444, construct.py:1016-1028)."""
    from detex_amd.getdata import trim_window
    ef = EventFetcher(sr / (dec or 1), ['E', 'N', 'Z'][:Nc])
    for r in recs:
        i0, cnt = trim_window(r['starttime'], r['data'].shape[1], sr, r['origintime'] - trim[0], r['origintime'] + trim[1])
        ch = [np.asarray(c, np.float32) for c in r['data'][:, i0:i0 + cnt]]
        if dec:
            ch = orc.decimate(ch, dec, sr)
        ch = orc.apply_filter([np.asarray(c, np.float32) for c in ch], filt, sr / (dec or 1))
        ef.add('XX.RAW', r['name'], np.array(ch), starttime=r['starttime'] + i0 / sr, origintime=r['origintime'],
               magnitude=r['magnitude'])
    return ef


@pytest.mark.parametrize('dec', [None, 2])
def test_raw_template_events_host_logic(dec):
    """createCluster on RAW template records (EventFetcher.add_raw): window [origin - trim[0], origin + trim[1]]
    -> decimate -> detrend -> band-pass -> multiplex (construct.py:852-925, 990-1030) gives the clusters,
    CCs, lags and start times of the run on waveforms pre-processed the same way beforehand.  CPU stand-in
    context: the host logic; the device kernels are held to the same comparison in test_gpu_preproc.py."""
    recs, sr, Nc = raw_event_records()
    filt, trim = [1, 8, 2, True], [5, 20]
    efr = EventFetcher(sr, ['E', 'N', 'Z'])
    for r in recs:
        efr.add_raw('XX.RAW', r['name'], r['data'], r['starttime'], r['origintime'], r['magnitude'])
    a = createCluster(CCreq=0.5, fetch_arg=efr, filt=filt, trim=trim, decimate=dec, ctx=OracleCtx(), saveclust=False)
    b = createCluster(CCreq=0.5, fetch_arg=prefiltered_from_raw(recs, sr, Nc, filt, trim, dec), filt=filt, trim=trim,
                      decimate=dec, ctx=OracleCtx(), saveclust=False)
    ra, rb = a.trdf.iloc[0], b.trdf.iloc[0]
    assert list(ra.Events) == list(rb.Events) and len(ra.Events) == 7
    assert np.allclose(ra.CCs.values.astype(float), rb.CCs.values.astype(float), atol=1e-6, equal_nan=True)
    assert np.array_equal(np.nan_to_num(ra.Lags.values.astype(float), nan=-1), np.nan_to_num(rb.Lags.values.astype(float), nan=-1))
    assert sorted(map(sorted, a['RAW'].clusts)) == sorted(map(sorted, b['RAW'].clusts))
    # (decimation runs BEFORE the detrend, as in the reference: the causal low-pass rings on the records' 1000-count
    # offset at the start of every window and the families blur -- two clusters only without it)
    assert len(a['RAW'].clusts) == (2 if dec is None else 1)
    q = dec or 1
    want_len = Nc * (((int(round(25 * sr)) + 1) + q - 1) // q)       # inclusive ObsPy window, then data[::q]
    for ev in ra.Events:
        assert len(a._mptd['XX.RAW'][ev]) == want_len
        sa, sb = ra.Stats[ev], rb.Stats[ev]
        assert sa['sampling_rate'] == sr / q and abs(sa['starttime'] - sb['starttime']) < 1e-9
        assert abs(sa['offset'] + trim[0]) < 1e-9                    # the window starts trim[0] before the origin
    # enforceOrigin (construct.py:884-885): the waveforms start at the origin time
    c = createCluster(CCreq=0.5, fetch_arg=efr, filt=filt, trim=trim, decimate=dec, enforceOrigin=True,
                      ctx=OracleCtx(), saveclust=False)
    for ev in c.trdf.iloc[0].Events:
        assert abs(c.trdf.iloc[0].Stats[ev]['offset']) < 1e-9
        assert np.array_equal(c._mptd['XX.RAW'][ev], a._mptd['XX.RAW'][ev][Nc * int(round(trim[0] * sr / q)):])


def test_raw_template_events_fragments_and_drop_rules(caplog):
    """Raw int32 count fragments as templates: a record with a gap keeps its longest gap-free stretch
    (_mergeChannels), a record still in several traces after that is dropped as fractured; a record that is
    much too short (< 20 % of the median) and one with an all-zero channel are removed (construct.py:897-916)."""
    import logging
    recs, sr, Nc = raw_event_records(nev=8)
    filt, trim = [1, 8, 2, True], [5, 20]
    ef = EventFetcher(sr, ['E', 'N', 'Z'])
    T = recs[0]['data'].shape[1]
    for i, r in enumerate(recs):
        cnt = r['data'].astype(np.int32)
        frs = [(c, 0, cnt[c]) for c in range(Nc)]
        if i == 1:      # a gap BEFORE the window on one channel: the window itself is whole
            frs = [(0, 0, cnt[0, :200]), (0, 260, cnt[0, 260:])] + [(c, 0, cnt[c]) for c in (1, 2)]
        if i == 2:      # a gap inside the window, near its start: the long stretch behind it is kept
            g0 = int((30 - 5) * sr) + 40
            frs = [(0, 0, cnt[0, :g0]), (0, g0 + 30, cnt[0, g0 + 30:])] + [(c, 0, cnt[c]) for c in (1, 2)]
        if i == 3:      # the record ends 3 s after the window opens: far too short
            frs = [(c, 0, cnt[c, :int((30 - 5 + 3) * sr)]) for c in range(Nc)]
        if i == 4:      # dead channel
            frs = [(0, 0, cnt[0]), (1, 0, cnt[1]), (2, 0, np.zeros(T, np.int32))]
        ef.add_counts('XX.RAW', r['name'], frs, T, r['starttime'], r['origintime'], r['magnitude'])
    with caplog.at_level(logging.WARNING, logger='detex_amd.construct'):
        cl = createCluster(CCreq=0.4, fetch_arg=ef, filt=filt, trim=trim, ctx=OracleCtx(), saveclust=False)
    row = cl.trdf.iloc[0]
    # ev03 (short) and ev04 (dead channel) are gone; ev02 survives shortened -> every event is trimmed to it
    # only if it is within 90 % of the median length, else it is dropped too (_testStreamLengths)
    assert 'ev03' not in row.Events and 'ev04' not in row.Events and 'ev01' in row.Events
    assert any('fractured or missing data' in m for m in caplog.messages)
    assert any('all zeros' in m for m in caplog.messages)
    full = Nc * (int(round(25 * sr)) + 1)
    assert 'ev02' in row.Events and len(cl._mptd['XX.RAW']['ev00']) == full - Nc * 70
    # the whole-window events equal the float-record run, cut to the common length
    efr = EventFetcher(sr, ['E', 'N', 'Z'])
    for r in recs:
        efr.add_raw('XX.RAW', r['name'], r['data'], r['starttime'], r['origintime'], r['magnitude'])
    ref = createCluster(CCreq=0.4, fetch_arg=efr, filt=filt, trim=trim, ctx=OracleCtx(), saveclust=False)
    assert np.allclose(cl._mptd['XX.RAW']['ev00'], ref._mptd['XX.RAW']['ev00'][:full - Nc * 70], atol=1e-6)


def test_utcSaves_rows_match_the_reference_layout(tmp_path, monkeypatch):
    """SubSpace.detex(utcSaves=[...]) (detect.py:63-75, 183-185, 297-318): one row per detector for every chunk
    that covers a listed time, holding the chunk's multiplexed data, its whole detection-statistic vector and
    the STA/LTA vector; written to UTCsaves.pkl, an earlier file appended."""
    monkeypatch.chdir(tmp_path)
    ctx = OracleCtx()
    ss, cf, plants, (SR, NC, L) = small_workflow(ctx)
    b = cf.chunk_bounds('TA.M17A')
    t_in = b[1][2] + 30.0                                  # inside chunk 1 only (chunks overlap by 20 s)
    res = ss.detex(subspaceDB=None, useSingles=True, ctx=ctx, utcSaves=[t_in, 5.0])
    df = res['utc_saves']
    names, kinds, U, off, offsets, thr = ss._detector_bank('TA.M17A', True, True)
    assert list(df.columns) == ['Station', 'Name', 'Threshold', 'offset', 'TS1', 'TS2', 'utcSaves', 'MPcon', 'SSdetect',
                                'STALTA', 'TimeStamp', 'SampRate', 'MaxDS', 'MaxSTALTA', 'Nc', 'File']
    assert sorted(df.Name) == sorted(names) and set(df.TS1) == {b[1][2]}
    chunk = cf.data['TA.M17A'][0][:, b[1][0]:b[1][0] + b[1][1]]
    mp = orc.multiplex(list(chunk))
    for _, r in df.iterrows():
        d = names.index(r.Name)
        assert np.array_equal(r.MPcon, mp.astype(np.float32)) and list(r.utcSaves) == [t_in]
        want = orc.scrub_ds(orc.mpxds_fft(mp.astype(np.float64), U[off[d]:off[d + 1]], NC))[0]
        assert np.allclose(r.SSdetect, want, atol=1e-6) and abs(r.MaxDS - want.max()) < 1e-6
        sl = orc.get_stalta(np.asarray(r.SSdetect, np.float64).copy(), 5 * SR, 0)
        assert np.allclose(r.STALTA, sl, rtol=1e-9, equal_nan=True)
        assert r.TS2 == r.TS1 + len(mp) / (SR * NC) and r.Threshold == np.float32(thr[r.Name])
    assert len(pd_read(tmp_path / 'UTCsaves.pkl')) == len(df)
    again = ss.detex(subspaceDB=None, useSingles=True, ctx=ctx, utcSaves=(t_in,))
    assert len(again['utc_saves']) == 2 * len(df)          # the earlier file is appended (detect.py:101-105)
    # the plain table is unchanged by the option
    plain = ss.detex(subspaceDB=None, useSingles=True, ctx=ctx)
    assert plain['ss_df'].values.tolist() == res['ss_df'].values.tolist() and 'utc_saves' not in plain
    with pytest.raises(Exception):
        ss.detex(subspaceDB=None, ctx=ctx, utcSaves='2015-01-01')
    # a datetime (timestamp() is a method) and an ObsPy-style object (timestamp is a property) read alike
    import datetime

    class _UTC(object):
        timestamp = t_in
    dt = datetime.datetime.fromtimestamp(t_in, tz=datetime.timezone.utc)
    for form in (dt, _UTC()):
        r2 = ss.detex(subspaceDB=None, useSingles=True, ctx=ctx, utcSaves=[form])
        assert len(r2['utc_saves']) % len(df) == 0 and len(r2['utc_saves']) > 2 * len(df)


def pd_read(path):
    import pandas as pd
    return pd.read_pickle(str(path))


def test_classifyEvents_scans_the_template_events(tmp_path, monkeypatch):
    """SubSpace.detex(classifyEvents=...) (detect.py:51-55, 151-152; subspace.py:1813-1825): the detectors
    run over the event waveforms of the cluster's event fetcher; the highest detection statistic of every
    (event, station, detector) is returned and pickled to eventCorFile_<station>.pkl."""
    monkeypatch.chdir(tmp_path)
    ctx = OracleCtx()
    ss, cf, plants, (SR, NC, L) = small_workflow(ctx)
    ss.cfetcher = None                                     # no continuous data needed
    res = ss.detex(subspaceDB=None, useSingles=False, ctx=ctx, classifyEvents=True, eventCorFile='EvCor')
    ec = res['event_cors']
    df = ss.subspaces['TA.M17A']
    assert set(ec.Event) == set('ev%02d' % i for i in range(9)) and set(ec.Name) == set(df.Name)
    assert len(ec) == 9 * len(df)
    # every member event of a subspace is best matched by its own detector, well above the threshold
    for _, row in df.iterrows():
        for ev in row.Events:
            mine = ec[(ec.Event == ev)].set_index('Name').MaxDS
            assert mine.idxmax() == row.Name and mine[row.Name] > 0.5
    # the values are the oracle's maxima over the event's own windows
    names, kinds, U, off, offsets, thr = ss._detector_bank('TA.M17A', True, False)
    rec = ss.clusters.fetcher.events['TA.M17A']['ev03']
    mp = orc.multiplex(list(np.asarray(rec['data'], np.float32)))
    for d, nm in enumerate(names):
        want = orc.scrub_ds(orc.mpxds_fft(mp.astype(np.float64), U[off[d]:off[d + 1]], NC))[0].max()
        got = float(ec[(ec.Event == 'ev03') & (ec.Name == nm)].MaxDS.iloc[0])
        assert abs(got - want) < 1e-6
    assert len(pd_read(tmp_path / 'EvCor_TA.M17A.pkl')) == len(ec)
    # a template key (DataFrame with NAME) restricts the run
    import pandas as pd
    few = ss.detex(subspaceDB=None, useSingles=False, ctx=ctx, classifyEvents=pd.DataFrame({'NAME': ['ev01', 'ev05']}),
                   eventCorFile=None)
    assert sorted(set(few['event_cors'].Event)) == ['ev01', 'ev05']
    # detections at the events' own times come out in the ordinary table too
    assert len(res['ss_df']) >= 8


def test_long_records_are_scanned_in_day_units(tmp_path, monkeypatch):
    """A record whose DS tensor exceeds the budget is scanned day by day (SubSpace._scan_units): the same tables,
    histograms and SQLite rows as the one-launch run, row for row; the SQLite file is fed by the writer
    thread while the units are scanned."""
    import sqlite3
    import pandas as pd
    ctx = OracleCtx()
    ss, cf, stas, _ = multi_station_workflow(ctx)
    db1, db2 = str(tmp_path / 'one.db'), str(tmp_path / 'units.db')
    ref = ss.detex(subspaceDB=db1, ctx=ctx)
    n_scans = len([c for c in ctx.calls if c[0] == 'scan'])
    hist = {sta: {k: v.copy() for k, v in ss.histSubSpaces[sta].items()} for sta in stas}
    monkeypatch.setenv('DETEX_AMD_DS_BUDGET_MB', '0')
    got = ss.detex(subspaceDB=db2, ctx=ctx)
    # five chunks per station, two per day: three units per station instead of one launch
    assert len([c for c in ctx.calls if c[0] == 'scan']) - n_scans == 3 * n_scans
    assert got['ss_df'].values.tolist() == ref['ss_df'].values.tolist() and len(ref['ss_df']) >= 10
    for sta in stas:
        for nm, h in hist[sta].items():
            assert np.array_equal(ss.histSubSpaces[sta][nm], h)
    rows = []
    for db in (db1, db2):
        con = sqlite3.connect(db)
        rows.append(pd.read_sql('select * from ss_df', con))
        tabs = set(r[0] for r in con.execute("select name from sqlite_master where type='table'"))
        assert {'ss_df', 'ss_info', 'ss_hist', 'filt_params'} <= tabs
        # the histogram rows go in as each station finishes (and once: the second call rewrote db2's from scratch)
        hrows = con.execute('select Sta, Name, Value from ss_hist').fetchall()
        assert sorted((s, n) for s, n, _ in hrows) == sorted((s, n) for s in stas for n in hist[s])
        for s, n, v in hrows:
            assert np.array_equal(np.array(v.split(), dtype=np.int64), hist[s][n])
        con.close()
    # a second run into an existing file replaces the histogram tables (to_sql(if_exists='replace'))
    ss.detex(subspaceDB=db2, ctx=ctx, delOldCorrs=False)
    con = sqlite3.connect(db2)
    assert con.execute('select count(*) from ss_hist').fetchone()[0] == len(hrows)
    con.close()
    assert rows[0].values.tolist() == rows[1].values.tolist() == ref['ss_df'].values.tolist()
    assert list(rows[0].columns) == list(ref['ss_df'].columns)
    # the units overlap by the chunks' 120 s buffer: a little more goes over the host link than the record holds
    total = sum(4 * 3 * cf.data[s][0].shape[1] for s in stas)
    assert total < ss.last_run['uploaded_bytes'] < 1.01 * total


def test_raw_template_fragments_with_fillZeros():
    """createCluster(fillZeros=True) on fragmented raw records: the gap is filled with zeros (_mergeChannelsFill,
    construct.py:1069-1071) and the event keeps its whole window, where the default cuts it to its longest
    gap-free stretch."""
    recs, sr, Nc = raw_event_records(nev=6)
    filt, trim = [1, 8, 2, True], [5, 20]
    T = recs[0]['data'].shape[1]
    g0 = int((30 - 5) * sr) + 300

    def fetcher():
        ef = EventFetcher(sr, ['E', 'N', 'Z'])
        for i, r in enumerate(recs):
            cnt = r['data'].astype(np.int32)
            frs = [(c, 0, cnt[c]) for c in range(Nc)]
            if i == 2:          # a 30-sample gap in the middle of the window, on one channel
                frs = [(0, 0, cnt[0, :g0]), (0, g0 + 30, cnt[0, g0 + 30:])] + [(c, 0, cnt[c]) for c in (1, 2)]
            ef.add_counts('XX.RAW', r['name'], frs, T, r['starttime'], r['origintime'], r['magnitude'])
        return ef
    full = Nc * (int(round(25 * sr)) + 1)
    filled = createCluster(CCreq=0.4, fetch_arg=fetcher(), filt=filt, trim=trim, fillZeros=True, ctx=OracleCtx(),
                           saveclust=False)
    assert 'ev02' in filled.trdf.iloc[0].Events and len(filled._mptd['XX.RAW']['ev02']) == full
    cut = createCluster(CCreq=0.4, fetch_arg=fetcher(), filt=filt, trim=trim, ctx=OracleCtx(), saveclust=False)
    # without fillZeros the longest gap-free stretch of ev02 (671 of 1001 samples, behind the gap) is all that is
    # left of it: below 90 % of the median length, _testStreamLengths drops the event (construct.py:679-694)
    assert 'ev02' not in cut.trdf.iloc[0].Events and len(cut._mptd['XX.RAW']['ev00']) == full
    # zero fill: the filtered waveform equals the one of the record with the gap samples set to zero
    rec = recs[2]['data'].copy()
    rec[0, g0:g0 + 30] = 0
    ef = EventFetcher(sr, ['E', 'N', 'Z'])
    for i, r in enumerate(recs):
        ef.add_raw('XX.RAW', r['name'], rec if i == 2 else r['data'], r['starttime'], r['origintime'], r['magnitude'])
    ref = createCluster(CCreq=0.4, fetch_arg=ef, filt=filt, trim=trim, ctx=OracleCtx(), saveclust=False)
    assert np.allclose(filled._mptd['XX.RAW']['ev02'], ref._mptd['XX.RAW']['ev02'], atol=1e-6)


def _updateReqCC_by_sets(link, key, ccReq):
    """The reference's dendrogram cut (subspace.py:305-346, 385-412) restated with its member lists and subset
    tests -- the O(N^2) form detex_amd.subspace.Cluster.updateReqCC replaces; kept here as the yardstick."""
    N = len(link)
    members = {i: [i] for i in range(N + 1)}
    rows = []
    for a in range(N):
        i1, i2 = int(link[a, 0]), int(link[a, 1])
        members[N + 1 + a] = members[i1] + members[i2]
        rows.append(dict(disSim=link[a, 2], clust=N + 1 + a, II=members[i1] + members[i2]))
    rest = [r for r in rows if r['disSim'] <= 1 - ccReq]
    rest.sort(key=lambda r: -r['disSim'])
    clustlinks, clusts = {}, []
    while rest:
        top = rest[0]
        inside = [r for r in rest if set(r['II']).issubset(top['II'])]
        clustlinks[len(clusts)] = [r['clust'] for r in inside]
        clusts.append([key[y] for y in sorted(set(y for r in inside for y in r['II']))])
        rest = [r for r in rest if r not in inside]
    cs = set(y for x in clusts for y in x)
    return clusts, [k for k in key if k not in cs], clustlinks


@pytest.mark.parametrize('seed', range(12))
def test_dendrogram_cut_equals_the_subset_form(seed):
    """Cluster.updateReqCC (one depth-first walk per cluster) against the reference's subset form on random
    single linkages, ties between merge heights included, at several ccReq."""
    from scipy.cluster.hierarchy import linkage
    from detex_amd.subspace import Cluster
    rs = np.random.RandomState(seed)
    N = int(rs.randint(3, 60))
    npair = N * (N - 1) // 2
    cc = rs.uniform(0, 1, npair)
    if seed % 3 == 0:
        cc = np.round(cc, 1)                           # many equal dissimilarities: ties between merges
    link = linkage(1.0000001 - cc)
    key = ['ev%02d' % i for i in range(N)]
    for ccReq in (0.0, 0.2, 0.5, 0.8, 0.95, 1.0):
        c = Cluster(None, 'XX.STA', None, key, link, ccReq, None, None, None, None)
        clusts, singles, clustlinks = _updateReqCC_by_sets(link, key, ccReq)
        assert c.clusts == clusts and c.singles == singles and c.clustlinks == clustlinks, (seed, ccReq)
        assert c.clustcount == sum(len(x) for x in clusts)
    assert sorted(c.members(2 * len(link))) == list(range(N))          # the root holds every leaf
    # dflink rows answer 'II' like the reference's frame (subspace.py:392-411: members of i1, then of i2), lazily;
    # a pickled cluster keeps them
    import pickle
    mem = {i: [i] for i in range(N)}
    for a in range(len(link)):
        mem[N + a] = mem[int(link[a, 0])] + mem[int(link[a, 1])]
    c2 = pickle.loads(pickle.dumps(c))
    for a in (0, len(link) // 2, len(link) - 1):
        assert 'II' not in dict.keys(c.dflink[a])
        assert c.dflink[a]['II'] == mem[N + a] == c2.dflink[a]['II']
    with pytest.raises(KeyError):
        c.dflink[0]['nope']


def test_db_sync_mode_follows_delOldCorrs(tmp_path, monkeypatch):
    """_DbWriter: synchronous=OFF only for a file this call creates from scratch (delOldCorrs removed the old one);
    a file that holds earlier runs' rows (append mode, subspace.py:1847-1854: delOldCorrs=False) is written with
    SQLite's default FULL, as the reference's to_sql does.  DETEX_AMD_DB_SYNC overrides and is validated."""
    import sqlite3
    import pandas as pd
    from detex_amd import subspace as ssmod
    monkeypatch.delenv('DETEX_AMD_DB_SYNC', raising=False)
    made = []
    orig = ssmod._DbWriter.__init__

    def spy(self, path, fresh=False):
        orig(self, path, fresh=fresh)
        made.append(self.sync)
    monkeypatch.setattr(ssmod._DbWriter, '__init__', spy)
    ctx = OracleCtx()
    ss, cf, plants, _ = small_workflow(ctx)
    db = str(tmp_path / 'SubSpace.db')
    r1 = ss.detex(subspaceDB=db, ctx=ctx, estimateMags=False)                  # no file yet, delOldCorrs=True
    r2 = ss.detex(subspaceDB=db, ctx=ctx, estimateMags=False)                  # file removed first: fresh again
    n1 = len(pd.read_sql('SELECT * FROM ss_df', sqlite3.connect(db)))
    r3 = ss.detex(subspaceDB=db, ctx=ctx, estimateMags=False, delOldCorrs=False)   # appends to r2's rows
    n3 = len(pd.read_sql('SELECT * FROM ss_df', sqlite3.connect(db)))
    assert made == ['OFF', 'OFF', 'FULL'], made
    assert n1 == len(r1['ss_df']) == len(r2['ss_df']) and n3 == 2 * n1 and n1 > 0
    # a file that exists and is NOT removed is never written without durability, whatever `fresh` says
    w = ssmod._DbWriter(db, fresh=True)
    w.close()
    assert w.sync == 'FULL'
    # delOldCorrs=False with no file yet: also FULL (nothing said the file is this call's alone)
    w = ssmod._DbWriter(str(tmp_path / 'new.db'), fresh=False)
    w.close()
    assert w.sync == 'FULL'
    monkeypatch.setenv('DETEX_AMD_DB_SYNC', 'normal')
    w = ssmod._DbWriter(db)
    w.close()
    assert w.sync == 'NORMAL'
    monkeypatch.setenv('DETEX_AMD_DB_SYNC', "OFF; DROP TABLE ss_df")
    with pytest.raises(ValueError):
        ssmod._DbWriter(db)


def test_loadClusters_loadSubSpace_round_trip(tmp_path, monkeypatch):
    """util.py:934-969: write() -> loadClusters / loadSubSpace -> the loaded objects give the same subspaces and the
    same detection tables; a file of the wrong kind is an error."""
    import pandas as pd
    import detex_amd
    monkeypatch.chdir(tmp_path)
    ctx = OracleCtx()
    ss, cf, plants, _ = small_workflow(ctx)
    ref = ss.detex(subspaceDB=None, ctx=ctx, useSingles=True)
    cl = ss.clusters
    cl.filename = 'clust.pkl'
    cl.write()
    ss.write()                                               # 'subspace.pkl'
    cl2 = detex_amd.loadClusters()
    ss2 = detex_amd.loadSubSpace()
    assert isinstance(cl2, detex_amd.ClusterStream) and isinstance(ss2, detex_amd.SubSpace)
    assert [c.station for c in cl2] == [c.station for c in cl]
    got = ss2.detex(subspaceDB=None, ctx=OracleCtx(), useSingles=True)
    for k in ('ss_df', 'sg_df'):
        pd.testing.assert_frame_equal(got[k], ref[k])
    assert len(ref['ss_df']) >= len(plants) - 1
    with pytest.raises(detex_amd.DetexHipError):
        detex_amd.loadClusters('subspace.pkl')
    with pytest.raises(detex_amd.DetexHipError):
        detex_amd.loadSubSpace('clust.pkl')


def test_lazy_mptd_pickles_without_its_cache_and_link_rows_answer_II():
    """a pickled ClusterStream holds the events once (the multiplexed arrays built so far are not written
    beside the channel planes), and a dflink row answers 'II' through get / in as well as [...]."""
    import pickle
    from detex_amd.construct import _LazyMPtd
    rs = np.random.RandomState(0)
    planes = {'e%d' % i: rs.standard_normal((3, 500)) for i in range(6)}
    m = _LazyMPtd(planes, 1500, np.float64)
    bare = len(pickle.dumps(m))
    for k in planes:
        assert m[k].shape == (1500,)                       # builds (and caches) the multiplexed arrays
    assert len(pickle.dumps(m)) == bare
    m2 = pickle.loads(pickle.dumps(m))
    assert all(np.array_equal(m2[k], m[k]) for k in planes)
    ctx = OracleCtx()
    ss, _, _, _ = small_workflow(ctx)
    c = ss.clusters[0]
    row = c.dflink.iloc[0] if hasattr(c.dflink, 'iloc') else c.dflink[0]
    row = row if isinstance(row, dict) else row.to_dict()
    if hasattr(row, '_cluster'):
        assert 'II' in row and row.get('II') == row['II'] and len(row['II']) >= 2
