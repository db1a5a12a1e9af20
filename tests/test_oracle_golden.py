"""Pin the CPU oracle (oracle/detex_oracle.py) against golden vectors produced
by running the reference itself (tests/golden/make_golden.py).  CPU only."""
import os

import numpy as np
import pytest

from oracle import detex_oracle as orc

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _load(name):
    return np.load(os.path.join(G, name), allow_pickle=False)


def _same(a, b, tol=1e-9):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape
    nan = np.isnan(a)
    assert np.array_equal(nan, np.isnan(b))
    inf = np.isinf(a)
    assert np.array_equal(a[inf], b[inf])
    ok = ~(nan | inf)
    if ok.any():
        assert np.max(np.abs(a[ok] - b[ok])) <= tol


@pytest.mark.parametrize('form', ['fft', 'direct'])
def test_ccx2_matches_reference(form):
    """Every _CCX2 golden case.  A raw CC value within 1e-9 of +-1 is a
    'knife-edge': the reference zeroes it iff float64 rounding put it above 1
    (construct.py:458-462), so such cases are compared away from those lags and
    their scalar outputs must equal the reference under ONE of the two scrub
    decisions."""
    z = _load('ccx2.npz')
    names = list(z['names'])
    assert len(names) == 98     # 70 + the (Nc, L) = (5, 40), (9, 30) families of round 5
    raw_fn = orc.ccx2_raw_fft if form == 'fft' else orc.ccx2_raw_direct
    nvec = nknife = 0
    for nm in names:
        key = nm.split(':')[0]
        x1, x2, Nc = z[key + '_x1'], z[key + '_x2'], int(z[key + '_Nc'])
        n = len(x1)
        trunc = n // (2 * Nc) - 1
        ref = z[key + '_out']
        refres = z[key + '_result']
        raw = raw_fn(x1, x2, Nc)
        tr = raw[trunc:-trunc]
        knife = np.abs(np.abs(tr) - 1.0) < 1e-9
        cands = [orc.ccx2_reduce(raw, n, Nc, trunc, 0.0)]
        if knife.any():
            nknife += 1
            cands.append(orc.ccx2_reduce(raw, n, Nc, trunc, 1e-9))
            forced = raw.copy()
            forced[trunc:-trunc][knife] = np.sign(tr[knife]) * (1 + 1e-7)
            cands.append(orc.ccx2_reduce(forced, n, Nc, trunc, 0.0))
        ok = False
        for cc, lag, ss, res in cands:
            good = abs(cc - ref[0]) <= 1e-9 and lag == ref[1]
            if abs(ref[2]) <= .5:
                good = good and abs(ss - ref[2]) <= 1e-6
            else:
                good = good and ss == ref[2]
            ok = ok or good
        assert ok, nm
        if len(refres):
            res = cands[0][3]
            # zero-variance windows: the reference's value there is FFT
            # round-off / 0 (inf -> scrubbed to 0, or NaN when the round-off
            # happens to be exactly 0); not comparable, so leave them out
            pad = np.pad(np.asarray(x2, dtype=np.float64), (n - 1, n - 1))
            wstd = np.lib.stride_tricks.sliding_window_view(pad, n).std(axis=1)
            dead = (wstd[Nc - 1::Nc] == 0)[trunc:-trunc]
            assert np.array_equal(np.isnan(refres)[~dead], np.isnan(res)[~dead]), nm
            fin = np.isfinite(refres) & np.isfinite(res) & ~knife & ~dead
            assert np.max(np.abs(refres[fin] - res[fin]), initial=0) <= 1e-9, nm
            nvec += 1
    assert nvec >= 60 and nknife >= 4


@pytest.mark.parametrize('form', ['fft', 'direct'])
def test_cclags_matches_reference(form):
    z = _load('cclags.npz')
    for tag in 'ab':
        ev, Nc = z[tag + '_events'], int(z[tag + '_Nc'])
        cc, lag, ss = orc.make_cclags(list(ev), Nc, form=form)
        _same(cc, z[tag + '_cc'])
        assert np.array_equal(np.nan_to_num(lag, nan=-1e9),
                              np.nan_to_num(z[tag + '_lag'], nan=-1e9))
        _same(ss, z[tag + '_subsamp'], 1e-6)
        _same(orc.condensed_dissimilarity(cc), z[tag + '_cx'])


def test_fast_normcorr_matches_reference():
    z = _load('normcorr.npz')
    for i in range(3):
        _same(orc.fast_normcorr(z['n%d_t' % i], z['n%d_s' % i]), z['n%d_out' % i])


@pytest.mark.parametrize('form', ['fft', 'direct'])
def test_mpxds_matches_reference(form):
    z = _load('mpxds.npz')
    fn = orc.mpxds_fft if form == 'fft' else orc.mpxds_direct
    for nm in z['names']:
        key = nm.split(':')[0]
        x, U, Nc = z[key + '_x'], z[key + '_U'], int(z[key + '_Nc'])
        ref = z[key + '_ds']
        ds = fn(x, U, Nc)
        assert ds.shape == ref.shape, nm
        # reference ran in float32 arithmetic for the 'f32' case
        tol = 2e-5 if x.dtype == np.float32 else 1e-9
        # windows of zero variance: reference divides by a pandas rolling_var
        # residue (~1e-30), giving meaningless huge values; compare elsewhere
        n = U.shape[1]
        win = np.lib.stride_tricks.sliding_window_view(
            np.asarray(x, dtype=np.float64), n)[::Nc]
        live = win.std(axis=1) > 1e-9
        assert live.sum() > 0.5 * len(ref)
        assert np.max(np.abs(ds[live] - ref[live])) <= tol, nm
        # planted event is the global maximum over live windows
        pos = int(z[key + '_pos'])
        if 'zero_gap' not in nm:     # gap-edge windows legitimately score high
            assert abs(int(np.argmax(np.where(live, ds, 0))) - pos) <= 1, nm


def test_trigger_picking_matches_reference():
    z = _load('trig.npz')
    sr, start, offsets = float(z['sr']), float(z['start']), z['offsets']
    for nm in z['names']:
        key = nm.split(':')[0]
        ds, thr = z[key + '_ds'], float(z[key + '_thr'])
        sl = orc.get_stalta(ds, 5 * sr, 0)
        _same(sl, z[key + '_stalta_sta0'])
        _same(orc.get_stalta(ds, 5 * sr, 0.5 * sr), z[key + '_stalta_sta05'])
        assert np.array_equal(orc.ds_histogram(ds), z[key + '_hist'])
        trig = orc.eval_trig_con(ds.max(), thr)
        assert trig == bool(z[key + '_trig']), nm
        rows = z[key + '_rows']
        if not trig:
            assert rows.shape[0] == 0
            continue
        got = orc.pick_triggers(ds, thr, sr, start, offsets, stalta=sl)
        assert len(got) == rows.shape[0], nm
        for g, r in zip(got, rows):
            assert g['DS'] == r[0]
            assert abs(g['DS_STALTA'] - r[1]) <= 1e-9 * max(1, abs(r[1]))
            assert g['STMP'] == r[2]
            assert g['MSTAMPmin'] == r[3] and g['MSTAMPmax'] == r[4]
            assert g['index'] == int(round((r[2] - start) * sr))


def test_svd_matches_reference_golden():
    """svd.npz: SubSpace.SVD of the reference itself (subspace.py:786-1054, run through the
    ListDict shim of tests/golden/_ref_import.py): singular values / vectors, fractional
    energy capture, selected basis, criteria-3 thresholds."""
    z = np.load(os.path.join(G, 'svd.npz'))
    for nm in z['names']:
        k = nm.split(':')[0]
        ev, t, ht = z[k + '_events'], z[k + '_trims'], bool(z[k + '_has_trims'])
        arr = orc.trim_groups(list(ev), max(t[0], 0) if ht else None, t[1] if ht else None)
        U, s = orc.svd_basis(arr, normalize=bool(z[k + '_normalize']))
        assert np.allclose(s, z[k + '_s'], rtol=1e-12), nm
        sg = np.sign(np.sum(U * z[k + '_U'], axis=0))            # singular vectors: sign is free
        assert np.allclose(U * sg, z[k + '_U'], atol=1e-9), nm
        wf = [e[t[0]:t[1]] if ht else e for e in ev]
        fr, av, mn = orc.frac_energy(U, wf)
        assert np.allclose(fr, z[k + '_frac'], atol=1e-12), nm
        assert np.allclose(mn, z[k + '_minimum'], atol=1e-12), nm
        crit, val = int(z[k + '_crit']), float(z[k + '_val'])
        nb = orc.used_basis_count(av, crit, val)
        assert nb == int(z[k + '_numbasis']), nm
        assert np.allclose(s[:nb], z[k + '_used'], rtol=1e-12), nm
        if float(z[k + '_thr_in']) == 0:                           # criteria 3 threshold
            assert np.isclose(orc.threshold_criteria3(mn, nb, val), float(z[k + '_threshold']),
                              rtol=1e-12), nm


def test_delays_match_reference_golden():
    """delays.npz: _getDelays / _alignTD of the reference itself (construct.py:710-760,
    486-504) on five clusters."""
    z = np.load(os.path.join(G, 'delays.npz'))
    for nm in z['names']:
        k = nm.split(':')[0]
        d = orc.get_delays(z[k + '_link'], z[k + '_cx'], z[k + '_lags_in'])
        assert np.array_equal(d, z[k + '_delays']), nm
        sd = d - d.min()
        assert np.array_equal(sd, z[k + '_sample_delays']), nm
        al = orc.align_td(sd, list(z[k + '_events']))
        assert np.array_equal(np.array(al), z[k + '_aligned']), nm


def test_svd_basis_properties():
    """Defining properties of the basis selection (subspace.py:886-1013), beside the
    reference-run vectors of test_svd_matches_reference_golden."""
    rs = np.random.RandomState(3)
    base = rs.randn(3, 240)
    ali = [rs.uniform(.5, 2) * base[i % 2] + 0.2 * base[2] * rs.randn() +
           0.05 * rs.randn(240) + 0.3 for i in range(5)]
    arr = orc.trim_groups(ali, 20, 200)
    assert arr.shape == (5, 180) and np.allclose(arr.mean(axis=1), 0)
    U, s = orc.svd_basis(arr)
    assert np.allclose(U.T @ U, np.eye(5), atol=1e-12)
    assert np.all(np.diff(s) <= 0)
    fr, av, mn = orc.frac_energy(U, [a[20:200] for a in ali])
    assert np.all(np.diff(av) >= -1e-15) and av[0] == 0
    nd = orc.used_basis_count(av, 2, 0.9)
    assert 1 <= nd <= 3
    assert av[nd] >= 0.9 > av[nd - 1]
    thr = orc.threshold_criteria3(mn, nd, 0.8)
    # a training event scores at least its own (demeaned) energy capture
    for a in ali:
        w = a[20:200]
        x = np.concatenate([rs.randn(300) * 0.01, w, rs.randn(300) * 0.01])
        ds = orc.mpxds_direct(x, U[:, :nd].T, 1)
        assert ds.max() > thr * 0.9


def fullsize_inputs(z):
    """The inputs of fullsize.npz, regenerated from its seeds with detex_amd/synth.py (the fixture stores
    the reference's outputs only: make_golden.gen_fullsize)."""
    from detex_amd import synth
    seed, nfam, nper, Nc, L = (int(v) for v in z['xc_params'])
    td = synth.event_families(seed, nfam, nper, Nc, L)
    bank_seed, day_seed, K, dNc, dL, M, nplant = (int(v) for v in z['ds_params'])
    U, off = synth.detector_bank(bank_seed, 1, K, dNc * dL)
    x, plants = synth.continuous_day(day_seed, 1, M, dNc, U, off, nplant=nplant)
    assert [p for (_, p, _) in plants] == z['ds_plants'].tolist()
    return td, Nc, U, off, x, dNc


def test_oracle_at_baseline_sizes_matches_reference():
    """The oracle at the BASELINE sizes against the reference run there (fullsize.npz): _CCX2 on four pairs
    of n = 39 000 samples, _MPXDS on 2000 windows of a 1 116 000-sample chunk x one 8-vector detector."""
    z = _load('fullsize.npz')
    td, Nc, U, off, x, dNc = fullsize_inputs(z)
    for (b, c), ref in zip(z['xc_pairs'], z['xc_out']):
        cc, lag, ss, _ = orc.ccx2_fft(td[b].astype(np.float64), td[c].astype(np.float64), Nc)
        assert abs(cc - ref[0]) <= 1e-9 and lag == ref[1] and abs(ss - ref[2]) <= 1e-6, ((b, c), cc, lag, ss, ref)
    ds = orc.mpxds_fft(x[0].astype(np.float64), U.astype(np.float64), dNc)
    assert np.max(np.abs(ds[z['ds_index']] - z['ds_value'])) <= 1e-9
    assert int(np.argmax(ds)) == int(z['ds_max'][1]) and abs(float(ds.max()) - z['ds_max'][0]) <= 1e-9
