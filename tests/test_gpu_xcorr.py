"""GPU parity of the all-pairs normalised cross-correlation (HIP, through the C
ABI) against the reference-generated golden vectors and the CPU oracle.
Tolerances: cc 1e-5, lag exact, subsamp 1e-4."""
import os

import numpy as np
import pytest

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu

# DETEX_TEST_SEED_BASE=k re-draws every seeded sweep of this file (soak runs: scripts/soak.sh); 0 = the committed cases
SOAK = 100000 * int(__import__('os').environ.get('DETEX_TEST_SEED_BASE', 0))

from oracle import detex_oracle as orc  # noqa: E402

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
SCRUB = 1e-5


@pytest.fixture(scope='module')
def ctx():
    from detex_amd.hip import HipContext
    c = HipContext(0)
    yield c
    c.close()


def _gpu(ctx, td, Nc, **kw):
    t = torch.from_numpy(np.ascontiguousarray(td, dtype=np.float32)).cuda()
    cc, lag, ss = ctx.xcorr_allpairs(t, Nc, **kw)
    ctx.sync()
    return cc.cpu().numpy(), lag.cpu().numpy(), ss.cpu().numpy()


def _knife(raw, trunc):
    """True if the oracle's raw CC vector has a value within the scrub band of
    +-1, where float32 vs float64 rounding legitimately flips the decision."""
    tr = raw[trunc:-trunc]
    tr = tr[np.isfinite(tr)]
    return bool(np.any(np.abs(np.abs(tr) - 1.0) < 2 * SCRUB))


def _check_pair(x1, x2, Nc, got, what):
    x1 = np.asarray(x1, dtype=np.float32).astype(np.float64)
    x2 = np.asarray(x2, dtype=np.float32).astype(np.float64)
    n = len(x1)
    trunc = n // (2 * Nc) - 1
    raw = orc.ccx2_raw_direct(x1, x2, Nc)
    cc, lag, ss, res = orc.ccx2_reduce(raw, n, Nc, trunc, SCRUB)
    if _knife(raw, trunc):
        # e.g. identical events: cc must still be the scrub-tolerant answer
        assert abs(got[0] - cc) <= 2e-5, what
        return 'knife'
    # near-ties between two lags can legitimately resolve differently in f32
    order = np.sort(res[np.isfinite(res)])
    if len(order) > 1 and order[-1] - order[-2] < 5e-6:
        assert abs(got[0] - cc) <= 1e-5, what
        return 'tie'
    assert abs(got[0] - cc) <= 1e-5, (what, got, cc, lag, ss)
    assert got[1] == lag, (what, got, cc, lag, ss)
    if abs(ss) <= .5:
        assert abs(got[2] - ss) <= 1e-4, (what, got, ss)
    else:
        # (_subSamp's arccos of a ratio outside [-1, 1] is NaN in the reference too -- construct.py:397-422 -- and
        # NaN > .5 is false, so NaN is what it returns: the same NaN is parity; found by a soak run, the committed
        # draw of pairs holds none)
        assert got[2] == np.float32(ss) or (np.isnan(got[2]) and np.isnan(ss)), (what, got, ss)
    return 'ok'


# The 15 golden cases on which the build deliberately differs from the reference.  The two
# events are identical (or identical up to a constant offset): the true maximum is cc = 1 at lag 0, but the
# reference's float64 round-off lands a hair ABOVE 1 there, its scrub (construct.py:458-462: every value > 1 or
# < -1 is set to 0 and the maximum taken again) removes the true peak and a side lobe wins.  The build scrubs at
# |cc| > 1 + 1e-5 and keeps the peak.  name: (reference cc, reference lag) -- read from the fixture and asserted
# below -- against the build's (1.0, 0).  Every OTHER case (83, the 20 zero_tail / zeroed_chan ones included) must
# equal the reference's own output.
KNIFE = {
    'c008': (0.7901441, -1), 'c010': (0.5618626, -1), 'c016': (0.8101324, 3), 'c018': (0.5822324, 3),
    'c024': (0.5824542, 3), 'c030': (0.8168820, -3), 'c032': (0.8192597, -3), 'c036': (0.8231192, 3),
    'c038': (0.7938896, 3), 'c052': (0.6192905, 6), 'c058': (0.4627540, -2), 'c060': (0.2692443, -8),
    'c064': (0.4729424, -2), 'c066': (0.3145494, -8),
    'c080': (0.6002927, 5),                 # round 5's (Nc, L) = (5, 40) family
}


def test_golden_ccx2_pairs(ctx):
    """Every golden _CCX2 case as a 2-event problem against the REFERENCE's output (cc 1e-5, lag exact,
    subsamp 1e-4), except the 15 listed knife-edge cases, which are held to the build's documented answer
    with the reference's value beside it."""
    z = np.load(os.path.join(G, 'ccx2.npz'))
    seen = set()
    for nm in z['names']:
        key = nm.split(':')[0]
        x1, x2, Nc = z[key + '_x1'], z[key + '_x2'], int(z[key + '_Nc'])
        cc, lag, ss = _gpu(ctx, np.stack([x1, x2]), Nc)
        got = (float(cc[0]), int(lag[0]), float(ss[0]))
        ref = z[key + '_out']
        if key in KNIFE:
            seen.add(key)
            assert 'identical' in nm or 'offset' in nm
            assert abs(ref[0] - KNIFE[key][0]) <= 1e-6 and int(ref[1]) == KNIFE[key][1], (nm, ref)   # the reference's answer
            assert abs(got[0] - 1.0) <= 1e-5 and got[1] == 0, (nm, got)                              # the build's
            continue
        assert abs(got[0] - ref[0]) <= 1e-5 and got[1] == int(ref[1]), (nm, got, ref)
        if abs(ref[2]) <= .5:
            assert abs(got[2] - ref[2]) <= 1e-4, (nm, got, ref)
        _check_pair(x1, x2, Nc, got, nm)                    # and the oracle agrees
    assert seen == set(KNIFE) and len(z['names']) == 98


def test_golden_cclags_matrix(ctx):
    z = np.load(os.path.join(G, 'cclags.npz'))
    for tag in 'ab':
        ev, Nc = z[tag + '_events'], int(z[tag + '_Nc'])
        N = len(ev)
        cc, lag, ss = _gpu(ctx, ev, Nc)
        iu = np.triu_indices(N, 1)
        rcc = z[tag + '_cc'][iu[0], iu[1] - 1]
        rlag = z[tag + '_lag'][iu[0], iu[1] - 1]
        rss = z[tag + '_subsamp'][iu[0], iu[1] - 1]
        assert np.max(np.abs(cc - rcc)) <= 1e-5
        assert np.array_equal(lag, rlag.astype(np.int32))
        assert np.max(np.abs(ss - rss)) <= 1e-4
        # the dissimilarity vector feeds scipy linkage in condensed order
        assert np.max(np.abs((1.0000001 - cc) - z[tag + '_cx'])) <= 1e-5


@pytest.mark.parametrize('N,Nc,L', [(5, 3, 64), (17, 1, 100), (40, 3, 37), (9, 6, 50),
                                    (230, 2, 24), (12, 3, 333),
                                    # channel counts without a specialised form; 65 > the finishing kernel's window
                                    (9, 5, 40), (12, 7, 33), (20, 8, 64), (6, 9, 30), (7, 12, 50), (5, 65, 12),
                                    (215, 5, 24), (6, 9, 300)])
def test_allpairs_vs_oracle(ctx, N, Nc, L):
    """Row tiles of every size (N > 208 spans two tiles), odd lengths."""
    from detex_amd import synth
    td = synth.event_families(N + L, max(1, N // 4), 4, Nc, L, maxshift=L // 8, snr=4.0)[:N]
    if len(td) < N:
        td = np.concatenate([td, synth.event_families(7, N - len(td), 1, Nc, L, L // 8)])
    cc, lag, ss = _gpu(ctx, td, Nc)
    rs = np.random.RandomState(N + SOAK)
    pairs = [(b, c) for b in range(N) for c in range(b + 1, N)]
    if len(pairs) > 150:
        pairs = [pairs[i] for i in rs.choice(len(pairs), 150, replace=False)]
    nok = 0
    for (b, c) in pairs:
        idx = N * b - b * (b + 1) // 2 + (c - b - 1)
        got = (float(cc[idx]), int(lag[idx]), float(ss[idx]))
        nok += _check_pair(td[b], td[c], Nc, got, (b, c)) == 'ok'
    assert nok >= 0.9 * len(pairs)


@pytest.mark.parametrize('N,Nc,L', [(30, 3, 48), (21, 3, 1500)])
def test_pair_range_sharding(ctx, N, Nc, L):
    """Disjoint [pair_begin, pair_end) ranges reproduce the full result (the
    multi-GPU partition) and leave other entries untouched -- on the MFMA contraction
    (short events) and on the frequency-domain lag search (L >= 1024)."""
    from detex_amd import synth
    td = synth.event_families(3, 6, (N + 5) // 6, Nc, L, maxshift=6)[:N]
    full = _gpu(ctx, td, Nc)
    npairs = N * (N - 1) // 2
    cuts = [0, 1, 57, 200, 201, npairs]
    acc = [np.zeros(npairs, np.float32), np.zeros(npairs, np.int32), np.zeros(npairs, np.float32)]
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        part = _gpu(ctx, td, Nc, pair_begin=lo, pair_end=hi)
        for a, p in zip(acc, part):
            assert not p[:lo].any() and not p[hi:].any()
            a[lo:hi] = p[lo:hi]
    for a, f in zip(acc, full):
        assert np.array_equal(a, f)


def test_dead_and_degenerate_events(ctx):
    rs = np.random.RandomState(2 + SOAK)
    Nc, L = 3, 40
    td = rs.standard_normal((4, Nc * L)).astype(np.float32)
    td[1] = 0.0                       # all-zero event -> (0, 0, 0) (construct.py:463-464)
    td[3] = 2.5                       # constant event: zero variance
    cc, lag, ss = _gpu(ctx, td, Nc)
    N = 4
    for (b, c) in [(0, 1), (1, 2), (1, 3)]:
        idx = N * b - b * (b + 1) // 2 + (c - b - 1)
        assert cc[idx] == 0 and lag[idx] == 0 and ss[idx] == 0
    # too short for the central-half truncation (L < 4): all zeros
    cc, lag, ss = _gpu(ctx, rs.standard_normal((3, 6)).astype(np.float32), 2)
    assert not cc.any() and not lag.any() and not ss.any()


@pytest.mark.parametrize('Nc,L,N', [(3, 1024, 9), (1, 2500, 12), (3, 3000, 23), (2, 5200, 8),
                                    (3, 6500, 7), (1, 9000, 10), (3, 13000, 6),
                                    # round 3: the host picks segment count / length / lag-block advance by cost
                                    # (dtx_xcfft_run): just past the single-segment limit, odd step counts, the
                                    # longest events (16 segments), many channels
                                    (3, 5700, 6), (2, 7777, 5), (1, 20000, 5), (1, 32768, 4), (6, 6000, 5),
                                    # round 5: channel counts without a specialised form (one segment / several)
                                    (5, 1024, 8), (7, 2000, 5), (8, 3000, 4), (9, 1500, 9), (12, 1100, 5), (5, 6000, 4),
                                    (65, 1024, 3)])
def test_long_events_frequency_domain(ctx, Nc, L, N):
    """Events of >= 1024 samples per channel take the frequency-domain lag search under the
    default engine (dtx_xcfft.hip; past 5633 samples cut into segments): same answers as the
    contraction engine and the float64 oracle -- cc 1e-5, lag exact, subsamp 1e-4; includes an
    odd event count (a pair with a zero row), an identical pair and a dead event."""
    from detex_amd import synth
    rs = np.random.RandomState(L + N + SOAK)
    nfam = 2
    td = synth.event_families(L, nfam, (N + nfam - 1) // nfam, Nc, L, maxshift=max(1, L // 40))[:N]
    td = td * np.float32(10 ** rs.uniform(-3, 3))
    # catalogue events differ by orders of magnitude: neighbours (which share a transform) by 10^6
    td *= (10.0 ** rs.choice([-3, 0, 3], size=N)).astype(np.float32)[:, None]
    if N >= 8:
        td[5] = td[2]                      # identical pair: cc = 1, lag 0
        td[7] = 0.0                        # dead event: cc 0 with everything
        if Nc > 1:
            td[6].reshape(L, Nc)[:, 1] = 0.0   # one channel of an event zeroed (a dead component: the golden 'zeroed_chan' cases, long)
    if N >= 9:
        td[8] = 2.5                        # constant event: zero variance -> (0, 0, 0) with everything (construct.py:463-464)
    ctx.set_engine('auto')
    cc, lag, ss = _gpu(ctx, td, Nc)
    ctx.set_engine('bf16x3')
    cc2, lag2, ss2 = _gpu(ctx, td, Nc)
    ctx.set_engine('auto')
    n = Nc * L
    trunc = n // (2 * Nc) - 1
    k = 0
    nchecked = 0
    for b in range(N):
        for c in range(b + 1, N):
            x1 = td[b].astype(np.float64)
            x2 = td[c].astype(np.float64)
            raw = orc.ccx2_raw_fft(x1, x2, Nc)
            rc, rl, rss, res = orc.ccx2_reduce(raw, n, Nc, trunc, SCRUB)
            assert abs(cc[k] - rc) <= 1e-5, (b, c, cc[k], rc)
            assert abs(cc[k] - cc2[k]) <= 1e-5, (b, c)
            fin = np.sort(res[np.isfinite(res)])
            if len(fin) > 1 and fin[-1] - fin[-2] > 5e-6 and abs(abs(rc) - 1) > 2e-5:
                assert lag[k] == rl, (b, c, lag[k], rl)
                assert lag[k] == lag2[k], (b, c)
                if abs(rss) <= .5:
                    assert abs(ss[k] - rss) <= 1e-4, (b, c)
                nchecked += 1
            k += 1
    assert nchecked >= N


@pytest.mark.parametrize('Nc,L,N', [(3, 3000, 23), (1, 2500, 12), (3, 13000, 9), (2, 7777, 5), (6, 6000, 6), (3, 1024, 2),
                                    (5, 2000, 9), (9, 1024, 8)])
def test_both_lag_search_kernels(ctx, monkeypatch, Nc, L, N):
    """The frequency-domain lag search has two kernels: the 2 x 2 workgroup (two later events x two pair spectra,
    DETEX_AMD_XC=22) and the half-size workgroup (one pair spectrum x two later events, two workgroups per CU:
    DETEX_AMD_XC=12); the library picks between them by the lag blocks' step count.  (Round 3's kernel, a third
    bit-identical variant until round 5, was removed in round 6.)  The same products summed in the same order and the same
    transform: identical answers, on the whole matrix and on disjoint pair ranges (odd and even event counts: a
    last later event without a partner, a last pair spectrum with one row)."""
    from detex_amd import synth
    td = synth.event_families(L + 1, 2, (N + 1) // 2, Nc, L, maxshift=max(1, L // 40))[:N]
    if N >= 8:
        td[5] = td[2]
        td[7] = 0.0
        # windows of next to no variance (an offset a million times the signal): float32 transforms leave values
        # far beyond +-1 there, which the search sets to 0 (construct.py:458-462) -- the 2 x 2 kernel on a path
        # of its own, taken only by the waves that meet one
        td[3] = 1000.0 + 1e-3 * td[3]
        td[6, : Nc * (L // 3)] = 500.0
    npairs = N * (N - 1) // 2
    res = {}
    for v in ('22', '12'):
        monkeypatch.setenv('DETEX_AMD_XC', v)
        full = _gpu(ctx, td, Nc)
        cuts = sorted({0, 1, npairs // 3, npairs // 3 + 1, (2 * npairs) // 3, npairs})
        acc = [np.zeros(npairs, np.float32), np.zeros(npairs, np.int32), np.zeros(npairs, np.float32)]
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            part = _gpu(ctx, td, Nc, pair_begin=lo, pair_end=hi)
            for a, p_ in zip(acc, part):
                assert not p_[:lo].any() and not p_[hi:].any()
                a[lo:hi] = p_[lo:hi]
        for a, f in zip(acc, full):
            assert np.array_equal(a, f, equal_nan=True)       # (a NaN subsample shift is an answer: _subSamp's arccos, construct.py:397-422)
        res[v] = full
    for a, b in zip(res['22'], res['12']):
        assert np.array_equal(a, b, equal_nan=True)
    monkeypatch.delenv('DETEX_AMD_XC')


@pytest.mark.parametrize('seed', range(16))
def test_lag_search_random_shapes(ctx, monkeypatch, seed):
    """Seeded random event counts, channel counts and lengths across the frequency-domain lag search's layouts (one
    segment / several, four to fifteen lags per thread, odd and even counts): both kernels and a random split of the
    pair range give one answer, and a sample of pairs agrees with the float64 oracle (cc 1e-5, lag exact away from
    ties)."""
    from detex_amd import synth
    rs = np.random.RandomState(500 + seed + SOAK)
    N = int(rs.randint(2, 12))
    Nc = int(rs.choice([1, 2, 3, 4, 6]))
    if seed >= 10:
        Nc = [5, 7, 8, 9, 12, 5][seed - 10]
    L = int(rs.choice([1024, 1300, 2049, 3100, 4000, 5633, 5634, 7000, 9001, 14000]))
    td = synth.event_families(seed + 3, 2, (N + 1) // 2, Nc, L, maxshift=max(1, L // 50))[:N]
    td = td * (10.0 ** rs.choice([-2, 0, 2], size=N)).astype(np.float32)[:, None]
    npairs = N * (N - 1) // 2
    cut = int(rs.randint(0, npairs + 1))
    res = {}
    for v in ('22', '12'):
        monkeypatch.setenv('DETEX_AMD_XC', v)
        full = _gpu(ctx, td, Nc)
        acc = [np.zeros(npairs, np.float32), np.zeros(npairs, np.int32), np.zeros(npairs, np.float32)]
        for lo, hi in ((0, cut), (cut, npairs)):
            if lo == hi:
                continue
            part = _gpu(ctx, td, Nc, pair_begin=lo, pair_end=hi)
            for a, p_ in zip(acc, part):
                a[lo:hi] = p_[lo:hi]
        for a, f in zip(acc, full):
            assert np.array_equal(a, f, equal_nan=True)       # (a NaN subsample shift is an answer: _subSamp's arccos, construct.py:397-422)
        res[v] = full
    for a, b in zip(res['22'], res['12']):
        assert np.array_equal(a, b, equal_nan=True)
    monkeypatch.delenv('DETEX_AMD_XC')
    cc, lag, ss = res['22']
    n = Nc * L
    trunc = n // (2 * Nc) - 1
    pairs = [(b, c) for b in range(N) for c in range(b + 1, N)]
    for (b, c) in [pairs[i] for i in rs.choice(len(pairs), min(4, len(pairs)), replace=False)]:
        k = N * b - b * (b + 1) // 2 + (c - b - 1)
        raw = orc.ccx2_raw_fft(td[b].astype(np.float64), td[c].astype(np.float64), Nc)
        rc, rl, rss, rr = orc.ccx2_reduce(raw, n, Nc, trunc, SCRUB)
        assert abs(cc[k] - rc) <= 1e-5, (seed, b, c, cc[k], rc)
        fin = np.sort(rr[np.isfinite(rr)])
        if len(fin) > 1 and fin[-1] - fin[-2] > 5e-6 and abs(abs(rc) - 1) > 2e-5:
            assert lag[k] == rl, (seed, b, c, lag[k], rl)


def test_fast_normcorr_golden(ctx):
    """fast_normcorr (construct.py:469-483) on the device against the reference's own outputs
    (tests/golden/normcorr.npz), through the context method and the NumPy-facing function."""
    from detex_amd import fast_normcorr
    z = np.load(os.path.join(G, 'normcorr.npz'))
    for i in range(3):
        t, s, ref = z['n%d_t' % i], z['n%d_s' % i], z['n%d_out' % i]
        got = ctx.normcorr(torch.from_numpy(t.astype(np.float32)).cuda(),
                           torch.from_numpy(s.astype(np.float32)).cuda()).cpu().numpy()
        assert got.shape == ref.shape
        assert np.max(np.abs(got - ref)) <= 2e-6, i
        got2 = fast_normcorr(s, t, ctx=ctx)              # arguments in either order (473-474)
        assert np.max(np.abs(got2 - ref)) <= 2e-6, i
    # a long series: every window against the float64 oracle
    rs = np.random.RandomState(9 + SOAK)
    t, s = rs.standard_normal(3000), rs.standard_normal(40000) + 2.0
    s[7000:10000] += 3 * t
    got = fast_normcorr(t, s, ctx=ctx)
    ref = orc.fast_normcorr(t.astype(np.float32).astype(np.float64), s.astype(np.float32).astype(np.float64))
    assert np.max(np.abs(got - ref)) <= 2e-6 and int(np.argmax(got)) == 7000
