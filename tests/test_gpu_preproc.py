"""Device pre-processing (detrend + zero-phase Butterworth band-pass + multiplex,
detex/construct.py:990-1030, 928-987) against the SciPy arithmetic ObsPy
delegates to.  Output is float32: tolerance 2e-6 of the series' peak amplitude."""
import numpy as np
import pytest

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu

from oracle import detex_oracle as orc  # noqa: E402


@pytest.fixture(scope='module')
def ctx():
    from detex_amd.hip import HipContext
    c = HipContext(0)
    yield c
    c.close()


@pytest.mark.parametrize('T,Nc,nchunk,filt,sr,detrend', [
    (5000, 3, 2, [1, 10, 2, True], 100.0, True),      # the reference defaults (construct.py:27)
    (5000, 3, 2, [1, 10, 2, False], 100.0, True),
    (1023, 1, 1, [2, 8, 4, True], 40.0, True),        # shorter than one block
    (1025, 6, 3, [5, 40, 3, True], 500.0, False),
    (4097, 2, 1, None, 100.0, True),                  # detrend + multiplex only
    (372000, 3, 2, [1, 10, 2, True], 100.0, True),    # full chunk length
])
def test_preprocess_matches_scipy(ctx, T, Nc, nchunk, filt, sr, detrend):
    rs = np.random.RandomState(T + Nc)
    t = np.arange(T)
    raw = (rs.standard_normal((nchunk, Nc, T)) * 50 + 300.0 + 0.01 * t[None, None, :] *
           rs.uniform(-1, 1, (nchunk, Nc, 1)) +
           40 * np.sin(2 * np.pi * 0.2 * t / sr)[None, None, :]).astype(np.float32)
    out = ctx.preprocess(torch.from_numpy(raw).cuda(), filt=filt, sampling_rate=sr, detrend=detrend)
    ctx.sync()
    got = out.cpu().numpy()
    assert got.shape == (nchunk, Nc * T)
    for b in range(nchunk):
        chans = orc.apply_filter(list(raw[b]), filt, sr, detrend=detrend)
        want = orc.multiplex(chans)
        scale = np.max(np.abs(want))
        assert np.max(np.abs(got[b] - want)) <= 2e-6 * scale + 1e-7 * 400, (b, scale)


def test_preprocess_feeds_the_scan(ctx):
    """Raw counts -> device pre-processing -> detection statistic, against the
    oracle pipeline on the same raw data."""
    rs = np.random.RandomState(3)
    Nc, T, sr, L = 3, 6000, 100.0, 150
    filt = [1, 10, 2, True]
    raw = (rs.standard_normal((1, Nc, T)) * 100 + 1000).astype(np.float32)
    tmpl = orc.multiplex(orc.apply_filter(list(rs.standard_normal((Nc, L + 400))), filt, sr))[
        Nc * 200: Nc * (200 + L)]
    U = (tmpl / np.linalg.norm(tmpl))[None, :].astype(np.float32)
    mp = ctx.preprocess(torch.from_numpy(raw).cuda(), filt=filt, sampling_rate=sr)
    ctx.upload_basis(U, [0, 1], Nc)
    ds = ctx.subspace_ds(mp)
    ctx.sync()
    want = orc.mpxds_fft(orc.multiplex(orc.apply_filter(list(raw[0]), filt, sr)), U, Nc)
    assert np.max(np.abs(ds[0, 0].cpu().numpy() - want)) <= 1e-5


@pytest.mark.parametrize('dec', [None, 2])
def test_detex_on_raw_continuous_data(ctx, dec):
    """SubSpace.detex / getFAS with a raw ContinuousFetcher: per-chunk device
    decimation (dec) + filtering feeds the scan; triggers equal those of the oracle's
    decimate -> filter -> DS chain run chunk by chunk, and everything downstream runs
    at the decimated rate (detect.py:239)."""
    from detex_amd import ContinuousFetcher, EventFetcher, createCluster, createSubSpace
    from detex_amd.getdata import multiplex
    rs = np.random.RandomState(12)
    q = dec or 1
    sr, Nc, L = 20.0, 3, 160                 # rate / template length AFTER decimation
    sr_raw, Lr = sr * q, L * q
    filt = [1, 8, 2, True]
    wav = [np.convolve(rs.standard_normal(Lr + 6 * q), np.hanning(6 * q + 1), 'valid') *
           np.exp(-np.arange(Lr) / (50.0 * q)) for _ in range(Nc)]

    def prep(chans):
        chans = [np.asarray(c, dtype=np.float32) for c in chans]
        if dec:
            chans = orc.decimate(chans, dec, sr_raw)
        return orc.apply_filter(chans, filt, sr)
    wavf = [w[:L] for w in prep(wav)]
    ef = EventFetcher(sr, ['E', 'N', 'Z'])
    for i in range(4):
        ef.add('XX.STA', 'ev%d' % i, np.array([rs.uniform(.5, 2) * w + 0.02 * rs.standard_normal(L)
                                                for w in wavf]), starttime=100. * i)
    T = int(1500 * sr_raw)
    con = rs.standard_normal((Nc, T)) * 0.3 + 500.0 + 0.001 * np.arange(T)
    for t in (4000 * q, 13000 * q, 22000 * q):
        for c in range(Nc):
            con[c, t:t + Lr] += 3.0 * wav[c]
    cf = ContinuousFetcher(sr_raw, ['E', 'N', 'Z'], conDatDuration=600, conBuff=60, raw=True)
    cf.add('XX.STA', con.astype(np.float32), starttime=5.0e5)
    cl = createCluster(CCreq=0.5, fetch_arg=ef, filt=filt, decimate=dec, ctx=ctx)
    ss = createSubSpace(clust=cl, conDatFetcher=cf)
    ss.attachPickTimes(0.0, 6.0)
    ss.SVD(selectCriteria=4, selectValue=0, threshold=0.35, useSingles=False)
    res = ss.detex(subspaceDB=None, ctx=ctx, estimateMags=True)
    names, kinds, U, off, offsets, thr = ss._detector_bank('XX.STA', useSingles=False)
    want, alld = [], []
    for chunk, t0 in cf.getConData('XX.STA'):
        x = multiplex(prep(list(chunk)))
        ds = orc.mpxds_fft(x, U.astype(np.float32), Nc)
        alld.append(ds)
        if orc.eval_trig_con(ds.max(), np.float32(thr['SS0'])):
            want += [(t['STMP'], t['DS']) for t in orc.pick_triggers(ds, np.float32(thr['SS0']), sr, t0)]
    got = res['ss_df'].sort_values('STMP')
    assert len(got) == len(want) >= 3
    for (_, g), w in zip(got.iterrows(), sorted(want)):
        assert g.STMP == w[0] and abs(g.DS - w[1]) <= 1e-5 and np.isfinite(g.SNR)
    # false-alarm statistics over every chunk of the same raw fetcher (fas.py:88-117)
    ss.getFAS(conDatNum=len(alld), staltalimit=None, seed=0)
    fas = ss.subspaces['XX.STA'].FAS.iloc[0]
    ref = np.histogram(np.concatenate(alld), bins=np.linspace(-.01, 1, 401))[0]
    assert fas['hist'].sum() == ref.sum()
    assert np.max(np.abs(np.cumsum(fas['hist']) - np.cumsum(ref))) <= 3    # bin-edge ties only


@pytest.mark.parametrize('T,factor,sr,shape', [
    (5000, 2, 100.0, (2, 3)),
    (5001, 4, 100.0, (3,)),
    (1023, 5, 500.0, (1, 6)),      # shorter than one block per series after the split
    (372000, 4, 100.0, (3,)),      # full chunk length
    (777, 1, 40.0, (2,)),          # factor 1: causal low-pass only
    (4096, 16, 200.0, (1,)),
])
def test_decimate_matches_scipy(ctx, T, factor, sr, shape):
    """dtx_decimate (construct.py:1016-1017) against the SciPy arithmetic ObsPy's
    Trace.decimate delegates to; float32 output, 2e-6 of the peak amplitude."""
    rs = np.random.RandomState(T + factor)
    raw = (rs.standard_normal(shape + (T,)) * 50 + 300.0).astype(np.float32)
    got = ctx.decimate(torch.from_numpy(raw).cuda(), factor, sr)
    ctx.sync()
    got = got.cpu().numpy()
    assert got.shape == shape + ((T + factor - 1) // factor,)
    flat, gflat = raw.reshape(-1, T), got.reshape(-1, got.shape[-1])
    want = orc.decimate(list(flat), factor, sr)
    for s in range(len(flat)):
        assert np.max(np.abs(gflat[s] - want[s])) <= 2e-6 * np.max(np.abs(want[s])), s


def test_decimate_rejects_large_factor(ctx):
    with pytest.raises(ValueError):
        ctx.decimate(torch.zeros((1, 100), device='cuda'), 17, 100.0)
