"""Device pre-processing (detrend + zero-phase Butterworth band-pass + multiplex,
detex/construct.py:990-1030, 928-987) against the SciPy arithmetic ObsPy
delegates to.  Output is float32: tolerance 2e-6 of the series' peak amplitude."""
import numpy as np
import pytest

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu

# DETEX_TEST_SEED_BASE=k re-draws every seeded sweep of this file (soak runs: scripts/soak.sh); 0 = the committed cases
SOAK = 100000 * int(__import__('os').environ.get('DETEX_TEST_SEED_BASE', 0))

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
    rs = np.random.RandomState(T + Nc + SOAK)
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


@pytest.mark.parametrize('Nc,T,step,nchunk,filt,detrend', [
    (3, 3720, 3600, 5, [1, 10, 2, True], True),       # the reference's chunks: conDatDuration + conBuff, overlapping
    (3, 3720, 3600, 5, [1, 10, 2, False], True),
    (3, 4000, 4000, 3, None, False),                  # no filter, no trend: the 16-byte multiplex kernel
    (3, 3721, 3601, 4, None, True),                   # ... with a trend, strides that are no multiples of four
    (1, 999, 10, 6, [2, 8, 4, True], True),           # one channel, heavy overlap
    (2, 2048, 0, 3, [5, 20, 3, True], False),         # the same chunk three times
])
def test_preprocess_of_chunks_where_they_lie_equals_packed_copy(ctx, Nc, T, step, nchunk, filt, detrend):
    """dtx_preprocess_planes: chunks taken as views of one [Nc, Tday] record (channel stride Tday, chunk stride =
    their spacing, overlapping as the reference's chunks do) against the same chunks copied into a packed
    [nchunk, Nc, T] array first -- bit-identical; a view the entry point cannot express (sample stride 2) falls
    back to the packed copy."""
    rs = np.random.RandomState(Nc * 1000 + T + SOAK)
    Tday = step * (nchunk - 1) + T + 37
    t = np.arange(Tday)
    rec = (rs.standard_normal((Nc, Tday)) * 50 + 300.0 + 0.01 * t[None, :]).astype(np.float32)
    recd = torch.from_numpy(rec).cuda()
    first = 5
    view = recd.as_strided((nchunk, Nc, T), (step, Tday, 1), first)
    packed = torch.stack([recd[:, first + k * step:first + k * step + T] for k in range(nchunk)])
    assert view.data_ptr() == recd.data_ptr() + 4 * first and torch.equal(view, packed)
    a = ctx.preprocess(view, filt=filt, sampling_rate=100.0, detrend=detrend)
    b = ctx.preprocess(packed, filt=filt, sampling_rate=100.0, detrend=detrend)
    ctx.sync()
    assert torch.equal(a, b)
    odd = torch.from_numpy(np.repeat(rec, 2, axis=1)).cuda()[:, ::2].as_strided((nchunk, Nc, T), (2 * step, 2 * Tday, 2), 2 * first)
    c = ctx.preprocess(odd, filt=filt, sampling_rate=100.0, detrend=detrend)
    ctx.sync()
    assert torch.equal(c, b)


@pytest.mark.parametrize('seed', range(26))
def test_preprocess_random_shapes(ctx, seed):
    """Seeded random lengths around the block structure of the filter kernels (blocks of 256 samples: shorter than
    one, whole multiples, one sample off), channel / chunk counts, band-pass orders 1 .. 4 (two to eight filter
    states), causal and zero-phase, with and without the trend: the scipy restatement within 2e-6 of the output's
    scale -- the zero-state block states are dot products with an impulse table here, a filter run there."""
    rs = np.random.RandomState(1000 + seed + SOAK)
    base = int(rs.choice([1, 2, 3, 5, 9, 40]))
    T = max(40, base * 256 + int(rs.choice([-1, 0, 1, 17, 128, 255])))
    if seed == 0:
        T = 200                                         # shorter than one block
    Nc = int(rs.choice([1, 2, 3, 6]))
    if seed >= 16:                                      # channel counts the multiplex has no vector form for
        Nc = [5, 7, 8, 9, 12][(seed - 16) % 5]
    nchunk = int(rs.choice([1, 2, 5]))
    corners = int(rs.choice([1, 2, 3, 4]))
    sr = float(rs.choice([40.0, 100.0, 500.0]))
    lo = sr * rs.uniform(0.01, 0.05)
    hi = sr * rs.uniform(0.1, 0.4)
    zerophase = bool(rs.randint(2))
    detrend = bool(rs.randint(2))
    filt = [lo, hi, corners, zerophase]
    t = np.arange(T)
    raw = (rs.standard_normal((nchunk, Nc, T)) * 50 + rs.uniform(-500, 500, (nchunk, Nc, 1)) +
           0.05 * t[None, None, :] * rs.uniform(-1, 1, (nchunk, Nc, 1))).astype(np.float32)
    out = ctx.preprocess(torch.from_numpy(raw).cuda(), filt=filt, sampling_rate=sr, detrend=detrend)
    ctx.sync()
    got = out.cpu().numpy()
    assert got.shape == (nchunk, Nc * T)
    for b in range(nchunk):
        want = orc.multiplex(orc.apply_filter(list(raw[b]), filt, sr, detrend=detrend))
        scale = np.max(np.abs(want))
        assert np.max(np.abs(got[b] - want)) <= 2e-6 * scale + 1e-7 * 1000, (seed, T, Nc, filt, detrend, b)


def test_preprocess_feeds_the_scan(ctx):
    """Raw counts -> device pre-processing -> detection statistic, against the
    oracle pipeline on the same raw data."""
    rs = np.random.RandomState(3 + SOAK)
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
    rs = np.random.RandomState(12 + SOAK)
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
    # the same call under another current stream (the short chunk groups are filtered on helper contexts with
    # streams of their own, ordered against the station's stream) and with the per-station contexts of the
    # SubSpace instead of the caller's: the same table
    with torch.cuda.stream(torch.cuda.Stream()):
        again = ss.detex(subspaceDB=None, estimateMags=True)
    assert len(again['ss_df']) == len(got)
    a2 = again['ss_df'].sort_values('STMP')
    assert np.array_equal(a2.STMP.values, got.STMP.values) and np.allclose(a2.DS.values, got.DS.values, atol=1e-6)
    assert np.allclose(a2.Mag.values.astype(float), got.Mag.values.astype(float), rtol=1e-5, equal_nan=True)
    # false-alarm statistics over every chunk of the same raw fetcher (fas.py:88-117)
    ss.getFAS(conDatNum=len(alld), staltalimit=None, seed=0)
    fas = ss.subspaces['XX.STA'].FAS.iloc[0]
    ref = np.histogram(np.concatenate(alld), bins=np.linspace(-.01, 1, 401))[0]
    assert fas['hist'].sum() == ref.sum()
    # bin-edge ties only: the counts below an edge differ by no more than the values that lie within the parity
    # tolerance (1e-5) of that edge (a fixed "<= 3" held for the committed draw and failed, with 4, on one of 30
    # re-draws of a soak run: the noise windows' statistic crowds the first bins, ~9000 values per bin)
    allv = np.sort(np.concatenate(alld))
    edges = np.linspace(-.01, 1, 401)[1:]
    near = np.searchsorted(allv, edges + 1e-5, 'right') - np.searchsorted(allv, edges - 1e-5, 'left')
    flips = np.abs(np.cumsum(fas['hist']) - np.cumsum(ref))
    assert np.all(flips <= near) and flips.max() <= 8, (int(flips.max()), int(near[np.argmax(flips)]))


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
    rs = np.random.RandomState(T + factor + SOAK)
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


def _random_fragments(rs, Nc, T, holes, dead_channel=False):
    """Non-overlapping fragments of int32 counts per channel with `holes` gaps each."""
    frags = []
    for c in range(Nc):
        if dead_channel and c == Nc - 1:
            continue
        cuts = np.sort(rs.choice(np.arange(20, T - 20), size=2 * holes, replace=False))
        edges = [0] + list(cuts) + [T]
        for k in range(0, len(edges), 2):
            a, b = int(edges[k]), int(edges[k + 1])
            if b > a:
                x = rs.randint(-2000, 2000, size=b - a).astype(np.int32)
                x[x == 0] = 7                      # zeros only where planted below
                frags.append((c, a, x))
    return frags


@pytest.mark.parametrize('Nc,T,holes,seed', [(3, 5000, 0, 0), (3, 5000, 2, 1), (1, 777, 3, 2),
                                             (6, 20000, 4, 3), (3, 372000, 1, 4), (2, 1500, 0, 5),
                                             (5, 3000, 2, 6), (7, 1500, 1, 7), (8, 4097, 3, 8), (9, 2500, 2, 9),
                                             (12, 1000, 1, 10)])
def test_ingest_counts_matches_merge_restatement(ctx, Nc, T, holes, seed):
    """dtx_ingest_counts / dtx_gapfree_run against the NumPy restatement of the merge with zero
    fill (construct.py:1038 / 1070) and of _mergeChannels' longest-run rule (construct.py:1033-1066),
    incl. an exact zero sample acting as a gap, ties between equal runs (the first wins), a
    run on a column slice of a larger buffer.  ObsPy itself is not installable here: parity vs
    ObsPy's Stream.merge / Trace.split unpinned."""
    rs = np.random.RandomState(seed + SOAK)
    frags = _random_fragments(rs, Nc, T, holes)
    if seed == 5:                                  # two equal runs: sample 500 and 1001 are zero
        frags = [(c, 0, np.where(np.isin(np.arange(T), [500, 1001]), 0, 9).astype(np.int32)) for c in range(Nc)]
        frags[1] = (1, 0, np.full(T, 5, np.int32))
    if seed == 0:                                  # a genuine zero count in one channel
        c, a, x = frags[1]
        x[T // 3] = 0
    flat = np.concatenate([x for (_, _, x) in frags])
    off, desc = 0, []
    for (c, a, x) in frags:
        desc.append((c, off, a, len(x)))
        off += len(x)
    planes, run = ctx.ingest_counts(torch.from_numpy(flat).cuda(), desc, Nc, T)
    ctx.sync()
    want = orc.merge_fill(frags, Nc, T)
    assert np.array_equal(planes.cpu().numpy(), want.astype(np.float32))
    r0, rl = (int(v) for v in run.cpu().numpy())
    assert (r0, rl) == orc.merge_channels_run(want)
    if seed == 5:
        assert (r0, rl) == (0, 500)                # 0..499 and 501..1000 tie: the first is kept
    # the run of a chunk = columns [a, b) of the day buffer
    a, b = T // 5, T - T // 7
    sub = ctx.gapfree_run(planes[:, a:b])
    ctx.sync()
    assert tuple(int(v) for v in sub.cpu().numpy()) == orc.merge_channels_run(want[:, a:b])
    # a channel without any data: nothing survives
    frags2 = _random_fragments(rs, max(Nc, 2), T, 1, dead_channel=True)
    flat2 = np.concatenate([x for (_, _, x) in frags2])
    off, desc2 = 0, []
    for (c, a_, x) in frags2:
        desc2.append((c, off, a_, len(x)))
        off += len(x)
    _, run2 = ctx.ingest_counts(torch.from_numpy(flat2).cuda(), desc2, max(Nc, 2), T)
    assert tuple(int(v) for v in run2.cpu().numpy()) == (-1, 0)


def test_detex_on_raw_counts_with_gaps(ctx):
    """SubSpace.detex on a fetcher that holds raw int32 counts in fragments (add_counts): merge
    with zero fill, per-chunk longest gap-free stretch, detrend + band-pass, scan -- against the
    oracle's merge -> trim -> filter -> DS chain; a chunk whose kept stretch is still in two
    pieces on one channel is dropped ('too fractured', construct.py:1011-1018); fillZeros=True
    scans the zero-filled chunks whole."""
    from detex_amd import ContinuousFetcher, EventFetcher, createCluster, createSubSpace
    from detex_amd.getdata import multiplex
    rs = np.random.RandomState(21 + SOAK)
    sr, Nc, L = 20.0, 3, 160
    filt = [1, 8, 2, True]
    wav = [np.convolve(rs.standard_normal(L + 6), np.hanning(7), 'valid') * np.exp(-np.arange(L) / 50.0)
           for _ in range(Nc)]
    wavf = orc.apply_filter([w.astype(np.float32) for w in wav], filt, sr)
    ef = EventFetcher(sr, ['E', 'N', 'Z'])
    for i in range(4):
        ef.add('XX.STA', 'ev%d' % i, np.array([rs.uniform(.5, 2) * w + 0.02 * rs.standard_normal(L)
                                                for w in wavf]), starttime=100. * i)
    T = int(1500 * sr)                                   # chunks of 600 s + 60 s: 0, 12000, 24000
    con = rs.standard_normal((Nc, T)) * 300
    for t in (4000, 9000, 16000, 22000, 27000):
        for c in range(Nc):
            con[c, t:t + L] += 3000.0 * wav[c]
    cnt = np.rint(con).astype(np.int32)
    # zero-mean counts: exact zeros are routine.  In chunks 0 and 1, which arrive in several traces, they
    # would be gaps of _mergeChannels (construct.py:1047-1050) -> removed there; chunk 2 (24000..29999) is
    # ONE trace per channel and keeps its zeros: the reference does not merge it (construct.py:1005)
    head = cnt[:, :25200]
    head[head == 0] = 1
    assert (cnt[:, 25200:] == 0).sum() >= 2
    # channel 1 loses samples 6000..6999 (inside chunk 0); channel 2 is delivered in two abutting
    # traces that meet at 20000 (inside chunk 1) and has a hole at 13000..13099 as well
    frags = [(0, 0, cnt[0]), (1, 0, cnt[1, :6000]), (1, 7000, cnt[1, 7000:]),
             (2, 0, cnt[2, :13000]), (2, 13100, cnt[2, 13100:20000]), (2, 20000, cnt[2, 20000:])]
    cf = ContinuousFetcher(sr, ['E', 'N', 'Z'], conDatDuration=600, conBuff=60)
    cf.add_counts('XX.STA', frags, T, starttime=6.0e5)
    cl = createCluster(CCreq=0.5, fetch_arg=ef, filt=filt, ctx=ctx)
    ss = createSubSpace(clust=cl, conDatFetcher=cf)
    ss.attachPickTimes(0.0, 6.0)
    ss.SVD(selectCriteria=4, selectValue=0, threshold=0.35, useSingles=False)
    res = ss.detex(subspaceDB=None, ctx=ctx, estimateMags=False)
    names, kinds, U, off, offsets, thr = ss._detector_bank('XX.STA', useSingles=False)
    merged = orc.merge_fill(frags, Nc, T)
    want = []
    kept = []
    for (s0, n, ts) in cf.chunk_bounds('XX.STA'):
        sp = orc.chunk_span(frags, Nc, s0, n)            # construct.py:1003-1023, gated like the reference
        if sp is None:
            continue
        a, rl = sp
        kept.append((a, rl))
        x = multiplex(orc.apply_filter([m.astype(np.float32) for m in merged[:, a:a + rl]], filt, sr))
        if len(x) <= U.shape[1]:
            continue
        ds = orc.mpxds_fft(x, U.astype(np.float32), Nc)
        if orc.eval_trig_con(ds.max(), np.float32(thr['SS0'])):
            want += [(t['STMP'], t['DS']) for t in orc.pick_triggers(ds, np.float32(thr['SS0']), sr, ts + (a - s0) / sr)]
    # chunk 0 (0..13199) has two holes that leave equal stretches 0..5999 and 7000..12999: the
    # first is kept; chunk 1 (12000..25199) is cut to 13100..25199, which holds the seam at 20000
    # on channel 2 -> dropped; chunk 2 is whole, zeros and all
    assert kept == [(0, 6000), (24000, 6000)]
    got = res['ss_df'].sort_values('STMP')
    assert len(got) == len(want) >= 2
    for (_, g), w in zip(got.iterrows(), sorted(want)):
        assert g.STMP == w[0] and abs(g.DS - w[1]) <= 1e-5
    # fillZeros: the merged chunks are scanned whole (construct.py:1007-1008)
    resz = ss.detex(subspaceDB=None, ctx=ctx, estimateMags=False, fillZeros=True)
    wz = []
    for (s0, n, ts) in cf.chunk_bounds('XX.STA'):
        x = multiplex(orc.apply_filter([m.astype(np.float32) for m in merged[:, s0:s0 + n]], filt, sr))
        ds = orc.mpxds_fft(x, U.astype(np.float32), Nc)
        if orc.eval_trig_con(ds.max(), np.float32(thr['SS0'])):
            wz += [(t['STMP'], t['DS']) for t in orc.pick_triggers(ds, np.float32(thr['SS0']), sr, ts)
                   if t['DS'] <= 1.05]
    gz = resz['ss_df'].sort_values('STMP')
    assert len(gz) == len(wz) >= len(want)
    for (_, g), w in zip(gz.iterrows(), sorted(wz)):
        assert g.STMP == w[0] and abs(g.DS - w[1]) <= 1e-5 and g.DS_STALTA == 0.0
    # false-alarm statistics from the same fragments (fas.py:88-117: every sampled chunk through _applyFilter with
    # fillZeros=False, i.e. trimmed to its gap-free stretch; the fractured chunk is drawn, counted and skipped): the
    # histogram of the DS values of the two kept stretches (this raised until round 6)
    ss.getFAS(conDatNum=3, staltalimit=None, seed=0)
    fas = ss.subspaces['XX.STA'].FAS.iloc[0]
    alld = []
    for (a, rl) in kept:
        x = multiplex(orc.apply_filter([m.astype(np.float32) for m in merged[:, a:a + rl]], filt, sr))
        alld.append(orc.mpxds_fft(x, U.astype(np.float32), Nc))
    allv = np.sort(np.concatenate(alld))
    edges = np.linspace(-.01, 1, 401)
    ref = np.histogram(allv, bins=edges)[0]
    assert fas['hist'].sum() == ref.sum() == sum(len(d) for d in alld)
    near = np.searchsorted(allv, edges[1:] + 1e-5, 'right') - np.searchsorted(allv, edges[1:] - 1e-5, 'left')
    flips = np.abs(np.cumsum(fas['hist']) - np.cumsum(ref))
    assert np.all(flips <= near), (int(flips.max()), int(near[np.argmax(flips)]))


def test_detex_on_raw_counts_with_resent_packets(ctx):
    """Overlapping (re-sent) packets in ContinuousFetcher.add_counts through the device path: resolved on the host
    by ObsPy's merge rule (getdata.resolve_overlaps; restated, parity vs ObsPy unpinned), merged / trimmed /
    filtered / scanned on the device -- against the oracle chain on the ORIGINAL traces, and the same detections
    as a record that was delivered clean wherever the duplicates were identical."""
    from detex_amd import ContinuousFetcher, EventFetcher, createCluster, createSubSpace
    from detex_amd.getdata import multiplex
    rs = np.random.RandomState(34 + SOAK)
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
    clean = [(0, 0, cnt[0, :12000]), (0, 12000, cnt[0, 12000:24000]), (0, 24000, cnt[0, 24000:]),
             (1, 0, cnt[1, :12000]), (1, 12000, cnt[1, 12000:24000]), (1, 24000, cnt[1, 24000:]),
             (2, 0, cnt[2, :12000]), (2, 12000, cnt[2, 12000:24000]),
             (2, 24000, cnt[2, 24000:25000]), (2, 25100, cnt[2, 25100:])]
    frags = clean + [(0, 5000, cnt[0, 5000:5600]),       # chunk 0: identical duplicate -> merged, scanned whole
                     (1, 18000, bad),                    # chunk 1: differs -> overlap zero-filled -> longest stretch
                     (1, 30000, cnt[1, 30000:30500])]    # chunk 2: hole + duplicate in the kept stretch -> too fractured

    def run(fr):
        cf = ContinuousFetcher(sr, ['E', 'N', 'Z'], conDatDuration=600, conBuff=0)
        cf.add_counts('XX.STA', fr, T, starttime=6.0e5)
        cl = createCluster(CCreq=0.5, fetch_arg=ef, filt=filt, ctx=ctx)
        ss = createSubSpace(clust=cl, conDatFetcher=cf)
        ss.attachPickTimes(0.0, 6.0)
        ss.SVD(selectCriteria=4, selectValue=0, threshold=0.35, useSingles=False)
        return cf, ss, ss.detex(subspaceDB=None, ctx=ctx, estimateMags=False)['ss_df'].sort_values('STMP')

    cf, ss, got = run(frags)
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
    assert spans[:3] == [(0, 12000), (12000, 6000), None]
    assert len(got) == len(want) >= 3
    for (_, g), w in zip(got.iterrows(), sorted(want)):
        assert g.STMP == w[0] and abs(g.DS - w[1]) <= 1e-5
    # chunk 0 with its identical duplicate = chunk 0 of the clean record, bit for bit
    _, _, ref = run(clean)
    t1 = 6.0e5 + 600
    assert got[got.STMP < t1][['STMP', 'DS']].values.tolist() == ref[ref.STMP < t1][['STMP', 'DS']].values.tolist()


@pytest.mark.parametrize('dec', [None, 2])
def test_raw_templates_and_raw_continuous_equal_prefiltered_run(ctx, dec):
    """f4 end to end: RAW template records + RAW continuous data through
    createCluster(filt, trim, decimate) -> createSubSpace -> SVD -> detex, everything cut / decimated /
    detrended / band-passed on the device, against the same workflow on waveforms and chunks pre-processed
    beforehand with the CPU restatement (oracle.decimate / apply_filter): same events and clusters, lags
    exact, CCs and detection statistics to 1e-5, the same detections."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from detex_amd import ContinuousFetcher, EventFetcher, createCluster, createSubSpace
    from test_host_logic import prefiltered_from_raw, raw_event_records
    recs, sr, Nc = raw_event_records(seed=9, nev=9)
    q = dec or 1
    # (without the 1000-count offset of the CPU test's records when decimating: the causal low-pass would ring
    # on it at the start of every window -- reference behaviour, but it blurs the families)
    if dec:
        for r in recs:
            r['data'] = (r['data'] - 1000.0).astype(np.float32)
    filt, trim = [1, 8, 2, True], [5, 20]
    efr = EventFetcher(sr, ['E', 'N', 'Z'])
    for r in recs:
        efr.add_raw('XX.RAW', r['name'], r['data'], r['starttime'], r['origintime'], r['magnitude'])
    # continuous data: noise + drift + copies of the two families' records around their arrivals
    rs = np.random.RandomState(2 + SOAK)
    T = int(2400 * sr)
    con = rs.standard_normal((Nc, T)) * 6.0 + (0.0 if dec else 1000.0) + 0.002 * np.arange(T)[None, :]
    pre = int(30 * sr)
    plants = [int(200 * sr), int(700 * sr), int(1150 * sr), int(1700 * sr), int(2100 * sr)]
    for j, t in enumerate(plants):
        src = recs[j % 2]['data']
        seg = src[:, pre:pre + int(20 * sr)] - np.median(src, axis=1, keepdims=True)
        con[:, t:t + seg.shape[1]] += 1.3 * seg
    con = np.round(con).astype(np.float32)
    cfr = ContinuousFetcher(sr, ['E', 'N', 'Z'], conDatDuration=600, conBuff=60, raw=True)
    cfr.add('XX.RAW', con, starttime=9.0e5)
    # the pre-filtered twin: chunk by chunk on the CPU restatement
    efp = prefiltered_from_raw(recs, sr, Nc, filt, trim, dec)

    def run(ef, cf, **kw):
        cl = createCluster(CCreq=0.6, fetch_arg=ef, filt=filt, trim=trim, decimate=dec, ctx=ctx, saveclust=False)
        ss = createSubSpace(clust=cl, conDatFetcher=cf)
        ss.attachPickTimes(6.0, 10.0)
        ss.SVD(selectCriteria=4, selectValue=0, threshold=0.4, useSingles=False)
        return cl, ss
    cla, ssa = run(efr, cfr)
    clb, ssb = run(efp, None)
    ra, rb = cla.trdf.iloc[0], clb.trdf.iloc[0]
    assert list(ra.Events) == list(rb.Events) and len(ra.Events) == 9
    for ev in ra.Events:
        wa, wb = cla._mptd['XX.RAW'][ev], clb._mptd['XX.RAW'][ev]
        assert len(wa) == len(wb) and np.max(np.abs(wa - wb)) <= 2e-6 * np.max(np.abs(wb)) + 1e-4
    assert np.allclose(ra.CCs.values.astype(float), rb.CCs.values.astype(float), atol=1e-5, equal_nan=True)
    assert np.array_equal(np.nan_to_num(ra.Lags.values.astype(float), nan=-1),
                          np.nan_to_num(rb.Lags.values.astype(float), nan=-1))
    assert sorted(map(sorted, cla['RAW'].clusts)) == sorted(map(sorted, clb['RAW'].clusts)) and len(cla['RAW'].clusts) == 2
    res = ssa.detex(subspaceDB=None, ctx=ctx, estimateMags=True)
    # the reference's per-chunk _applyFilter on the CPU restatement, scanned with the pre-filtered run's detectors
    names, kinds, U, off, offsets, thr = ssb._detector_bank('XX.RAW', useSingles=False)
    srd = sr / q
    want = []
    for chunk, t0 in cfr.getConData('XX.RAW'):
        ch = [np.asarray(c, np.float32) for c in chunk]
        if dec:
            ch = [np.asarray(c, np.float32) for c in orc.decimate(ch, dec, sr)]
        x = orc.multiplex(orc.apply_filter(ch, filt, srd))
        for d, nm in enumerate(names):
            ds = orc.mpxds_fft(x, U[off[d]:off[d + 1]].astype(np.float32), Nc)
            if orc.eval_trig_con(ds.max(), np.float32(thr[nm])):
                want += [(t['STMP'], nm, t['DS']) for t in orc.pick_triggers(ds, np.float32(thr[nm]), srd, t0)]
    got = res['ss_df'].sort_values(['STMP', 'Name'])
    want.sort()
    assert len(got) == len(want) >= len(plants)
    for (_, g), w in zip(got.iterrows(), want):
        assert abs(g.STMP - w[0]) < 1e-6 and g.Name == w[1] and abs(g.DS - w[2]) <= 1e-5, (g.STMP, g.Name, g.DS, w)
    # classifyEvents on RAW records: the events are cut / decimated / filtered on the device like the templates
    # were (SubSpace._event_chunks), every member of a cluster is best matched by its own subspace
    ec = ssa.detex(subspaceDB=None, ctx=ctx, estimateMags=False, classifyEvents=True, eventCorFile=None)['event_cors']
    assert set(ec.Event) == set(ra.Events)
    for _, row in ssa.subspaces['XX.RAW'].iterrows():
        for e in row.Events:
            mine = ec[ec.Event == e].set_index('Name').MaxDS
            assert mine.idxmax() == row.Name and mine[row.Name] > 0.6, (e, dict(mine))
