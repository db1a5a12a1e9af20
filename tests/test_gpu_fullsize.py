"""BASELINE.json's full-size configurations on the GPU, checked through
size-independent properties and spot checks against the oracle (the oracle
cannot finish the full problems in seconds)."""
import numpy as np
import pytest

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu

from detex_amd import synth  # noqa: E402
from oracle import detex_oracle as orc  # noqa: E402


@pytest.fixture(scope='module')
def ctx():
    from detex_amd.hip import HipContext
    c = HipContext(0)
    yield c
    c.close()


def _spot_ds(x, U, Nc, p):
    """Closed form at one window, float64."""
    n = U.shape[1]
    w = x[p * Nc: p * Nc + n].astype(np.float64)
    mu = w.mean()
    den = n / (n - 1.0) * np.sum((w - mu) ** 2)
    Ud = U.astype(np.float64)
    return np.sum((Ud @ w - Ud.sum(axis=1) * mu) ** 2) / den


@pytest.mark.parametrize('cfg', ['config2', 'config5'])
def test_scan_fullsize_properties(ctx, cfg):
    if cfg == 'config2':      # 200 basis vectors, 3-ch 100 Hz, 30 s templates, 3720 s chunk
        Nc, L, D, k, T, nchunk = 3, 3000, 25, 8, 372000, 2
    else:                     # 500 Hz 6-ch borehole array, 64-dim basis, 2 s window, 1 h
        Nc, L, D, k, T, nchunk = 6, 1000, 1, 64, 1800000, 1
    n, M = Nc * L, Nc * T
    U, off = synth.detector_bank(5, D, k, n)
    x, plants = synth.continuous_day(9, nchunk, M, Nc, U, off, nplant=8)
    xt = torch.from_numpy(x).cuda()
    ctx.upload_basis(U, off, Nc)
    ds = ctx.subspace_ds(xt)
    P = T - L + 1
    assert tuple(ds.shape) == (nchunk, D, P)
    # range: orthonormal zero-mean basis -> 0 <= DS <= (n-1)/n
    assert float(ds.min()) >= 0.0 and float(ds.max()) <= (n - 1.0) / n + 1e-5
    assert bool(torch.isfinite(ds).all())
    # scale invariance (linearity of the numerator and the denominator)
    ds2 = ctx.subspace_ds(xt * 37.5)
    assert float((ds - ds2).abs().max()) <= 5e-6
    # planted events are the block maxima of their detector
    res = ctx.ds_post(ds, thr=np.full(D, 0.3, np.float32))
    ctx.sync()
    cb, cd, cp, cv = res['cand']
    for (b, p, d) in plants:
        assert float(ds[b, d, p]) > 0.9
        assert ((cb == b) & (cd == d) & (cp == p)).any()
    # histogram is a partition of the windows with 0 <= DS <= 1
    assert int(res['hist'].sum()) == nchunk * D * P
    assert np.array_equal(res['maxds'].cpu().numpy(), ds.amax(dim=2).cpu().numpy())
    # spot checks against the float64 closed form
    rs = np.random.RandomState(0)
    dsh = ds.cpu().numpy()
    pts = [(b, p, d) for (b, p, d) in plants[:3]] + \
          [(int(rs.randint(nchunk)), int(rs.randint(P)), int(rs.randint(D))) for _ in range(12)] + \
          [(0, 0, 0), (nchunk - 1, P - 1, D - 1)]
    for (b, p, d) in pts:
        ref = _spot_ds(x[b], U[off[d]:off[d + 1]], Nc, p)
        assert abs(dsh[b, d, p] - ref) <= 1e-5, (b, p, d, dsh[b, d, p], ref)
    # regrouping the same rows into other detectors: energies add (the in-register
    # membership-MFMA epilogue at full size)
    if D % 5 == 0:
        off2 = (np.arange(D // 5 + 1) * 5 * k).astype(np.int32)
        ctx.upload_basis(U, off2, Nc)
        dsg = ctx.subspace_ds(xt[:1])
        want = ds[:1].reshape(1, D // 5, 5, P).sum(dim=2)
        assert float((dsg - want).abs().max()) <= 5e-6


def test_xcorr_config3_scale(ctx):
    """configs[2] shape (Nc=3, L=13000 -> n=39000) at a reduced event count:
    known-answer pairs and spot checks against the reference FFT algorithm."""
    Nc, L, N = 3, 13000, 230
    td = synth.event_families(2, 23, 10, Nc, L)
    rs = np.random.RandomState(1)
    # known answers: event 1 = event 0 delayed by 37 samples; event 3 = 2.5 * event 2
    base = td[0].reshape(L, Nc)
    sh = 37
    td[1] = np.vstack([rs.standard_normal((sh, Nc)).astype(np.float32) * 1e-3, base[:-sh]]).reshape(-1)
    td[3] = 2.5 * td[2]
    cc, lag, ss = ctx.xcorr_allpairs(torch.from_numpy(td).cuda(), Nc)
    ctx.sync()
    cc, lag, ss = cc.cpu().numpy(), lag.cpu().numpy(), ss.cpu().numpy()
    idx = lambda b, c: N * b - b * (b + 1) // 2 + (c - b - 1)  # noqa: E731
    assert lag[idx(0, 1)] == sh * Nc and cc[idx(0, 1)] > 0.99
    assert lag[idx(2, 3)] == 0 and abs(cc[idx(2, 3)] - 1) <= 1e-5 and abs(ss[idx(2, 3)]) < 1e-3
    assert np.all(np.abs(cc) <= 1 + 1e-5) and np.all(lag % Nc == 0)
    assert np.all(np.abs(lag) <= (L - L // 2) * Nc)
    pairs = [(0, 1), (4, 5), (10, 11), (0, 229), (228, 229), (7, 150), (100, 209), (207, 208),
             (208, 209), (13, 14)]
    fds = {}
    for (b, c) in pairs:
        for e in (b, c):
            if e not in fds:
                fds[e] = orc.get_freq_domain(td[e].astype(np.float64))
        rc, rl, rss, res = orc.ccx2_fft(td[b], td[c], Nc, fds[b], fds[c], scrub_tol=1e-5)
        i = idx(b, c)
        srt = np.sort(res[np.isfinite(res)])
        assert abs(cc[i] - rc) <= 1e-5, (b, c)
        if srt[-1] - srt[-2] > 5e-6:
            assert lag[i] == rl, (b, c)
            if abs(rss) <= .5:
                assert abs(ss[i] - rss) <= 1e-4, (b, c)
