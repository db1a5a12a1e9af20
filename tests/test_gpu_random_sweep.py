"""Seeded random sweep of problem shapes (channels, template length, detector
sizes, block length / stride, engine) for the detection statistic and of event
sets for the similarity matrix, each checked against the oracle."""
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


@pytest.mark.parametrize('seed', range(24))
def test_random_scan(ctx, seed):
    rs = np.random.RandomState(1000 + seed)
    Nc = int(rs.choice([1, 2, 3, 4, 6]))
    L = int(rs.randint(4, 260))
    n = Nc * L
    nd = int(rs.randint(1, 7))
    ks = [int(rs.choice([1, 1, 2, 3, 5, 8, 13, 16, 17, 40, 100])) for _ in range(nd)]
    if rs.rand() < 0.2:
        ks = [1] * int(rs.randint(200, 450))           # many singles -> several K-tiles
    off = np.concatenate([[0], np.cumsum(ks)]).astype(np.int32)
    U = rs.standard_normal((off[-1], n))
    U /= np.linalg.norm(U, axis=1, keepdims=True)
    if rs.rand() < 0.5:
        U -= U.mean(axis=1, keepdims=True)
    nchunk = int(rs.randint(1, 4))
    T = int(rs.randint(L + 1, L + 2500))
    M = Nc * T - int(rs.randint(0, Nc))                # ragged tail
    stride = int(rs.randint(Nc, M + 1) // Nc * Nc)
    buf = rs.standard_normal((nchunk - 1) * stride + M) * float(10 ** rs.uniform(-6, 3)) + \
        float(rs.choice([0, 0, 5]))
    ctx.set_engine('bf16x3' if seed % 2 else 'fp32')
    ctx.upload_basis(U.astype(np.float32), off, Nc)
    ds = ctx.subspace_ds(torch.from_numpy(buf.astype(np.float32)).cuda(), M=M,
                         chunk_stride=stride, nchunk=nchunk)
    ctx.sync()
    ds = ds.cpu().numpy()
    Uf = U.astype(np.float32).astype(np.float64)
    bf = buf.astype(np.float32).astype(np.float64)
    P = (M - n) // Nc + 1
    assert ds.shape == (nchunk, len(ks), P)
    rows = rs.choice(len(ks), size=min(len(ks), 6), replace=False)
    for b in range(nchunk):
        xb = bf[b * stride: b * stride + M]
        for d in rows:
            ref = orc.mpxds_fft(xb, Uf[off[d]:off[d + 1]], Nc)
            scale = max(1.0, float(np.max(ref)))       # non-orthonormal rows can exceed 1
            assert np.max(np.abs(ds[b, d] - ref)) <= 1e-5 * scale, (seed, b, d, Nc, L, ks[d])
    ctx.set_engine('bf16x3')


@pytest.mark.parametrize('seed', range(8))
def test_random_xcorr(ctx, seed):
    from detex_amd import synth
    rs = np.random.RandomState(2000 + seed)
    Nc = int(rs.choice([1, 2, 3, 6]))
    L = int(rs.randint(8, 400))
    N = int(rs.choice([2, 3, 7, 20, 64, 209, 215]))
    nfam = max(1, N // 5)
    td = synth.event_families(seed, nfam, (N + nfam - 1) // nfam, Nc, L, maxshift=max(1, L // 10))[:N]
    td = td * np.float32(10 ** rs.uniform(-5, 4))
    ctx.set_engine('bf16x3' if seed % 2 else 'fp32')
    cc, lag, ss = ctx.xcorr_allpairs(torch.from_numpy(td).cuda(), Nc)
    ctx.sync()
    cc, lag, ss = cc.cpu().numpy(), lag.cpu().numpy(), ss.cpu().numpy()
    pairs = [(b, c) for b in range(N) for c in range(b + 1, N)]
    if len(pairs) > 60:
        pairs = [pairs[i] for i in rs.choice(len(pairs), 60, replace=False)]
    n = Nc * L
    trunc = n // (2 * Nc) - 1
    for (b, c) in pairs:
        i = N * b - b * (b + 1) // 2 + (c - b - 1)
        raw = orc.ccx2_raw_direct(td[b].astype(np.float64), td[c].astype(np.float64), Nc)
        rc, rl, rss, res = orc.ccx2_reduce(raw, n, Nc, trunc, 1e-5)
        assert abs(cc[i] - rc) <= 1e-5, (seed, b, c)
        fin = np.sort(res[np.isfinite(res)])
        if len(fin) > 1 and fin[-1] - fin[-2] > 5e-6 and abs(abs(rc) - 1) > 2e-5:
            assert lag[i] == rl, (seed, b, c)
            if abs(rss) <= .5:
                assert abs(ss[i] - rss) <= 1e-4, (seed, b, c)
    ctx.set_engine('bf16x3')
