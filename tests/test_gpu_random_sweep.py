"""Seeded random sweep of problem shapes (channels, template length, detector
sizes, block length / stride, engine) for the detection statistic and of event
sets for the similarity matrix, each checked against the oracle."""
import numpy as np
import pytest

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu

# DETEX_TEST_SEED_BASE=k re-draws every seeded sweep of this file (soak runs: scripts/soak.sh); 0 = the committed cases
SOAK = 100000 * int(__import__('os').environ.get('DETEX_TEST_SEED_BASE', 0))

from oracle import detex_oracle as orc  # noqa: E402

# channel counts for which the library has no compile-time form (the reference takes any Nc:
# construct.py:428-439, detect.py:577): the seeds past each test's original range draw from these
NC_EXTRA = [5, 7, 8, 9, 12]


@pytest.fixture(scope='module')
def ctx():
    from detex_amd.hip import HipContext
    c = HipContext(0)
    yield c
    c.close()


@pytest.mark.parametrize('engine', ['mfma', 'fft', 'auto'])
@pytest.mark.parametrize('seed', range(34))
def test_random_scan(ctx, seed, engine):
    """engine 'mfma': the two contraction engines alternate over the seeds; 'fft': the
    frequency-domain engine forced on the same shapes (dead stretches, spikes, offsets);
    'auto': the library's own choice (float32 MFMA below 64 samples per channel, split-bf16 with its
    guard + float32 repair up to 255, the frequency-domain engine from 256).  Every window is held to
    1e-5 -- the windows that hold the 10^4 sigma spike included."""
    rs = np.random.RandomState(1000 + seed + SOAK)
    Nc = int(rs.choice([1, 2, 3, 4, 6]))
    if seed >= 24:
        Nc = NC_EXTRA[(seed - 24) % 5]
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
    amp = float(10 ** rs.uniform(-6, 3))
    dc = float(rs.choice([0, 0, 5]))
    buf = rs.standard_normal((nchunk - 1) * stride + M) * amp + dc
    if rs.rand() < 0.35 and len(buf) > 6 * n:
        # dead stretch: the reference zero-fills gaps BEFORE detrend + band-pass
        # (construct.py:1005-1028), so in the scan input a gap sits at the data's DC level
        g0 = int(rs.randint(0, len(buf) - 3 * n))
        buf[g0: g0 + int(rs.randint(n, 3 * n))] = dc
    spike = -1
    if rs.rand() < 0.35:                                    # spike 10^4 x the noise level
        spike = int(rs.randint(len(buf)))
        buf[spike] += 1e4 * amp
    split = engine == 'mfma' and seed % 2 == 1
    ctx.set_engine({'fft': 'fft', 'auto': 'auto'}.get(engine, 'bf16x3' if split else 'fp32'))
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
            # (the reference's frequency-domain form -- rolling E[x^2] - mean^2 in float64 -- loses (dc / amp)^2 * 1e-16:
            # 6e-5 when the data ride 5 on top of 6.5e-6 of noise.  The statistic does not change when a constant is
            # added to the data (detect.py:559-578: the mean is taken out of every window), so that form is given
            # the data minus their mean; the direct form subtracts every window's own mean and needs no help.)
            ref = orc.mpxds_direct(xb, Uf[off[d]:off[d + 1]], Nc) if P * n < 3e6 else \
                orc.mpxds_fft(xb - xb.mean(), Uf[off[d]:off[d + 1]], Nc)
            live = np.lib.stride_tricks.sliding_window_view(xb, n)[::Nc].std(axis=1) > 0
            assert np.all(ds[b, d][~live] == 0)        # zero-variance windows score 0
            scale = max(1.0, float(np.max(ref[live])) if live.any() else 1.0)  # non-orthonormal rows can exceed 1 (no live window: a chunk inside the dead stretch)
            # a window that holds the spike is that one sample times a basis column: nothing averages
            # the split products' 2^-17 rounding there -- the split engine's guard hands such blocks to
            # the float32 engine (split_guard_kernel), so those windows are held to 1e-5 like all others
            bad = (np.abs(ds[b, d] - ref) > 1e-5 * scale) & live
            assert not bad.any(), (seed, b, d, Nc, L, ks[d],
                                   float(np.max(np.abs(ds[b, d] - ref)[live])), scale)
    ctx.set_engine('auto')


@pytest.mark.parametrize('seed', range(24))
def test_random_scan_long(ctx, seed):
    """Default engine on templates long enough for the frequency-domain engine: every block
    advance V = 2048 .. 7680, several blocks per chunk with a ragged last one, overlapping
    chunks, Nc = 1 / 3 (block spectra in LDS) and 2 / 4 / 6 (streamed), single-vector
    detectors sharing a pair, a 300-sigma event (guard), and templates past 5633 samples that
    are cut into 2 .. 4 segments of 4096 (virtual channels)."""
    rs = np.random.RandomState(3000 + seed + SOAK)
    Nc = int([3, 1, 2, 3, 6, 3, 4, 1, 3, 2, 3, 3, 6, 3, 1, 3, 2, 5, 7, 8, 9, 12, 5, 9][seed])
    L = int([256, 300, 700, 1100, 1500, 2100, 2600, 3000, 3700, 4200, 5000, 5633, 5634, 7681, 9000, 13000, 16384,
             256, 900, 2100, 3000, 1500, 5634, 9000][seed])
    n = Nc * L
    ks = [[8, 3], [1, 1, 1, 2], [5], [16, 1], [2, 2], [7], [1], [8] * 3, [4, 1, 1], [3], [9], [2, 5], [4], [2],
          [3, 1, 1], [4], [2], [8, 3], [1, 1, 2], [5], [2, 2], [7], [3], [1, 4]][seed]
    off = np.concatenate([[0], np.cumsum(ks)]).astype(np.int32)
    U = rs.standard_normal((off[-1], n))
    U -= U.mean(axis=1, keepdims=True) * (seed % 2)
    U /= np.linalg.norm(U, axis=1, keepdims=True)
    V = (8192 - L + 1) // 512 * 512 if L <= 5633 else 4096
    nchunk = 1 + seed % 2
    T = L + int(rs.randint(int(1.2 * V), int(2.7 * V)))
    M = Nc * T - int(rs.randint(0, Nc))
    stride = int(rs.randint(M // 2, M + 1) // Nc * Nc)
    buf = rs.standard_normal((nchunk - 1) * stride + M) * float(10 ** rs.uniform(-3, 3)) + float(rs.choice([0, 7]))
    p_ev = int(rs.randint(0, (M - n) // Nc))
    buf[p_ev * Nc: p_ev * Nc + n] += (300.0 if seed % 3 == 0 else 6.0) * np.std(buf) * np.sqrt(n) * U[off[1] - 1]
    ctx.set_engine('auto')
    ctx.upload_basis(U.astype(np.float32), off, Nc)
    ds = ctx.subspace_ds(torch.from_numpy(buf.astype(np.float32)).cuda(), M=M,
                         chunk_stride=stride, nchunk=nchunk)
    ctx.sync()
    ds = ds.cpu().numpy()
    Uf = U.astype(np.float32).astype(np.float64)
    bf = buf.astype(np.float32).astype(np.float64)
    P = (M - n) // Nc + 1
    assert ds.shape == (nchunk, len(ks), P)
    for b in range(nchunk):
        xb = bf[b * stride: b * stride + M]
        for d in range(len(ks)):
            ref = orc.mpxds_fft(xb, Uf[off[d]:off[d + 1]], Nc)
            err = float(np.max(np.abs(ds[b, d] - ref)))
            assert err <= 1e-5, (seed, b, d, Nc, L, ks[d], err)
    if seed % 3 == 0 and nchunk == 1:
        assert abs(int(np.argmax(ds[0, 0])) - p_ev) <= 0


@pytest.mark.parametrize('seed', range(14))
def test_random_xcorr(ctx, seed):
    from detex_amd import synth
    rs = np.random.RandomState(2000 + seed + SOAK)
    Nc = int(rs.choice([1, 2, 3, 6]))
    if seed >= 8:
        Nc = (NC_EXTRA + [65])[seed - 8]      # 65: past the finishing kernel's 64-channel window (xc_finish_kernel)
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
    ctx.set_engine('auto')
