"""Dirty records through the frequency-domain engine: zero-filled gaps (getdata fillZeros: detect.py:231,
construct.py:1038), single-sample glitches of 10^5 .. 10^7 noise sigmas, short clipped bursts -- alone and beside
planted events.  The reference's float64 transforms pay nothing for such data (detect.py:559-578); the float32
engine's dynamic-range guard lists the blocks they touch, splits isolated outliers off the transforms
(spike_split_kernel -> xhat_clean_kernel -> ds_fft_rescan_kernel) and hands what is still out of range to the
float32 contraction in tasks of 128 windows x a row group (ds_mfma_task_kernel).  Every window is held to 1e-5 of
the oracle's direct form, and the counters say which path ran."""
import numpy as np
import pytest

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu

SOAK = 100000 * int(__import__('os').environ.get('DETEX_TEST_SEED_BASE', 0))

from oracle import detex_oracle as orc  # noqa: E402
from detex_amd import synth  # noqa: E402


@pytest.fixture(scope='module')
def ctx():
    from detex_amd.hip import HipContext
    c = HipContext(0)
    yield c
    c.close()


def direct_at(x, U, Nc, idx):
    """oracle.mpxds_direct (detect.py:559-578 in closed form) at the listed windows only."""
    x = np.asarray(x, dtype=np.float64)
    U = np.asarray(U, dtype=np.float64)
    n = U.shape[1]
    S = U.sum(axis=1)
    out = np.empty(len(idx))
    with np.errstate(all='ignore'):
        for i, p in enumerate(idx):
            w = x[p * Nc: p * Nc + n]
            mu = w.mean()
            den = n / (n - 1.0) * np.sum((w - mu) ** 2)
            out[i] = np.sum((U @ w - S * mu) ** 2) / den if den > 0 else 0.0
    return out


def bank(rs, Nc, L, ks):
    n = Nc * L
    rows = [synth.orthonormal_basis(rs, n, k) for k in ks]
    off = np.concatenate([[0], np.cumsum(ks)]).astype(np.int32)
    return np.concatenate(rows).astype(np.float32), off


CASES = {
    # name: (Nc, L, ks, frames, what)
    'spike':        (3, 400, [8, 4, 1], 30000, dict(spikes=[(9000, 1, 1e5)])),
    'two_sizes':    (3, 400, [8, 4], 30000, dict(spikes=[(9000, 0, 1e5), (9100, 2, -1e7), (22000, 1, 3e4)])),
    'glitch3':      (3, 300, [5, 5, 5], 26000, dict(spikes=[(5000, 0, 2e5), (5001, 0, -3e5), (5002, 1, 1e5)])),
    'spike_event':  (3, 400, [8, 4], 30000, dict(spikes=[(9000, 1, 1e6)], events=[(8800, 0, 12.0), (20000, 1, 8.0)])),
    'gap':          (3, 400, [8, 4], 30000, dict(gaps=[(12000, 3000)])),
    'gap_event':    (3, 400, [8, 4], 30000, dict(gaps=[(12000, 3000)], events=[(15050, 0, 20.0), (11500, 1, 15.0)])),
    'gap_spike':    (3, 400, [8, 4], 30000, dict(gaps=[(12000, 3000)], spikes=[(15300, 2, 1e5)], events=[(15050, 0, 20.0)])),
    'clipped':      (3, 400, [8], 30000, dict(burst=(7000, 10, 4e5))),            # 10 frames x 3 channels = 30 outliers
    'long_burst':   (3, 400, [8, 4], 30000, dict(burst=(7000, 200, 4e5))),        # more than kMaxSpk: the contraction's
    'loud_onset':   (3, 400, [8, 4], 30000, dict(onset=(14000, 3000.0))),         # a large event: amplitude x 3000 for good
    'one_channel':  (1, 900, [6, 2], 40000, dict(spikes=[(17000, 0, 1e6)], gaps=[(30000, 2500)])),
    'six_channels': (6, 300, [16, 3], 20000, dict(spikes=[(4000, 5, 1e5), (15000, 0, -1e6)])),   # streamed block spectra
    'singles':      (3, 400, [1] * 24, 30000, dict(spikes=[(9000, 1, 1e5)], gaps=[(20000, 1200)])),  # two detectors per pair
    'five_channels': (5, 280, [7, 2], 18000, dict(spikes=[(6000, 3, 1e5)])),      # the generic-Nc kernels
}


def make(case, seed):
    Nc, L, ks, T, what = CASES[case]
    rs = np.random.RandomState(4000 + seed + SOAK)
    U, off = bank(rs, Nc, L, ks)
    n = Nc * L
    x = rs.standard_normal((T, Nc))
    for (t, d, a) in what.get('events', []):
        rows = U[off[d]:off[d + 1]].astype(np.float64)
        coef = rs.standard_normal(rows.shape[0])
        coef /= np.linalg.norm(coef)
        x[t:t + L] += (a * np.sqrt(n) * (coef @ rows)).reshape(L, Nc)
    if 'onset' in what:
        t, a = what['onset']
        x[t:] *= a
    for (t, glen) in what.get('gaps', []):
        x[t:t + glen] = 0.0
    for (t, c, a) in what.get('spikes', []):
        x[t, c] += a
    if 'burst' in what:
        t, blen, a = what['burst']
        x[t:t + blen] = a * np.sign(rs.standard_normal((blen, Nc)))
    return Nc, L, U, off, x.reshape(-1).astype(np.float32)


@pytest.mark.parametrize('case', sorted(CASES))
def test_dirty_block_vs_oracle(ctx, case):
    Nc, L, U, off, x = make(case, 0)
    n = Nc * L
    ctx.set_engine('fft')
    ctx.upload_basis(U, off, Nc)
    ds = ctx.subspace_ds(torch.from_numpy(x).cuda())
    cnt = ctx.repair_counts()
    ds = ds[0].cpu().numpy()
    ctx.set_engine('auto')
    P = (len(x) - n) // Nc + 1
    assert ds.shape == (len(off) - 1, P)
    what = CASES[case][4]
    # the path: isolated outliers are split off; a burst longer than the list, a gap's edge beside an event and a loud
    # onset go to the contraction in tasks
    if case != 'gap':                                      # (a gap alone lists nothing: its edge windows hold 1 / n of a full one's energy)
        assert cnt['blocks'] > 0, cnt
    if ('spikes' in what or case == 'clipped') and case != 'gap_spike':
        assert cnt['split'] > 0, cnt
    if case in ('spike', 'two_sizes', 'glitch3', 'clipped', 'six_channels', 'five_channels'):
        assert cnt['tiles'] == 0, cnt                      # nothing left for the contraction
    if case in ('long_burst', 'loud_onset', 'gap_event'):
        assert cnt['tiles'] > 0, cnt
    if case == 'long_burst':
        assert cnt['split'] == 0, cnt
    xd = x.astype(np.float64)
    Uf = U.astype(np.float64)
    for d in range(min(len(off) - 1, 4)):
        ref = orc.mpxds_direct(xd, Uf[off[d]:off[d + 1]], Nc)
        live = np.lib.stride_tricks.sliding_window_view(xd, n)[::Nc].std(axis=1) > 0
        assert np.all(ds[d][~live] == 0)
        err = np.abs(ds[d] - ref)
        scale = max(1.0, float(np.max(ref[live])))
        bad = (err > 1e-5 * scale) & live
        assert not bad.any(), (case, d, int(bad.sum()), int(np.argmax(err * live)), float(np.max(err[live])), cnt)


@pytest.mark.parametrize('seed', range(8))
def test_dirty_random(ctx, seed):
    """Random mixtures (re-drawn under DETEX_TEST_SEED_BASE): 0 .. 4 glitches of 10^4.5 .. 10^7 sigma, 0 .. 2 gaps,
    0 .. 3 events, over a few blocks; every engine the library would choose itself (auto)."""
    rs = np.random.RandomState(7000 + seed + SOAK)
    Nc = int(rs.choice([1, 3, 3, 4]))
    L = int(rs.randint(256, 1200))
    ks = [int(rs.choice([1, 2, 5, 8])) for _ in range(int(rs.randint(1, 4)))]
    T = int(rs.randint(3 * L + 9000, 3 * L + 30000))
    U, off = bank(rs, Nc, L, ks)
    n = Nc * L
    amp = float(10 ** rs.uniform(-3, 2))
    x = rs.standard_normal((T, Nc)) * amp
    for _ in range(int(rs.randint(0, 4))):
        t, d = int(rs.randint(0, T - L)), int(rs.randint(len(ks)))
        rows = U[off[d]:off[d + 1]].astype(np.float64)
        coef = rs.standard_normal(rows.shape[0])
        x[t:t + L] += (amp * rs.uniform(5, 20) * np.sqrt(n) * (coef / np.linalg.norm(coef) @ rows)).reshape(L, Nc)
    for _ in range(int(rs.randint(0, 3))):
        t = int(rs.randint(0, T - 10))
        x[t:t + int(rs.randint(1, 2 * L))] = 0.0
    for _ in range(int(rs.randint(0, 5))):
        x[int(rs.randint(T)), int(rs.randint(Nc))] += amp * 10 ** rs.uniform(4.5, 7) * rs.choice([-1, 1])
    x = x.reshape(-1).astype(np.float32)
    ctx.set_engine('auto')
    ctx.upload_basis(U, off, Nc)
    ds = ctx.subspace_ds(torch.from_numpy(x).cuda())
    cnt = ctx.repair_counts()
    ds = ds[0].cpu().numpy()
    xd, Uf = x.astype(np.float64), U.astype(np.float64)
    live = np.lib.stride_tricks.sliding_window_view(xd, n)[::Nc].std(axis=1) > 0
    for d in range(len(ks)):
        ref = orc.mpxds_direct(xd, Uf[off[d]:off[d + 1]], Nc)
        err = np.abs(ds[d] - ref)
        scale = max(1.0, float(np.max(ref[live])) if live.any() else 1.0)
        bad = (err > 1e-5 * scale) & live
        assert np.all(ds[d][~live] == 0)
        assert not bad.any(), (seed, d, Nc, L, ks, int(bad.sum()), int(np.argmax(err * live)), float(np.max(err[live])), cnt)


def test_config2_dirty_fullsize(ctx):
    """bench.py's `config2_dirty` generator at BASELINE configs[1] size (four of its 24 hourly chunks): 10 gaps of 60 s,
    10 glitches of 10^5 sigma, the planted events; the windows around every glitch, gap edge and plant plus a random
    sample against the oracle's direct form, the chunk-by-chunk result equal to scanning a chunk alone, and the
    repair path's volume (a handful of blocks, not the day)."""
    Nc, L, D, k, chunks = 3, 3000, 25, 8, 4
    n, M = Nc * L, 372000 * Nc
    U, off = synth.detector_bank(1, D, k, n)
    x, plants = synth.continuous_day(1, chunks, M, Nc, U, off, nplant=12)
    gaps, spikes = synth.make_dirty(x, Nc, 11, ngap=4, gap_frames=6000, nspike=4, spike_amp=1e5)
    ctx.set_engine('auto')
    ctx.upload_basis(U, off, Nc)
    xt = torch.from_numpy(x).cuda()
    ds = ctx.subspace_ds(xt)
    cnt = ctx.repair_counts()
    assert 0 < cnt['split'] <= 3 * len(spikes), cnt
    assert cnt['blocks'] <= 3 * (len(spikes) + 2 * len(gaps)), cnt
    P = (M - n) // Nc + 1
    rs = np.random.RandomState(5)
    Uf = U.astype(np.float64)
    for b in range(chunks):
        pts = set(int(v) for v in rs.randint(0, P, 40))
        for (bb, t, c) in spikes:
            if bb == b:
                pts.update(int(v) for v in (t - L + 1, t - L, t - L // 2, t - 1, t, t + 1, t + 5000, t - 5000))
        for (bb, t, glen) in gaps:
            if bb == b:
                for e in (t, t + glen):
                    pts.update(int(v) for v in (e - L - 1, e - L, e - L + 1, e - L + 30, e - 1, e, e + 1, e - 40, e + 40))
        for (bb, p, d) in plants:
            if bb == b:
                pts.update((p - 1, p, p + 1))
        idx = np.array(sorted(v for v in pts if 0 <= v < P))
        got = ds[b].cpu().numpy()
        xd = x[b].astype(np.float64)
        for d in sorted(set([0, D - 1] + [dd for (bb, _, dd) in plants if bb == b]))[:5]:
            ref = direct_at(xd, Uf[off[d]:off[d + 1]], Nc, idx)
            err = np.abs(got[d][idx] - ref)
            assert err.max() < 1e-5, (b, d, int(idx[np.argmax(err)]), float(err.max()), cnt)
        # ... and the batch equals the chunk scanned alone (the listings of one chunk do not leak into another)
        alone = ctx.subspace_ds(xt[b:b + 1])[0].cpu().numpy()
        assert np.array_equal(alone, got), b
    for (b, p, d) in plants:
        if any(bb == b and t - L < p < t + glen for (bb, t, glen) in gaps):
            continue                                       # (a plant a gap has eaten into)
        row = ds[b, d].cpu().numpy()
        lo = max(0, p - 50)
        assert lo + int(np.argmax(row[lo:p + 50])) == p


def test_dirty_ragged_batch_equals_chunk_by_chunk(ctx):
    """Dirty chunks of different lengths in ONE ragged launch (dtx_subspace_scan_ragged): glitches, a gap's edge beside
    an event and a clipped burst spread over three chunks -- each chunk comes out bit for bit as when it is scanned
    alone (the guard's lists, the outliers and the repair tasks of one chunk do not leak into another), and within
    1e-5 of the oracle's direct form."""
    Nc, L = 3, 400
    n = Nc * L
    rs = np.random.RandomState(911 + SOAK)
    U, off = bank(rs, Nc, L, [8, 4, 1])
    frames = [30000, 21000, 26000]
    chunks = []
    for i, T in enumerate(frames):
        x = rs.standard_normal((T, Nc))
        if i == 0:
            x[9000, 1] += 1e5
            x[20000, 2] -= 3e6
        if i == 1:
            x[8000:9500] = 0.0
            rows = U[off[0]:off[1]].astype(np.float64)
            c = rs.standard_normal(rows.shape[0])
            x[9520:9520 + L] += (18.0 * np.sqrt(n) * (c / np.linalg.norm(c) @ rows)).reshape(L, Nc)
        if i == 2:
            x[5000:5012] = 4e5 * np.sign(rs.standard_normal((12, Nc)))
        chunks.append(x.reshape(-1).astype(np.float32))
    starts, pos = [], 0
    for c in chunks:
        starts.append(pos)
        pos += len(c) + 7 * Nc                              # (a few frames of someone else's data between the chunks)
    flat = rs.standard_normal(pos).astype(np.float32) * 50.0
    for s0, c in zip(starts, chunks):
        flat[s0:s0 + len(c)] = c
    ctx.set_engine('auto')
    ctx.upload_basis(U, off, Nc)
    xt = torch.from_numpy(flat).cuda()
    tab = ctx.chunk_table(starts, [len(c) for c in chunks])
    thr = np.full(len(off) - 1, 0.3, np.float32)
    ds, res = ctx.subspace_scan(xt, thr=thr, chunk_tab=tab, cand_cap=1 << 16)
    cnt = ctx.repair_counts()
    assert cnt['split'] >= 2 and cnt['tiles'] >= 1, cnt
    dsr = ds.cpu().numpy()
    for b, c in enumerate(chunks):
        Pb = (len(c) - n) // Nc + 1
        alone = ctx.subspace_ds(torch.from_numpy(c).cuda())[0].cpu().numpy()
        assert np.array_equal(dsr[b, :, :Pb], alone), (b, float(np.abs(dsr[b, :, :Pb] - alone).max()))
        xd = c.astype(np.float64)
        live = np.lib.stride_tricks.sliding_window_view(xd, n)[::Nc].std(axis=1) > 0
        for d in range(len(off) - 1):
            ref = orc.mpxds_direct(xd, U[off[d]:off[d + 1]].astype(np.float64), Nc)
            err = np.abs(alone[d] - ref)
            assert not ((err > 1e-5 * max(1.0, float(ref[live].max()))) & live).any(), (b, d, float(err[live].max()))
