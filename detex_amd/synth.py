"""Seeded synthetic inputs of the BASELINE.json configurations.
Pure NumPy, float32 outputs; shared by bench.py and the tests."""
import numpy as np


def orthonormal_basis(rs, n, k):
    """k orthonormal, zero-mean basis vectors of length n (rows): QR of
    demeaned Gaussian columns -- the shape SubSpace.SVD produces
    (detex/subspace.py:886-895: left singular vectors of demeaned events)."""
    a = rs.standard_normal((n, k))
    a -= a.mean(axis=0, keepdims=True)
    q, _ = np.linalg.qr(a)
    return np.ascontiguousarray(q.T)


def detector_bank(seed, D, k, n):
    """D detectors of k basis vectors each -> U [D*k, n] float32, det_off [D+1]."""
    rs = np.random.RandomState(seed)
    U = np.concatenate([orthonormal_basis(rs, n, k) for _ in range(D)], axis=0)
    return U.astype(np.float32), (np.arange(D + 1) * k).astype(np.int32)


def continuous_day(seed, nchunk, M, Nc, U=None, det_off=None, nplant=50, amp=(5.0, 20.0)):
    """nchunk data blocks of M multiplexed samples: N(0,1) noise plus planted
    events (random combinations of one detector's basis, scaled to `amp` times
    the noise sigma per sample) at channel-aligned positions; planted events never
    overlap each other (a 20-sigma event on top of a 5-sigma one would bury it: every
    plant must be detectable on its own).  Returns x [nchunk, M] float32 and the list
    of (chunk, window index, detector)."""
    rs = np.random.RandomState(seed)
    x = rs.standard_normal((nchunk, M)).astype(np.float32)
    plants = []
    if U is not None and nplant > 0:
        n = U.shape[1]
        D = len(det_off) - 1
        for e in range(nplant):
            b = int(rs.randint(nchunk))
            d = int(rs.randint(D))
            rows = U[det_off[d]:det_off[d + 1]]
            coef = rs.standard_normal(rows.shape[0])
            coef /= np.linalg.norm(coef)
            a = rs.uniform(*amp) * np.sqrt(n)
            while True:
                p = int(rs.randint(0, (M - n) // Nc))
                if all(pb != b or abs(pp - p) >= 2 * (n // Nc) for (pb, pp, _) in plants):
                    break
            pos = p * Nc
            plants.append((b, p, d))
            x[b, pos:pos + n] += (a * (coef @ rows)).astype(np.float32)
    return x, plants


def event_families(seed, nfam, nper, Nc, L, maxshift=50, snr=5.0):
    """Template events for the similarity matrix: nfam families x nper events,
    each family a random band-limited wavelet with exponential envelope, events
    are scaled / shifted / noisy copies.  Returns td [N, Nc*L] float32 (multiplexed)."""
    rs = np.random.RandomState(seed)
    N = nfam * nper
    td = np.empty((N, Nc * L), dtype=np.float32)
    env = np.exp(-np.arange(L + 2 * maxshift) / (0.3 * L))
    kern = np.hanning(9)[1:-1]
    kern /= kern.sum()
    e = 0
    for f in range(nfam):
        wav = [np.convolve(rs.standard_normal(L + 2 * maxshift + 6), kern, 'valid') * env
               for _ in range(Nc)]
        for _ in range(nper):
            sh = int(rs.randint(0, 2 * maxshift + 1))
            amp = rs.uniform(0.5, 2.0)
            ch = [amp * w[sh:sh + L] + rs.standard_normal(L) * (np.std(w) / snr) for w in wav]
            td[e] = np.vstack(ch).flatten(order='F')
            e += 1
    return td


def make_dirty(x, Nc, seed, ngap=10, gap_frames=6000, nspike=10, spike_amp=1e5):
    """What real records add to `continuous_day`'s clean noise (in place, returns the lists): `ngap` zero-filled gaps
    of `gap_frames` sample frames on all channels (getdata's fillZeros: detect.py:231, construct.py:1038) and `nspike`
    single-sample glitches of `spike_amp` noise sigmas on one channel each.  x: [nchunk, M] multiplexed."""
    rs = np.random.RandomState(seed)
    nchunk, M = x.shape
    T = M // Nc
    gaps, spikes = [], []
    for _ in range(ngap):
        b, t = int(rs.randint(nchunk)), int(rs.randint(0, max(1, T - gap_frames)))
        x[b, t * Nc:(t + gap_frames) * Nc] = 0.0
        gaps.append((b, t, gap_frames))
    for _ in range(nspike):
        b, t, c = int(rs.randint(nchunk)), int(rs.randint(T)), int(rs.randint(Nc))
        x[b, t * Nc + c] += np.float32(spike_amp * (1 if rs.rand() < 0.5 else -1))
        spikes.append((b, t, c))
    return gaps, spikes
