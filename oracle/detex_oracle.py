"""CPU oracle for the Detex hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A plain NumPy (float64) restatement of the reference algorithm
(d-chambers/Detex v1.0.9, pure Python) for the two data-parallel hot paths:

  * the all-pairs normalised cross-correlation that fills the similarity / lag
    matrices (detex/construct.py:369-466), and
  * the sliding-window subspace detection statistic with its post-processing
    (detex/detect.py:220-296, 390-445, 501-578; detex/fas.py:120-134) and the
    SVD basis construction that feeds it (detex/subspace.py:886-1054).

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this module, and only as the checker / the timed CPU
baseline.  The product (``detex_amd``) never imports it and has no CPU
fallback: it fails loudly when the HIP library is missing.

Parity pin: the reference's own tests hold no numerical vectors for this path
(SURVEY.md section 4), so the oracle is pinned against outputs of the reference
itself, generated in the build container by ``tests/golden/make_golden.py``
(which imports /root/reference through a stub/shim preamble) and committed as
``tests/golden/*.npz``.  ``tests/test_oracle_golden.py`` checks every function
below against those vectors (<= 1e-9 abs in float64; integers exact).

Each function cites the reference file:line it follows.  Two forms are given
for the two kernels: ``*_fft`` follows the reference's FFT arithmetic step by
step, ``*_direct`` is the closed form that the HIP
kernels implement.
"""
import numpy as np
import scipy.fftpack
import scipy.linalg

# --------------------------------------------------------------------------
# small helpers standing in for pandas 0.17 rolling_* (third-party arithmetic)
# --------------------------------------------------------------------------


def _windows(x, n):
    return np.lib.stride_tricks.sliding_window_view(np.asarray(x), int(n))


def rolling_mean_valid(x, n):
    """pd.rolling_mean(x, n)[n-1:] -- mean of every full window."""
    x = np.asarray(x, dtype=np.float64)
    if len(x) * int(n) <= 5_000_000:
        return _windows(x, n).mean(axis=1)
    cs = np.concatenate([[0.0], np.cumsum(x.astype(np.longdouble))])
    return np.asarray((cs[n:] - cs[:-n]) / n, dtype=np.float64)


def rolling_var_valid(x, n, ddof=1):
    """pd.rolling_var(x, n)[n-1:] (ddof=1 as in pandas 0.17)."""
    x = np.asarray(x, dtype=np.float64)
    if len(x) * int(n) <= 5_000_000:
        return _windows(x, n).var(axis=1, ddof=ddof)
    xl = x.astype(np.longdouble)
    c1 = np.concatenate([[0.0], np.cumsum(xl)])
    c2 = np.concatenate([[0.0], np.cumsum(xl * xl)])
    s1 = c1[n:] - c1[:-n]
    s2 = c2[n:] - c2[:-n]
    out = (s2 - s1 * s1 / n) / (n - ddof)
    return np.asarray(out, dtype=np.float64)


def rolling_mean_centered(x, n):
    """pd.rolling_mean(x, n, center=True): NaN where the window is not full.

    pandas centres a window of n on index i as [i-(n-1)//2 ... ] shifted so the
    trailing-window result at index j is placed at j - (n-1)//2.
    """
    x = np.asarray(x, dtype=np.float64)
    n = int(n)
    out = np.full(len(x), np.nan)
    if n > len(x) or n < 1:
        return out
    valid = rolling_mean_valid(x, n)          # trailing windows ending at n-1..
    off = (n - 1) // 2
    # trailing result at index j (j >= n-1) moves to j - off
    out[n - 1 - off: n - 1 - off + len(valid)] = valid
    return out


# --------------------------------------------------------------------------
# multiplexing / spectra                                     construct.py
# --------------------------------------------------------------------------

def multiplex(chans, trim=True):
    """construct.py:928-987 -- interleave Nc channel arrays sample-wise
    (vstack(chans).flatten(order='F'), lines 977-978), trimming every channel
    to the shortest (970-974).  ``chans`` must already be in st.sort() order."""
    chans = [np.asarray(c) for c in chans]
    L = min(len(c) for c in chans)
    return np.vstack([c[:L] for c in chans]).flatten(order='F')


def fft_len_events(n):
    """construct.py:669-676 -- nfft = 2 ** (2*len(mp)).bit_length()."""
    return 2 ** int(2 * n).bit_length()


def get_freq_domain(mp):
    """construct.py:669-676."""
    return scipy.fftpack.fft(mp, n=fft_len_events(len(mp)))


# --------------------------------------------------------------------------
# pairwise normalised cross-correlation                      construct.py
# --------------------------------------------------------------------------

def subsamp(ceval, ind):
    """construct.py:397-422 -- cosine-fit sub-sample shift (Cespedes 1995).
    Returns ``ind`` itself when |tau| > .5 (reference quirk, 418-421)."""
    if ind == 0 or ind == len(ceval) - 1:
        return 0.0
    cb4 = ceval[ind - 1]
    caf = ceval[ind + 1]
    cn = ceval[ind]
    with np.errstate(all='ignore'):
        alpha = np.arccos((cb4 + caf) / (2 * cn))
        alsi = np.sin(alpha)
        tau = -(np.arctan((cb4 - caf) / (2 * cn * alsi)) / alpha)
    if abs(tau) > .5:
        return ind
    return tau


def ccx2_reduce(raw, n, Nc, trunc, scrub_tol=0.0):
    """construct.py:453-466 -- truncate, scrub, max/argmax, sub-sample, lag.

    ``raw`` is the channel-aligned CC vector ([Nc-1::Nc], 2L-1 lags).
    ``scrub_tol`` = 0 is the reference rule (> 1 or < -1 -> 0, line 461); the
    HIP path uses 1e-5 so that a correlation of 1 + one fp32 ulp (identical
    events) is not zeroed -- the reference itself zeroes or keeps such a value
    depending on float64 rounding luck, see DESIGN.md "knife-edge scrub".
    Returns (maxcc, sampleLag, subsamp, scrubbed_vector)."""
    result = np.array(raw[trunc: -trunc], dtype=np.float64)
    hi = 1.0 + scrub_tol
    try:
        with np.errstate(all='ignore'):
            maxcc = np.nanmax(result)
            mincc = np.nanmin(result)
            maxind = np.nanargmax(result)
            if maxcc > hi or mincc < -hi:
                result[(result > hi) | (result < -hi)] = 0
                maxcc = np.nanmax(result)
                maxind = np.nanargmax(result)
    except ValueError:
        return 0.0, 0.0, 0.0, result
    ss = subsamp(result, maxind)
    return maxcc, (maxind + 1 + trunc) * Nc - n, ss, result


def ccx2_raw_fft(x1, x2, Nc, fd1=None, fd2=None):
    """construct.py:438-452 step by step (FFT form): the channel-aligned CC
    vector before truncation / scrubbing."""
    x1 = np.asarray(x1, dtype=np.float64)
    x2 = np.asarray(x2, dtype=np.float64)
    n = len(x1)
    if fd1 is None:
        fd1 = get_freq_domain(x1)
    if fd2 is None:
        fd2 = get_freq_domain(x2)
    pad = np.pad(x2, (n - 1, n - 1), 'constant')                 # 443-445
    a = rolling_mean_valid(pad, n)                               # 446
    with np.errstate(all='ignore'):
        b = np.sqrt(rolling_var_valid(pad, n, ddof=1))           # 447
        b = b * np.sqrt((n - 1.0) / n)                           # 448
        c = np.real(scipy.fftpack.ifft(np.conj(fd1) * fd2))      # 449
        c1 = np.concatenate([c[-(n - 1):], c[:n]])               # 450
        return ((c1 - x1.sum() * a) / (n * b * np.std(x1)))[Nc - 1::Nc]  # 452


def ccx2_raw_direct(x1, x2, Nc):
    """Closed form of construct.py:438-452.

    For sample lag s in [-(L-1), L-1], k = s*Nc: w_s[j] = x2[j+k] (0 outside),
    cc(s) = (x1.w_s - sum(x1) mean(w_s)) / (n std_pop(w_s) std_pop(x1)).
    This is the arithmetic the HIP kernel performs."""
    x1 = np.asarray(x1, dtype=np.float64)
    x2 = np.asarray(x2, dtype=np.float64)
    n = len(x1)
    L = n // Nc
    pad = np.pad(x2, (n - 1, n - 1), 'constant')
    # element i of the reference's (pre-slice) vector is the window starting at
    # pad[i]; [Nc-1::Nc] keeps i = Nc-1 + Nc*m, i.e. lag (m+1)*Nc - n
    idx = np.arange(Nc - 1, 2 * n - 1, Nc)
    sx1 = x1.sum()
    sd1 = np.std(x1)
    res = np.empty(len(idx))
    with np.errstate(all='ignore'):
        for o, i in enumerate(idx):
            w = pad[i: i + n]
            sd = w.std()
            if sd == 0:
                # zero-variance (all-zero / constant) window: the reference
                # divides FFT round-off by ~0 and gets +-inf, which its scrub
                # (construct.py:458-462) turns into 0.  State that as +inf.
                # (All-zero x2, or zero-variance x1, gives exact 0/0 = NaN in the
                # reference too -> all-NaN -> its ValueError path -> (0, 0, 0).)
                res[o] = np.inf if (x2.any() and sd1 > 0) else np.nan
            else:
                res[o] = (np.dot(x1, w) - sx1 * w.mean()) / (n * sd * sd1)
    assert len(res) == 2 * L - 1
    return res


def ccx2_fft(x1, x2, Nc, fd1=None, fd2=None, scrub_tol=0.0):
    """construct.py:425-466 (FFT form) -> (maxcc, sampleLag, subsamp, vector)."""
    n = len(x1)
    return ccx2_reduce(ccx2_raw_fft(x1, x2, Nc, fd1, fd2), n, Nc,
                       n // (2 * Nc) - 1, scrub_tol)             # 439


def ccx2_direct(x1, x2, Nc, scrub_tol=0.0):
    """construct.py:425-466 (closed form) -> (maxcc, sampleLag, subsamp, vector)."""
    n = len(x1)
    return ccx2_reduce(ccx2_raw_direct(x1, x2, Nc), n, Nc,
                       n // (2 * Nc) - 1, scrub_tol)


def make_cclags(events_td, Nc, form='fft'):
    """construct.py:369-394 -- all pairs b < c.  Returns three (N-1)x(N-1)
    float arrays (index 0..N-2, columns 1..N-1 -> array column c-1), NaN below
    the diagonal, exactly the layout of DFcc / DFlag / DFsubsamp."""
    N = len(events_td)
    cc = np.full((N - 1, N - 1), np.nan)
    lag = np.full((N - 1, N - 1), np.nan)
    ss = np.full((N - 1, N - 1), np.nan)
    fds = [get_freq_domain(np.asarray(e, dtype=np.float64)) for e in events_td] \
        if form == 'fft' else None
    for b in range(N - 1):
        for c in range(b + 1, N):
            if form == 'fft':
                m, l, s, _ = ccx2_fft(events_td[b], events_td[c], Nc,
                                      fds[b], fds[c])
            else:
                m, l, s, _ = ccx2_direct(events_td[b], events_td[c], Nc)
            cc[b, c - 1] = m
            lag[b, c - 1] = l
            ss[b, c - 1] = s
    return cc, lag, ss


def condensed_dissimilarity(cc):
    """construct.py:153-157, 604-612 -- 1.0000001 - cc flattened row-major with
    the NaNs (below-diagonal cells) dropped = SciPy condensed order."""
    cx = (1.0000001 - np.asarray(cc, dtype=np.float64)).flatten()
    return cx[~np.isnan(cx)]


def fast_normcorr(t, s):
    """construct.py:469-483."""
    t = np.asarray(t, dtype=np.float64)
    s = np.asarray(s, dtype=np.float64)
    if len(t) > len(s):
        t, s = s, t
    n = len(t)
    nt = (t - np.mean(t)) / (np.std(t) * n)
    sum_nt = nt.sum()
    a = rolling_mean_valid(s, n)
    b = np.sqrt(rolling_var_valid(s, n, ddof=1)) * np.sqrt((n - 1.0) / n)
    c = np.convolve(nt[::-1], s, mode='valid')
    return (c - sum_nt * a) / b


# --------------------------------------------------------------------------
# subspace detection statistic                      detect.py / fas.py
# --------------------------------------------------------------------------

def ds_fft_len(data_len_samples, n):
    """detect.py:367-370 / fas.py:165-167 -- reqlen and its power of two."""
    reqlen = int(data_len_samples + n)
    return reqlen, 2 ** reqlen.bit_length()


def mpxds_fft(x, U, Nc, rbi=None, xFD=None, ssFD=None):
    """detect.py:559-578 == fas.py:120-134 (FFT form, step by step).

    x : multiplexed data block (M,), U : basis (K, n).  Returns DS of length
    (M - n)//Nc + 1.  ``xFD`` / ``ssFD`` may carry the precomputed spectra the
    reference passes in (detect.py:255-256, 367-371)."""
    x = np.asarray(x, dtype=np.float64)
    U = np.atleast_2d(np.asarray(U, dtype=np.float64))
    n = U.shape[1]
    if rbi is None:
        rbi = 2 ** int(len(x) + n).bit_length()
    a = rolling_mean_valid(x, n)                                 # 566
    b = rolling_var_valid(x, n, ddof=1) * n                      # 567-568
    sum_ss = U.sum(axis=1)                                       # 569
    av_norm = a.reshape(1, -1) * sum_ss.reshape(-1, 1)           # 570-572
    if ssFD is None:
        ssFD = np.array([scipy.fftpack.fft(u[::-1], n=rbi) for u in U])
    if xFD is None:
        xFD = scipy.fftpack.fft(x, n=rbi)
    m1 = ssFD * xFD                                              # 573
    if1 = np.real(scipy.fftpack.ifft(m1))[:, n - 1:len(x)] - av_norm  # 575
    with np.errstate(all='ignore'):
        result = np.sum(np.square(if1), axis=0) / b              # 576
    return result[::Nc]                                          # 577


def mpxds_direct(x, U, Nc):
    """Closed form of detect.py:559-578:
    DS[p] = sum_k (U_k.w - S_k mu)^2 / (n/(n-1) sum_q (w_q - mu)^2),
    w = x[p*Nc : p*Nc+n].  This is the arithmetic the HIP kernel performs."""
    x = np.asarray(x, dtype=np.float64)
    U = np.atleast_2d(np.asarray(U, dtype=np.float64))
    K, n = U.shape
    P = (len(x) - n) // Nc + 1
    S = U.sum(axis=1)
    out = np.empty(P)
    with np.errstate(all='ignore'):
        for p in range(P):
            w = x[p * Nc: p * Nc + n]
            mu = w.mean()
            den = n / (n - 1.0) * np.sum((w - mu) ** 2)
            out[p] = np.sum((U @ w - S * mu) ** 2) / den
    return out


def scrub_ds(ssd):
    """detect.py:275-281 -- MaxDS = ssd.max(); if MaxDS > 1.1 zero the infs and
    recompute.  Returns (ssd, MaxDS); mutates a copy."""
    ssd = np.array(ssd, dtype=np.float64)
    maxds = ssd.max()
    if maxds > 1.1:
        ssd[np.isinf(ssd)] = 0
        maxds = ssd.max()
    return ssd, maxds


HIST_BINS = np.linspace(0, 1, num=401)          # detect.py:80
FAS_BINS = np.linspace(-.01, 1, num=401)        # fas.py:31


def ds_histogram(ds, bins=HIST_BINS):
    """detect.py:178-181 -- np.histogram(DS, bins)."""
    return np.histogram(np.asarray(ds, dtype=np.float64), bins=bins)[0]


def replace_nan_with_mean(arg):
    """detect.py:516-524 (note arg[first + 1], reference quirk)."""
    ind = np.where(~np.isnan(arg))[0]
    first, last = ind[0], ind[-1]
    arg[:first] = arg[first + 1]
    arg[last + 1:] = arg[last]
    return arg


def get_stalta(C, LTA, STA):
    """detect.py:501-514."""
    C = np.asarray(C, dtype=np.float64)
    if STA == 0:
        STA = 1
        st = np.abs(C)
    else:
        st = replace_nan_with_mean(rolling_mean_centered(np.abs(C), STA))
    lt = replace_nan_with_mean(rolling_mean_centered(np.abs(C), LTA))
    with np.errstate(all='ignore'):
        return np.divide(st, lt)


def downplay_around_max(C, sr, dpv=0, buff=20):
    """detect.py:545-557 -- zero +-buff s around the max, in place."""
    index = C.argmax()
    if index < buff * sr + 1:
        C[0:int(index + buff * sr)] = dpv
    elif index > len(C) - buff * sr:
        C[int(index - sr * buff):] = dpv
    else:
        C[int(index - sr * buff):int(sr * buff + index)] = dpv
    return C


def eval_trig_con(maxds, threshold):
    """detect.py:526-543, trigCon == 0: strict '>'."""
    return bool(maxds > threshold)


def pick_triggers(ds, threshold, sr, start, offsets=(0.0,), stalta=None,
                  maxcount=4000):
    """detect.py:390-445 -- trigger-picking loop (trigCon == 0, magnitudes off).

    Returns a list of dicts with the reference's columns that do not depend on
    magnitude estimation: index, DS, DS_STALTA, STMP, MSTAMPmin, MSTAMPmax."""
    ds = np.asarray(ds, dtype=np.float64)
    ceval = ds.copy()
    out = []
    count = 0
    minof, maxof = np.min(offsets), np.max(offsets)
    while ceval.max() >= threshold:                              # 410
        trig = int(ceval.argmax())                               # 411
        coef = ds[trig]
        times = float(trig) / sr + start                         # 413
        sl = 0.0 if stalta is None else stalta[trig]
        ceval = downplay_around_max(ceval, sr, 0)                # 421
        if count > maxcount:                                     # 433
            raise RuntimeError('over 4000 events found in single data block')
        out.append(dict(index=trig, DS=coef, DS_STALTA=sl, STMP=times,
                        MSTAMPmin=times - maxof, MSTAMPmax=times - minof))
        count += 1
    return out


# --------------------------------------------------------------------------
# SVD basis construction                                     subspace.py
# --------------------------------------------------------------------------

def trim_groups(aligned, start=None, end=None):
    """subspace.py:921-943 -- rows = aligned[start:end] - mean."""
    rows = []
    for w in aligned:
        w = np.asarray(w, dtype=np.float64)
        if start is not None:
            s = max(int(start), 0)
            w = w[s:int(end)]
        rows.append(w - np.mean(w))
    return np.vstack(rows)


def svd_basis(arr, normalize=False):
    """subspace.py:886-895 -- U, s of svd(arr.T, full_matrices=False)."""
    arr = np.asarray(arr, dtype=np.float64)
    if normalize:
        arr = np.array([x / np.linalg.norm(x) for x in arr])
    U, s, Vh = scipy.linalg.svd(arr.T, full_matrices=False)
    return U, s


def frac_energy(U, waveforms):
    """subspace.py:968-997 -- cumulative fractional energy capture per event
    (waveforms NOT demeaned here, 979-988), plus Average / Minimum."""
    fr = []
    for w in waveforms:
        w = np.asarray(w, dtype=np.float64)
        nu = np.dot(U.T, w) / scipy.linalg.norm(w)
        rep = np.insert(np.square(nu), 0, 0)
        fr.append([np.sum(rep[:x + 1]) for x in range(len(rep))])
    fr = np.array(fr)
    return fr, np.average(fr, axis=0), np.min(fr, axis=0)


def used_basis_count(average, selectCriteria, selectValue):
    """subspace.py:999-1013 -- number of basis vectors kept."""
    if selectCriteria in (1, 2, 3):
        av = np.array(average, dtype=np.float64)
        av[-1] = 1.00
        return int(np.argmax(av >= selectValue))
    if selectCriteria == 4:             # keys[:selectValue + 1]: at most all of them
        return min(int(selectValue) + 1, len(average) - 1)
    raise ValueError('selectCriteria of %s is not supported' % selectCriteria)


def threshold_criteria3(minimum, numbasis, selectValue):
    """subspace.py:1049-1054."""
    return minimum[numbasis] * selectValue


# --------------------------------------------------------------------------
# alignment through the dendrogram (Harris 2006 App. B)   construct.py:710-812
# --------------------------------------------------------------------------

def _triangular(n):
    """construct.py:790-795."""
    return n * (n + 1) // 2


def get_delays(link, cx, lags):
    """construct.py:710-760 (_getDelays, _traceEventDendro, _updateLags, _getDow,
    _getAcr, _getClustDict) for one cluster of N + 1 events.

    link : linkage(cx); cx : condensed dissimilarities (unique values); lags :
    condensed pair lags.  Returns the accumulated lag of every event (lagSeries)."""
    N = len(link)                                            # merges (events - 1)
    lags = np.array(lags, dtype=np.float64)
    cx = np.asarray(cx, dtype=np.float64)
    clust = {i: [i] for i in range(N + 1)}                   # 800-812
    for a in range(N):
        clust[N + 1 + a] = clust[int(link[a, 0])] + clust[int(link[a, 1])]
    iu = np.triu_indices(N + 1, 1)
    total = np.zeros(N + 1, dtype=np.int64)
    for a in range(N):
        height = link[a, 2]
        k = int(np.flatnonzero(cx == height)[0])             # 727-729: cxdf[cxdf == row.cc]
        ev1 = int(iu[0][k])                                  # tempdf.index[0]
        cl22 = clust[int(link[a, 1])] if ev1 in clust[int(link[a, 0])] else clust[int(link[a, 0])]
        currentLag = int(np.round(lags[k]))                  # 752: CCtoLag[cc]
        for b in cl22:                                       # 754-756
            total[b] += currentLag
            nn = N                                           # _updateLags(b, lags, len(dflink), .)
            acr = [_triangular(nn) - _triangular(nn - b) + q for q in range(nn - b)]       # 782-788
            dow = [_triangular(nn - 1) - 1 + b - _triangular(nn - 1 - q) for q in range(b)]  # 773-779
            for q in acr:
                lags[q] += currentLag
            for q in dow:
                lags[q] -= currentLag
    return total


def align_td(sample_delays, events_td):
    """construct.py:486-504: events sliced by their (non-negative) sample delays and cut to
    the common length."""
    sample_delays = [int(d) for d in sample_delays]
    TDlengths = len(events_td[0]) - max(sample_delays)
    return [np.asarray(e)[d:][:TDlengths] for e, d in zip(events_td, sample_delays)]


# --------------------------------------------------------------------------
# magnitude estimates at a trigger                 detect.py:447-499, 637-664
# --------------------------------------------------------------------------

def rolling_std_valid(x, n):
    """pd.rolling_std(x, n)[n-1:] (ddof=1)."""
    return np.sqrt(rolling_var_valid(x, n, ddof=1))


def est_mag(trigIndex, Nc, MPcon, mags, U, ewf, issubspace=True):
    """detect.py:447-499 literally, including the n x n UtU of detect.py:367 and
    WFU = WFs.UtU of 381.  Returns (projectedEnergyMag, stdMag, SNR)."""
    MPcon = np.asarray(MPcon, dtype=np.float64)
    ewf = np.asarray(ewf, dtype=np.float64)
    mags = np.asarray(mags, dtype=np.float64)
    if issubspace:
        U = np.asarray(U, dtype=np.float64)
        UtU = np.dot(U.T, U)
        WFU = np.dot(ewf, UtU)
    else:
        WFU = ewf
    WFlen = np.shape(WFU)[1]
    ConDat = MPcon[trigIndex * Nc: trigIndex * Nc + WFlen]
    if issubspace:
        ssCon = np.dot(UtU, ConDat)
        proEn = np.var(ssCon) / np.var(WFU, axis=1)
    if trigIndex * Nc > 5 * WFlen:
        pe = MPcon[trigIndex * Nc - 5 * WFlen: trigIndex * Nc]
    else:
        pe = MPcon[trigIndex * Nc: trigIndex * Nc + WFlen + 6 * WFlen]
    baseNoise = np.median(rolling_std_valid(pe, WFlen))
    SNR = np.std(ConDat) / baseNoise
    touse = mags > -15
    if issubspace:
        ecor = np.array([fast_normcorr(x, ConDat)[0] for x in ewf])
        weDenom = np.sum(np.square(ecor[touse]))
        pem = stm = 0.0
        for x in range(len(ewf)):
            if mags[x] > -15:
                we = np.square(ecor[x])
                pem += (mags[x] + np.log10(np.sqrt(proEn[x]))) * we
                stm += (mags[x] + np.log10(np.std(ConDat) / np.std(ewf[x]))) * we
        return pem / weDenom, stm / weDenom, SNR
    if np.isnan(mags[0]) or mags[0] < -15:
        return np.nan, np.nan, SNR
    d1 = np.dot(ConDat, WFU[0])
    d2 = np.dot(WFU[0], WFU[0])
    return mags[0] + d1 / d2, mags[0] + np.log10(np.std(ConDat) / np.std(WFU[0])), SNR


def trigger_stats_row(trigIndex, Nc, MPcon, U, ewf, ds=None, sta_lo=-1, sta_len=1, lta_lo=-1,
                      lta_len=1, literal_utu=True):
    """What dtx_trigger_stats writes for one trigger, in NumPy float64 (the checker of that
    kernel): means of |DS| over the STA / LTA windows, mean / std of the triggering window,
    var(UtU . ConDat) with the literal n x n UtU of detect.py:367, the median rolling std of
    the pre-event noise (detect.py:466-472) and the dot products with the training events.
    literal_utu=False forms U^T (U ConDat) instead (the same product; for n where n x n does not fit)."""
    MPcon = np.asarray(MPcon, dtype=np.float64)
    U = np.atleast_2d(np.asarray(U, dtype=np.float64))
    ewf = np.atleast_2d(np.asarray(ewf, dtype=np.float64))
    n = U.shape[1]
    a = trigIndex * Nc
    ConDat = MPcon[a:a + n]
    ssCon = np.dot(np.dot(U.T, U), ConDat) if literal_utu else np.dot(U.T, np.dot(U, ConDat))
    if a > 5 * n:
        pe = MPcon[a - 5 * n:a]
    else:
        pe = MPcon[a:a + n + 6 * n]
    rs = rolling_std_valid(pe, n) if len(pe) >= n else np.array([np.nan])
    st = np.mean(np.abs(ds[sta_lo:sta_lo + sta_len])) if (ds is not None and sta_lo >= 0) else np.nan
    lt = np.mean(np.abs(ds[lta_lo:lta_lo + lta_len])) if (ds is not None and lta_lo >= 0) else np.nan
    return np.concatenate([[st, lt, ConDat.mean(), ConDat.std(), np.var(ssCon), np.median(rs)],
                           ewf @ ConDat])


# --------------------------------------------------------------------------
# pre-processing                                   construct.py:990-1030
# --------------------------------------------------------------------------

def merge_fill(frags, Nc, T):
    """Stream.merge(fill_value=0.0) of construct.py:1038 / 1070 on arrays: fragments
    (channel, first sample, samples) placed in zeroed [Nc, T] planes (third-party ObsPy
    behaviour, restated for non-overlapping fragments)."""
    planes = np.zeros((Nc, T))
    for (c, t0, x) in frags:
        planes[c, t0:t0 + len(x)] = np.asarray(x, dtype=np.float64)
    return planes


def merge_fill_overlaps(frags, Nc, T):
    """The same merge with OVERLAPPING traces, as ObsPy 1.0.2 resolves them (Stream.merge(method=0,
    fill_value=0) -> Trace.__add__; third-party behaviour restated, ObsPy is not installable here): per
    channel the traces in (start, end) order are added to a running trace; an overlap (partial, or a contained
    trace) whose samples are all identical is kept once, otherwise the whole overlap is set to the fill value
    0 -- and stays data for later additions.  Dense and sample-wise on purpose: the checker of
    detex_amd.getdata.resolve_overlaps, which works on fragment lists."""
    planes = np.zeros((Nc, T))
    for c in range(Nc):
        trs = sorted([(int(t0), np.asarray(x, dtype=np.float64)) for (cc, t0, x) in frags if cc == c and len(x)],
                     key=lambda f: (f[0], f[0] + len(f[1])))
        end = None                                       # one past the last sample of the running trace
        for t0, x in trs:
            e = t0 + len(x)
            if end is None or t0 >= end:                 # first trace, gap (zeros between) or exact fit
                planes[c, t0:e] = x
                end = e
                continue
            lo, hi = t0, min(e, end)                     # the overlap with the running trace
            if not np.array_equal(planes[c, lo:hi], x[:hi - lo]):
                planes[c, lo:hi] = 0.0
            if e > end:
                planes[c, end:e] = x[end - t0:]
                end = e
    return planes


def merge_channels_run(planes):
    """_mergeChannels (construct.py:1033-1066) on the merged planes: the product of the channels
    (1047-1049) masked where it is zero (1050), split into its contiguous unmasked pieces (1056),
    the longest kept, the first among equals (1059-1063).  Returns (first sample, length),
    (-1, 0) if everything is masked."""
    ar = np.prod(np.asarray(planes, dtype=np.float64), axis=0)
    ok = np.concatenate([[False], ar != 0.0, [False]])
    edges = np.flatnonzero(ok[1:] != ok[:-1])
    starts, stops = edges[0::2], edges[1::2]
    if len(starts) == 0:
        return -1, 0
    k = int(np.argmax(stops - starts))                # argmax: first of the longest
    return int(starts[k]), int(stops[k] - starts[k])


def chunk_span(frags, Nc, s0, cnt):
    """Front half of _applyFilter (construct.py:1003-1023) for the chunk [s0, s0 + cnt) of a record
    that arrives as fragments (channel, first sample, samples), one ObsPy trace each: returns
    (first sample, length) of what is filtered and scanned, or None if the chunk is dropped.

    * ``len(st) > len(nc)`` (1005) -- some channel has several traces in the chunk: _mergeChannels
      (1033-1066) on the zero-filled merge (overlapping traces: merge_fill_overlaps), longest run without
      a zero in any channel; if the kept stretch still holds several of the ORIGINAL traces of a channel:
      'too fractured' (1009-1014) -> None; no zero inside the common span: the merged stream, whole.
    * otherwise the traces are NOT merged and zeros are ordinary samples: only the trim to
      [max start, min end] of 1019-1023.
    * a channel without any trace in the chunk: None (the build drops it; the reference would
      continue with fewer channels)."""
    per = {}
    for (c, t0, x) in frags:
        lo, hi = max(int(t0), s0), min(int(t0) + len(x), s0 + cnt)
        if hi > lo:
            per.setdefault(c, []).append((lo, hi))
    if len(per) < Nc:
        return None
    if all(len(v) == 1 for v in per.values()):
        a = max(v[0][0] for v in per.values())
        b = min(v[0][1] for v in per.values())
        return (a, b - a) if b > a else None
    merged = merge_fill_overlaps([(c, t0 - s0, x) for (c, t0, x) in
                         [(c, max(t0, s0), np.asarray(x)[max(t0, s0) - t0:min(t0 + len(x), s0 + cnt) - t0])
                          for (c, t0, x) in frags if min(t0 + len(x), s0 + cnt) > max(t0, s0)]], Nc, cnt)
    r0, rl = merge_channels_run(merged)
    if rl <= 0:
        return None
    # zeros are looked for inside the span all merged channels cover (construct.py:1039-1050); without one the
    # merged stream goes on whole (1065-1066) and is trimmed to that span by 1019-1023
    A = max(min(lo for lo, _ in v) for v in per.values())
    B = min(max(hi for _, hi in v) for v in per.values())
    if (s0 + r0, rl) != (A, B - A):
        a, b = s0 + r0, s0 + r0 + rl
        pieces = {}
        for (c, t0, x) in frags:
            if t0 < b and t0 + len(x) > a:
                pieces[c] = pieces.get(c, 0) + 1
        if max(pieces.values()) > 1:
            return None
    return s0 + r0, rl


def bandpass_sos(freqmin, freqmax, df, corners):
    """obspy.signal.filter.bandpass design (ObsPy 1.0.2, third-party, absent here;
    restated from its published algorithm): Butterworth band-pass as zpk ->
    zpk2sos."""
    from scipy.signal import iirfilter, zpk2sos
    fe = 0.5 * df
    z, p, k = iirfilter(corners, [freqmin / fe, freqmax / fe], btype='band', ftype='butter',
                        output='zpk')
    return zpk2sos(z, p, k)


def apply_filter(chans, filt, df, detrend=True):
    """The numerical part of construct.py:1024-1028 on one stream: per channel
    st.detrend('linear') (ObsPy delegates to scipy.signal.detrend) and
    st.filter('bandpass', freqmin, freqmax, corners, zerophase) (sosfilt forward
    and, for zerophase, again over the reversed first pass), float64.
    PARITY UNPINNED against ObsPy itself (not installable here); pinned against
    the SciPy routines ObsPy calls."""
    from scipy.signal import detrend as sp_detrend, sosfilt
    out = []
    for c in chans:
        y = np.asarray(c, dtype=np.float64)
        if detrend:
            y = sp_detrend(y, type='linear')
        if filt is not None:
            sos = bandpass_sos(filt[0], filt[1], df, filt[2])
            if filt[3]:
                first = sosfilt(sos, y)
                y = sosfilt(sos, first[::-1])[::-1]
            else:
                y = sosfilt(sos, y)
        out.append(y)
    return out


def decimate(chans, factor, df, maxorder=12):
    """construct.py:1016-1017 ``st.decimate(decimate)`` on one stream: ObsPy 1.0.2
    Trace.decimate (third-party, absent here; restated from its published
    algorithm) = filter('lowpass_cheby_2', freq=df/2/factor, maxorder=12), a causal
    sosfilt, then ``data[::factor]``.  lowpass_cheby_2: stop band at ``freq`` with
    96 dB attenuation, 1 dB ripple, pass-band edge lowered by 1 % steps until
    cheb2ord gives an order <= maxorder.  PARITY UNPINNED against ObsPy itself."""
    from scipy.signal import cheb2ord, cheby2, zpk2sos, sosfilt
    nyquist = df * 0.5
    ws = min((df * 0.5 / float(factor)) / nyquist, 1.0)
    wp, order = ws, 1e99
    while order > maxorder:
        wp *= 0.99
        order, wn = cheb2ord(wp, ws, 1, 96, analog=0)
    z, p, k = cheby2(order, 96, wn, btype='low', analog=0, output='zpk')
    sos = zpk2sos(z, p, k)
    return [sosfilt(sos, np.asarray(c, dtype=np.float64))[::factor] for c in chans]
