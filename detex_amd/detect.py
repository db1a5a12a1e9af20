"""Host side of the detection scan: the reference's trigger-picking loop
(detex/detect.py:390-445, 526-557) run on the sparse list of above-threshold
samples the GPU returns instead of on the full detection-statistic vector."""
import numpy as np


def pick_triggers_sparse(idx, val, P, threshold, sr, maxds, buff=20, maxcount=4000):
    """Exact restatement of ``_evalTrigCon`` + ``_CreateCoeffArray`` +
    ``_downPlayArrayAroundMax`` (trigCon == 0) on sparse candidates.

    idx / val : window indices (ascending) and DS values of every sample with
    DS >= threshold in one data block of P windows; maxds : DS.max() of the block.
    Returns [(trigIndex, DS), ...] in the order the reference finds them.

    Only samples >= threshold can ever be the arg-max while the loop runs
    (detect.py:410), and zeroing a range only removes candidates, so the sparse
    loop visits exactly the indices the dense loop does; ``argmax`` picks the
    first occurrence of the maximum in both (idx is ascending)."""
    out = []
    if not (maxds > threshold):                      # detect.py:534, strict '>'
        return out
    idx = np.asarray(idx, dtype=np.int64)
    val = np.array(val, dtype=np.float64)
    alive = np.ones(len(idx), dtype=bool)
    w = buff * sr
    count = 0
    while alive.any():
        cur = np.where(alive, val, -np.inf)
        j = int(np.argmax(cur))                      # first occurrence of the max
        if not (cur[j] >= threshold):                # detect.py:410
            break
        index = int(idx[j])
        out.append((index, float(val[j])))
        # detect.py:551-556
        if index < w + 1:
            lo, hi = 0, int(index + w)
        elif index > P - w:
            lo, hi = int(index - w), P
        else:
            lo, hi = int(index - w), int(w + index)
        alive &= ~((idx >= lo) & (idx < hi))
        if count > maxcount:                         # detect.py:433-436
            raise RuntimeError('over 4000 events found in single data block')
        count += 1
    return out


# --------------------------------------------------------------------------
# STA/LTA of the detection statistic at a trigger       detect.py:501-524
# --------------------------------------------------------------------------

def _centered_mean_at(fetch, i, n, P):
    """Value at index i of ``replace_nan_with_mean(pd.rolling_mean(|C|, n,
    center=True))`` (detect.py:509-511, 516-524) without forming the whole
    vector: ``fetch(lo, hi)`` returns |C|[lo:hi]."""
    n = int(n)
    if n < 1 or n > P:
        raise IndexError('window longer than the detection statistic')   # reference: except -> 0.0
    off = (n - 1) // 2
    first = n - 1 - off                 # first index with a full window
    last = first + (P - n)
    if i < first:                       # arg[:first] = arg[first + 1]   (reference quirk)
        i = first + 1
        if i > P - 1:
            raise IndexError('arg[first + 1] is out of range')
        if i > last:
            return float('nan')         # arg[first + 1] is itself a NaN of the rolling mean
    elif i > last:                      # arg[last + 1:] = arg[last]
        i = last
    lo = i - first
    return float(np.mean(fetch(lo, lo + n)))


def stalta_at(fetch, i, P, LTA, STA):
    """detect.py:501-514 evaluated at one index."""
    if STA == 0:
        st = float(fetch(i, i + 1)[0])
    else:
        st = _centered_mean_at(fetch, i, STA, P)
    lt = _centered_mean_at(fetch, i, LTA, P)
    with np.errstate(all='ignore'):
        return float(np.divide(st, lt))


# --------------------------------------------------------------------------
# one station: chunks -> GPU detection statistic -> triggers   detect.py:137-218
# --------------------------------------------------------------------------

def _chunk_batches(bounds):
    """Group consecutive chunks of equal length and equal spacing so that each
    group is one dtx_subspace_ds launch with a chunk_stride."""
    batches = []
    i = 0
    while i < len(bounds):
        j = i + 1
        stride = bounds[j][0] - bounds[i][0] if j < len(bounds) else bounds[i][1]
        while (j < len(bounds) and bounds[j][1] == bounds[i][1] and
               bounds[j][0] - bounds[j - 1][0] == stride):
            j += 1
        batches.append((i, j, stride))
        i = j
    return batches


def _dec_len(cnt, dec):
    """samples left of cnt after data[::dec]"""
    return (cnt + dec - 1) // dec


def scan_station(ss, sta, utcStart, utcEnd, triggerLTATime, triggerSTATime, calcHist,
                 useSubSpaces, useSingles, fillZeros, device=0, ctx=None, estimateMags=False):
    """_SSDetex._corDat (detect.py:137-218) for one station with all its
    detectors in one GPU basis: per batch of chunks the detection statistic
    (dtx_subspace_scan: detection statistic + histogram / max / candidates), then the
    reference's trigger loop on the sparse candidates.  Returns dict(ss=DataFrame,
    sg=DataFrame, hist_ss={name: counts}, hist_sg={...}) or None."""
    import pandas as pd
    import torch
    from .hip import HipContext
    names, kinds, U, det_off, offsets, thr = ss._detector_bank(sta, useSubSpaces, useSingles)
    if not names:
        return None
    fetcher = ss.cfetcher
    if sta not in fetcher.data:
        return None
    sr = fetcher.sampling_rate
    Nc = len(fetcher.channels)
    n = U.shape[1]
    D = len(names)
    own = ctx is None
    if own:
        ctx = HipContext(device)
    cols = ['DS', 'DS_STALTA', 'STMP', 'Name', 'Sta', 'MSTAMPmin', 'MSTAMPmax', 'Mag', 'SNR',
            'ProEnMag']
    rows = {'ss': [], 'sg': []}
    try:
        ctx.upload_basis(U, det_off, Nc)
        data, t0 = fetcher.data[sta]
        raw = bool(getattr(fetcher, 'raw', False))
        dec = int(getattr(ss.clusters, 'decimate', None) or 1) if raw else 1
        if raw:
            # unfiltered planes stay on the device; each chunk is decimated, detrended and
            # band-passed separately, like the reference's per-chunk _applyFilter
            # (detect.py:231); after decimation everything runs at sr / decimate (detect.py:239)
            rawdev = torch.from_numpy(np.ascontiguousarray(data, dtype=np.float32)).to(ctx.device)
            xdev = None
            sr_raw, sr = sr, sr / dec
        else:
            # channel planes go up as they are and are interleaved on the device
            # (construct.py:977-978; dtx_preprocess without filter or detrend)
            planes = torch.from_numpy(np.ascontiguousarray(data, dtype=np.float32)).to(ctx.device)
            xdev = ctx.preprocess(planes[None], filt=None, detrend=False)[0]
            del planes
        thr_arr = np.array([thr[k] for k in names], dtype=np.float32)
        hist = torch.zeros((D, 400), dtype=torch.int32, device=ctx.device)
        bounds = fetcher.chunk_bounds(sta, utcStart, utcEnd)
        # detect.py:262-274: blocks not longer than the template, or giving < 10 windows, are skipped
        bounds = [b for b in bounds
                  if _dec_len(b[1], dec) * Nc > n and (_dec_len(b[1], dec) * Nc - n) // Nc + 1 >= 10]
        LTA = triggerLTATime * sr
        STA = triggerSTATime * sr
        for (i0, i1, stride) in _chunk_batches(bounds):
            s0, cnt = bounds[i0][0], bounds[i0][1]
            M = _dec_len(cnt, dec) * Nc
            if raw:
                planes = torch.stack([rawdev[:, bounds[k][0]:bounds[k][0] + cnt]
                                      for k in range(i0, i1)])
                if dec > 1:                                      # construct.py:1016-1017
                    planes = ctx.decimate(planes, dec, sr_raw)
                mp = ctx.preprocess(planes, filt=ss.clusters.filt, sampling_rate=sr)
                ds, res = ctx.subspace_scan(mp, thr=thr_arr, hist=hist if calcHist else None)
                mp_host = mp.cpu().numpy() if estimateMags else None
            else:
                ds, res = ctx.subspace_scan(xdev[s0 * Nc:], M=M, chunk_stride=stride * Nc,
                                            nchunk=i1 - i0, thr=thr_arr,
                                            hist=hist if calcHist else None)
            P = int(ds.shape[2])
            maxds = res['maxds'].cpu().numpy()
            cb, cd, cp, cv = res['cand']
            if res['ncand'] > len(cv):
                raise RuntimeError('candidate buffer overflow (%d > %d): threshold far too low'
                                   % (res['ncand'], len(cv)))
            mp_cache = {}
            # candidates grouped by (chunk, detector); pairs without one cannot trigger
            key = cb.astype(np.int64) * D + cd
            order = np.argsort(key, kind='stable')               # keeps p ascending per pair
            ks, first = np.unique(key[order], return_index=True)
            last = np.append(first[1:], len(order))
            for kk, f0, f1 in zip(ks, first, last):
                b, d = int(kk // D), int(kk % D)
                name = names[d]
                start = bounds[i0 + b][2]
                sel = order[f0:f1]
                trig = pick_triggers_sparse(cp[sel], cv[sel], P, float(thr_arr[d]), sr,
                                            float(maxds[b, d]))
                row_dev = ds[b, d]

                def fetch(lo, hi, _r=row_dev):
                    return np.abs(_r[lo:hi].cpu().numpy().astype(np.float64))
                for (ti, coef) in trig:
                    if coef > 1.05:                          # detect.py:199-204
                        continue
                    if fillZeros:                            # detect.py:414-415
                        sl = 0.0
                    else:
                        try:
                            sl = stalta_at(fetch, ti, P, LTA, STA)
                        except IndexError:                   # detect.py:289-295, 417-420
                            sl = 0.0
                    times = float(ti) / sr + start           # detect.py:413
                    of = offsets[name]
                    pem = stm = snr = np.nan
                    if estimateMags:                         # detect.py:422-429
                        if raw:
                            mpc = mp_host[b]
                        else:
                            if b not in mp_cache:            # the chunk as the scan saw it
                                c0 = bounds[i0 + b][0] * Nc
                                mp_cache[b] = xdev[c0:c0 + M].cpu().numpy()
                            mpc = mp_cache[b]
                        pem, stm, snr = est_mag(ti, Nc, mpc, ss._bank_mags[name],
                                                U[det_off[d]:det_off[d + 1]],
                                                ss._bank_ewf[name], kinds[d] == 'ss')
                    rows[kinds[d]].append([coef, sl, times, name, sta, times - np.max(of),
                                           times - np.min(of), stm, snr, pem])
        ctx.sync()
        h = hist.cpu().numpy()
    finally:
        if own:
            ctx.close()
    out = {k: pd.DataFrame(v, columns=cols) for k, v in rows.items()}
    out['hist_ss'] = {nm: h[i] for i, nm in enumerate(names) if kinds[i] == 'ss'}
    out['hist_sg'] = {nm: h[i] for i, nm in enumerate(names) if kinds[i] == 'sg'}
    return out


# --------------------------------------------------------------------------
# magnitude estimates at a trigger                 detect.py:447-499, 637-664
# --------------------------------------------------------------------------

def _rolling_std_median(pe, n):
    """median(pd.rolling_std(pe, n)[n-1:]) (ddof=1) by prefix sums in float64."""
    pe = np.asarray(pe, dtype=np.float64)
    pe = pe - pe.mean()                      # shift invariance: kills the cancellation
    c1 = np.concatenate([[0.0], np.cumsum(pe)])
    c2 = np.concatenate([[0.0], np.cumsum(pe * pe)])
    s1 = c1[n:] - c1[:-n]
    s2 = c2[n:] - c2[:-n]
    var = np.maximum((s2 - s1 * s1 / n) / (n - 1.0), 0.0)
    return float(np.median(np.sqrt(var)))


def est_mag(trigIndex, Nc, MPcon, mags, U, ewf, issubspace=True):
    """Projected-energy and standard-deviation magnitude estimates and the SNR
    of a detection (Chambers et al. 2015; detect.py:447-499).  Same results as
    the reference without its n x n ``UtU`` (648 MB per detector at n = 9000,
    detect.py:367): U^T (U w) replaces UtU w, and (WFs U^T) U replaces WFs UtU.
    Returns (projectedEnergyMag, stdMag, SNR); per trigger, rare, on the host."""
    MPcon = np.asarray(MPcon)
    ewf = np.asarray(ewf, dtype=np.float64)
    mags = np.asarray(mags, dtype=np.float64)
    WFlen = ewf.shape[1]
    a = trigIndex * Nc
    ConDat = np.asarray(MPcon[a: a + WFlen], dtype=np.float64)
    if a > 5 * WFlen:                                            # detect.py:466-471
        pe = MPcon[a - 5 * WFlen: a]
    else:
        pe = MPcon[a: a + WFlen + 6 * WFlen]
    SNR = np.std(ConDat) / _rolling_std_median(pe, WFlen)        # detect.py:472-473
    touse = mags > -15
    if issubspace:
        U = np.asarray(U, dtype=np.float64)
        if not touse.any():                                      # detect.py:478-482
            return np.nan, np.nan, SNR
        ssCon = U.T @ (U @ ConDat)                               # = UtU . ConDat
        WFU = (ewf @ U.T) @ U                                    # = WFs . UtU
        proEn = np.var(ssCon) / np.var(WFU, axis=1)              # detect.py:463
        cd = ConDat - ConDat.mean()
        ec = np.array([np.dot(x - x.mean(), cd) / (len(x) * np.std(x) * np.std(ConDat))
                       for x in ewf])                            # fast_normcorr(x, ConDat)[0]
        we = np.square(ec) * touse
        pem = np.sum((mags + np.log10(np.sqrt(proEn)))[touse] * we[touse]) / we.sum()
        stm = np.sum((mags + np.log10(np.std(ConDat) / np.std(ewf, axis=1)))[touse] *
                     we[touse]) / we.sum()
        return float(pem), float(stm), float(SNR)
    if np.isnan(mags[0]) or mags[0] < -15:                       # detect.py:490-492
        return np.nan, np.nan, float(SNR)
    d1 = np.dot(ConDat, ewf[0])
    d2 = np.dot(ewf[0], ewf[0])
    return (float(mags[0] + d1 / d2),
            float(mags[0] + np.log10(np.std(ConDat) / np.std(ewf[0]))), float(SNR))
