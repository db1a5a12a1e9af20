"""Host side of the detection scan: the reference's trigger-picking loop
(detex/detect.py:390-445, 526-557) run on the sparse list of above-threshold
samples the GPU returns instead of on the full detection-statistic vector."""
import contextlib
import os

import numpy as np


def _stream_of(ctx, device):
    """The stream a context's kernels run on (events that order against them are recorded there)."""
    import torch
    st = getattr(ctx, 'stream', None)
    return st if st is not None else torch.cuda.current_stream(device)


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


def pick_triggers_grouped(cp, cv, first, last, Pg, thr_g, maxds_g, sr, buff=20):
    """pick_triggers_sparse for many (chunk, detector) groups at once: group g's candidates are
    cp / cv[first[g]:last[g]] (window indices ascending), Pg / thr_g / maxds_g its window count,
    threshold and DS.max().  Returns one list of (trigIndex, DS) per group.

    A group whose candidates span fewer than buff * sr windows -- one detection's peak, the
    ordinary case -- has exactly one trigger: the first arg-max, whose zeroed range
    [index - w, index + w) (clipped at the block's ends, detect.py:551-556) holds every other
    candidate.  Those groups are answered with a few array operations for all of them; the
    others (and any whose maximum is below threshold) go through the loop of
    pick_triggers_sparse."""
    G = len(first)
    out = [[] for _ in range(G)]
    if G == 0:
        return out
    cp = np.asarray(cp, dtype=np.int64)
    cv64 = np.asarray(cv, dtype=np.float64)
    first = np.asarray(first, dtype=np.int64)
    last = np.asarray(last, dtype=np.int64)
    w = buff * sr
    live = np.asarray(maxds_g, dtype=np.float64) > np.asarray(thr_g, dtype=np.float64)   # detect.py:534
    gmax = np.maximum.reduceat(cv64, first)
    single = live & ((cp[last - 1] - cp[first]) < w) & (gmax >= thr_g)
    # first occurrence of the group's maximum
    pos = np.arange(len(cp), dtype=np.int64)
    at = np.minimum.reduceat(np.where(cv64 == np.repeat(gmax, last - first), pos, len(cp)), first)
    ti = cp[np.minimum(at, len(cp) - 1)]
    for g in np.nonzero(live)[0]:
        if single[g]:
            out[g] = [(int(ti[g]), float(gmax[g]))]
        else:
            out[g] = pick_triggers_sparse(cp[first[g]:last[g]], cv64[first[g]:last[g]], int(Pg[g]), float(thr_g[g]),
                                          sr, float(maxds_g[g]), buff=buff)
    return out


# --------------------------------------------------------------------------
# STA/LTA of the detection statistic at a trigger       detect.py:501-524
# --------------------------------------------------------------------------

def _centered_lo(i, n, P):
    """First sample of the window whose mean is element i of
    ``replace_nan_with_mean(pd.rolling_mean(|C|, n, center=True))`` (detect.py:509-511,
    516-524): -1 if that element is itself a NaN of the rolling mean (the ``first + 1``
    quirk on very short vectors); IndexError where the reference's indexing fails
    (its caller then records 0.0, detect.py:289-295, 417-420)."""
    n = int(n)
    if n < 1 or n > P:
        raise IndexError('window longer than the detection statistic')
    off = (n - 1) // 2
    first = n - 1 - off                 # first index with a full window
    last = first + (P - n)
    if i < first:                       # arg[:first] = arg[first + 1]   (reference quirk)
        i = first + 1
        if i > P - 1:
            raise IndexError('arg[first + 1] is out of range')
        if i > last:
            return -1                   # arg[first + 1] is itself a NaN of the rolling mean
    elif i > last:                      # arg[last + 1:] = arg[last]
        i = last
    return i - first


def _centered_lo_many(i, n, P):
    """_centered_lo for arrays of indices i and window counts P: (lo, bad) with bad where the
    scalar form raises IndexError (lo there is -1)."""
    i = np.asarray(i, dtype=np.int64)
    P = np.asarray(P, dtype=np.int64)
    n = int(n)
    first = n - 1 - (n - 1) // 2
    last = first + (P - n)
    bad = (n < 1) | (n > P)
    front = i < first
    bad = bad | (front & (first + 1 > P - 1))
    lo = np.where(front, np.where(first + 1 > last, -1, 1), np.where(i > last, last - first, i - first))
    return np.where(bad, -1, lo), bad


def _centered_mean_at(fetch, i, n, P):
    """Value at index i of that vector without forming it: ``fetch(lo, hi)`` returns |C|[lo:hi]."""
    lo = _centered_lo(i, n, P)
    if lo < 0:
        return float('nan')
    return float(np.mean(fetch(lo, lo + int(n))))


def stalta_at(fetch, i, P, LTA, STA):
    """detect.py:501-514 evaluated at one index."""
    if STA == 0:
        st = float(fetch(i, i + 1)[0])
    else:
        st = _centered_mean_at(fetch, i, STA, P)
    lt = _centered_mean_at(fetch, i, LTA, P)
    with np.errstate(all='ignore'):
        return float(np.divide(st, lt))


def stalta_array(C, LTA, STA):
    """detect.py:501-524 as a whole vector (the product evaluates it at the triggers only, on the device:
    dtx_trigger_stats; this host form serves the utcSaves rows): |C| (STA == 0) or its centred rolling mean
    over STA samples, divided by the centred rolling mean over LTA samples, the NaN ends of a rolling mean
    replaced as _replaceNanWithMean does (the front by element first + 1, the back by the last valid one)."""
    C = np.abs(np.asarray(C, dtype=np.float64))
    P = len(C)

    def centred(n):
        n = int(n)
        out = np.full(P, np.nan)
        if n < 1 or n > P:
            raise IndexError('window longer than the detection statistic')
        cs = np.concatenate([[0.0], np.cumsum(C)])
        rm = (cs[n:] - cs[:-n]) / n                        # rm[j] = mean(C[j:j + n])
        first = n - 1 - (n - 1) // 2
        last = first + (P - n)
        out[first:last + 1] = rm
        if first > 0:
            out[:first] = out[first + 1] if first + 1 <= P - 1 else np.nan
        out[last + 1:] = out[last]
        return out
    st = C if STA == 0 else centred(STA)
    with np.errstate(all='ignore'):
        return np.divide(st, centred(LTA))


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


# capacity of the device candidate list of one batch (samples at or above threshold); a batch
# that exceeds it is listed again at the size the device counted
CAND_CAP = 1 << 18


def _warn(msg):
    """detex.log(level='warning', pri=True): the scan goes on (detect.py:160-174)."""
    import logging
    logging.getLogger('detex_amd.detect').warning(msg)


def mag_constants(ewf, mags, U, issub):
    """Per-detector constants of the magnitude estimates (detect.py:367-381, 447-499): sums,
    stds and self dot products of the training events (as float32, the precision the device
    holds them in), and var(WFU, axis=1) with WFU = WFs . UtU formed as (WFs U^T) U."""
    ewf = np.asarray(ewf, dtype=np.float32).astype(np.float64)
    c = dict(ewf=ewf, mags=np.asarray(mags, dtype=np.float64), sum_e=ewf.sum(axis=1),
             std_e=ewf.std(axis=1), d2=np.einsum('ij,ij->i', ewf, ewf), var_wfu=None)
    if issub:
        Ud = np.asarray(U, dtype=np.float64)
        c['var_wfu'] = np.var((ewf @ Ud.T) @ Ud, axis=1)
    return c


def mags_from_stats(c, issub, n, o):
    """(ProEnMag, Mag, SNR) of one trigger (detect.py:447-499, _estPEMag / _estSTDMag 637-664)
    from its dtx_trigger_stats row ``o``: o[2] mean and o[3] std of the triggering window,
    o[4] var(U^T U w), o[5] median rolling std of the pre-event noise, o[6 + e] w . event e."""
    mean_c, std_c, var_s, noise = o[2], o[3], o[4], o[5]
    with np.errstate(all='ignore'):
        snr = float(np.divide(std_c, noise))                      # detect.py:472-473
    mags = c['mags']
    dots = np.asarray(o[6:6 + len(mags)], dtype=np.float64)
    if issub:
        touse = mags > -15
        if not touse.any():                                       # detect.py:478-482
            return np.nan, np.nan, snr
        with np.errstate(all='ignore'):
            # fast_normcorr(x, ConDat)[0] (construct.py:469-483) from the dot products
            ec = (dots - c['sum_e'] * mean_c) / (n * c['std_e'] * std_c)
            we = np.square(ec) * touse
            proEn = var_s / c['var_wfu']                          # detect.py:463
            pem = np.sum((mags + np.log10(np.sqrt(proEn)))[touse] * we[touse]) / we.sum()
            stm = np.sum((mags + np.log10(std_c / c['std_e']))[touse] * we[touse]) / we.sum()
        return float(pem), float(stm), snr
    if np.isnan(mags[0]) or mags[0] < -15:                        # detect.py:490-492
        return np.nan, np.nan, snr
    with np.errstate(all='ignore'):
        return (float(mags[0] + dots[0] / c['d2'][0]),
                float(mags[0] + np.log10(std_c / c['std_e'][0])), snr)


def mags_from_stats_many(c, issub, n, O):
    """mags_from_stats for all the rows O [k, 6 + emax] of one detector at once: (ProEnMag, Mag, SNR)
    arrays of length k.  The same formulas row by row (sums over the events in the same order)."""
    O = np.asarray(O, dtype=np.float64)
    k = len(O)
    mean_c, std_c, var_s, noise = O[:, 2], O[:, 3], O[:, 4], O[:, 5]
    with np.errstate(all='ignore'):
        snr = np.divide(std_c, noise)
    mags = c['mags']
    nan = np.full(k, np.nan)
    dots = O[:, 6:6 + len(mags)]
    if issub:
        touse = mags > -15
        if not touse.any():
            return nan, nan, snr
        with np.errstate(all='ignore'):
            ec = (dots - c['sum_e'][None, :] * mean_c[:, None]) / (n * c['std_e'][None, :] * std_c[:, None])
            we = np.square(ec) * touse[None, :]
            proEn = var_s[:, None] / c['var_wfu'][None, :]
            wsum = we.sum(axis=1)
            pem = np.sum(((mags[None, :] + np.log10(np.sqrt(proEn))) * we)[:, touse], axis=1) / wsum
            stm = np.sum(((mags[None, :] + np.log10(std_c[:, None] / c['std_e'][None, :])) * we)[:, touse], axis=1) / wsum
        return pem, stm, snr
    if np.isnan(mags[0]) or mags[0] < -15:
        return nan, nan, snr
    with np.errstate(all='ignore'):
        return mags[0] + dots[:, 0] / c['d2'][0], mags[0] + np.log10(std_c / c['std_e'][0]), snr


class _MagBank(object):
    """The training events of every detector of a station on the device (for
    dtx_trigger_stats) and their host-side constants."""

    def __init__(self, ss, sta, names, kinds, U, det_off, device):
        import torch
        ewf, mags = ss._bank_aux[sta]
        self.consts = [mag_constants(ewf[nm], mags[nm],
                                     U[det_off[d]:det_off[d + 1]], kinds[d] == 'ss')
                       for d, nm in enumerate(names)]
        eoff = np.concatenate([[0], np.cumsum([len(c['mags']) for c in self.consts])])
        self.emax = int(max(len(c['mags']) for c in self.consts))
        self.ewf = torch.from_numpy(np.concatenate([c['ewf'] for c in self.consts])
                                    .astype(np.float32)).to(device)
        self.eoff = torch.from_numpy(eoff.astype(np.int32)).to(device)

        # the constants padded to emax events per detector, for estimate_all
        D = len(self.consts)

        def padded(key, fill):
            a = np.full((D, max(self.emax, 1)), fill, dtype=np.float64)
            for d, c in enumerate(self.consts):
                if c[key] is not None:
                    a[d, :len(c[key])] = c[key]
            return a
        self.issub = np.array([k == 'ss' for k in kinds], dtype=bool)
        self.mags_p, self.sum_e_p, self.std_e_p = padded('mags', np.nan), padded('sum_e', 0.0), padded('std_e', 1.0)
        self.var_wfu_p, self.d2_p = padded('var_wfu', 1.0), padded('d2', 1.0)
        with np.errstate(invalid='ignore'):
            self.touse_p = self.mags_p > -15                    # (NaN padding compares False)

    def estimate(self, d, issub, n, o):
        return mags_from_stats(self.consts[d], issub, n, o)

    def estimate_all(self, dets, n, O):
        """mags_from_stats for the rows O [k, 6 + emax] of triggers of ANY of the bank's detectors (dets[r] =
        row r's): the formulas of mags_from_stats_many on the padded constants, the padding masked out of
        every sum.  One call per batch of chunks instead of one per detector with a trigger in it."""
        O = np.asarray(O, dtype=np.float64)
        dets = np.asarray(dets, dtype=np.int64)
        k = len(O)
        mean_c, std_c, var_s, noise = O[:, 2], O[:, 3], O[:, 4], O[:, 5]
        pem, stm = np.full(k, np.nan), np.full(k, np.nan)
        with np.errstate(all='ignore'):
            snr = np.divide(std_c, noise)
            sub = self.issub[dets]
            r = np.nonzero(sub)[0]
            if len(r):
                d = dets[r]
                tu, mg, se = self.touse_p[d], self.mags_p[d], self.std_e_p[d]
                dots = O[r, 6:6 + tu.shape[1]]
                sc = std_c[r, None]
                ec = (dots - self.sum_e_p[d] * mean_c[r, None]) / (n * se * sc)
                we = np.where(tu, np.square(ec), 0.0)
                proEn = var_s[r, None] / self.var_wfu_p[d]
                wsum = we.sum(axis=1)
                some = tu.any(axis=1)                                  # detect.py:478-482
                pem[r] = np.where(some, np.where(tu, (mg + np.log10(np.sqrt(proEn))) * we, 0.0).sum(axis=1) / wsum, np.nan)
                stm[r] = np.where(some, np.where(tu, (mg + np.log10(sc / se)) * we, 0.0).sum(axis=1) / wsum, np.nan)
            r = np.nonzero(~sub)[0]
            if len(r):
                d = dets[r]
                m0 = self.mags_p[d, 0]
                ok = ~(np.isnan(m0) | (m0 < -15))                      # detect.py:490-492
                pem[r] = np.where(ok, m0 + O[r, 6] / self.d2_p[d, 0], np.nan)
                stm[r] = np.where(ok, m0 + np.log10(std_c[r] / self.std_e_p[d, 0]), np.nan)
        return pem, stm, snr

    def estimate_many(self, d, issub, n, O):
        return mags_from_stats_many(self.consts[d], issub, n, O)


def scan_station(ss, sta, utcStart, utcEnd, triggerLTATime, triggerSTATime, calcHist,
                 useSubSpaces, useSingles, fillZeros, device=0, ctx=None, estimateMags=True,
                 shard=None, bank=None):
    """_SSDetex._corDat (detect.py:137-218) for one station with all its
    detectors in one GPU basis: per batch of chunks the detection statistic
    (dtx_subspace_scan: detection statistic + histogram / max / candidates), the
    reference's trigger loop on the sparse candidates, then ONE dtx_trigger_stats launch
    for the STA/LTA and magnitude ingredients of all the batch's triggers.  Returns
    dict(ss=DataFrame, sg=DataFrame, hist_ss={name: counts}, hist_sg={...}) or None.

    shard: what SubSpace.detex's uploader prepared -- dict(planes=[Nc, T] device tensor or None,
    bounds=[(first sample in planes, samples, start time, chunk index)], T, counts=fragments in the
    same coordinates or None): the whole station, or in a multi-process run only the rank's own
    chunks, packed; bank: the station's detector bank if the
    caller already built (or received) it; a failing batch is logged and skipped like a
    failing chunk of the reference (detect.py:160-174).  (StationScan.start / .pick / .finish are
    the three stages: SubSpace.detex keeps several units in flight, one or more in each.)"""
    job = StationScan(ss, sta, utcStart, utcEnd, triggerLTATime, triggerSTATime, calcHist, useSubSpaces,
                      useSingles, fillZeros, device, ctx, estimateMags, shard, bank)
    import pandas as pd
    try:
        job.start()
        res = job.finish()
    finally:
        job.close()
    if res is not None:
        for k in ('ss', 'sg'):
            res[k] = pd.DataFrame(res[k], columns=StationScan.COLS)
    return res


class StationScan(object):
    """One station's scan in three stages: start() enqueues everything the GPU does for the
    detection statistic (basis upload if needed, interleaving / filtering, one dtx_subspace_scan
    per batch of chunks, the small results on their way to pinned host memory); pick() takes the
    triggers from the candidate lists and enqueues dtx_trigger_stats; finish() waits for its rows
    (by event: nothing here synchronises the stream, which holds the next stations' scans) and
    builds the tables."""

    COLS = ['DS', 'DS_STALTA', 'STMP', 'Name', 'Sta', 'MSTAMPmin', 'MSTAMPmax', 'Mag', 'SNR', 'ProEnMag']

    def __init__(self, ss, sta, utcStart, utcEnd, triggerLTATime, triggerSTATime, calcHist, useSubSpaces,
                 useSingles, fillZeros, device=0, ctx=None, estimateMags=True, shard=None, bank=None,
                 fetcher=None, utcSaves=None):
        from .hip import HipContext
        self.ss, self.sta = ss, sta
        self.args = (utcStart, utcEnd, triggerLTATime, triggerSTATime, calcHist, fillZeros, estimateMags)
        self.bank = bank if bank is not None else ss._detector_bank(sta, useSubSpaces, useSingles)
        self.shard = shard
        # the data scanned: the SubSpace's continuous fetcher, or the template events of a classifyEvents run
        self.fetcher = fetcher if fetcher is not None else ss.cfetcher
        self.utc_saves = None if utcSaves is None else np.asarray(utcSaves, dtype=np.float64)
        self.empty = (not self.bank[0]) or sta not in self.fetcher.data
        self.own = ctx is None and not self.empty
        self.ctx = HipContext(device) if self.own else ctx
        self.batches = []
        self.picked = None
        self.scan_ev = None
        self.packed = False

    def close(self):
        if self.own and self.ctx is not None:
            self.ctx.close()
            self.ctx = None

    def _guard(self):
        """The context's stream made current (HipContext.on_stream): buffers, events and kernels of
        this scan agree on the stream whatever stream the caller works on."""
        g = getattr(self.ctx, 'on_stream', None)
        return g() if g is not None else contextlib.nullcontext()

    def start(self):
        if self.empty:
            return
        with self._guard():
            self._start()

    def _start(self):
        import torch
        from ._lib import DetexHipError
        ss, sta, ctx = self.ss, self.sta, self.ctx
        utcStart, utcEnd, triggerLTATime, triggerSTATime, calcHist, fillZeros, estimateMags = self.args
        names, kinds, U, det_off, offsets, thr = self.bank
        fetcher = self.fetcher
        sr = fetcher.sampling_rate
        Nc = len(fetcher.channels)
        n = U.shape[1]
        D = len(names)
        if getattr(ctx, '_basis_of', None) is not U:      # the same bank object: images are resident
            ctx.upload_basis(U, det_off, Nc)
            ctx._basis_of = U
        data, t0 = fetcher.data[sta]
        raw = bool(getattr(fetcher, 'raw', False))
        dec = int(getattr(ss.clusters, 'decimate', None) or 1) if raw else 1
        shard = self.shard
        self.shard = None
        if shard is None:                                 # the whole station, uploaded here
            counts = getattr(fetcher, 'counts', {}).get(sta)
            bounds = [(int(b[0]), int(b[1]), b[2], i)
                      for i, b in enumerate(fetcher.chunk_bounds(sta, utcStart, utcEnd))]
            planes_dev, T = None, data.shape[1]
        else:
            counts, bounds, planes_dev, T = shard['counts'], list(shard['bounds']), shard['planes'], shard['T']
        xdev_up = None if shard is None else shard.get('xdev')    # the uploader interleaved the planes already
        del shard
        trimmed = False
        if counts is not None:
            # raw int32 counts in fragments: merged with zero fill on the device
            # (construct.py:1038 / 1070), then the per-chunk trimming.  Both on a helper context with a stream
            # of its own: the upload runs beside the scans queued on this station's stream, and the host's
            # wait for the longest gap-free runs is a wait for that helper stream only
            pinned = getattr(fetcher, 'counts_pinned', {}).get(sta)
            main = getattr(ctx, 'stream', None)
            side = getattr(ss, '_side_ctx', None) if main is not None else None
            if side is not None:
                h = side(ctx.device, 2)
                with torch.cuda.stream(h.stream):
                    planes_dev = _ingest_station(h, counts, Nc, T, pinned)
                    if not fillZeros:
                        bounds = _trim_to_gapfree(h, planes_dev, counts, bounds, fetcher.sampling_rate, sta)
                        trimmed = True
                planes_dev.record_stream(main)
                main.wait_stream(h.stream)
            else:
                planes_dev = _ingest_station(ctx, counts, Nc, T, pinned)
        elif planes_dev is None:
            planes_dev = torch.from_numpy(np.ascontiguousarray(data, dtype=np.float32)).to(ctx.device)
        sr_raw = sr
        if raw:
            # unfiltered planes stay on the device; each chunk is decimated, detrended and
            # band-passed separately, like the reference's per-chunk _applyFilter
            # (detect.py:231); after decimation everything runs at sr / decimate (detect.py:239)
            rawdev, xdev = planes_dev, None
            sr = sr / dec
        else:
            # channel planes are interleaved on the device
            # (construct.py:977-978; dtx_preprocess without filter or detrend)
            rawdev = None
            xdev = xdev_up if xdev_up is not None else ctx.preprocess(planes_dev[None], filt=None, detrend=False)[0]
        del planes_dev
        thr_arr = np.array([thr[k] for k in names], dtype=np.float32)
        from .hip import h2d_async
        # the thresholds stay on the device with the bank they belong to (a day unit re-sent 100 bytes)
        held = getattr(ctx, '_thr_dev', None)
        if held is None or held[0] is not U or held[1] != thr_arr.tobytes():
            held = (U, thr_arr.tobytes(), h2d_async(thr_arr, ctx.device))
            ctx._thr_dev = held
        thr_dev = held[2]
        # a HipContext scan owns its (zeroed) histogram and hands it back on the host with the candidates:
        # one fill and two copies per scan instead of two fills and five copies (hip._post_buffers)
        self.packed = bool(getattr(ctx, 'packed_results', False))
        hist = None if self.packed else torch.zeros((D, 400), dtype=torch.int32, device=ctx.device)
        if counts is not None and not fillZeros and not trimmed:
            bounds = _trim_to_gapfree(ctx, rawdev, counts, bounds, fetcher.sampling_rate, sta)
        # detect.py:262-274: blocks not longer than the template, or giving < 10 windows, are skipped
        bounds = [b for b in bounds
                  if _dec_len(b[1], dec) * Nc > n and (_dec_len(b[1], dec) * Nc - n) // Nc + 1 >= 10]
        need_stats = estimateMags or not fillZeros
        mbank = None
        if need_stats:
            # the training events on the device + their constants: kept with the bank they
            # belong to (rebuilt only when the bank is)
            held = getattr(ctx, '_magbank', None)
            if held is None or held[0] is not U:
                held = (U, _MagBank(ss, sta, names, kinds, U, det_off, ctx.device))
                ctx._magbank = held
            mbank = held[1]
        self.st = dict(names=names, kinds=kinds, offsets=offsets, thr_arr=thr_arr, thr_dev=thr_dev, hist=hist,
                       bounds=bounds, Nc=Nc, n=n, D=D, sr=sr, LTA=triggerLTATime * sr, STA=triggerSTATime * sr,
                       mbank=mbank)
        groups = _chunk_batches(bounds)
        if len(groups) > 1 and os.environ.get('DETEX_AMD_RAGGED', '1') != '0':
            # chunks of several lengths (the first and the last chunk of a request, chunks trimmed to
            # their data): ONE ragged launch instead of one set of launches per length
            try:
                self._start_ragged(groups, rawdev, xdev, dec, sr, sr_raw, thr_dev, hist if calcHist else None)
                groups = []
            except DetexHipError as e:
                # the reference loses a failing chunk, not the station (detect.py:160-174): launch by launch
                _warn('ragged launch failed on %s (%s): scanning one chunk length at a time' % (sta, e))
        for (i0, i1, stride) in groups:
            try:
                s0, cnt = bounds[i0][0], bounds[i0][1]
                M = _dec_len(cnt, dec) * Nc
                if raw:
                    xb = self._filtered(rawdev, i0, i1, dec, sr, sr_raw)
                    xarg, xM, xstride, cstride = xb, None, None, int(xb.stride(0))
                else:
                    xb = xdev[s0 * Nc:]
                    xarg, xM, xstride, cstride = xb, M, stride * Nc, stride * Nc
                ds, fetch = ctx.subspace_scan(xarg, M=xM, chunk_stride=xstride, nchunk=None if raw else i1 - i0,
                                              thr=thr_dev, hist=hist if calcHist else None, cand_cap=CAND_CAP,
                                              lazy=True, **({'packed': True} if self.packed else {}))
                self.batches.append((i0, i1, xb, M, cstride, ds, fetch, None))
            except DetexHipError as e:
                self._skip(i0, i1, e)
        # the histograms follow the last scan to pinned host memory; finish() waits for this event
        # only (a stream-wide synchronisation there would also wait for the NEXT stations' scans,
        # which are queued behind: it drained the queue once per station-day)
        self.hist_host = self.hist_ev = None
        if self.packed:
            self.scan_ev = torch.cuda.Event()
            self.scan_ev.record(_stream_of(ctx, ctx.device))
        elif hist.is_cuda:
            # behind this unit's scans: what the trigger-time kernel (on a stream of its own) waits for
            self.scan_ev = torch.cuda.Event()
            self.scan_ev.record(_stream_of(ctx, hist.device))
            self.hist_host = torch.empty(hist.shape, dtype=hist.dtype, pin_memory=True)
            self.hist_host.copy_(hist, non_blocking=True)
            self.hist_ev = torch.cuda.Event()
            self.hist_ev.record(_stream_of(ctx, hist.device))

    def _filtered(self, rawdev, i0, i1, dec, sr, sr_raw, ctx=None, out=None):
        """Chunks i0 .. i1 - 1 (equal lengths) of the unfiltered planes: decimated, detrended, band-passed
        and multiplexed like the reference's per-chunk _applyFilter -> [i1 - i0, Nc * samples]."""
        import torch
        ctx = ctx or self.ctx
        b = self.st['bounds']
        cnt = b[i0][1]
        step = b[i0 + 1][0] - b[i0][0] if i1 - i0 > 1 else 0
        if rawdev.stride(1) == 1 and all(b[k + 1][0] - b[k][0] == step for k in range(i0, i1 - 1)) and step >= 0:
            # equally spaced chunks of one record: a view of it (the reference's chunks overlap by conBuff, so they
            # are not a reshape; a packed copy of a station-day cost 0.14 ms ahead of every unit's filter)
            planes = rawdev.as_strided((i1 - i0, rawdev.shape[0], cnt), (step, rawdev.stride(0), 1),
                                       rawdev.storage_offset() + b[i0][0])
        else:
            planes = torch.stack([rawdev[:, b[k][0]:b[k][0] + cnt] for k in range(i0, i1)])
        if dec > 1:                                      # construct.py:1016-1017
            planes = ctx.decimate(planes, dec, sr_raw)
        return ctx.preprocess(planes, filt=self.ss.clusters.filt, sampling_rate=sr, out=out)

    def _start_ragged(self, groups, rawdev, xdev, dec, sr, sr_raw, thr_dev, hist):
        """All chunks of the station in one dtx_subspace_scan_ragged launch."""
        import torch
        from ._lib import DetexHipError
        ctx, b, Nc = self.ctx, self.st['bounds'], self.st['Nc']
        lens = np.array([_dec_len(c[1], dec) * Nc for c in b], dtype=np.int64)
        if rawdev is not None:
            # every group of equal length is filtered on its own, into rows of one buffer
            Mmax = int(lens.max())
            # (rows of the longest chunk's length; what lies behind a shorter chunk's samples is never read)
            xb = torch.empty((len(b), Mmax), dtype=torch.float32, device=ctx.device)
            # The filter kernels are chains of dependent steps: a group of ONE chunk takes nearly as long as
            # the group of 23 (0.8 against 1.2 ms).  The short groups therefore run beside the longest one, on
            # helper contexts with streams of their own; this context's stream waits for them before the scan.
            main = getattr(ctx, 'stream', None)          # (None: not a HipContext -- the tests' stand-in)
            big = max(groups, key=lambda g: g[1] - g[0])
            side = getattr(self.ss, '_side_ctx', None) if main is not None else None
            waits = []
            for k, (i0, i1, _) in enumerate(g for g in groups if g is not big):
                try:
                    if side is None:
                        self._filtered(rawdev, i0, i1, dec, sr, sr_raw, out=xb[i0:i1])
                        continue
                    h = side(ctx.device, k % 2)
                    h.stream.wait_stream(main)           # the planes and xb are this stream's
                    with torch.cuda.stream(h.stream):
                        self._filtered(rawdev, i0, i1, dec, sr, sr_raw, ctx=h, out=xb[i0:i1])
                    waits.append(h.stream)
                except DetexHipError as e:
                    self._skip(i0, i1, e)
                    lens[i0:i1] = 0                      # (a chunk without windows: nothing is counted)
            try:
                i0, i1, _ = big
                self._filtered(rawdev, i0, i1, dec, sr, sr_raw, out=xb[i0:i1])
            except DetexHipError as e:
                self._skip(i0, i1, e)
                lens[i0:i1] = 0
            for hs in waits:
                main.wait_stream(hs)
            starts = np.arange(len(b), dtype=np.int64) * Mmax
        else:
            xb = xdev
            starts = np.array([c[0] * Nc for c in b], dtype=np.int64)
        ctab = ctx.chunk_table(starts, lens)
        ds, fetch = ctx.subspace_scan(xb, thr=thr_dev, hist=hist, cand_cap=CAND_CAP, lazy=True, chunk_tab=ctab,
                                      **({'packed': True} if self.packed else {}))
        self.batches.append((0, len(b), xb, int(lens.max()), None, ds, fetch, ctab))

    def _skip(self, i0, i1, e):
        # detect.py:169-174: "failing to run detector on ... from ... to ..." and go on
        b = self.st['bounds']
        _warn('failing to run detector on %s from %s to %s: %s'
              % (self.sta, b[i0][2], b[i1 - 1][2] + b[i1 - 1][1] / self.st['sr'], e))

    def pick(self, behind=None):
        """First host half, once the scans' candidate lists are back: trigger picking, and the
        trigger-time kernel enqueued (its results are not waited for).  behind: the StationScan whose scans
        are queued next on the device -- the trigger-time kernel starts when ITS scan kernel has ended."""
        from ._lib import DetexHipError
        if self.empty or self.picked is not None:
            return
        bctx = getattr(behind, 'ctx', None) if behind is not None and not behind.empty else None
        st = self.st
        self.picked = []
        with self._guard():
            for (i0, i1, xb, M, cstride, ds, fetch, ctab) in self.batches:
                try:
                    self.picked.append((i0,) + _batch_triggers(self.ctx, self.sta, st, self.args[5], self.args[6],
                                                               i0, i1, xb, M, cstride, ds, fetch, ctab,
                                                               after=self.scan_ev, behind=bctx))
                except DetexHipError as e:
                    self._skip(i0, i1, e)

    def finish(self):
        """Second host half: trigger-time estimates back from the device -> rows."""
        if self.empty:
            return None
        self.pick()
        st = self.st
        rows = {'ss': [], 'sg': []}
        keys = {'ss': [], 'sg': []}
        for (i0, trigs, bad, stats) in self.picked:
            _batch_rows(self.sta, st, self.args[5], self.args[6], i0, trigs, bad, stats, rows, keys)
        utc_rows = self._utc_rows() if self.utc_saves is not None else None
        self.batches, self.picked = [], None
        if self.packed:
            # every scan of the unit counted into a histogram of its own (fetched with its candidates)
            h = np.zeros((st['D'], 400), dtype=np.int64)
            if self.args[4]:                                 # calcHist
                for part in st.get('hist_parts', ()):
                    h += part.numpy()
        elif self.hist_ev is not None:
            self.hist_ev.synchronize()
            h = self.hist_host.numpy()
        else:
            h = st['hist'].cpu().numpy()
        names, kinds = st['names'], st['kinds']
        # rows as plain lists (column order COLS): SubSpace.detex makes ONE DataFrame per table at the end of the
        # call (a DataFrame per unit and kind cost the host a millisecond per station-day)
        out = dict(rows)
        # (chunk index in the station's chunk list, detector, order found) per row: a multi-process
        # run merges the ranks' tables by it into the single-process order
        for k, v in keys.items():
            out['key_' + k] = np.asarray(v, dtype=np.int64).reshape(-1, 3)
        out['hist_ss'] = {nm: h[i] for i, nm in enumerate(names) if kinds[i] == 'ss'}
        out['hist_sg'] = {nm: h[i] for i, nm in enumerate(names) if kinds[i] == 'sg'}
        # the largest detection statistic of every (chunk, detector): {chunk index: [D] array} (classifyEvents)
        out['maxds'] = dict(st.get('maxds', {}))
        out['names'] = list(names)
        if utc_rows is not None:
            out['utc'] = utc_rows
        return out

    def _utc_rows(self):
        """detect.py:297-318 _makeUTCSaveDF: for every chunk that covers one of the utcSaves times, one row
        per detector with the chunk's multiplexed data and its whole detection-statistic vector, copied
        back from the resident DS tensor (a debugging aid: these copies wait for the device)."""
        st = self.st
        names, thr, offsets, Nc, n, sr = st['names'], st['thr_arr'], st['offsets'], st['Nc'], st['n'], st['sr']
        fillZeros = self.args[5]
        rows = []
        for (i0, i1, xb, M, cstride, ds, fetch, ctab) in self.batches:
            for b in range(i1 - i0):
                if ctab is not None:
                    a, m = (int(v) for v in ctab[1][b])
                    mp = xb.reshape(-1)[a:a + m]
                elif xb.dim() == 2:
                    mp = xb[b, :M]
                else:
                    mp = xb[b * cstride: b * cstride + M]
                m = int(mp.numel())
                if m <= n:
                    continue
                ts1 = st['bounds'][i0 + b][2]
                ts2 = ts1 + m / (sr * float(Nc))
                inside = (self.utc_saves > ts1) & (self.utc_saves < ts2)
                if not inside.any():
                    continue
                Pb = (m - n) // Nc + 1
                mp_h = mp.cpu().numpy()
                ds_h = ds[b, :, :Pb].cpu().numpy()
                for d, nm in enumerate(names):
                    C = ds_h[d].astype(np.float64)
                    sl = None
                    if not fillZeros:
                        try:
                            sl = stalta_array(C, st['LTA'], st['STA'])
                        except IndexError:
                            sl = None                           # detect.py:289-295: logged, left empty
                    rows.append(dict(Station=self.sta, Name=nm, Threshold=float(thr[d]), offset=offsets[nm],
                                     TS1=ts1, TS2=ts2, utcSaves=self.utc_saves[inside], MPcon=mp_h, SSdetect=C,
                                     STALTA=sl, TimeStamp=ts1, SampRate=sr, MaxDS=float(C.max()),
                                     MaxSTALTA=(float(np.nanmax(sl)) if sl is not None else np.nan), Nc=Nc,
                                     File=np.nan, chunk=st['bounds'][i0 + b][3]))
        return rows


def _ingest_station(ctx, counts, Nc, T, pinned=None):
    """All fragments of a station in one dtx_ingest_counts call -> zero-filled [Nc, T] planes.
    pinned: (page-locked int32 buffer of the fragments end to end, their table, the fragment list it was
    built from) -- ContinuousFetcher.pin()."""
    import torch
    if pinned is not None and pinned[2] is counts:
        cdev = pinned[0].to(ctx.device, non_blocking=True)
        planes, _ = ctx.ingest_counts(cdev, pinned[1], Nc, T, want_run=False)
        return planes
    off, frs = 0, []
    for (c, t0, x) in counts:
        frs.append((c, off, t0, len(x)))
        off += len(x)
    flat = np.concatenate([x for (_, _, x) in counts]) if counts else np.zeros(0, np.int32)
    cdev = torch.from_numpy(flat).to(ctx.device)
    planes, _ = ctx.ingest_counts(cdev, frs, Nc, T, want_run=False)
    return planes


def chunk_traces(counts, s0, cnt):
    """The traces a chunk [s0, s0 + cnt) of a fragmented record consists of, per channel: the
    fragments that reach into it, clipped to it -- what the reference's fetcher hands to
    _applyFilter as ``st`` (one ObsPy trace per fragment).  {channel: [(first, end), ...]}"""
    from .getdata import trace_extents
    per = {}
    for (c, t0, m) in trace_extents(counts):
        lo, hi = max(int(t0), s0), min(int(t0) + m, s0 + cnt)
        if hi > lo:
            per.setdefault(c, []).append((lo, hi))
    return per


def _trim_to_gapfree(ctx, planes, counts, bounds, sr, sta):
    """The front half of _applyFilter per chunk (construct.py:1003-1022).

    A chunk with ONE trace per channel (``len(st) == len(nc)``, construct.py:1005) is not merged at
    all: it is only trimmed to the span all its traces cover (1019-1023) -- exact zero counts in it
    are ordinary samples.  Only a chunk in which some channel arrives in several traces goes
    through _mergeChannels (1033-1066): cut to its longest stretch without a gap or a zero in any
    channel (one device pass, dtx_gapfree_run); if the kept stretch is still in several traces on
    some channel it is dropped like the reference's 'Stream is too fractured' (1011-1018).  A chunk
    in which a channel has no data at all is dropped with a warning (the reference would go on
    with fewer channels than the templates have; deliberate deviation)."""
    import torch
    if not bounds:
        return bounds
    Nc = int(planes.shape[0])
    plan, fractured = [], []
    for k, (s0, cnt, ts, gi) in enumerate(bounds):
        per = chunk_traces(counts, s0, cnt)
        if len(per) < Nc:
            _warn('no data on %d of %d channels of %s from %s' % (Nc - len(per), Nc, sta, ts))
            plan.append(None)
        elif all(len(v) == 1 for v in per.values()):
            a = max(v[0][0] for v in per.values())            # startTrim / endTrim, construct.py:1019-1023
            b = min(v[0][1] for v in per.values())
            plan.append((a - s0, b - a) if b > a else None)
        else:
            plan.append('merge')
            fractured.append(k)
    runs = None
    if fractured:
        runs = torch.zeros((len(fractured), 2), dtype=torch.int64, device=planes.device)
        for j, k in enumerate(fractured):
            s0, cnt = bounds[k][0], bounds[k][1]
            ctx.gapfree_run(planes[:, s0:s0 + cnt], out=runs[j])
        runs = runs.cpu().numpy()
    out, j = [], 0
    for (s0, cnt, ts, gi), pl in zip(bounds, plan):
        if pl is None:
            continue
        if pl == 'merge':
            r0, rl = (int(v) for v in runs[j])
            j += 1
            if rl <= 0:
                _warn('no gap-free data on %s from %s' % (sta, ts))
                continue
            # _mergeChannels looks for zeros inside the span ALL merged channels cover (construct.py:1039-1050);
            # without one the merged stream itself goes on (1065-1066): no fracture check, trimmed to that span
            whole = chunk_traces(counts, s0, cnt)
            A = max(min(lo for lo, _ in v) for v in whole.values())
            B = min(max(hi for _, hi in v) for v in whole.values())
            if (s0 + r0, rl) != (A, B - A):
                per = chunk_traces(counts, s0 + r0, rl)       # traces that reach into the kept window
                if any(len(v) > 1 for v in per.values()):     # len(st) != len(nc)
                    _warn('Stream is too fractured around %s on %s' % (ts, sta))
                    continue
        else:
            r0, rl = pl
        out.append((s0 + int(r0), int(rl), ts + int(r0) / sr, gi))
    return out


def _batch_triggers(ctx, sta, st, fillZeros, estimateMags, i0, i1, xb, M, cstride, ds, fetch, ctab=None, after=None,
                    behind=None):
    """One launch group of chunks (equal lengths, or a ragged batch with its chunk table), first
    host half: candidates -> triggers; the trigger-time kernel is enqueued and its (device) result
    returned with them: (trigs, bad, stats)."""
    import torch
    names, kinds, offsets, thr_arr, thr_dev = st['names'], st['kinds'], st['offsets'], st['thr_arr'], st['thr_dev']
    bounds, Nc, n, D, sr, LTA, STA, bank = (st['bounds'], st['Nc'], st['n'], st['D'], st['sr'], st['LTA'],
                                            st['STA'], st['mbank'])
    nb = i1 - i0
    res = fetch()
    if st.get('hist') is None and res.get('hist') is not None:
        st.setdefault('hist_parts', []).append(res['hist'])      # packed results: this scan's own histogram, on the host
    if res['ncand'] > CAND_CAP:
        # the candidate list is sized for ordinary thresholds; one that overflows is listed
        # again from the DS tensor at the size the device counted (no second scan; the
        # histogram already holds this batch)
        _warn('%d samples above threshold on %s in one batch of %d chunks: thresholds very low?'
              % (res['ncand'], sta, nb))
        again = ctx.ds_post(ds, thr=thr_dev, want_max=False, cand_cap=int(res['ncand']) + 1024)
        res['cand'], res['ncand'] = again['cand'], again['ncand']
    P = int(ds.shape[2])
    maxds = res['maxds'].cpu().numpy() if res['maxds'].is_cuda else res['maxds'].numpy()
    for b in range(nb):                                  # (kept per chunk: classifyEvents reports them)
        st.setdefault('maxds', {})[bounds[i0 + b][3]] = np.array(maxds[b], dtype=np.float64)
    cb, cd, cp, cv = res['cand']
    # windows of every chunk of the group (ragged batch: its own)
    Pb = np.full(nb, P, dtype=np.int64)
    if ctab is not None:
        Pb = np.where(ctab[1][:, 1] > n, (ctab[1][:, 1] - n) // Nc + 1, 0)
        live = cp < Pb[cb]                               # (the re-listed candidates of an overflow are not cut)
        cb, cd, cp, cv = cb[live], cd[live], cp[live], cv[live]
    # candidates grouped by (chunk, detector); pairs without one cannot trigger
    key = cb.astype(np.int64) * D + cd
    order = np.argsort(key, kind='stable')               # keeps p ascending per pair
    ks, first = np.unique(key[order], return_index=True)
    last = np.append(first[1:], len(order))
    trigs = []                                           # (chunk, det, index, coef)
    gb, gd = ks // D, ks % D
    picked = pick_triggers_grouped(cp[order], cv[order], first, last, Pb[gb], np.asarray(thr_arr, dtype=np.float64)[gd],
                                   np.asarray(maxds, dtype=np.float64)[gb, gd], sr)
    for b, d, found in zip(gb.tolist(), gd.tolist(), picked):
        for (ti, coef) in found:
            if coef > 1.05:                              # detect.py:199-204
                continue
            trigs.append((b, d, ti, coef))
    bad = np.zeros(len(trigs), dtype=bool)               # IndexError of the reference's indexing -> 0.0
    stats = None
    if trigs and bank is not None:
        tab = np.full((len(trigs), 5), -1, dtype=np.int32)
        tab[:, :3] = np.array([t[:3] for t in trigs], dtype=np.int64)
        if not fillZeros:
            Pt = Pb[tab[:, 0]]
            lo_l, bad_l = _centered_lo_many(tab[:, 2], LTA, Pt)
            if STA == 0:
                lo_s, bad_s = tab[:, 2].astype(np.int64), np.zeros(len(trigs), dtype=bool)
            else:
                lo_s, bad_s = _centered_lo_many(tab[:, 2], STA, Pt)
            bad = bad_s | bad_l                          # (the reference's try covers both windows: neither is kept)
            tab[:, 3] = np.where(bad, -1, lo_s)
            tab[:, 4] = np.where(bad, -1, lo_l)
        targs = (xb, M, cstride, ds, tab, 1 if STA == 0 else int(STA), int(LTA),
                 bank.ewf if estimateMags else None, bank.eoff, bank.emax)
        tkw = {} if ctab is None else {'chunk_tab': ctab}
        if hasattr(ctx, 'trigger_stats_async'):
            # the kernel and the copy of its rows to pinned host memory on the context's trigger stream, beside
            # the scans queued behind this batch's; _batch_rows waits for the event
            stats = ctx.trigger_stats_async(*targs, after=after, behind=behind, **tkw)
        else:
            stats = ctx.trigger_stats(*targs, **tkw)
            if stats.is_cuda:
                host = torch.empty(stats.shape, dtype=stats.dtype, pin_memory=True)
                host.copy_(stats, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(_stream_of(ctx, stats.device))
                stats = (host, ev, stats)
    return trigs, bad, stats


def _batch_rows(sta, st, fillZeros, estimateMags, i0, trigs, bad, stats, rows, keys=None):
    """Second host half: the trigger-time estimates -> rows of the detection tables."""
    names, kinds, offsets = st['names'], st['kinds'], st['offsets']
    bounds, n, sr, bank = st['bounds'], st['n'], st['sr'], st['mbank']
    if not trigs:
        return
    if isinstance(stats, tuple):
        stats[1].synchronize()
        stats = stats[0].numpy()
    elif stats is not None:
        stats = stats.cpu().numpy()
    nt = len(trigs)
    pem_a, stm_a, snr_a = np.full(nt, np.nan), np.full(nt, np.nan), np.full(nt, np.nan)
    if estimateMags:                                     # detect.py:422-429, all the batch's triggers at once
        pem_a, stm_a, snr_a = bank.estimate_all([t[1] for t in trigs], n, stats)
    if stats is not None:
        with np.errstate(all='ignore'):
            sl_a = np.divide(stats[:, 0], stats[:, 1])
    # the detectors' largest / smallest offsets (MSTAMPmin / MSTAMPmax): kept with the station's bank, like its constants
    ofmm = getattr(bank, 'offset_minmax', None) if bank is not None else st.get('offset_minmax')
    if ofmm is None:
        ofmm = {nm: (np.max(offsets[nm]), np.min(offsets[nm])) for nm in names}
        if bank is not None:
            bank.offset_minmax = ofmm
        else:
            st['offset_minmax'] = ofmm
    stm_l, snr_l, pem_l = stm_a.tolist(), snr_a.tolist(), pem_a.tolist()
    for r, (b, d, ti, coef) in enumerate(trigs):
        name = names[d]
        start = bounds[i0 + b][2]
        sl = 0.0 if (fillZeros or bad[r]) else float(sl_a[r])     # detect.py:414-420, 289-295
        times = float(ti) / sr + start                   # detect.py:413
        omax, omin = ofmm[name]
        rows[kinds[d]].append([coef, sl, times, name, sta, times - omax, times - omin, stm_l[r], snr_l[r], pem_l[r]])
        if keys is not None:
            keys[kinds[d]].append((bounds[i0 + b][3], d, r))
