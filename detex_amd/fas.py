"""False-alarm statistics (detex/fas.py:23-117 _initFAS / _getDSVect, driven by
SubSpace.getFAS, detex/subspace.py:1652-1743): the detection statistic of every
detector over randomly chosen continuous chunks without high-amplitude signals,
its histogram, and a beta distribution fitted to it by maximum likelihood; the
threshold is then beta.isf(Pf) (subspace.py:1033-1047).

The statistic is the same HIP kernel as the scan (fas.py:120-134 is a twin of
detect.py:559-578).  The beta fit with floc=0, fscale=1 depends on the data only
through n, sum log x and sum log1p(-x) (and the moment-based start values), which
the GPU accumulates in float64 (dtx_ds_moments); the two-parameter root solve is
SciPy's, restated from scipy.stats.beta.fit's fixed-loc/scale branch.
"""
import numpy as np
import scipy.optimize
import scipy.special as sc

from ._lib import DetexHipError


def classic_sta_lta(a, nsta, nlta):
    """obspy.signal.trigger.classic_sta_lta (third-party, restated from its
    documented pure-Python form): ratio of trailing means of a**2."""
    a = np.asarray(a, dtype=np.float64)
    nsta, nlta = int(nsta), int(nlta)
    sta = np.cumsum(a ** 2)
    lta = sta.copy()
    sta[nsta:] = sta[nsta:] - sta[:-nsta]
    sta /= nsta
    lta[nlta:] = lta[nlta:] - lta[:-nlta]
    lta /= nlta
    sta[:nlta - 1] = 0
    dtiny = np.finfo(0.0).tiny
    lta[lta < dtiny] = dtiny
    return sta / lta


def _checkSTALTA(chunk, channels, sr, STATime, LTATime, limit):
    """fas.py:175-205: the Z (else first) channel must not exceed ``limit``."""
    if limit is None:
        return True
    zi = [i for i, c in enumerate(channels) if str(c).upper().endswith('Z')]
    z = chunk[zi[0] if zi else 0]
    if len(z) <= LTATime * sr:
        return False
    return bool(np.max(classic_sta_lta(z, STATime * sr, LTATime * sr)) <= limit)


def beta_mle(n, sx, sxx, s1, s2):
    """scipy.stats.beta.fit(data, floc=0, fscale=1) from sufficient statistics:
    method-of-moments start, then fsolve of the likelihood equations
    s1 = n (psi(a) - psi(a+b)), s2 = n (psi(b) - psi(a+b))."""
    xbar = sx / n
    var = sxx / n - xbar * xbar
    fac = xbar * (1 - xbar) / var - 1
    a0, b0 = xbar * fac, (1 - xbar) * fac

    def eqs(theta):
        a, b = theta
        psiab = sc.psi(a + b)
        return [s1 - n * (-psiab + sc.psi(a)), s2 - n * (-psiab + sc.psi(b))]
    theta, info, ier, mesg = scipy.optimize.fsolve(eqs, [a0, b0], full_output=True)
    if ier != 1:
        raise DetexHipError('beta fit did not converge: %s' % mesg)
    a, b = float(theta[0]), float(theta[1])
    nnlf = -((a - 1) * s1 + (b - 1) * s2 - n * sc.betaln(a, b))
    return (a, b, 0, 1), float(nnlf)


def _declen(cnt, dec):
    return (int(cnt) + dec - 1) // dec


def _log(msg):
    """detex.log(..., pri=True)"""
    import logging
    logging.getLogger('detex_amd.fas').info(msg)


def run_fas(ss, conDatNum, LTATime=5, STATime=0.5, staltalimit=7.5, useSubSpaces=True,
            useSingles=False, numBins=401, recalc=False, device=None, seed=None):
    """fas._initFAS for every station of a SubSpace; stores a dict with 'bins',
    'hist', 'betadist', 'nnlf' in the FAS column of each subspace / single."""
    import torch
    from .getdata import multiplex
    from .hip import HipContext
    rs = np.random.RandomState(seed)
    fetcher = ss.cfetcher
    if fetcher is None:
        raise DetexHipError('no continuous-data fetcher attached (conDatFetcher)')
    bins = np.linspace(-.01, 1, num=numBins)                        # fas.py:31
    from . import parallel
    ctx = HipContext(parallel.local_device() if device is None else device)     # one process per GPU: LOCAL_RANK
    try:
        for sta in ss.Stations:
            names, kinds, U, det_off, _, _ = ss._detector_bank_nothr(sta, useSubSpaces, useSingles)
            if not names or sta not in fetcher.data:
                continue
            # subspace.py:1700-1708, 1722-1729: a FAS that is there already is kept unless recalc is passed

            def have(d):
                df = ss.subspaces[sta] if kinds[d] == 'ss' else ss.singles[sta]
                return isinstance(df.loc[df.index[df.Name == names[d]][0], 'FAS'], dict)
            todo = [d for d in range(len(names)) if recalc or not have(d)]
            if not todo:
                _log('FAS for station %s already calculated, to recalculate pass True to the parameter recalc'
                     % sta)
                continue
            Nc = len(fetcher.channels)
            sr = fetcher.sampling_rate
            n = U.shape[1]
            counts = getattr(fetcher, 'counts', {}).get(sta)
            data, _t0 = fetcher.data[sta]
            raw = bool(getattr(fetcher, 'raw', False))
            dec = int(getattr(ss.clusters, 'decimate', None) or 1) if raw else 1
            sr_raw, sr = sr, sr / dec                              # detect.py:239
            dropped = set()
            if counts is not None:
                # Raw count fragments (ContinuousFetcher.add_counts: the record as it came off the digitiser).  The
                # reference's FAS filters every sampled chunk like a scanned one -- fas.py:101: _applyFilter(st, filt,
                # deci, dtype), i.e. fillZeros=False: merged, cut to its longest stretch without a gap, dropped when
                # still fractured (construct.py:1003-1023) -- and a chunk that comes back empty still counts as drawn
                # (fas.py:99-103).  The same front half the scan runs (detect.StationScan): one ingest of the
                # station's fragments, then the per-chunk trimming on the device.
                from . import detect
                T = int(data.shape[1])
                planes = detect._ingest_station(ctx, counts, Nc, T,
                                                pinned=getattr(fetcher, 'counts_pinned', {}).get(sta))
                every = fetcher.chunk_bounds(sta)
                kept = detect._trim_to_gapfree(ctx, planes, counts,
                                               [(b[0], b[1], b[2], i) for i, b in enumerate(every)], sr_raw, sta)
                kept = {b[3]: (b[0], b[1], b[2]) for b in kept if _declen(b[1], dec) * Nc > n}
                if not kept:
                    raise DetexHipError('Could not get any data for %s' % sta)       # fas.py:113-115
                # (the draw is over ALL chunks, as the reference's getConData(randSamps=...) draws before it filters)
                drawn = [int(k) for k in rs.permutation(len(every))[:conDatNum * 4]]
                dropped = set(k for k in drawn if k not in kept)
                bounds = {k: kept[k] for k in drawn if k in kept}
                order = [k for k in drawn if k in kept]
                order_all = drawn
            else:
                bounds = [b for b in fetcher.chunk_bounds(sta) if _declen(b[1], dec) * Nc > n]
                if not bounds:
                    raise DetexHipError('Could not get any data for %s' % sta)       # fas.py:113-115
                # fas.py:92-93: ask for 4x more chunks than needed, in random order
                order = [int(k) for k in rs.permutation(len(bounds))[:conDatNum * 4]]
                order_all = order
                pinned = getattr(fetcher, 'pinned', {}).get(sta)
                planes = (pinned if pinned is not None else
                          torch.from_numpy(np.ascontiguousarray(data, dtype=np.float32))).to(ctx.device, non_blocking=True)
            zs = [i for i, c in enumerate(fetcher.channels) if str(c).upper().endswith('Z')]
            zi = zs[0] if zs else 0                                   # fas.py:175-205: the Z (else first) channel
            nsta, nlta = int(STATime * sr), int(LTATime * sr)
            T = int(planes.shape[1])
            # candidate chunks on the device, multiplexed, grouped by length: rows[M] = (tensor [G, M], [chunk ids])
            groups = {}
            if raw:
                # every candidate is decimated / detrended / band-passed like a scanned chunk (fas.py:101 ->
                # _applyFilter); chunks of one length go through the kernels as one batch
                bylen = {}
                for k in order:
                    bylen.setdefault(int(bounds[k][1]), []).append(k)
                for cnt, ks in bylen.items():
                    pl = torch.stack([planes[:, bounds[k][0]:bounds[k][0] + cnt] for k in ks])
                    if dec > 1:
                        pl = ctx.decimate(pl, dec, sr_raw)
                    mp = ctx.preprocess(pl, filt=ss.clusters.filt, sampling_rate=sr)
                    groups.setdefault(int(mp.shape[1]), []).append((mp, ks))
                ratio = {}
                for M, lst in groups.items():
                    for mp, ks in lst:
                        r = ctx.stalta_max(mp, [g * M + zi for g in range(len(ks))], [M // Nc] * len(ks), nsta, nlta,
                                           stride=Nc)
                        ratio.update(zip(ks, r.cpu().numpy()))
                lens = {k: _declen(bounds[k][1], dec) for k in order}
            else:
                # the screen reads the Z plane where it lies; the scan takes the interleaved record
                # (construct.py:977-978; dtx_preprocess without filter or detrend)
                r = ctx.stalta_max(planes, [zi * T + bounds[k][0] for k in order], [bounds[k][1] for k in order],
                                   nsta, nlta)
                xdev = ctx.preprocess(planes[None], filt=None, detrend=False)[0]
                ratio = dict(zip(order, r.cpu().numpy()))
                lens = {k: int(bounds[k][1]) for k in order}

            def ok(k, limit):
                # fas._checkSTALTA: a chunk not longer than the LTA window is rejected, else max(cft) <= limit
                return lens[k] > LTATime * sr and ratio[k] <= limit

            def pick(limit):
                chosen, count = [], 0
                for k in order_all:
                    count += 1
                    if k in dropped or (limit is not None and not ok(k, limit)):
                        continue
                    if len(chosen) >= conDatNum:
                        break
                    chosen.append(k)
                return chosen, count
            chosen, count = pick(staltalimit)
            if count and float(len(chosen)) / count <= .25:           # fas.py:60-68
                chosen, count = pick(None)
            ctx.upload_basis(U, det_off, Nc)
            D = len(names)
            hist = torch.zeros((D, numBins - 1), dtype=torch.int32, device=ctx.device)
            sums = torch.zeros((D, 5), dtype=torch.float64, device=ctx.device)
            # chunks of equal length go through the scan together (one launch fills the chip
            # better than one launch per chunk); batches bounded to ~4 GB of DS
            sel = set(chosen)
            batches = []                                             # (tensor [G, M] multiplexed chunks)
            if raw:
                for M, lst in groups.items():
                    for mp, ks in lst:
                        idx = [g for g, k in enumerate(ks) if k in sel]
                        if idx:
                            batches.append(mp if len(idx) == len(ks) else
                                           mp[torch.tensor(idx, device=mp.device)])
            else:
                bylen = {}
                for k in chosen:
                    bylen.setdefault(int(bounds[k][1]), []).append(k)
                for cnt, ks in sorted(bylen.items()):
                    batches.append(torch.stack([xdev[bounds[k][0] * Nc:(bounds[k][0] + cnt) * Nc] for k in ks]))
            for mp in batches:
                M = int(mp.shape[1])
                per = max(1, int(4e9 // (4.0 * D * max(1, (M - n) // Nc + 1))))
                for i in range(0, int(mp.shape[0]), per):
                    ds = ctx.subspace_ds(mp[i:i + per])
                    ctx.ds_moments(ds, lo=-.01, hi=1.0, nbins=numBins - 1, hist=hist, sums=sums)
                    del ds
            ctx.sync()
            h = hist.cpu().numpy()
            sm = sums.cpu().numpy()
            for d, name in enumerate(names):
                if d not in todo:
                    continue
                params, nnlf = beta_mle(*sm[d])
                rec = {'bins': bins, 'hist': h[d], 'betadist': params, 'nnlf': nnlf}
                df = ss.subspaces[sta] if kinds[d] == 'ss' else ss.singles[sta]
                ind = df.index[df.Name == name][0]
                df.at[ind, 'FAS'] = rec
    finally:
        ctx.close()
