"""createCluster / createSubSpace with the reference's names and argument meaning
(detex/construct.py:25-171, 177-301), on in-memory pre-processed waveforms.

The numerics seam is ``_makeDFcclags`` (detex/construct.py:369-394): the
reference's Python pair loop is replaced by ONE call into the HIP library
(dtx_xcorr_allpairs); everything around it (dissimilarity, SciPy single linkage,
dendrogram cut, alignment by accumulated lags, Harris 2006 App. B) is host logic
on tiny arrays and follows the reference line by line.
"""
from collections.abc import Mapping

import numpy as np
import pandas as pd
from scipy.cluster.hierarchy import linkage

from ._lib import DetexHipError
from .getdata import EventFetcher, clip_fragments, multiplex, trim_window


def _error(msg, exc=DetexHipError):
    """detex.log(level='error') raises (detex/__init__.py:131-138)."""
    raise exc(msg)


# --------------------------------------------------------------------------
# event loading                                   construct.py:615-706
# --------------------------------------------------------------------------

def _warn(msg):
    """detex.log(level='warning', pri=True)."""
    import logging
    logging.getLogger('detex_amd.construct').warning(msg)


def _preprocess_events(ctx, fetcher, sta, filt, trim, decimate, fillZeros=False):
    """Raw template records of one station -> filtered waveforms, on the device: what the reference's
    ``fetcher.getTemData(temkey, csta, trim[0], trim[1])`` + ``_applyFilter(st, filt, decimate, dtype)``
    do per event (construct.py:866-869, 990-1030; getdata.py:441-444), in the reference's order:

      cut [origin - trim[0], origin + trim[1]] out of the record (ObsPy trim, getdata.trim_window)
      -> a record in several traces per channel: merged with zero fill and cut to its longest stretch
         without a gap or a zero in any channel (_mergeChannels, 1033-1066; dtx_ingest_counts +
         dtx_gapfree_run), dropped if that stretch is still in several traces ('Stream is too fractured',
         1011-1018); fillZeros: merged only (_mergeChannelsFill)
      -> decimate (1016-1017, dtx_decimate) -> detrend('linear') + band-pass (1024-1028, dtx_preprocess)

    Events of equal length go through the kernels as one batch.  Returns [(name, [Nc, L'] float32 array,
    time of the first sample)] in event-name order."""
    import torch
    from .detect import _trim_to_gapfree
    sr = fetcher.sampling_rate
    Nc = len(fetcher.channels)
    dec = int(decimate or 1)
    cut = []                                   # (name, [Nc, cnt] host array or device planes, start time)
    for name, rec in fetcher.getTemData(sta):
        t0, T = rec['starttime'], rec['nsamples']
        i0, cnt = trim_window(t0, T, sr, rec['origintime'] - trim[0], rec['origintime'] + trim[1])
        if cnt <= 0:
            _warn('no data on %s around %s' % (sta, name))
            continue
        if rec['counts'] is None:
            cut.append((name, np.asarray(rec['data'][:, i0:i0 + cnt], dtype=np.float32), t0 + i0 / sr))
            continue
        # the traces that reach into the window, clipped to it and re-based to its first sample
        frs = clip_fragments(rec['counts'], [(i0, i0 + cnt, 0)])
        off, tab = 0, []
        for (c, f0, x) in frs:
            tab.append((c, off, f0, len(x)))
            off += len(x)
        flat = np.concatenate([x for (_, _, x) in frs]) if frs else np.zeros(0, np.int32)
        planes, _ = ctx.ingest_counts(torch.from_numpy(flat).to(ctx.device), tab, Nc, cnt, want_run=False)
        bounds = [(0, cnt, t0 + i0 / sr, 0)]
        if fillZeros:
            if len(set(c for (c, _, _) in frs)) < Nc:
                _warn('%s on %s is fractured or channels are missing' % (name, sta))
                continue
        else:
            bounds = _trim_to_gapfree(ctx, planes, frs, bounds, sr, sta)
        if not bounds:
            _warn('%s on %s is fractured or channels are missing, skipping' % (name, sta))   # construct.py:878-883
            continue
        s0, m, ts, _ = bounds[0]
        cut.append((name, planes[:, s0:s0 + m], ts))
    # batches of equal length through decimate / detrend / band-pass / multiplex
    out = {}
    groups = {}
    for k, (name, arr, ts) in enumerate(cut):
        groups.setdefault(int(arr.shape[1]), []).append(k)
    for m, ks in groups.items():
        if all(isinstance(cut[k][1], np.ndarray) for k in ks):
            planes = torch.from_numpy(np.stack([cut[k][1] for k in ks])).to(ctx.device)     # one upload
        else:
            planes = torch.stack([cut[k][1] if torch.is_tensor(cut[k][1])
                                  else torch.from_numpy(cut[k][1]).to(ctx.device) for k in ks])
        if dec > 1:
            planes = ctx.decimate(planes, dec, sr)
        mp = ctx.preprocess(planes, filt=list(filt) if isinstance(filt, (list, tuple)) else None,
                            sampling_rate=sr / dec)
        ctx.sync()
        mp = mp.cpu().numpy()
        Lm = mp.shape[1] // Nc
        for row, k in zip(mp, ks):
            out[cut[k][0]] = (row.reshape(Lm, Nc).T, cut[k][2])
    return [(name, out[name][0], out[name][1]) for (name, _, _) in cut]


class _LazyMPtd(Mapping):
    """{event: multiplexed array} of a station (the reference's ``MPtd`` column, construct.py:639-655), multiplexed
    when an event is ASKED for: createCluster itself only needs the event matrix on the device, which is
    assembled from the channel planes (``planes_f32``) and interleaved there -- at 2000 events the host-side
    multiplex of every event cost more than the correlation kernels.  ``le``: the common length the station's
    events are trimmed to (_testStreamLengths, construct.py:679-694), in multiplexed samples."""

    def __init__(self, planes, le, dtype):
        self._planes, self._le, self._dtype, self._done = planes, int(le), dtype, {}

    def __getitem__(self, ev):
        if ev not in self._done:
            self._done[ev] = multiplex(list(self._planes[ev]), dtype=self._dtype)[:self._le]
        return self._done[ev]

    def __iter__(self):
        return iter(self._planes)

    def __len__(self):
        return len(self._planes)

    def __getstate__(self):
        """Pickled without the multiplexed arrays built so far (ClusterStream.write after createSubSpace had read
        every event wrote both copies: twice the file) -- they are rebuilt when asked for."""
        d = dict(self.__dict__)
        d['_done'] = {}
        return d

    def length(self, ev):
        """len(self[ev]) without building it"""
        p = self._planes[ev]
        return min(self._le, int(p.shape[0]) * int(p.shape[1]))

    def planes_f32(self, ev, out):
        """The event's channel planes, trimmed like the multiplexed array, into out [Nc, le / Nc] float32."""
        p = self._planes[ev]
        np.copyto(out, p[:, :out.shape[1]], casting='unsafe')


def _loadEvents(fetcher, dtype, filt=None, trim=None, decimate=None, enforceOrigin=False, fillZeros=False,
                ctx=None):
    """construct.py:615-655 (+ _loadStream 852-925) on an EventFetcher: one row per station with
    Events (sorted), MPtd {event: multiplexed float array}, Channels, Stats.  Raw records
    (EventFetcher.add_raw / add_counts) are cut, decimated and filtered on the device first
    (_preprocess_events); waveforms added with ``add`` are taken as already processed."""
    np_dtype = np.float32 if dtype == 'single' else np.float64
    rows = []
    Nc = len(fetcher.channels)
    for sta in fetcher.stations():
        mptd, stats, chans = {}, {}, {}
        raw = fetcher.has_raw(sta)
        if raw:
            if not all(rec.get('raw') for _, rec in fetcher.getTemData(sta)):
                _error('station %s mixes raw and pre-processed events' % sta, ValueError)
            if ctx is None:
                _error('raw template events need a GPU context')
            done = _preprocess_events(ctx, fetcher, sta, filt, trim, decimate, fillZeros)
            sr = fetcher.sampling_rate / int(decimate or 1)
        else:
            done = [(name, rec['data'], rec['starttime']) for name, rec in fetcher.getTemData(sta)]
            sr = fetcher.sampling_rate
        allzeros, lens, planes = [], {}, {}
        for name, data, start in done:
            rec = fetcher.events[sta][name]
            data = np.asarray(data)
            if enforceOrigin and raw:
                # construct.py:884-885: st.trim(starttime=originTime, pad=True, fill_value=0.0)
                k = int(round((rec['origintime'] - start) * sr))
                data = data[:, k:] if k >= 0 else np.concatenate(
                    [np.zeros((data.shape[0], -k), dtype=data.dtype), data], axis=1)
                start = start + k / sr
            if data.shape[1] == 0:
                continue
            # construct.py:896-897: a channel that is all zeros (looked for only where a channel STARTS with a zero)
            if not np.all(data[:, 0]) and not data.any(axis=1).all():
                allzeros.append(name)
            planes[name] = data                                   # [Nc, L]: multiplexed when asked for (_LazyMPtd)
            lens[name] = data.size                                # STlens: total samples of the stream
            stats[name] = dict(Nc=Nc, sampling_rate=sr, starttime=start, origintime=rec['origintime'],
                               magnitude=rec['magnitude'], offset=start - rec['origintime'])
            chans[name] = list(fetcher.channels)
        if raw and planes:
            # construct.py:901-916: fractured / very short waveforms (< 20 % of the median length) and
            # waveforms with an all-zero channel are removed
            mlen = np.median(list(lens.values()))
            for key in [k for k in planes if lens[k] < mlen * .2]:
                _warn('%s is fractured or missing data, removing' % key)
                planes.pop(key), stats.pop(key), chans.pop(key)
            for key in set(allzeros):
                if key in planes:
                    _warn('%s has at least one channel that is all zeros, deleting' % key)
                    planes.pop(key), stats.pop(key), chans.pop(key)
            if len(planes) < 2:                                   # construct.py:918-922
                _warn('Less than 2 events survived preprocessing for station %s' % sta)
                continue
        if not planes:
            continue
        # construct.py:679-694 _testStreamLengths: trim to the smallest length that is
        # within 90 % of the median, drop shorter events (lengths of the multiplexed arrays)
        mlens = {k: int(v.shape[0]) * int(v.shape[1]) for k, v in planes.items()}
        lv = np.array(list(mlens.values()))
        le = int(np.min(lv[lv > np.median(lv) * .9]))
        events = sorted(k for k in planes if mlens[k] >= le)
        mptd = _LazyMPtd({k: planes[k] for k in events}, le, np_dtype)
        rows.append(dict(Station=sta, Events=events, MPtd=mptd,
                         Channels={k: chans[k] for k in events},
                         Stats={k: stats[k] for k in events},
                         Link=None, CCs=None, Lags=None, Subsamp=None, numEvents=len(events)))
    TRDF = pd.DataFrame(rows, dtype=object)
    if len(TRDF):
        TRDF = TRDF.sort_values(by='Station').reset_index(drop=True)
    return TRDF


# --------------------------------------------------------------------------
# the numerics seam                                construct.py:369-394
# --------------------------------------------------------------------------

def _frames_from_condensed(N, cc, lag, ss):
    """Condensed (b < c, row-major) vectors -> DFcc, DFlag, DFsubsamp with the
    reference's layout: index 0..N-2, columns 1..N-1, NaN below the diagonal
    (construct.py:373-378)."""
    cols = np.arange(1, N)
    idx = np.arange(0, N - 1)
    # pair (b, c), b < c, sits at [b, c - 1]: the upper triangle (diagonal included) of the frame in row-major
    # order IS the condensed order -- one masked assignment (a fancy-indexed one took 0.2 s per frame at 2000 events)
    upper = np.triu(np.ones((N - 1, N - 1), dtype=bool))
    out = []
    for v in (cc, lag, ss):
        m = np.full((N - 1, N - 1), np.nan, dtype=np.float64)
        m[upper] = np.asarray(v, dtype=np.float64)
        out.append(pd.DataFrame(m, columns=cols, index=idx))
    return out[0], out[1], out[2]


# seconds per phase of the last createCluster call in this process (ClusterStream.timings; bench.py's
# e2e_createCluster_2000 leg): plain perf_counter stamps around the phases, device work synchronised at their ends
_PHASES = {}


class _phase(object):
    def __init__(self, name):
        self.name = name

    def __enter__(self):
        import time
        self.t0 = time.perf_counter()

    def __exit__(self, *exc):
        import time
        _PHASES[self.name] = _PHASES.get(self.name, 0.0) + time.perf_counter() - self.t0
        return False


def _sync_copy(ctx):
    """Wait for the copies torch queued on its current stream of the context's device (phase timing only)."""
    import torch
    dev = getattr(ctx, 'device', None)
    if dev is not None and str(dev).startswith('cuda') and torch.cuda.is_available():
        torch.cuda.current_stream(dev).synchronize()


def _xcorr_condensed(eventList, row, ctx):
    """construct.py:369-394 on the GPU: all pairs b < c of ``eventList`` in one dtx_xcorr_allpairs call.
    Returns (N, cc, lag, subsamp) as condensed NumPy vectors (pair (b, c), b < c, row-major: the order of the
    reference's double loop)."""
    import torch
    if len(set(len(row.Channels[e]) for e in eventList)) != 1:
        _error('Number of Channels not equal, cannot perform correlation')      # 429-431
    lazy = isinstance(row.MPtd, _LazyMPtd)
    if len(set((row.MPtd.length(e) if lazy else len(row.MPtd[e])) for e in eventList)) != 1:
        _error('Lengths not equal on multiplexed data, cannot correlate')       # 433-436
    from . import parallel
    Nc = len(row.Channels[eventList[0]])
    N = len(eventList)
    with _phase('fill'):
        # The event matrix, assembled in page-locked memory and uploaded slab by slab: a slab is on its way to the
        # device -- at the link's rate, not through a pageable staging copy -- while the next one is copied (and
        # converted) event by event.  (Round 5: all events, then one upload: 19 + 6 ms at 2000 events; a list of 2000
        # float64 arrays through ascontiguousarray took 1.2 s in round 4.  Four threads copying, round 6: 54 ms -- 2000
        # short copies hand the interpreter lock back and forth.)
        n = row.MPtd.length(eventList[0]) if lazy else len(row.MPtd[eventList[0]])
        lazy = lazy and n % Nc == 0
        on_gpu = True
        try:
            tdt = torch.empty((N, n), dtype=torch.float32, pin_memory=True)
        except RuntimeError:                                                # (no GPU runtime: the CPU stand-in of the tests)
            tdt = torch.empty((N, n), dtype=torch.float32)
            on_gpu = False
        td = tdt.numpy()
        td3 = td.reshape(N, Nc, n // Nc) if lazy else None

        def copy_rows(a, b):
            for i in range(a, b):
                if lazy:
                    # channel planes [N][Nc][n / Nc]: one contiguous copy per channel; interleaved on the device
                    # below (construct.py:977-978; dtx_preprocess without filter or detrend)
                    row.MPtd.planes_f32(eventList[i], td3[i])
                else:
                    td[i] = row.MPtd[eventList[i]]
        nslab = 8 if (on_gpu and N >= 64) else 1
        tdev = torch.empty((N, n), dtype=torch.float32, device=ctx.device) if on_gpu else None
        for sl in range(nslab):
            a, b = sl * N // nslab, (sl + 1) * N // nslab
            copy_rows(a, b)
            if on_gpu:
                tdev[a:b].copy_(tdt[a:b], non_blocking=True)
    with _phase('h2d'):
        if not on_gpu:
            tdev = tdt.to(ctx.device, non_blocking=True)
        _sync_copy(ctx)
    rank, world = parallel.dist_info()
    if parallel.dist_active():
        # one process per GPU: every rank correlates a contiguous range of the condensed pair
        # list (every pair costs the same), the slices are all-gathered
        npairs = N * (N - 1) // 2
        ranges = [parallel.shard_pairs(npairs, r, world) for r in range(world)]
        b, e = ranges[rank]
        with _phase('kernels'):
            if lazy:
                tdev = ctx.preprocess(tdev.reshape(N, Nc, n // Nc), filt=None, detrend=False)
            cc, lag, ss = ctx.xcorr_allpairs(tdev, Nc, b, e)
            ctx.sync()
        cc, lag, ss = (parallel.gather_condensed(v, ranges, npairs, ctx.device) for v in (cc, lag, ss))
    else:
        with _phase('kernels'):
            if lazy:
                tdev = ctx.preprocess(tdev.reshape(N, Nc, n // Nc), filt=None, detrend=False)
            cc, lag, ss = ctx.xcorr_allpairs(tdev, Nc)
            ctx.sync()
    with _phase('d2h'):
        cc, lag, ss = cc.cpu().numpy(), lag.cpu().numpy(), ss.cpu().numpy()
    return N, cc, lag, ss


def _makeDFcclags(eventList, row, ctx):
    """construct.py:369-394: (DFcc, DFlag, DFsubsamp) of all pairs b < c of ``eventList``."""
    N, cc, lag, ss = _xcorr_condensed(eventList, row, ctx)
    with _phase('frames'):
        return _frames_from_condensed(N, cc, lag, ss)


def fast_normcorr(t, s, ctx=None, device=None):
    """detex/construct.py:469-483 on the GPU (dtx_normcorr): normalised correlation of the
    shorter array with every same-length window of the longer one; NumPy in, NumPy out
    (float32 arithmetic inputs, float64 accumulation)."""
    import torch
    from .hip import HipContext
    own = ctx is None
    if own:
        from . import parallel
        ctx = HipContext(parallel.local_device() if device is None else device)
    try:
        tt = torch.from_numpy(np.ascontiguousarray(t, dtype=np.float32)).to(ctx.device)
        st = torch.from_numpy(np.ascontiguousarray(s, dtype=np.float32)).to(ctx.device)
        out = ctx.normcorr(tt, st)
        ctx.sync()
        return out.cpu().numpy().astype(np.float64)
    finally:
        if own:
            ctx.close()


def _flatNoNan(df):
    """construct.py:702-708."""
    ar = np.asarray(df.values, dtype=np.float64).flatten()
    return ar[~np.isnan(ar)]


# --------------------------------------------------------------------------
# createCluster                                    construct.py:25-171
# --------------------------------------------------------------------------

def _checkClusterInputs(filt, dtype, trim, decimate):
    """construct.py:304-... argument validation."""
    if filt is not None and len(filt) != 4:
        _error('filt must be a list of length 4: [freqmin, freqmax, corners, zerophase]',
               ValueError)
    if dtype not in ('double', 'single'):
        _error('dtype must be "double" or "single"', ValueError)
    if trim is not None and len(trim) != 2:
        _error('trim must be a list of length 2', ValueError)
    if decimate is not None and (not isinstance(decimate, int) or decimate < 1):
        _error('decimate must be None or a positive int', ValueError)


_WARNED_DOUBLE = [False]


def _warn_double_once(dtype):
    """construct.py:34, 199-201: dtype='double' is the reference's default.  The device path computes in
    float32 (float64 window statistics and accumulations) whatever is asked for."""
    if dtype == 'double' and not _WARNED_DOUBLE[0]:
        _WARNED_DOUBLE[0] = True
        _warn("dtype='double': host arrays are float64, the HIP kernels compute in float32 with float64 "
              "accumulations (detection statistics and CC coefficients within 1e-5 of the float64 path)")


def createCluster(CCreq=0.5, fetch_arg=None, filt=[1, 10, 2, True], stationKey=None,
                  templateKey=None, trim=[10, 120], saveclust=True, fileName='clust.pkl',
                  decimate=None, dtype='double', eventsOnAllStations=False,
                  enforceOrigin=False, fillZeros=False, phases=None, device=None, ctx=None):
    """Create a ClusterStream (detex/construct.py:25-171).

    ``fetch_arg`` is an ``EventFetcher``.  Raw records in it (``add_raw`` / ``add_counts``) are cut to
    [origin - trim[0], origin + trim[1]], decimated by ``decimate``, detrended and band-passed with ``filt``
    on the device before they are correlated (construct.py:852-925, 990-1030); waveforms it holds as already
    processed (``add``) are taken as they are, and ``filt`` / ``trim`` / ``decimate`` then only describe how
    they WERE processed (they are still what SubSpace.detex applies to raw continuous data).  Outside the
    hot path: ``stationKey`` / ``templateKey`` / ``phases`` are accepted and stored, the station and event
    lists come from the fetcher (ObsPy I/O is out of scope).  ``dtype='double'`` (the reference's default)
    selects float64 host arrays; the device arithmetic is float32 with float64 accumulations, within 1e-5
    of the float64 path (BASELINE north_star) -- warned about once.  ``device`` / ``ctx`` select the GPU."""
    from .subspace import ClusterStream
    from .hip import HipContext
    _checkClusterInputs(filt, dtype, trim, decimate)
    if not isinstance(fetch_arg, EventFetcher):
        _error('fetch_arg must be a detex_amd.getdata.EventFetcher', TypeError)
    _warn_double_once(dtype)
    own = ctx is None
    if own:
        if device is None:
            # one process per GPU: the launcher's LOCAL_RANK names this process's device
            from . import parallel
            device = parallel.local_device()
        ctx = HipContext(device)
    import time
    _PHASES.clear()
    t_all = time.perf_counter()
    try:
        with _phase('load_events'):
            TRDF = _loadEvents(fetch_arg, dtype, filt, trim, decimate, enforceOrigin, fillZeros, ctx)
        if len(TRDF) < 1:
            _error('No events survived pre-processing, check DataFetcher and event quality')
        if eventsOnAllStations:
            eventList = sorted(set.intersection(*[set(x) for x in TRDF.Events]))
            if len(eventList) < 2:
                _error('less than 2 events in population have required stations')
        for ind, row in TRDF.iterrows():
            if not eventsOnAllStations:
                eventList = list(row.Events)
            if len(row.Events) < 2:       # construct.py:143-146: skip the station
                continue
            N, cc, lag, ss = _xcorr_condensed(eventList, row, ctx)
            # the three frames (construct.py:373-378; 3 x 32 MB at 2000 events) are laid out by a second thread while
            # this one clusters: the condensed vector IS the frame's upper triangle in row-major order, i.e. what
            # _flatNoNan(1.0000001 - DFcc) (construct.py:153-156) flattens it back to
            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(max_workers=1) as ex:
                fut = ex.submit(_frames_from_condensed, N, cc, lag, ss)
                with _phase('linkage'):
                    cx = 1.0000001 - np.asarray(cc, dtype=np.float64)
                    cx = cx[~np.isnan(cx)]
                    TRDF.at[ind, 'Link'] = linkage(cx)         # default: single
                with _phase('frames'):
                    DFcc, DFlag, DFsubsamp = fut.result()
            TRDF.at[ind, 'Lags'] = DFlag
            TRDF.at[ind, 'CCs'] = DFcc
            TRDF.at[ind, 'Subsamp'] = DFsubsamp
    finally:
        if own:
            ctx.close()
    colstk = ['Station', 'Link', 'CCs', 'Lags', 'Subsamp', 'Events', 'Stats']
    keep = TRDF[[x is not None for x in TRDF.Link]].reset_index(drop=True)
    if len(keep) < 1:
        _error('Less than 2 valid events on every station')
    eventListAll = sorted(set.union(*[set(x) for x in keep.Events]))
    with _phase('cluster_stream'):                             # Cluster objects: updateReqCC on every station
        clust = ClusterStream(keep[colstk], templateKey, stationKey, fetch_arg, eventListAll,
                              CCreq, filt, decimate, trim, fileName, eventsOnAllStations,
                              enforceOrigin)
    clust._mptd = {r.Station: r.MPtd for _, r in TRDF.iterrows()}
    clust._channels = {r.Station: r.Channels for _, r in TRDF.iterrows()}
    if saveclust:
        from . import parallel
        if parallel.dist_info()[0] == 0:       # one process per GPU: rank 0 writes the pickle
            with _phase('write_pickle'):
                clust.write()
    clust.timings = dict(_PHASES, total=time.perf_counter() - t_all)
    return clust


# --------------------------------------------------------------------------
# alignment (Harris 2006 App. B)                   construct.py:710-812
# --------------------------------------------------------------------------

def _triangular(n):
    return n * (n + 1) // 2


def _getClustDict(link):
    """construct.py:800-812: members (leaf indices) of every dendrogram node."""
    N = len(link)
    d = {i: [i] for i in range(N + 1)}
    for a in range(N):
        d[N + 1 + a] = d[int(link[a, 0])] + d[int(link[a, 1])]
    return d


def _getDelays(link, cx, lags, n_events):
    """construct.py:710-760 (_getDelays + _traceEventDendro + _updateLags).

    link : SciPy single linkage of the condensed dissimilarities ``cx``;
    lags : condensed pair lags (multiplexed samples, positive = second event
    later).  Walk the merges in order; the pair whose dissimilarity equals the
    merge height supplies the lag between the two merged groups; every event of
    the second group is shifted by it and all pair lags touching that event are
    updated.  Returns the accumulated shift per event (array of n_events)."""
    N = n_events
    cx = np.asarray(cx, dtype=np.float64)
    lags = np.array(lags, dtype=np.float64)
    iu = np.triu_indices(N, 1)
    pair_of = {}                            # condensed position -> (b, c)
    for k, (b, c) in enumerate(zip(iu[0], iu[1])):
        pair_of[k] = (int(b), int(c))
    pos = {bc: k for k, bc in pair_of.items()}
    clust = _getClustDict(link)
    total = np.zeros(N, dtype=np.int64)
    for a in range(len(link)):
        i1, i2, height = int(link[a, 0]), int(link[a, 1]), link[a, 2]
        # construct.py:727-729: first (row, column) whose dissimilarity == merge height
        k = int(np.nonzero(cx == height)[0][0])
        ev1, _ev2 = pair_of[k]
        cl22 = clust[i2] if ev1 in clust[i1] else clust[i1]          # 747-750
        currentLag = int(np.round(lags[k]))                           # 752
        for b in cl22:                                                # 754-756
            total[b] += currentLag
            for c in range(b + 1, N):     # _getAcr: pairs (b, c): lag += currentLag
                lags[pos[(b, c)]] += currentLag
            for r in range(0, b):         # _getDow: pairs (r, b): lag -= currentLag
                lags[pos[(r, b)]] -= currentLag
    return total


def _alignTD(sample_delays, events, mptd):
    """construct.py:486-504: slice each multiplexed event by its delay."""
    TDlengths = len(mptd[events[0]]) - int(np.max(sample_delays))
    aligned = {}
    for ev, d in zip(events, sample_delays):
        orig = mptd[ev][int(d):][:TDlengths]
        if len(orig) == 0:
            _error('Alignment of multiplexed stream failing, try raising ccreq or '
                   'widening trim window')
        aligned[ev] = orig
    return aligned


def _sub_condensed(df, sel):
    """construct.py:304-320 _getInfoFromClust: CC / lag sub-matrix of the events
    ``sel`` (positions in the station's event list, ascending) in condensed order."""
    vals = np.asarray(df.values, dtype=np.float64)
    out = []
    for a in range(len(sel)):
        for b in range(a + 1, len(sel)):
            out.append(vals[sel[a], sel[b] - 1])
    return np.array(out)


class _CastView(object):
    """mptd[ev] -> np.asarray(source[ev], dtype), remembered"""

    def __init__(self, src, dtype):
        self.src, self.dtype, self.done = src, dtype, {}

    def __getitem__(self, ev):
        if ev not in self.done:
            self.done[ev] = np.asarray(self.src[ev], dtype=self.dtype)
        return self.done[ev]


def createSubSpace(Pf=10 ** -12, clust=None, minEvents=2, dtype='double', conDatFetcher=None):
    """Create a SubSpace from a ClusterStream (detex/construct.py:177-301): for
    every cluster with >= minEvents events re-use the stored CC / lag values (no
    recomputation, construct.py:270), re-link, accumulate the alignment delays
    through the dendrogram and slice the waveforms; unclustered events become
    singles (construct.py:525-559)."""
    from .subspace import ClusterStream, SubSpace
    if not isinstance(clust, ClusterStream):
        _error('Invalid clust type, must be a ClusterStream instance.', ValueError)
    cl = clust
    np_dtype = np.float32 if dtype == 'single' else np.float64
    ssDict, singDic = {}, {}
    for _, trow in cl.trdf.iterrows():
        sta = trow.Station
        events_all = list(trow.Events)
        src = cl._mptd[sta]
        mptd = _CastView(src, np_dtype)                             # (cast on access: only what the clusters use)
        stats_all = trow.Stats
        c = cl[sta]
        rows = []
        for num, evelist in enumerate(c.clusts):
            evelist = sorted(evelist)                               # construct.py:584-585
            if len(evelist) < minEvents:
                continue
            sel = [events_all.index(e) for e in evelist]            # _makeEventListKey
            cx = 1.0000001 - _sub_condensed(trow.CCs, sel)
            lags = _sub_condensed(trow.Lags, sel)
            link = linkage(cx)
            delays = _getDelays(link, cx, lags, len(evelist))
            sdel = delays - np.min(delays)                          # construct.py:282-283
            aligned = _alignTD(sdel, evelist, mptd)
            stats = {}
            for ev, d in zip(evelist, sdel):                        # _updateStartTimes 346-366
                s = dict(stats_all[ev])
                s['starttime'] = s['starttime'] + d / (s['sampling_rate'] * s['Nc'])
                s['offset'] = s['starttime'] - s['origintime']
                stats[ev] = s
            offs = [stats[e]['offset'] for e in evelist]
            rows.append(dict(Name='SS%d' % num, Station=sta, Events=evelist, AlignedTD=aligned,
                             SVD=None, UsedSVDKeys=None, FracEnergy=None, SVDdefined=False,
                             SampleTrims={}, Threshold=np.nan, FAS=None, NumBasis=0,
                             Offsets=[np.min(offs), np.median(offs), np.max(offs)],
                             Stats=stats, Channels={e: cl._channels[sta][e] for e in evelist},
                             numEvents=len(evelist)))
        if rows:
            ssDict[sta] = pd.DataFrame(rows, dtype=object)
        srows = []
        for sn, ev in enumerate(c.singles):                         # construct.py:525-559
            srows.append(dict(Name='SG%d' % sn, Station=sta, Events=[ev],
                              MPtd={ev: mptd[ev]}, Stats={ev: dict(stats_all[ev])},
                              Channels={ev: cl._channels[sta][ev]}, SampleTrims={}, FAS=None,
                              Threshold=np.nan,
                              Offsets=[stats_all[ev]['offset']] * 3))
        if srows:
            singDic[sta] = pd.DataFrame(srows, dtype=object)
    return SubSpace(singDic, ssDict, cl, dtype, Pf, conDatFetcher)
