"""In-memory stand-ins for the reference's DataFetcher (detex/getdata.py) on the
hot path.  ObsPy I/O, response removal and filtering are out of scope
(SURVEY.md section 8): the fetchers below hand out arrays that are already
pre-processed (what ``construct._applyFilter`` returns), with the reference's
chunk geometry restated.

Channel order inside every [Nc, L] array is the reference's ``st.sort()`` order
(detex/construct.py:998); ``multiplex`` then interleaves channels sample-wise
exactly as detex/construct.py:928-987.
"""
import numpy as np


def multiplex(chans, dtype=None):
    """detex/construct.py:928-987: trim every channel to the shortest (970-974)
    and interleave sample-wise, ``vstack(chans).flatten(order='F')`` (977-978).  dtype: cast on the way
    (construct.py:1000-1002 casts the traces first; one pass instead of two)."""
    chans = [np.asarray(c) for c in chans]
    L = min(len(c) for c in chans)
    # (column writes into a [L, Nc] array: vstack(...).flatten(order='F') spends 42 ns per element -- 3.3 s for the
    # 2000 events of configs[2] -- on NumPy's generic strided copy)
    out = np.empty((L, len(chans)), dtype=np.result_type(*chans) if dtype is None else dtype)
    for c, x in enumerate(chans):
        out[:, c] = x[:L]
    return out.reshape(-1)


def trim_window(t0, T, sr, start, end):
    """Samples ObsPy's ``Trace.trim(starttime=start, endtime=end)`` (nearest_sample=True, no padding) keeps
    of a trace of T samples starting at time t0: (first sample, count).  Third-party behaviour, restated:
    the left cut is ``round((start - t0) sr)`` samples, the right cut ``round((t_last - end) sr)``; negative
    cuts (the window reaches past the data) cut nothing.  The reference's fetcher hands every template
    through it (detex/getdata.py:441-444 -> getStream -> trim)."""
    T = int(T)
    if T <= 0:
        return 0, 0
    lo = max(int(round((float(start) - t0) * sr)), 0)
    hi = max(int(round((t0 + (T - 1) / sr - float(end)) * sr)), 0)
    return lo, max(T - hi - lo, 0)


class Fragments(list):
    """[(channel, first sample, int32 counts)], non-overlapping -- what the device merges.  ``traces``: the
    extents [(channel, first sample, number of samples)] of the ObsPy traces the record ARRIVED in, when they
    differ from the fragments (overlaps were resolved): the reference decides 'is this chunk fragmented' and
    'too fractured' on the unmerged stream (construct.py:1005, 1009-1014, 1064), so those decisions keep
    reading the original traces (trace_extents)."""
    traces = None


def trace_extents(counts):
    tr = getattr(counts, 'traces', None)
    return tr if tr is not None else [(c, t0, len(x)) for (c, t0, x) in counts]


def clip_fragments(counts, spans):
    """Fragments (and their trace extents) clipped to spans [(first, end, new offset of first)], re-based."""
    out = Fragments()
    for (c, t0, x) in counts:
        for (a, b, off) in spans:
            lo, hi = max(t0, a), min(t0 + len(x), b)
            if hi > lo:
                out.append((c, off + lo - a, x[lo - t0:hi - t0]))
    out.sort(key=lambda f: (f[0], f[1]))
    if getattr(counts, 'traces', None) is not None:
        out.traces = sorted((c, off + max(t0, a) - a, min(t0 + m, b) - max(t0, a))
                            for (c, t0, m) in counts.traces for (a, b, off) in spans
                            if min(t0 + m, b) > max(t0, a))
    return out


def resolve_overlaps(fragments):
    """Overlapping count fragments of a channel resolved the way ObsPy's default merge does it, so that what goes
    to the device (dtx_ingest_counts: non-overlapping fragments, zero fill) is what ``st.merge(fill_value=0.0)``
    (detex/construct.py:1038, 1070) would have produced.  Third-party behaviour, restated from ObsPy 1.0.2
    ``Stream.merge(method=0)`` -> ``Trace.__add__`` (ObsPy is not installable here: parity vs ObsPy unpinned):
    the traces of a channel are taken in (start, end) order and added one by one to the running trace;

    * partial overlap (the next trace starts inside the running one and ends past it): if ALL samples of the
      overlap are identical the data are kept once, else the WHOLE overlap becomes the fill value (0);
    * contained trace (ends inside the running one): identical -> dropped, else its whole extent in the running
      trace becomes the fill value;
    * the filled zeros are data for every later comparison (fill_value is applied in each addition).

    fragments: [(channel, first sample, int32 array)], any order.  Returns the same triples (a ``Fragments``
    list), sorted by (channel, first sample), non-overlapping; without any overlap every input trace comes
    back as it is (abutting traces stay separate: a seam is a seam for 'too fractured').
    Re-sent telemetry packets -- the ordinary case -- cost one comparison and no copy of anything else."""
    by_ch = {}
    for c, t0, x in fragments:
        by_ch.setdefault(int(c), []).append((int(t0), np.asarray(x)))
    out = Fragments()
    overlapped = False
    for c, frs in by_ch.items():
        end = None
        for t0, n in sorted((t0, len(x)) for t0, x in frs if len(x)):
            overlapped = overlapped or (end is not None and t0 < end)
            end = t0 + n if end is None else max(end, t0 + n)
    if not overlapped:                                    # the ordinary record: every trace as it came
        out.extend(sorted(((int(c), int(t0), x) for c, t0, x in fragments if len(x)), key=lambda f: (f[0], f[1])))
        return out
    for c in sorted(by_ch):
        frs = sorted(by_ch[c], key=lambda f: (f[0], f[0] + len(f[1])))
        cur_t0, parts, cur_end = None, [], None           # the running trace: pieces end to end
        own = False                                       # parts[-1] is a private copy (may be written)

        def flush():
            if parts:
                out.append((c, cur_t0, parts[0] if len(parts) == 1 else np.concatenate(parts)))

        for t0, x in frs:
            if len(x) == 0:
                continue
            if cur_t0 is None or t0 >= cur_end:           # gap or exact fit: a new piece
                if cur_t0 is not None and t0 > cur_end:
                    flush()
                    cur_t0, parts = None, []
                if cur_t0 is None:
                    cur_t0 = t0
                parts.append(x)
                cur_end, own = t0 + len(x), False
                continue
            # overlap with the running trace: make it one private array (only traces that overlap are copied)
            run = parts[0] if len(parts) == 1 and own else np.concatenate(parts)
            a = t0 - cur_t0
            if t0 + len(x) > cur_end:                     # partial overlap
                d = cur_end - t0
                if not np.array_equal(run[a:], x[:d]):
                    run[a:] = 0
                parts, own = [run, x[d:]], False
                cur_end = t0 + len(x)
            else:                                         # contained
                b = a + len(x)
                if not np.array_equal(run[a:b], x):
                    run[a:b] = 0
                parts, own = [run], True
        flush()
    out.traces = sorted((int(c), int(t0), len(x)) for c, t0, x in fragments if len(x))
    return out


class EventFetcher(object):
    """Template-event waveforms of one or more stations.

    ``add``: waveforms already filtered and trimmed (the ``fetcher.getTemData`` + ``_applyFilter`` output
    of detex/construct.py:852-869).  ``add_raw`` / ``add_counts``: an UNPROCESSED record around the event
    (float samples, or int32 counts in fragments as they come off the digitiser); createCluster then cuts
    [origin - trim[0], origin + trim[1]] out of it (getdata.py:441-444), merges / decimates / detrends /
    band-passes it on the device (construct.py:990-1030 _applyFilter) and multiplexes it.

    events[station][name] = dict(data=[Nc, L] array, starttime=float epoch seconds of the first sample,
    origintime=float, magnitude=float[, raw=True, counts=[(channel, first sample, int32 array)],
    nsamples=int])
    """
    method = 'memory'

    def __init__(self, sampling_rate, channels):
        self.sampling_rate = float(sampling_rate)
        self.channels = list(channels)
        self.events = {}

    def add(self, station, name, data, starttime=0.0, origintime=None, magnitude=np.nan):
        data = np.atleast_2d(np.asarray(data))
        if data.shape[0] != len(self.channels):
            raise ValueError('event %s on %s has %d channels, fetcher has %d'
                             % (name, station, data.shape[0], len(self.channels)))
        self.events.setdefault(station, {})[name] = dict(
            data=data, starttime=float(starttime),
            origintime=float(starttime if origintime is None else origintime),
            magnitude=float(magnitude))
        return self

    def add_raw(self, station, name, data, starttime, origintime, magnitude=np.nan):
        """An unfiltered [Nc, T] record that holds the event: sample 0 at ``starttime``."""
        self.add(station, name, data, starttime, origintime, magnitude)
        rec = self.events[station][name]
        rec.update(raw=True, counts=None, nsamples=int(rec['data'].shape[1]))
        return self

    def add_counts(self, station, name, fragments, nsamples, starttime, origintime, magnitude=np.nan):
        """The record as int32 count fragments [(channel index, first sample, counts), ...] -- one entry
        per ObsPy trace, gaps between them -- of a record of ``nsamples`` samples starting at
        ``starttime`` (cf. ContinuousFetcher.add_counts)."""
        frs = [(int(c), int(t0), np.ascontiguousarray(x, dtype=np.int32)) for c, t0, x in fragments]
        for c, t0, x in frs:
            if not (0 <= c < len(self.channels)) or t0 < 0 or t0 + len(x) > nsamples:
                raise ValueError('fragment of channel %d at sample %d does not fit the record' % (c, t0))
        # overlapping traces (re-sent packets): ObsPy's merge rule, on the host, before the device sees them
        frs = resolve_overlaps(frs)
        self.events.setdefault(station, {})[name] = dict(
            data=None, starttime=float(starttime), origintime=float(origintime), magnitude=float(magnitude),
            raw=True, counts=frs, nsamples=int(nsamples))
        return self

    def has_raw(self, station=None):
        stas = [station] if station is not None else list(self.events)
        return any(rec.get('raw') for sta in stas for rec in self.events.get(sta, {}).values())

    def stations(self):
        return sorted(self.events.keys())

    def getTemData(self, station):
        """Yield (name, record) in sorted event-name order."""
        evs = self.events[station]
        for name in sorted(evs.keys()):
            yield name, evs[name]


class EventChunks(object):
    """Template events laid end to end as the "continuous data" of a ``SubSpace.detex(classifyEvents=...)``
    run (detex/detect.py:51-55, 151-152: the detectors are run over event waveforms instead of continuous
    data): one chunk per event, each with the start time of its waveform.  The waveforms are pre-processed
    already (raw records went through construct._preprocess_events)."""
    method = 'memory'
    raw = False
    conDatDuration = 86400          # one event = one sharding unit in a multi-process run
    conBuff = 0

    def __init__(self, sampling_rate, channels):
        self.sampling_rate = float(sampling_rate)
        self.channels = list(channels)
        self.data, self.bounds, self.names = {}, {}, {}

    def add(self, station, events):
        """events: [(name, [Nc, L] array, time of the first sample)]"""
        events = [(nm, np.asarray(d, dtype=np.float32), float(t)) for (nm, d, t) in events if np.shape(d)[1] > 0]
        if not events:
            return self
        off, bounds = 0, []
        for (_, d, t) in events:
            bounds.append((off, int(d.shape[1]), t))
            off += int(d.shape[1])
        self.data[station] = (np.concatenate([d for (_, d, _) in events], axis=1), events[0][2])
        self.bounds[station] = bounds
        self.names[station] = [nm for (nm, _, _) in events]
        return self

    def chunk_bounds(self, station, utcStart=None, utcEnd=None):
        return list(self.bounds[station])


class ContinuousFetcher(object):
    """Continuous pre-processed data of each station as one [Nc, T] array with a
    start time.  ``getConData`` cuts it into the reference's chunks: starts on
    multiples of ``conDatDuration`` (detex/getdata.py:869-886 _divideIntoChunks),
    each ``conDatDuration + conBuff`` seconds long (getdata.py:518), trimmed to
    the available data.  Defaults 3600 s + 120 s (getdata.py:299)."""
    method = 'memory'

    def __init__(self, sampling_rate, channels, conDatDuration=3600, conBuff=120, raw=False):
        self.sampling_rate = float(sampling_rate)
        self.channels = list(channels)
        self.conDatDuration = conDatDuration
        self.conBuff = conBuff
        # raw=True: the arrays are unfiltered; every chunk is detrended and band-passed
        # on the GPU with the cluster's ``filt`` before the scan (construct.py:1024-1028)
        self.raw = bool(raw)
        self.data = {}

    def add(self, station, data, starttime):
        data = np.atleast_2d(np.asarray(data))
        if data.shape[0] != len(self.channels):
            raise ValueError('continuous data on %s has %d channels, fetcher has %d'
                             % (station, data.shape[0], len(self.channels)))
        self.data[station] = (data, float(starttime))
        return self

    def add_counts(self, station, fragments, nsamples, starttime):
        """Continuous data of a station as it comes off the digitiser: ``fragments`` is a list
        of (channel index, first sample, int32 counts) -- one entry per ObsPy trace --, with
        gaps between them; ``nsamples`` the extent of the record.  Marks the fetcher raw: the
        scan merges (zero fill), finds each chunk's longest gap-free stretch, decimates,
        detrends and band-passes on the device (construct.py:990-1030, _applyFilter)."""
        self.raw = True
        self.counts = getattr(self, 'counts', {})
        frs = [(int(c), int(t0), np.ascontiguousarray(x, dtype=np.int32)) for c, t0, x in fragments]
        for c, t0, x in frs:
            if not (0 <= c < len(self.channels)) or t0 < 0 or t0 + len(x) > nsamples:
                raise ValueError('fragment of channel %d at sample %d does not fit the record' % (c, t0))
        # overlapping traces (re-sent packets): ObsPy's merge rule, on the host, before the device sees them
        frs = resolve_overlaps(frs)
        self.counts[station] = frs
        # chunk geometry only needs the extent: a zero-stride placeholder, no memory
        self.data[station] = (np.broadcast_to(np.float32(0), (len(self.channels), int(nsamples))),
                              float(starttime))
        return self

    def pin(self):
        """Move every station's array into page-locked host memory (as float32): the scan then
        uploads it at PCIe rate on a copy stream, under the previous station's kernels
        (SubSpace.detex), instead of through a pageable staging copy."""
        import torch
        self.pinned = getattr(self, 'pinned', {})
        # raw counts: the station's fragments end to end in ONE page-locked int32 buffer (+ their table), built
        # once -- the scan then uploads it as it is instead of concatenating 100 MB per call on the host
        self.counts_pinned = getattr(self, 'counts_pinned', {})
        for sta, frs in getattr(self, 'counts', {}).items():
            if sta in self.counts_pinned:
                continue
            off, tab = 0, []
            for (c, t0, x) in frs:
                tab.append((c, off, t0, len(x)))
                off += len(x)
            flat = torch.empty((off,), dtype=torch.int32).pin_memory()
            for (c, o, t0, m), (_, _, x) in zip(tab, frs):
                flat[o:o + m] = torch.from_numpy(x)
            self.counts_pinned[sta] = (flat, tab, frs)
        for sta, (data, t0) in list(self.data.items()):
            if sta in self.pinned or sta in getattr(self, 'counts', {}):
                continue
            t = torch.from_numpy(np.ascontiguousarray(data, dtype=np.float32)).pin_memory()
            self.pinned[sta] = t
            self.data[sta] = (t.numpy(), t0)          # same memory, NumPy face
        return self

    def chunk_bounds(self, station, utcStart=None, utcEnd=None):
        """[(first sample, number of samples, chunk start time)] of every chunk."""
        data, t0 = self.data[station]
        sr = self.sampling_rate
        T = data.shape[1]
        t_end = t0 + T / sr
        ts1 = t0 if utcStart is None else float(utcStart)
        ts2 = t_end if utcEnd is None else float(utcEnd)
        dur = self.conDatDuration
        a = ts1 - ts1 % dur                       # getdata.py:877-878
        b = ts2 - ts2 % dur
        out = []
        t = a
        while t <= b:                             # getdata.py:880-883
            s = int(round((max(t, t0) - t0) * sr))
            e_time = min(t + dur + self.conBuff, ts2, t_end)
            e = int(round((e_time - t0) * sr))
            if e - s > 0 and s < T:
                out.append((s, min(e, T) - s, t0 + s / sr))
            t += dur
        return out

    def getConData(self, station, utcStart=None, utcEnd=None):
        """Yield ([Nc, Tc] chunk view, chunk start time)."""
        data, _ = self.data[station]
        for s, cnt, ts in self.chunk_bounds(station, utcStart, utcEnd):
            yield data[:, s:s + cnt], ts
