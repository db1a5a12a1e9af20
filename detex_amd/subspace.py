"""ClusterStream / Cluster / SubSpace with the reference's names for the hot path
(detex/subspace.py:46-707, 715-2037): dendrogram cut, SVD basis construction and
selection, thresholds, and the detection scan ``SubSpace.detex`` that drives the
HIP kernels.  Plotting, GUI picking, HypoDD/KML writers are out of scope.
"""
import functools
import os
import pickle
import sqlite3

import numpy as np
import pandas as pd
import scipy.linalg
import scipy.stats
from scipy.cluster.hierarchy import fcluster  # noqa: F401  (kept for parity checks)

from ._lib import DetexHipError


def _collector_set_aside(fn):
    """Run ``fn`` with the objects alive at its start set aside from Python's garbage collector (gc.freeze /
    gc.unfreeze; the collector stays on for what the call itself allocates).  A full collection walks every
    tracked object of the process -- 70 ms with torch, pandas and numpy imported -- and the rows a long scan
    builds start one every few hundred units: one call in three of 240 station-days stopped for it with its
    scans queued behind (3.9 against 3.5 ms per station-day; scripts/probe/e2e_unit_times.py).  Left alone if the
    caller runs without the collector or has frozen objects of its own (unfreeze would release those too)."""
    import gc

    @functools.wraps(fn)
    def call(*args, **kwargs):
        if not gc.isenabled() or gc.get_freeze_count() > 0:
            return fn(*args, **kwargs)
        gc.freeze()
        try:
            return fn(*args, **kwargs)
        finally:
            gc.unfreeze()
    return call


def _error(msg, exc=DetexHipError):
    raise exc(msg)


# --------------------------------------------------------------------------
# ClusterStream / Cluster                          subspace.py:46-707
# --------------------------------------------------------------------------

class Cluster(object):
    """One station's dendrogram (subspace.py:290-412)."""

    def __init__(self, clustStream, station, temkey, eventList, link, ccReq, filt, decimate,
                 trim, DFcc):
        self.link = link
        self.DFcc = DFcc
        self.station = station
        self.temkey = temkey
        self.key = list(eventList)
        self.updateReqCC(ccReq)
        self.trim = trim
        self.decimate = decimate

    def _makeDFLINK(self):
        """subspace.py:385-412: the merges as rows {i1, i2, disSim, num, clust} (clust = the node a merge
        creates).  The member lists the reference keeps with every row ('II') are not materialised: for a
        single-linkage chain of N events they hold N^2 / 2 indices (98 ms at 2000 events); ``row['II']`` (and
        ``members(node)``) walk the tree when asked."""
        N = len(self.link)
        return [_LinkRow(self, i1=int(self.link[a, 0]), i2=int(self.link[a, 1]), disSim=self.link[a, 2],
                         num=self.link[a, 3], clust=N + 1 + a) for a in range(N)]

    def members(self, node):
        """Leaves (positions in ``key``) under a dendrogram node: the reference's dflink.II of that merge."""
        N = len(self.link)
        out, stack = [], [int(node)]
        while stack:
            v = stack.pop()
            if v <= N:
                out.append(v)
            else:
                stack.append(int(self.link[v - N - 1, 1]))
                stack.append(int(self.link[v - N - 1, 0]))
        return out

    def updateReqCC(self, newccReq):
        """subspace.py:305-346: cut the dendrogram at 1 - ccReq.  Clusters are numbered by their top merge,
        highest dissimilarity first; ``clustlinks[k]`` lists the merges inside cluster k in that order.

        The reference takes, again and again, the highest remaining merge and every remaining merge whose
        member set is a subset of its -- set algebra over member lists, O(N^2) and more.  Restated on the tree:
        the highest remaining merge takes every selected merge of its subtree that is still free (a merge's
        children are never higher than the merge: SciPy linkage heights are monotone for the single linkage
        used here, so a subtree's selected merges are exactly the ones whose members are a subset), one
        depth-first walk per cluster, every merge visited once."""
        if newccReq < 0. or newccReq > 1.:
            _error('Parameter ccReq must be between 0 and 1')
        self.ccReq = newccReq
        self.dflink = self._makeDFLINK()
        N = len(self.link)
        thr = 1 - self.ccReq
        sel = [r['disSim'] <= thr for r in self.dflink]
        order = sorted((a for a in range(N) if sel[a]), key=lambda a: -self.dflink[a]['disSim'])   # stable
        rank = {a: k for k, a in enumerate(order)}
        assigned = [False] * N
        clustlinks, clustEvents = {}, {}
        for a in order:
            if assigned[a]:
                continue
            # a new cluster under merge a: every selected merge of its subtree that no earlier (higher, or equally
            # high and earlier in the linkage) cluster has taken -- what "II is a subset of the top's II" selects
            inside, stack = [], [a]
            while stack:
                v = stack.pop()
                if assigned[v] or not sel[v]:
                    continue
                assigned[v] = True
                inside.append(v)
                for ch in (self.dflink[v]['i1'], self.dflink[v]['i2']):
                    if ch > N:
                        stack.append(ch - N - 1)
            inside.sort(key=rank.__getitem__)
            clnum = len(clustlinks)
            clustlinks[clnum] = [self.dflink[v]['clust'] for v in inside]
            clustEvents[clnum] = sorted(self.members(N + 1 + a))
        self.clustlinks = clustlinks
        self.clusts = [[self.key[y] for y in clustEvents[x]] for x in sorted(clustEvents)]
        clustset = set(y for x in self.clusts for y in x)
        self.singles = [k for k in self.key if k not in clustset]
        self.clustcount = int(np.sum([len(x) for x in self.clusts])) if self.clusts else 0

    def __getitem__(self, index):
        return self.clusts[index]

    def __len__(self):
        return len(self.clusts)


class ClusterStream(object):
    """Container of per-station clusters (subspace.py:46-288).  ``trdf`` has the
    reference's columns Station, Link, CCs, Lags, Subsamp, Events, Stats
    (construct.py:159)."""

    def __init__(self, trdf, temkey, stakey, fetcher, eventList, ccReq, filt, decimate, trim,
                 fileName, eventsOnAllStations, enforceOrigin):
        self.trdf = trdf
        self.temkey = temkey
        self.stakey = stakey
        self.fetcher = fetcher
        self.eventList = eventList
        self.filt = filt
        self.decimate = decimate
        self.trim = trim
        self.filename = fileName
        self.eventsOnAllStations = eventsOnAllStations
        self.enforceOrigin = enforceOrigin
        self.ccReq = None
        self.stalist = trdf.Station.values.tolist()
        self.stalist2 = [x.split('.')[1] if '.' in x else x for x in self.stalist]
        self.clusters = []
        for _, row in trdf.iterrows():
            evlist = eventList if eventsOnAllStations else row.Events
            self.clusters.append(Cluster(self, row.Station, temkey, evlist, row.Link, ccReq, filt,
                                         decimate, trim, row.CCs))

    def updateReqCC(self, reqCC):
        """subspace.py:174-201: float for all stations, or {station: ccReq}."""
        if isinstance(reqCC, dict):
            for sta, v in reqCC.items():
                self[sta].updateReqCC(v)
        else:
            for c in self.clusters:
                c.updateReqCC(reqCC)

    def write(self):
        """subspace.py:261-267 (pickle).  The newest protocol: the reference's protocol 2 spends a second on
        the 2 000-event similarity frames (createCluster: 1.34 s of which 1.0 here; the matrices take 37 ms)."""
        with open(self.filename, 'wb') as f:
            pickle.dump(self, f, pickle.HIGHEST_PROTOCOL)

    def __getitem__(self, key):
        if isinstance(key, int):
            return self.clusters[key]
        if key in self.stalist:
            return self.clusters[self.stalist.index(key)]
        if key in self.stalist2:
            return self.clusters[self.stalist2.index(key)]
        _error('%s is not a station in this ClusterStream' % key, KeyError)

    def __len__(self):
        return len(self.clusters)


# --------------------------------------------------------------------------
# SubSpace                                         subspace.py:715-2037
# --------------------------------------------------------------------------

# units SubSpace.detex keeps in flight on the device (scans queued ahead of the host's picking / row building)
INFLIGHT = max(3, int(os.environ.get('DETEX_AMD_INFLIGHT', '4')))


class _StationUploader(object):
    """Station data to the device ahead of its scan.  Arrays the fetcher holds in pinned
    memory (ContinuousFetcher.pin) are copied on a second stream, under the kernels of the
    station being scanned, into one of three device buffers the uploader owns (detex keeps three
    stations in flight); pageable arrays go up synchronously.  In a multi-process run only the
    sample ranges of the rank's own chunks are uploaded, packed end to end
    (parallel.plan_shard): the host link is the one resource the ranks share.  The uploader and
    its buffers live as long as the SubSpace (one per device)."""

    def __init__(self, device):
        import torch
        self.device = device
        self.stream = torch.cuda.Stream(device) if device.type == 'cuda' else None
        self.pending = {}
        # one device buffer per unit in flight: [buffer, event recorded when its last reader was enqueued]
        self.slots = [None] * INFLIGHT
        self.turn = 0
        self.cur = None
        self.bytes_uploaded = 0           # host -> device bytes of station data since reset()
        self._mux = None                  # a HipContext on the copy stream: interleaves a unit's planes behind its upload

    def _mux_ctx(self):
        import torch
        from .hip import HipContext
        if self._mux is None:
            with torch.cuda.stream(self.stream):            # (a context keeps the stream current at its creation)
                self._mux = HipContext(self.device.index or 0)
        return self._mux

    def reset(self):
        self.pending = {}
        self.cur = None
        self.bytes_uploaded = 0

    def _slot(self, numel):
        """The next of the three device buffers, grown if needed.  Buffers are allocated with the
        copy stream current: their memory comes from that stream's pool, so a block a kernel queued
        on another stream still reads is never handed out; a buffer that is replaced is freed only
        behind its last reader."""
        import torch
        k = self.turn
        self.turn = (k + 1) % len(self.slots)
        slot = self.slots[k]
        if slot is None or slot[0].numel() < numel:
            with torch.cuda.stream(self.stream):
                if slot is not None and slot[1] is not None:
                    self.stream.wait_event(slot[1])
                self.slots[k] = slot = None
                slot = [torch.empty(numel, dtype=torch.float32, device=self.device), None]
            self.slots[k] = slot
        return k, slot

    def start(self, fetcher, sta, bounds, sel=None, key=None):
        """bounds: the station's chunks (ContinuousFetcher.chunk_bounds); sel: the chunk indices
        scanned here (None: all, the whole record is uploaded).  The chunks are handed to the scan
        as (first sample in the uploaded buffer, samples, start time, chunk index).  key: what get()
        asks for (default: the station; detex scans a long record in several units)."""
        import torch
        from . import parallel
        data = fetcher.data[sta][0]
        Nc = data.shape[0]
        counts = getattr(fetcher, 'counts', {}).get(sta)
        if sel is None:
            runs = [(0, data.shape[1], 0)]
            local = [(int(b[0]), int(b[1]), b[2], i) for i, b in enumerate(bounds)]
            T = data.shape[1]
        else:
            runs, local, T = parallel.plan_shard(bounds, sel)
            if counts is not None:
                counts = parallel.rebase_fragments(counts, runs)
        job = dict(planes=None, ready=None, slot=None, bounds=local, T=T, counts=counts)
        self.pending[sta if key is None else key] = job
        if counts is not None or T == 0:
            # raw counts: scan_station ingests the fragments (their upload is counted here)
            self.bytes_uploaded += sum(4 * len(x) for (_, _, x) in (counts or []))
            return
        pinned = getattr(fetcher, 'pinned', {}).get(sta)
        self.bytes_uploaded += 4 * Nc * T
        if pinned is None or self.stream is None:
            host = data if sel is None else np.concatenate([data[:, a:b] for (a, b, _) in runs], axis=1)
            job['planes'] = torch.from_numpy(np.ascontiguousarray(host, dtype=np.float32)).to(self.device)
            return
        k, slot = self._slot(Nc * T)
        dst = slot[0][:Nc * T].view(Nc, T)
        ready = torch.cuda.Event()
        with torch.cuda.stream(self.stream):
            if slot[1] is not None and not slot[1].query():
                self.stream.wait_event(slot[1])        # the scan that last read this buffer
            if sel is None:
                dst.copy_(pinned, non_blocking=True)
            else:
                for (a, b, off) in runs:               # contiguous rows: plain asynchronous copies
                    for c in range(Nc):
                        dst[c, off:off + b - a].copy_(pinned[c, a:b], non_blocking=True)
            xdev = None
            if not getattr(fetcher, 'raw', False):
                # already filtered planes: the multiplexed stream the scan takes (construct.py:977-978) is made
                # HERE, behind the upload on the copy stream -- under the previous unit's scan kernel -- instead
                # of in front of this unit's scan on the station's stream (45 us per station-day)
                xdev = self._mux_ctx().preprocess(dst[None], filt=None, detrend=False)[0]
            ready.record(self.stream)
        job.update(planes=dst, ready=ready, slot=k, xdev=xdev)

    def get(self, sta, stream=None):
        """The station's upload job; `stream` (the scanning context's) waits for the copy."""
        job = self.pending.pop(sta)
        self.cur = job['slot']
        if job['ready'] is not None and stream is not None:
            stream.wait_event(job['ready'])
            if job.get('xdev') is not None:
                job['xdev'].record_stream(stream)          # (memory of the copy stream's pool, read on the scan's)
        return job

    def release(self, stream=None):
        """The scan of the station handed out last has been enqueued on `stream`: its buffer may
        be overwritten once the work queued there so far is done."""
        import torch
        if self.cur is not None and stream is not None:
            ev = torch.cuda.Event()
            ev.record(stream)
            self.slots[self.cur][1] = ev
        self.cur = None


def _same_unit(a, b):
    """(station, chunk indices or None) twice the same?"""
    return a[0] == b[0] and ((a[1] is None and b[1] is None) or
                             (a[1] is not None and b[1] is not None and list(a[1]) == list(b[1])))


def _sum_hists(parts, hk):
    """{detector: counts} of several parts added."""
    acc = {}
    for p in parts:
        for nm, h in p[hk].items():
            acc[nm] = acc[nm] + h if nm in acc else np.array(h, copy=True)
    return acc


def _merge_ranks(parts, names_order=None):
    """Results of one station in several parts -- from every rank (each scanned its share of the chunks)
    and / or from the units a long record is scanned in: detection rows in the order one launch over the
    whole record produces them -- by (chunk, detector, order found), the key StationScan.finish returns
    next to the tables --, histogram counts added."""
    parts = [p for p in parts if p is not None]
    if not parts:
        return None
    if len(parts) == 1:
        return parts[0]
    out = {}
    for kind in ('ss', 'sg'):
        rows = [r for p in parts for r in p[kind]]
        keys = np.array([k for p in parts for k in np.asarray(p['key_' + kind], dtype=np.int64).reshape(-1, 3)],
                        dtype=np.int64).reshape(-1, 3)
        order = np.lexsort((keys[:, 2], keys[:, 1], keys[:, 0])) if len(rows) else []
        out[kind] = [rows[i] for i in order]
        out['key_' + kind] = keys[order] if len(rows) else keys
    for hk in ('hist_ss', 'hist_sg'):
        out[hk] = _sum_hists(parts, hk)
    out['names'] = parts[0].get('names', [])
    out['maxds'] = {}
    for p in parts:                                  # every chunk was scanned by exactly one rank
        out['maxds'].update(p.get('maxds', {}))
    if any('utc' in p for p in parts):
        out['utc'] = sorted((r for p in parts for r in p.get('utc', [])), key=lambda r: r['chunk'])
    return out


class _LinkRow(dict):
    """One merge of Cluster.dflink.  'II' -- the member list the reference stores with every row
    (subspace.py:392-406) -- is computed on first access instead of for all rows up front."""

    def __init__(self, cluster, **kw):
        dict.__init__(self, **kw)
        self._cluster = cluster

    def __missing__(self, key):
        if key != 'II':
            raise KeyError(key)
        self['II'] = self._cluster.members(self['clust'])
        return self['II']

    # ('II' answers through every way a dict is asked: row['II'], row.get('II'), 'II' in row, keys / items of a
    # row that was asked once -- ADVICE r5: only __getitem__ went through __missing__)
    def get(self, key, default=None):
        if key == 'II':
            return self['II']
        return dict.get(self, key, default)

    def __contains__(self, key):
        return key == 'II' or dict.__contains__(self, key)


class _DbWriter(object):
    """The SQLite file of a SubSpace.detex call, written by a thread of its own: the detection rows of
    every finished unit are queued and appended there while the next units are scanned (the reference
    appends every 500 rows from inside its chunk loop, detect.py:207-210, util.saveSQLite).  Rows go in
    through one executemany per flush into tables with the schema pandas' to_sql gives them (REAL / TEXT
    columns, no index) -- results.py and the reference's readers use pd.read_sql on them."""
    FLUSH = 500
    COLS = ['DS', 'DS_STALTA', 'STMP', 'Name', 'Sta', 'MSTAMPmin', 'MSTAMPmax', 'Mag', 'SNR', 'ProEnMag']
    TYPES = ['REAL', 'REAL', 'REAL', 'TEXT', 'TEXT', 'REAL', 'REAL', 'REAL', 'REAL', 'REAL']

    SYNC_MODES = ('OFF', 'NORMAL', 'FULL', 'EXTRA')

    def __init__(self, path, fresh=False):
        """fresh: the caller has just removed the file (delOldCorrs) -- everything in it will be this call's."""
        import queue
        import threading
        self.path = path
        # Durability.  A commit is an fsync (2.5 ms each on the test boxes' file system); the last flushes of a
        # call sit on its critical path (close() joins), and an 8-station call made five of them: 10 ms of a 37 ms
        # call.  Only a file this call creates from scratch (delOldCorrs removed the old one, or there was none)
        # is written with synchronous=OFF: a crash of the operating system can then cost nothing but this call's
        # own rows, which a re-run writes again.  A file that already holds earlier runs' detections (the
        # reference's append mode, delOldCorrs=False, subspace.py:1847-1854) is written with SQLite's default,
        # FULL, like the reference's to_sql.  DETEX_AMD_DB_SYNC overrides either way.
        mode = os.environ.get('DETEX_AMD_DB_SYNC')
        if mode is None:
            mode = 'OFF' if (fresh and not os.path.exists(path)) else 'FULL'
        mode = mode.strip().upper()
        if mode not in self.SYNC_MODES:
            raise ValueError('DETEX_AMD_DB_SYNC must be one of %s (got %r)' % ('/'.join(self.SYNC_MODES), mode))
        self.sync = mode
        self.q = queue.Queue()
        self.error = None
        self.thread = threading.Thread(target=self._run, name='detex-db-writer', daemon=True)
        self.thread.start()

    def put(self, table, rows):
        """rows: a list of row lists in COLS order, or a DataFrame with those columns"""
        if self.error is not None:                          # the writer thread has died: say so now, not at close()
            raise self.error
        if rows is not None and len(rows):
            if isinstance(rows, pd.DataFrame):
                rows = rows[self.COLS].values.tolist()
            self.q.put(('rows', table, list(rows)))

    def call(self, fn):
        """fn(connection) on the writer thread, behind everything queued so far."""
        self.q.put(('call', fn, None))

    def _run(self):
        try:
            con = sqlite3.connect(self.path)
            con.execute('PRAGMA synchronous=%s' % self.sync)      # (validated in __init__)
            pending = {}

            def flush(table=None):
                wrote = False
                for tab in ([table] if table else list(pending)):
                    rows = pending.pop(tab, [])
                    if rows:
                        con.execute('CREATE TABLE IF NOT EXISTS "%s" (%s)' % (
                            tab, ', '.join('"%s" %s' % ct for ct in zip(self.COLS, self.TYPES))))
                        con.executemany('INSERT INTO "%s" VALUES (%s)' % (tab, ','.join('?' * len(self.COLS))),
                                        rows)
                        wrote = True
                if wrote:
                    con.commit()                            # one commit for everything flushed together
            try:
                while True:
                    kind, a, b = self.q.get()
                    if kind == 'stop':
                        if not a:
                            flush()
                        break
                    if kind == 'rows':
                        pending.setdefault(a, []).extend(b)
                        if len(pending[a]) >= self.FLUSH:
                            flush(a)
                    else:
                        flush()
                        a(con)
                        con.commit()
            finally:
                con.close()
        except Exception as e:                              # noqa: BLE001 -- reported by close()
            self.error = e

    def close(self, abort=False):
        self.q.put(('stop', abort, None))
        self.thread.join()
        if self.error is not None and not abort:
            raise self.error


class SubSpace(object):
    """Subspaces (clusters of aligned events) and singles of every station
    (subspace.py:715-733)."""

    def __init__(self, singlesDict, subSpaceDict, cl, dtype, Pf, cfetcher):
        self.cfetcher = cfetcher
        self.clusters = cl
        self.subspaces = subSpaceDict
        self.singles = singlesDict
        self.dtype = dtype
        self.Pf = Pf
        self.ssStations = sorted(self.subspaces.keys())
        self.singStations = sorted(self.singles.keys())
        self.Stations = sorted(set(self.ssStations) | set(self.singStations))
        self.histSubSpaces = {}
        self.histSingles = {}

    # ------------------------------------------------------------ validation
    def validateClusters(self, device=None, ctx=None):
        """subspace.py:737-783: drop events whose aligned (and trimmed) waveform no longer
        correlates with any later event of its cluster at the cluster's ccReq (misaligned
        events, reference issue 25).  The correlations are fast_normcorr at lag 0
        (construct.py:469-483) on the device (dtx_normcorr).  As in the reference, without
        trim times the last sample is left out (``[0:-1]``, subspace.py:752-753)."""
        import torch
        from .hip import HipContext
        own = ctx is None
        if own:
            from . import parallel
            ctx = HipContext(parallel.local_device() if device is None else device)
        try:
            for sta in list(self.subspaces.keys()):
                subs = self.subspaces[sta]
                ccreq = self.clusters[sta].ccReq
                for ind, row in subs.iterrows():
                    st = row.SampleTrims or {}
                    start, stop = (st['Starttime'], st['Endtime']) if ('Starttime' in st and
                                                                       'Endtime' in st) else (0, -1)
                    dev = {e: torch.from_numpy(np.ascontiguousarray(row.AlignedTD[e][start:stop],
                                                                    dtype=np.float32)).to(ctx.device)
                           for e in row.Events}
                    events = list(row.Events)
                    cc = {}
                    for i, ev1 in enumerate(events[:-1]):
                        cc[ev1] = [ctx.normcorr(dev[ev1], dev[ev2]) for ev2 in events[i + 1:]]
                    ctx.sync()
                    for ev1 in events[:-1]:              # subspace.py:761-771
                        ccs = [float(c.cpu().numpy()[0]) for c in cc[ev1]]
                        if ccs and max(ccs) < ccreq:
                            row.Events.remove(ev1)       # _removeEvent (775-783)
                            row.AlignedTD.pop(ev1, None)
        finally:
            if own:
                ctx.close()

    # ------------------------------------------------------------ trims
    def attachPickTimes(self, startSeconds=0.0, defaultDuration=30):
        """Window of the (aligned) waveforms the detectors use, as
        SampleTrims = {Starttime, Endtime} in multiplexed samples, both multiples
        of Nc (subspace.py:1461-1615; default duration 30 s, line 1462).  The
        reference derives the start from a phase-pick file; here it is given
        directly as seconds from the start of the aligned waveforms.  Existing
        trims are not overridden (reference behaviour)."""
        for dfs in (self.subspaces, self.singles):
            for sta, df in dfs.items():
                for ind, row in df.iterrows():
                    if row.SampleTrims:
                        continue
                    ev = row.Events[0]
                    Nc = row.Stats[ev]['Nc']
                    sr = row.Stats[ev]['sampling_rate']
                    wf = row.AlignedTD[ev] if 'AlignedTD' in row.index else row.MPtd[ev]
                    s = int(round(startSeconds * sr)) * Nc
                    e = min(s + int(round(defaultDuration * sr)) * Nc, len(wf) - len(wf) % Nc)
                    df.at[ind, 'SampleTrims'] = {'Starttime': s, 'Endtime': e,
                                                 'DurationSeconds': (e - s) / (sr * Nc)}

    # ------------------------------------------------------------ SVD
    def _trimGroups(self, row, keys):
        """subspace.py:921-943."""
        ali = row.AlignedTD
        st = row.SampleTrims
        if 'Starttime' in st and 'Endtime' in st:
            stim, etim = max(st['Starttime'], 0), st['Endtime']
            arr = np.vstack([ali[x][stim:etim] - np.mean(ali[x][stim:etim]) for x in keys])
        else:
            arr = np.vstack([ali[x] - np.mean(ali[x]) for x in keys])
        return np.asarray(arr, dtype=np.float64), arr.shape[1]

    def _getFracEnergy(self, row, U):
        """subspace.py:968-997: cumulative fractional energy capture."""
        st = row.SampleTrims
        frac = {}
        for key in row.Events:
            w = np.asarray(row.AlignedTD[key], dtype=np.float64)
            if 'Starttime' in st and 'Endtime' in st:
                w = w[st['Starttime']:st['Endtime']]
            nu = np.dot(U.T, w) / scipy.linalg.norm(w)
            rep = np.insert(np.square(nu), 0, 0)
            frac[key] = [np.sum(rep[:x + 1]) for x in range(len(rep))]
        frac['Average'] = np.average([frac[x] for x in row.Events], axis=0)
        frac['Minimum'] = np.min([frac[x] for x in row.Events], axis=0)
        return frac

    @staticmethod
    def _getUsedBasis(nsv, fracEnergy, selectCriteria, selectValue):
        """subspace.py:999-1013: number of leading singular vectors kept."""
        if selectCriteria in (1, 2, 3):
            fracEnergy['Average'][-1] = 1.00
            return int(np.argmax(fracEnergy['Average'] >= selectValue))
        return min(int(selectValue) + 1, nsv)               # criteria 4: keys[:selectValue+1]

    def _checkSelection(self, selectCriteria, selectValue, threshold):
        """subspace.py:945-966."""
        if selectCriteria in (1, 2, 3):
            if selectValue > 1 or selectValue < 0:
                _error('When selectCriteria==%d selectValue must be a float between 0 and 1'
                       % selectCriteria, ValueError)
        elif selectCriteria == 4:
            if selectValue < 0 or not isinstance(selectValue, int):
                _error('When selectCriteria==4 selectValue must be an integer >= 0', ValueError)
        else:
            _error('selectCriteria of %s is not supported' % selectCriteria)
        if threshold is not None:
            if not isinstance(threshold, (int, float)) or threshold < 0:
                _error('Unsupported type for threshold, must be None or float', ValueError)

    def SVD(self, selectCriteria=2, selectValue=0.9, conDatNum=100, threshold=None,
            normalize=False, useSingles=True, validateWaveforms=True, backupThreshold=None,
            **kwargs):
        """subspace.py:786-912: SVD of the aligned, trimmed, demeaned events of
        every subspace; basis selection by fractional energy capture; thresholds.
        The SVD itself is LAPACK on the host exactly as in the reference (tiny:
        n x nEvents).  Singular vectors are kept as columns of ``SVD['U']`` in
        descending singular-value order (the reference's dict keyed by singular
        value sorted descending, lines 892-895 / 1004-1005)."""
        self._checkSelection(selectCriteria, selectValue, threshold)
        for sta in self.ssStations:
            df = self.subspaces[sta]
            for ind, row in df.iterrows():
                keys = sorted(row.Events)
                arr, basisLength = self._trimGroups(row, keys)
                if basisLength == 0:
                    df.drop(index=ind, inplace=True)       # subspace.py:880-885
                    continue
                if normalize:
                    arr = np.array([x / np.linalg.norm(x) for x in arr])
                U, s, _ = scipy.linalg.svd(arr.T, full_matrices=False)
                frac = self._getFracEnergy(row, U)
                nb = self._getUsedBasis(len(s), frac, selectCriteria, selectValue)
                df.at[ind, 'SVD'] = {'U': U, 's': s}
                df.at[ind, 'FracEnergy'] = frac
                df.at[ind, 'UsedSVDKeys'] = list(s[:nb])
                df.at[ind, 'SVDdefined'] = True
                df.at[ind, 'NumBasis'] = nb
        if self.ssStations:
            self._setThresholds(selectCriteria, selectValue, conDatNum, threshold,
                                backupThreshold, kwargs)
        if self.singStations and useSingles:
            self.setSinglesThresholds(conDatNum=conDatNum, threshold=threshold,
                                      backupThreshold=backupThreshold, **kwargs)

    def _setThresholds(self, selectCriteria, selectValue, conDatNum, threshold,
                       backupThreshold, kwargs):
        """subspace.py:1015-1054."""
        if threshold is not None and threshold > 0:
            for sta in self.ssStations:
                for ind in self.subspaces[sta].index:
                    self.subspaces[sta].at[ind, 'Threshold'] = threshold
        elif selectCriteria == 1:
            _error('selectCriteria 1 currently not supported', ValueError)
        elif selectCriteria in (2, 4):
            self.getFAS(conDatNum, **kwargs)
            for sta in self.ssStations:
                for ind, row in self.subspaces[sta].iterrows():
                    a, b = row.FAS['betadist'][0:2]
                    self.subspaces[sta].at[ind, 'Threshold'] = self._beta_threshold(
                        a, b, sta, row.Name, backupThreshold)
        elif selectCriteria == 3:
            for sta in self.ssStations:
                for ind, row in self.subspaces[sta].iterrows():
                    th = row.FracEnergy['Minimum'][row.NumBasis] * selectValue
                    self.subspaces[sta].at[ind, 'Threshold'] = th

    def _beta_threshold(self, a, b, sta, name, backupThreshold):
        """subspace.py:1033-1047 + _approxThld 1110-1140."""
        th = scipy.stats.beta.isf(self.Pf, a, b, 0, 1)
        if th > .9:
            th = self._approxThld(a, b, sta, name, self.Pf, 1000, 3, backupThreshold)
        return th

    @staticmethod
    def _approxThld(beta_a, beta_b, sta, name, target, numint, numloops, backupThreshold):
        startVal, stopVal = 0, 1
        loops = 0
        bestX = None
        while loops < numloops:
            Xs = np.linspace(startVal, stopVal, numint)
            pfs = scipy.stats.beta.sf(Xs, beta_a, beta_b)
            minind = int(np.abs(pfs - target).argmin())
            if minind == 0 or minind == numint - 1:
                if backupThreshold is None:
                    _error('Grid search for threshold failing for %s on %s, set it manually '
                           'or use default' % (sta, name), ValueError)
                return backupThreshold
            bestX = Xs[minind]
            startVal, stopVal = Xs[minind - 1], Xs[minind + 1]
            loops += 1
        return bestX

    def setSinglesThresholds(self, conDatNum=50, recalc=False, threshold=None,
                             backupThreshold=None, **kwargs):
        """subspace.py:1056-1108: singles without trims are dropped."""
        for sta in self.singStations:
            sing = self.singles[sta]
            sing = sing[[len(x) > 0 for x in sing.SampleTrims]].reset_index(drop=True)
            sing['Name'] = ['SG%d' % x for x in range(len(sing))]
            self.singles[sta] = sing
        if threshold is None:
            self.getFAS(conDatNum, useSingles=True, useSubSpaces=False, **kwargs)
        for sta in self.singStations:
            for ind, row in self.singles[sta].iterrows():
                if threshold:
                    th = threshold
                else:
                    a, b = row.FAS['betadist'][0:2]
                    th = self._beta_threshold(a, b, sta, row.Name, backupThreshold)
                self.singles[sta].at[ind, 'Threshold'] = th

    # ------------------------------------------------------------ detectors
    def _detector_bank(self, sta, useSubSpaces=True, useSingles=True, need_thr=True):
        """Stack the detectors acting on one station (detect.py:319-388
        _loadMPSubSpace): subspaces contribute their used left singular vectors,
        singles their trimmed, normalised, NOT demeaned template
        (detect.py:357-364).  Returns (names, kinds, U [Ktot, n], det_off,
        offsets {name: array}, thresholds {name: float})."""
        key, held = self._bank_key(sta, useSubSpaces, useSingles, need_thr)
        cache = self.__dict__.setdefault('_bank_cache', {})
        aux = self.__dict__.setdefault('_bank_aux', {})     # {station: (training events, magnitudes)}
        hit = cache.get((sta, useSubSpaces, useSingles, need_thr))
        if (hit is not None and hit[0] == key and len(hit[4]) == len(held) and
                all(a is b for a, b in zip(hit[4], held))):
            self._bank_ewf, self._bank_mags = aux[sta] = (hit[2], hit[3])
            return hit[1]
        bank = self._build_bank(sta, useSubSpaces, useSingles, need_thr)
        aux[sta] = (self._bank_ewf, self._bank_mags)
        cache[(sta, useSubSpaces, useSingles, need_thr)] = (key, bank, self._bank_ewf, self._bank_mags, held)
        return bank

    def _bank_key(self, sta, useSubSpaces, useSingles, need_thr):
        """What the bank depends on, read cheaply: (scalars, containers).  The scalars -- names, basis
        counts, trims, thresholds, the event lists, their magnitudes and offsets -- are compared by
        value; the SVD / waveform containers by identity, and the cache entry keeps them alive, so
        that a freed container's address can never be mistaken for a new one's.  SVD(),
        attachPickTimes(), validateClusters() and the threshold setters replace or change these:
        a changed SubSpace never sees a stale bank."""
        key, held = [], []

        def scal(name, st, th, ev, stats, offs):
            ev = tuple(ev)
            return (name, tuple(sorted((st or {}).items())), float(th) if need_thr else None, ev,
                    tuple(float(stats[e]['magnitude']) for e in ev),
                    tuple(np.asarray(offs, dtype=np.float64).ravel().tolist()))
        if useSubSpaces and sta in self.subspaces:
            df = self.subspaces[sta]                        # (columns zipped: iterrows costs 50 us a row)
            for name, svd, nb, ok, st, th, ali, ev, stats, offs in zip(
                    df.Name, df.SVD, df.NumBasis, df.SVDdefined, df.SampleTrims, df.Threshold,
                    df.AlignedTD, df.Events, df.Stats, df.Offsets):
                key.append(scal(name, st, th, ev, stats, offs) + (nb, bool(ok)))
                held += [svd, ali]
        if useSingles and sta in self.singles:
            df = self.singles[sta]
            for name, st, th, mp, ev, stats, offs in zip(df.Name, df.SampleTrims, df.Threshold, df.MPtd,
                                                         df.Events, df.Stats, df.Offsets):
                key.append(scal(name, st, th, ev, stats, offs))
                held.append(mp)
        return repr(key), held

    def _build_bank(self, sta, useSubSpaces, useSingles, need_thr):
        names, kinds, blocks, offsets, thr = [], [], [], {}, {}
        self._bank_ewf, self._bank_mags = {}, {}
        if useSubSpaces and sta in self.subspaces:
            for _, row in self.subspaces[sta].iterrows():
                if not row.SVDdefined or row.NumBasis < 1:
                    continue
                blocks.append(np.asarray(row.SVD['U'][:, :row.NumBasis].T, dtype=np.float64))
                names.append(row.Name)
                kinds.append('ss')
                st = row.SampleTrims                               # detect.py:347-355
                wfs = [np.asarray(row.AlignedTD[e], dtype=np.float64) for e in row.Events]
                if 'Starttime' in st:
                    wfs = [w[st['Starttime']:st['Endtime']] for w in wfs]
                self._bank_ewf[row.Name] = np.array(wfs)
                self._bank_mags[row.Name] = np.array([row.Stats[e]['magnitude']
                                                      for e in row.Events], dtype=np.float64)
                offsets[row.Name] = np.asarray(row.Offsets, dtype=np.float64)
                thr[row.Name] = float(row.Threshold) if need_thr else np.nan
        if useSingles and sta in self.singles:
            for _, row in self.singles[sta].iterrows():
                if not row.SampleTrims:
                    continue
                ev = row.Events[0]
                x = np.asarray(row.MPtd[ev], dtype=np.float64)
                x = x[row.SampleTrims['Starttime']:row.SampleTrims['Endtime']]
                blocks.append((x / np.linalg.norm(x))[None, :])
                names.append(row.Name)
                kinds.append('sg')
                self._bank_ewf[row.Name] = x[None, :]
                self._bank_mags[row.Name] = np.array([row.Stats[ev]['magnitude']])
                offsets[row.Name] = np.asarray(row.Offsets, dtype=np.float64)
                thr[row.Name] = float(row.Threshold) if need_thr else np.nan
        if not blocks:
            return [], [], None, None, {}, {}
        lens = set(b.shape[1] for b in blocks)
        if len(lens) != 1:
            # the reference mis-broadcasts here; refuse instead
            _error('all detectors of station %s must have the same length (got %s)'
                   % (sta, sorted(lens)))
        U = np.concatenate(blocks, axis=0)
        det_off = np.concatenate([[0], np.cumsum([b.shape[0] for b in blocks])]).astype(np.int32)
        return names, kinds, U, det_off, offsets, thr

    def _detector_bank_nothr(self, sta, useSubSpaces=True, useSingles=True):
        names, kinds, U, det_off, offsets, _ = self._detector_bank(sta, useSubSpaces, useSingles,
                                                                   need_thr=False)
        return names, kinds, U, det_off, offsets, None

    # ------------------------------------------------------------ FAS
    def getFAS(self, conDatNum, LTATime=5, STATime=0.5, staltalimit=7.5, useSubSpaces=True,
               useSingles=False, numBins=401, recalc=False, device=None, seed=None, **kwargs):
        """subspace.py:1652-1743 -> fas._initFAS: empirical null distribution of
        the detection statistic over random continuous chunks, beta fit."""
        from . import fas
        fas.run_fas(self, conDatNum, LTATime=LTATime, STATime=STATime, staltalimit=staltalimit,
                    useSubSpaces=useSubSpaces, useSingles=useSingles, numBins=numBins,
                    recalc=recalc, device=device, seed=seed)

    # ------------------------------------------------------------ detex
    @_collector_set_aside
    def detex(self, utcStart=None, utcEnd=None, subspaceDB='SubSpace.db', trigCon=0,
              triggerLTATime=5, triggerSTATime=0, multiprocess=False, delOldCorrs=True,
              calcHist=True, useSubSpaces=True, useSingles=False, estimateMags=True,
              classifyEvents=None, eventCorFile='EventCors', utcSaves=None, fillZeros=False,
              device=None, ctx=None):
        # (device / ctx: the GPU and, optionally, an existing HipContext to run on)
        """subspace.py:1745-1902: run every detector over the continuous data.

        Returns {'ss_df': DataFrame, 'sg_df': DataFrame} with the reference's
        columns (detect.py:397-398) and writes them, the histograms and the
        detector info to the SQLite file ``subspaceDB`` (None: no file).
        ``estimateMags`` (detect.py:447-499) fills Mag, SNR and ProEnMag at every
        trigger (dtx_trigger_stats + detect.mags_from_stats); they are NaN with estimateMags=False as in the
        reference.  A raw ContinuousFetcher is decimated (clusters.decimate),
        detrended and band-passed (clusters.filt) chunk by chunk on the device.
        ``classifyEvents`` (a template key, a list of event names, or True; detect.py:51-55, 151-152): the detectors
        run over the event waveforms of the cluster's event fetcher instead; the result also holds 'event_cors'
        (highest detection statistic per event, station and detector), pickled to ``eventCorFile_<station>.pkl``.
        ``utcSaves`` (epoch seconds; detect.py:63-75, 297-318): 'utc_saves' holds, per detector, the data and the
        whole detection-statistic vector of every chunk that covers a listed time, appended to UTCsaves.pkl."""
        import torch
        from . import parallel
        from .detect import StationScan
        if multiprocess or trigCon != 0:                    # subspace.py:1843-1845
            _error('multiprocessing and trigcon other than 0 not supported')
        utc = None
        if utcSaves is not None:                            # detect.py:63-75
            if isinstance(utcSaves, (str, bytes)) or not hasattr(utcSaves, '__iter__'):
                _error('UTCSaves must be a list or tupple')
            try:
                # epoch seconds, ObsPy UTCDateTime (.timestamp is a property) or datetime (.timestamp() a method)
                stamps = [getattr(x, 'timestamp', x) for x in utcSaves]
                utc = np.array([float(t() if callable(t) else t) for t in stamps], dtype=np.float64)
            except (TypeError, ValueError):
                _error('Not all elements in utcSaves are readable as epoch seconds')
        if classifyEvents is None and self.cfetcher is None:
            _error('no continuous-data fetcher attached (conDatFetcher)')
        rank, world = parallel.dist_info()
        active = parallel.dist_active()     # a process group exists (also of one rank): the collectives run
        if subspaceDB is not None and delOldCorrs and rank == 0 and os.path.exists(subspaceDB):
            os.remove(subspaceDB)                           # subspace.py:1847-1854
        own_ctx = ctx is None
        if device is None:
            # one process per GPU: the launcher's LOCAL_RANK names this process's device
            device = parallel.local_device()
        dev = torch.device('cuda', int(device)) if own_ctx else ctx.device
        if active and dev.type == 'cuda':
            torch.cuda.set_device(dev)          # RCCL's object collectives run on the current device
            if not parallel.BINDING:
                # one process per GPU: onto the CPU cores / NUMA node next to this rank's GPU, before the uploader
                # allocates its page-locked staging buffers (the host feed is what the ranks share);
                # once per process, recorded in last_run['binding']
                parallel.bind_to_device(dev.index)
        fetcher = self.cfetcher
        stations = []
        results, failure = {}, None
        bounds, banks = {}, {}
        early = None                                        # the unit whose upload was started ahead of the banks
        try:
            if classifyEvents is not None:
                # detect.py:51-55, 151-152: the detectors run over the event waveforms of the cluster's
                # own event fetcher instead of over continuous data
                fetcher = self._event_chunks(classifyEvents, device if own_ctx else None, ctx, fillZeros)
            stations = [sta for sta in self.Stations if sta in fetcher.data]
            bounds = {sta: fetcher.chunk_bounds(sta, utcStart, utcEnd) for sta in stations}
            if stations and not active:
                # the call's first upload goes out before the other stations' banks are built (a bank costs the
                # host ~0.2 ms, eight stations ~2 ms that the copy engine would otherwise idle through): the
                # first unit is known from the first station's bank alone
                sta0 = stations[0]
                banks[sta0] = self._detector_bank(sta0, useSubSpaces, useSingles)
                first = self._scan_units([sta0], bounds, {sta0: None}, banks, fetcher)
                if first:
                    up0 = self._uploader(dev)
                    up0.reset()
                    up0.start(fetcher, first[0][0], bounds[first[0][0]], first[0][1], key=0)
                    early = first[0]
            for sta in stations:
                if sta not in banks:
                    banks[sta] = self._detector_bank(sta, useSubSpaces, useSingles)
        except Exception as e:                              # noqa: BLE001
            if not active:
                raise
            # the other ranks are (or will be) waiting in the broadcast below: take part in every
            # collective of the call, then fail on every rank (the gather carries the marker)
            failure = '%s: %s' % (type(e).__name__, e)
        sel = {sta: None for sta in stations}
        if active:
            # One process per GPU.  (station, day) units of ALL stations are dealt round-robin
            # (parallel.deal_station_chunks, as bench.py --config 4 deals them); rank 0's bases and
            # thresholds go to every rank in ONE broadcast (RCCL over xGMI, the only collective on
            # the data path); the rows come back in ONE gather after the last station.
            try:
                payload = failure if failure is not None else [
                    (banks[s][2], banks[s][3], [banks[s][5][nm] for nm in banks[s][0]]) if banks[s][0]
                    else (None, None, None) for s in stations]
                sent = parallel.broadcast_banks(payload, dev)
                if failure is None:
                    per = max(1, int(round(86400.0 / fetcher.conDatDuration)))
                    dealt = parallel.deal_station_chunks([len(bounds[s]) for s in stations], per, rank, world)
                    sel = dict(zip(stations, dealt))
                    self._adopt_sent_banks(stations, banks, sent)
            except Exception as e:                          # noqa: BLE001
                failure = failure or '%s: %s' % (type(e).__name__, e)
        # A station's chunks are scanned in units: the whole record in one (ragged) launch while its DS tensor
        # fits the budget, else day by day (_scan_units) -- the pipeline below keeps INFLIGHT units in flight
        todo = [] if failure is not None else self._scan_units(stations, bounds, sel, banks, fetcher)
        up = self._uploader(dev)
        started0 = bool(todo) and early is not None and _same_unit(todo[0], early)
        if not started0:
            up.reset()
        parts = {}
        # single process: the SQLite file is written by a thread of its own, fed as the units finish
        # (detect.py:207-210 flushes every 500 rows); under a process group rank 0 writes after the gather
        writer = _DbWriter(subspaceDB, fresh=bool(delOldCorrs)) if (subspaceDB is not None and not active) else None
        units_left = {}
        for sta, _ in todo:
            units_left[sta] = units_left.get(sta, 0) + 1
        hist_written = set()
        if writer is not None:
            writer.call(self._write_static_tables)      # (nothing in them depends on the scan: written under it)

        def collect(job):
            """Second half of a unit: triggers, trigger-time estimates, rows."""
            try:
                res = job.finish()
            finally:
                job.close()
            if res is not None:
                parts.setdefault(job.sta, []).append(res)
                if writer is not None:
                    writer.put('ss_df', res['ss'])
                    writer.put('sg_df', res['sg'])
            units_left[job.sta] -= 1
            if writer is not None and units_left[job.sta] == 0 and parts.get(job.sta):
                # the station's histograms are final: their rows (400 counts as text per detector, ~30 us each:
                # 5 ms for 8 stations x 25 detectors) are rendered and written under the next station's scans
                hs = tuple(_sum_hists(parts[job.sta], hk) for hk in ('hist_ss', 'hist_sg'))
                writer.call(functools.partial(self._write_hist_tables, only={job.sta: hs}))
                hist_written.add(job.sta)

        # INFLIGHT (four) units are in flight: S_i (a unit's scans) is enqueued on its station's stream, then the host
        # picks unit i - 1's triggers and enqueues T_{i-1} (its trigger-time kernel) on that context's TRIGGER stream,
        # behind S_i's scan kernel (beside S_i's post pass and the preparation of S_{i+1}: a trigger-time workgroup
        # beside a scan kernel keeps the scan off its CU), and only then waits for the oldest unit's T and builds its
        # rows: whenever the host blocks or builds rows, whole scans are still queued.  (With two in flight the wait
        # for T_i drained the queue: 2.5 ms of idle GPU per station-day; with three a host shared with other jobs
        # left the GPU waiting now and then: 3.7 - 4.7 ms per station-day from call to call.)
        queue = []                                          # jobs started, oldest first
        try:
            if todo and not started0:
                up.start(fetcher, todo[0][0], bounds[todo[0][0]], todo[0][1], key=0)
            for i, (sta, chunks) in enumerate(todo):
                if own_ctx:
                    ctx = self._station_ctx(device, sta)
                stream = getattr(ctx, 'stream', None)
                shard = up.get(i, stream)
                if i + 1 < len(todo):                       # next unit's H2D under this scan
                    up.start(fetcher, todo[i + 1][0], bounds[todo[i + 1][0]], todo[i + 1][1], key=i + 1)
                job = StationScan(self, sta, utcStart, utcEnd, triggerLTATime, triggerSTATime, calcHist,
                                  useSubSpaces, useSingles, fillZeros, device, ctx, estimateMags,
                                  shard=shard, bank=banks[sta], fetcher=fetcher, utcSaves=utc)
                del shard
                job.start()
                up.release(stream)
                queue.append(job)
                if not own_ctx:
                    # a context handed in by the caller serves every station in turn: its basis and
                    # training events are replaced by the next start(), so this unit is finished first
                    collect(queue.pop(0))
                    continue
                if len(queue) >= 2:
                    queue[-2].pick(behind=job)              # T_{i-1} goes in behind S_i's scan kernel
                if len(queue) >= INFLIGHT:
                    collect(queue.pop(0))                   # rows of the oldest unit, the younger ones' scans still queued
            while queue:
                collect(queue.pop(0))
        except Exception as e:                              # noqa: BLE001
            for job in queue:
                job.close()
            if writer is not None:
                writer.close(abort=True)
            if not active:
                raise
            # the other ranks are (or will be) waiting in the gather below: take part in it with an
            # error marker instead of leaving them hung, then fail on every rank
            failure = '%s: %s' % (type(e).__name__, e)
        results = {sta: _merge_ranks(parts.get(sta, [])) for sta in stations}
        results = {sta: r for sta, r in results.items() if r is not None}
        self.last_run = dict(rank=rank, world=world, binding=dict(parallel.BINDING), uploaded_bytes=up.bytes_uploaded,
                             chunks={sta: (len(bounds.get(sta, ())) if sel[sta] is None else len(sel[sta]))
                                     for sta in stations})
        if active:
            parts = parallel.gather_rows((results, failure))
            bad = [(r, p[1]) for r, p in enumerate(parts) if p[1] is not None]
            if bad:
                _error('SubSpace.detex failed on rank(s) %s' % '; '.join('%d (%s)' % b for b in bad))
            results = {sta: _merge_ranks([p[0].get(sta) for p in parts]) for sta in stations}
        out = {'ss_df': [], 'sg_df': []}
        for sta in stations:
            res = results.get(sta)
            if res is None:
                continue
            for kind in ('ss', 'sg'):
                out[kind + '_df'] += res[kind]
            self.histSubSpaces[sta] = res['hist_ss']
            self.histSingles[sta] = res['hist_sg']
        cols = ['DS', 'DS_STALTA', 'STMP', 'Name', 'Sta', 'MSTAMPmin', 'MSTAMPmax', 'Mag', 'SNR',
                'ProEnMag']
        for k in out:                                       # one DataFrame per table (the units hand over row lists)
            out[k] = pd.DataFrame(out[k], columns=cols)
        if writer is not None:
            self._write_db(None, out, writer, hist_written)   # the remaining histogram rows; joins the thread
        elif subspaceDB is not None and rank == 0:
            self._write_db(subspaceDB, out, fresh=bool(delOldCorrs))
        if classifyEvents is not None:
            out['event_cors'] = self._event_cors(fetcher, stations, results, banks, eventCorFile, rank)
        if utc is not None:
            out['utc_saves'] = self._utc_saves(stations, results, rank)
        return out

    def _event_chunks(self, classifyEvents, device, ctx, fillZeros=False):
        """The events a classifyEvents run scans, as one chunk each (getdata.EventChunks).  classifyEvents:
        a template key -- DataFrame with a NAME column or the path of such a csv -- or a list of event names,
        or True for every event of the cluster's event fetcher (the reference reads the key and asks that
        fetcher for the events in it, detect.py:51-55).  Raw records are cut to the cluster's trim window and go
        through what the reference applies to every chunk it scans, events included (detect.py:231:
        _applyFilter(st, filt, decimate, dtype, fillZeros=<the detex call's>)) -- NOT through createCluster's
        template rules (enforceOrigin, the short / all-zero waveform drops of construct.py:884-922): the
        reference does not apply those here either."""
        from . import construct
        from .getdata import EventChunks
        ef = self.clusters.fetcher
        if ef is None or not hasattr(ef, 'getTemData'):
            _error('classifyEvents needs the event fetcher of the ClusterStream')
        if classifyEvents is True:
            wanted = None
        elif isinstance(classifyEvents, str):
            wanted = set(str(v) for v in pd.read_csv(classifyEvents).NAME)
        elif isinstance(classifyEvents, pd.DataFrame):
            wanted = set(str(v) for v in classifyEvents.NAME)
        else:
            wanted = set(str(v) for v in classifyEvents)
        cl = self.clusters
        dec = int(cl.decimate or 1)
        stas = [sta for sta in ef.stations() if sta in self.Stations]
        rawness = set(bool(ef.has_raw(sta)) for sta in stas)
        if dec > 1 and len(rawness) > 1:
            # (EventChunks runs at ONE rate: a decimated raw station beside a pre-processed one would get the wrong one)
            _error('classifyEvents: the event fetcher mixes raw and pre-processed stations under decimate=%d' % dec)
        out = EventChunks(ef.sampling_rate / (dec if True in rawness else 1), ef.channels)
        for sta in stas:
            if ef.has_raw(sta):
                own = ctx is None
                c = self._station_ctx(device, sta) if own else ctx
                done = construct._preprocess_events(c, ef, sta, cl.filt, cl.trim, cl.decimate, fillZeros)
            else:
                done = [(nm, rec['data'], rec['starttime']) for nm, rec in ef.getTemData(sta)]
            out.add(sta, [(nm, d, t) for (nm, d, t) in done if wanted is None or nm in wanted])
        return out

    def _event_cors(self, fetcher, stations, results, banks, eventCorFile, rank):
        """The table the reference documents for classifyEvents (subspace.py:1821-1825): the highest
        detection statistic of every event on every station for every detector; pickled per station to
        ``eventCorFile + '_<station>.pkl'`` (detect.py:88-95)."""
        rows = []
        for sta in stations:
            res = results.get(sta)
            if res is None:
                continue
            thr = banks[sta][5]
            srows = []
            for k, ev in enumerate(fetcher.names[sta]):
                mx = res['maxds'].get(k)
                if mx is None:                              # (the event was shorter than the detectors)
                    continue
                for d, nm in enumerate(res['names']):
                    srows.append(dict(Event=ev, Sta=sta, Name=nm, MaxDS=float(mx[d]), Threshold=float(thr[nm]),
                                      Triggered=bool(mx[d] > thr[nm]), STMP=fetcher.bounds[sta][k][2]))
            if srows and eventCorFile is not None and rank == 0:
                pd.DataFrame(srows).to_pickle('%s_%s.pkl' % (eventCorFile, sta))
            rows += srows
        return pd.DataFrame(rows, columns=['Event', 'Sta', 'Name', 'MaxDS', 'Threshold', 'Triggered', 'STMP'])

    def _utc_saves(self, stations, results, rank):
        """detect.py:97-110: the rows of every station, joined with what an earlier run left in
        'UTCsaves.pkl', written back there."""
        cols = ['Station', 'Name', 'Threshold', 'offset', 'TS1', 'TS2', 'utcSaves', 'MPcon', 'SSdetect', 'STALTA',
                'TimeStamp', 'SampRate', 'MaxDS', 'MaxSTALTA', 'Nc', 'File']
        rows = [r for sta in stations if results.get(sta) is not None for r in results[sta].get('utc', [])]
        df = pd.DataFrame([{c: r[c] for c in cols} for r in rows], columns=cols)
        if not len(df):
            from .detect import _warn
            _warn('Failed to save data in utcSaves')        # detect.py:108-110
            return df
        if rank == 0:
            try:
                df = pd.concat([df, pd.read_pickle('UTCsaves.pkl')], ignore_index=True)
            except Exception:                               # noqa: BLE001 -- "try and read, pass" (detect.py:101-105)
                pass
            df.to_pickle('UTCsaves.pkl')
        return df

    def _scan_units(self, stations, bounds, sel, banks, fetcher):
        """[(station, chunk indices or None)] in scan order.  A station goes through the kernels in ONE
        (ragged) launch -- None: its whole record is uploaded as it is -- while the DS tensor of that launch,
        chunks x detectors x windows x 4 bytes, stays within DETEX_AMD_DS_BUDGET_MB (default 1024: a day of
        hourly chunks x 25 detectors is 0.9 GB); a longer record is cut into day-sized runs of consecutive
        chunks (the units parallel.deal_station_chunks deals), each uploaded, scanned and picked on its own.
        The rows are merged per station by (chunk, detector, order found): the tables do not depend on the cut."""
        budget = int(os.environ.get('DETEX_AMD_DS_BUDGET_MB', '1024')) << 20
        per = max(1, int(round(86400.0 / fetcher.conDatDuration)))
        Nc = len(fetcher.channels)
        units = []
        for sta in stations:
            idx = list(range(len(bounds[sta]))) if sel[sta] is None else [int(i) for i in sel[sta]]
            if not idx:
                continue
            names, U = banks[sta][0], banks[sta][2]
            if not names or len(idx) <= per:
                units.append((sta, sel[sta]))
                continue
            P = max(0, (max(int(bounds[sta][i][1]) for i in idx) * Nc - U.shape[1]) // Nc + 1)
            if len(idx) * len(names) * P * 4 <= budget:
                units.append((sta, sel[sta]))
                continue
            days = {}
            for i in idx:
                days.setdefault(i // per, []).append(i)
            units += [(sta, days[d]) for d in sorted(days)]
        # (Short units at both ends of the pipeline -- a quarter of the first unit ahead, the last quarter of the last one
        # behind -- were built in round 4 and measured a loss once the call's first upload went out ahead of the banks:
        # 4.80 / 4.87 / 4.77 ms per station-day without against 4.84 / 4.88 / 4.94 with, profiles/p6_ramp_e2e.txt; removed.)
        return units

    def _adopt_sent_banks(self, stations, banks, sent):
        """Every rank (rank 0 too) scans with the numbers that were broadcast; a bank object is kept
        while they are the ones already resident (its basis images stay on the device)."""
        for sta, (Ub, offb, thrb) in zip(stations, sent):
            if Ub is None:
                continue
            names, kinds, U, det_off, offsets, thr = banks[sta]
            held = self.__dict__.setdefault('_rank_banks', {}).get(sta)
            if (held is not None and held[0][0] == names and np.array_equal(held[1], Ub) and
                    np.array_equal(held[2], thrb)):
                banks[sta] = held[0]
            else:
                banks[sta] = (names, kinds, Ub, offb, offsets, dict(zip(names, thrb)))
                self._rank_banks[sta] = (banks[sta], Ub, thrb)

    def _uploader(self, dev):
        """One _StationUploader (copy stream + three device buffers) per device, kept for the life
        of the object."""
        cache = self.__dict__.setdefault('_uploaders', {})
        key = str(dev)
        if key not in cache:
            cache[key] = _StationUploader(dev)
        return cache[key]

    def _side_ctx(self, device, k):
        """Helper context k of a device, on a stream of its own: pre-processing of a station's short chunk
        groups beside the long one (detect.StationScan._start_ragged)."""
        import torch
        from .hip import HipContext
        cache = self.__dict__.setdefault('_side_ctxs', {})
        dev = torch.device(device) if not isinstance(device, int) else torch.device('cuda', device)
        key = (str(dev), int(k))
        if key not in cache:
            with torch.cuda.stream(torch.cuda.Stream(dev)):
                cache[key] = HipContext(dev.index or 0)   # (a context keeps the stream current at its creation)
        return cache[key]

    def _station_ctx(self, device, sta, keep=16):
        """One HipContext per (device, station), kept for the life of the object (the `keep`
        most recently used): its workspaces (channel planes, window statistics, block
        spectra: ~0.4 GB at one station-day) are allocated once, and the station's basis
        images, pair spectra and training events stay resident between calls."""
        from .hip import HipContext
        cache = self.__dict__.setdefault('_hip_ctx', {})
        key = (device, sta)
        if key in cache:
            cache[key] = cache.pop(key)                     # most recently used last
        else:
            while len(cache) >= keep:
                cache.pop(next(iter(cache))).close()
            cache[key] = HipContext(device)
        return cache[key]

    def __getstate__(self):
        d = dict(self.__dict__)
        for k in ('_hip_ctx', '_bank_cache', '_bank_aux', '_uploaders', '_rank_banks', '_side_ctxs'):
            d.pop(k, None)                                  # device contexts / buffers are not picklable
        return d

    def _write_db(self, path, out, writer=None, hist_written=(), fresh=False):
        """subspace.py:1882-1902: ss_df / sg_df, ss_info / sg_info, ss_hist /
        sg_hist, filt_params.  writer: the call's _DbWriter -- the detection rows and the tables that do not
        depend on the scan (info, filter parameters: queued when the call began) are in, or on their way into,
        the file already; the histograms follow on its thread and the thread is joined."""
        if writer is None:
            writer = _DbWriter(path, fresh=fresh)
            for k, df in out.items():
                writer.put(k, df)
            writer.call(self._write_static_tables)
        writer.call(functools.partial(self._write_hist_tables, skip=frozenset(hist_written)))
        writer.close()

    @staticmethod
    def _replace_table(con, name, cols, rows):
        """pandas' to_sql(if_exists='replace', index=False) for a small table, without pandas (3 - 4 ms a call):
        cols = [(name, 'REAL' | 'TEXT' | 'INTEGER')]."""
        con.execute('DROP TABLE IF EXISTS "%s"' % name)
        con.execute('CREATE TABLE "%s" (%s)' % (name, ', '.join('"%s" %s' % c for c in cols)))
        con.executemany('INSERT INTO "%s" VALUES (%s)' % (name, ','.join('?' * len(cols))), rows)

    def _write_static_tables(self, con):
        # subspace.py:1905-1953 _getInfoDF: ss_info / sg_info (events comma-separated); filt_params
        def beta_of(fas):
            bd = (fas or {}).get('betadist') if isinstance(fas, dict) else None
            return (float(bd[0]), float(bd[1])) if bd is not None and len(bd) > 1 else (None, None)

        def num(v):
            v = float(v)
            return None if v != v else v                     # NaN -> NULL, as to_sql writes it
        info = []
        for sta in self.ssStations:
            for _, r in self.subspaces[sta].iterrows():
                info.append((r.Name, sta, ','.join(r.Events), num(r.Threshold), int(r.NumBasis)) + beta_of(r.FAS))
        if info:
            self._replace_table(con, 'ss_info', [('Name', 'TEXT'), ('Sta', 'TEXT'), ('Events', 'TEXT'),
                                                 ('Threshold', 'REAL'), ('NumBasisUsed', 'INTEGER'),
                                                 ('beta1', 'REAL'), ('beta2', 'REAL')], info)
        info = []
        for sta in self.singStations:
            for _, r in self.singles[sta].iterrows():
                info.append((r.Name, sta, ','.join(r.Events), num(r.Threshold)) + beta_of(r.FAS))
        if info:
            self._replace_table(con, 'sg_info', [('Name', 'TEXT'), ('Sta', 'TEXT'), ('Events', 'TEXT'),
                                                 ('Threshold', 'REAL'), ('beta1', 'REAL'), ('beta2', 'REAL')], info)
        for tab in ('ss_hist', 'sg_hist'):                  # (to_sql(if_exists='replace'): rows follow per station)
            con.execute('DROP TABLE IF EXISTS "%s"' % tab)
        cl = self.clusters
        self._replace_table(con, 'filt_params', [('FREQMIN', 'REAL'), ('FREQMAX', 'REAL'), ('CORNERS', 'INTEGER'),
                                                 ('ZEROPHASE', 'INTEGER'), ('DECIMATE', 'INTEGER')],
                            [(cl.filt[0], cl.filt[1], int(cl.filt[2]), int(bool(cl.filt[3])),
                              None if cl.decimate is None else int(cl.decimate))])

    def _write_hist_tables(self, con, only=None, skip=()):
        """subspace.py:1956-1995: one row per detector, the 400 counts as a blank-separated string.  Rows are
        appended (the tables were dropped with the static ones): only = {station: (hist_ss, hist_sg)} writes those
        stations' rows, else every station of self.hist* not in `skip`."""
        for i, (tab, hd) in enumerate((('ss_hist', self.histSubSpaces), ('sg_hist', self.histSingles))):
            if only is not None:
                hd = {sta: h[i] for sta, h in only.items()}
            rows = [(sta, name, str(np.asarray(h, dtype=np.int64).tolist())[1:-1].replace(',', ''))
                    for sta, d in hd.items() if sta not in skip for name, h in (d or {}).items()]
            if rows:
                con.execute('CREATE TABLE IF NOT EXISTS "%s" ("Sta" TEXT, "Name" TEXT, "Value" TEXT)' % tab)
                con.executemany('INSERT INTO "%s" VALUES (?,?,?)' % tab, rows)

    def write(self, filename='subspace.pkl'):
        """subspace.py:2018-2026."""
        with open(filename, 'wb') as f:
            pickle.dump(self, f, pickle.HIGHEST_PROTOCOL)


# --------------------------------------------------------------------------
# loaders                                          util.py:934-969
# --------------------------------------------------------------------------

def loadClusters(filename='clust.pkl'):
    """util.py:934-950: the ClusterStream that ``ClusterStream.write()`` (``createCluster(saveclust=True)``)
    pickled; anything else in the file is an error, as in the reference (``detex.log(level='error')`` raises)."""
    cl = pd.read_pickle(filename)
    if not isinstance(cl, ClusterStream):
        _error('%s is not a ClusterStream instance' % filename)
    return cl


def loadSubSpace(filename='subspace.pkl'):
    """util.py:953-969: the SubSpace that ``SubSpace.write()`` pickled (device contexts, banks and uploaders are
    not in the file: the first ``detex()`` of the loaded object builds them again)."""
    ss = pd.read_pickle(filename)
    if not isinstance(ss, SubSpace):
        _error('%s is not a SubSpace instance' % filename)
    return ss
