"""Thin Python face of the C ABI: one ``HipContext`` per GPU.

torch is used for device memory, streams and (elsewhere) torch.distributed --
plumbing only; all arithmetic happens in the hand-written HIP kernels of
libdetexhip.so.
"""
import contextlib
import ctypes as C
import functools
import os

import numpy as np
import torch

from . import _lib

HIST_BINS = 400


def bandpass_sos(freqmin, freqmax, df, corners):
    """Second-order sections of obspy.signal.filter.bandpass (third-party; restated:
    Butterworth band-pass designed as zpk and converted with zpk2sos)."""
    from scipy.signal import iirfilter, zpk2sos
    fe = 0.5 * df
    low, high = freqmin / fe, freqmax / fe
    if high - 1.0 > -1e-6 or low > 1:
        raise ValueError('band-pass corner frequencies must lie below Nyquist')
    z, p, k = iirfilter(corners, [low, high], btype='band', ftype='butter', output='zpk')
    return zpk2sos(z, p, k)


def decimate_sos(df, factor, maxorder=12):
    """Second-order sections of the anti-alias filter of obspy Trace.decimate(factor)
    (construct.py:1016-1017; third-party, restated): lowpass_cheby_2 with the stop band
    at the new Nyquist frequency df/(2 factor), 96 dB attenuation, 1 dB pass-band ripple;
    the pass band edge is lowered 1 % at a time until the order is <= maxorder."""
    from scipy.signal import cheb2ord, cheby2, zpk2sos
    if factor > 16:                      # obspy refuses: decimate in several steps
        raise ValueError('automatic filter design is unstable for decimation factors above 16')
    nyquist = df * 0.5
    freq = df * 0.5 / float(factor)
    rp, rs, order = 1, 96, 1e99
    ws = min(freq / nyquist, 1.0)
    wp = ws
    while order > maxorder:
        wp = wp * 0.99
        order, wn = cheb2ord(wp, ws, rp, rs, analog=0)
    z, p, k = cheby2(order, rs, wn, btype='low', analog=0, output='zpk')
    return zpk2sos(z, p, k)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def h2d_async(arr, device):
    """A small host array to the device without blocking the host: staged in pinned memory and
    copied on the current stream.  (A pageable torch.from_numpy(...).to(device) waits for
    everything already queued on that stream -- inside SubSpace.detex that is the next stations'
    scans.)"""
    t = torch.from_numpy(np.ascontiguousarray(arr))
    if device.type != 'cuda':
        return t.to(device)
    return t.pin_memory().to(device, non_blocking=True)


def _on_ctx_stream(fn):
    """Run a HipContext method with the context's stream as torch's current stream (see
    HipContext.on_stream)."""
    @functools.wraps(fn)
    def wrapped(self, *a, **kw):
        with self.on_stream():
            return fn(self, *a, **kw)
    return wrapped


class HipContext(object):
    """One HIP device + stream.  All tensors passed in must live on it.

    Stream contract: the kernels run on the stream that was torch's current stream when the
    context was created (``ctx.stream``).  Every method enters that stream before it allocates,
    records or waits for anything, so tensors, events and kernels always agree on the stream --
    also when the caller has since made another stream current (``with torch.cuda.stream(s)``).
    In that case the context's stream first waits for the caller's (inputs produced there are
    complete) and the caller's stream waits for the context's afterwards (results are ordered
    for whatever the caller enqueues next): a call behaves as if it had been enqueued on the
    caller's stream.  Result tensors are allocated from the context stream's pool: a caller that
    consumes them on yet another stream than the one the call was made from orders that stream
    itself (``t.record_stream(s)`` / ``s.wait_stream(ctx.stream)``), as with any torch tensor."""
    packed_results = True        # subspace_scan(packed=True): histogram / maxima / candidates in one allocation, one copy back

    def __init__(self, device=0):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise _lib.DetexHipError('no HIP device visible to torch; detex_amd '
                                     'has no CPU fallback')
        self.device = torch.device('cuda', int(device))
        torch.cuda.set_device(self.device)
        self._stream = torch.cuda.current_stream(self.device)
        h = C.c_void_p()
        rc = self.lib.dtx_ctx_create(int(device), C.c_void_p(self._stream.cuda_stream), 0,
                                     C.byref(h))
        _lib.check(self.lib, None, rc, 'dtx_ctx_create')
        self.h = h
        self.D = self.n = self.Nc = 0
        self.engine = 'auto'        # library default (dtx_set_engine)
        # DETEX_AMD_ENGINE=auto|fft|bf16x3|fp32 selects the arithmetic engine of new contexts
        env = os.environ.get('DETEX_AMD_ENGINE')
        if env:
            self.set_engine(env)

    @property
    def stream(self):
        """The torch stream the context enqueues on (the one handed to dtx_ctx_create)."""
        return self._stream

    @contextlib.contextmanager
    def on_stream(self):
        """Make the context's stream current for the body; order it against the caller's current
        stream on the way in and out if the two differ (class docstring)."""
        cur = torch.cuda.current_stream(self.device)
        if cur == self._stream:
            yield
            return
        self._stream.wait_stream(cur)
        try:
            with torch.cuda.stream(self._stream):
                yield
        finally:
            # also when the body raised (a failing batch is logged and skipped, detect.py:160-174):
            # whatever it did enqueue stays ordered in front of the caller's next work
            cur.wait_stream(self._stream)

    def close(self):
        if getattr(self, 'h', None):
            self.lib.dtx_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc, what):
        _lib.check(self.lib, self.h, rc, what)

    def set_engine(self, engine):
        """'auto' (default: 'fft' for templates of 256..32768 samples per channel,
        else 'bf16x3'), 'fft' (frequency-domain overlap-save engine, float32
        transforms), 'bf16x3' (split-bfloat16 MFMA contraction, float32-level
        accuracy) or 'fp32' (float32 MFMA, products bitwise an fma chain)."""
        code = {'fp32': 0, 'bf16x3': 1, 'fft': 2, 'auto': 3}[engine]
        self._ck(self.lib.dtx_set_engine(self.h, code), 'dtx_set_engine')
        self.engine = engine

    def sync(self):
        self._ck(self.lib.dtx_ctx_sync(self.h), 'dtx_ctx_sync')

    # ------------------------------------------------------------ subspace
    @_on_ctx_stream
    def upload_basis(self, U, det_off, Nc):
        """U: [Ktot, n] float32 (numpy or torch, host or device);
        det_off: D+1 ints (detector d owns rows det_off[d]:det_off[d+1])."""
        if isinstance(U, torch.Tensor):
            Ut = U.to(torch.float32).contiguous()
        else:
            Ut = torch.from_numpy(np.ascontiguousarray(U, dtype=np.float32))
        off = np.ascontiguousarray(det_off, dtype=np.int32)
        if Ut.dim() != 2 or off.ndim != 1 or off[-1] != Ut.shape[0]:
            raise ValueError('U must be [Ktot, n] and det_off[-1] == Ktot')
        D = len(off) - 1
        rc = self.lib.dtx_basis_upload(self.h, _ptr(Ut), off.ctypes.data_as(C.c_void_p),
                                       D, int(Ut.shape[1]), int(Nc))
        self._ck(rc, 'dtx_basis_upload')
        self.D, self.n, self.Nc = D, int(Ut.shape[1]), int(Nc)
        self._basis_of = None          # scan_station's residency token: any direct upload clears it

    def _check_out(self, out, nchunk, P):
        """A caller-supplied DS tensor is addressed as ds[(chunk * D + det) * ldp + p]: it must be a
        float32 CUDA tensor [nchunk, D, >= P] with unit window stride and rows packed per chunk."""
        if (out.dtype != torch.float32 or not out.is_cuda or out.dim() != 3 or
                int(out.shape[0]) != int(nchunk) or int(out.shape[1]) != self.D or
                int(out.shape[2]) < max(int(P), 0) or out.stride(2) != 1 or
                (out.shape[0] > 1 and out.stride(0) != self.D * out.stride(1)) or out.stride(1) < P):
            raise ValueError('out must be a float32 CUDA tensor [nchunk=%d, D=%d, >= P=%d] with unit window '
                             'stride and stride(0) == D * stride(1); got shape %s, strides %s, %s'
                             % (nchunk, self.D, P, tuple(out.shape), tuple(out.stride()), out.dtype))

    def num_windows(self, M):
        return int(self.lib.dtx_num_windows(int(M), self.n, self.Nc))

    @_on_ctx_stream
    def subspace_ds(self, x, M=None, chunk_stride=None, nchunk=None, out=None):
        """Detection statistic of every uploaded detector.

        x: float32 CUDA tensor.  Either [nchunk, M] (contiguous rows) or a flat
        buffer together with M / chunk_stride / nchunk (overlapping blocks).
        Returns ds [nchunk, D, P] (asynchronous on torch's current stream)."""
        if x.dtype != torch.float32 or not x.is_cuda:
            raise ValueError('x must be a float32 CUDA tensor')
        if M is None:
            if x.dim() == 1:
                x = x[None, :]
            if x.stride(-1) != 1:
                x = x.contiguous()
            nchunk, M = int(x.shape[0]), int(x.shape[1])
            chunk_stride = int(x.stride(0)) if nchunk > 1 else M
        else:
            need = (nchunk - 1) * chunk_stride + M
            if x.numel() < need or not x.is_contiguous():
                raise ValueError('flat buffer too small for the chunk layout')
        P = self.num_windows(M)
        if out is None:
            # row pitch padded to 64 floats: 16-byte aligned rows for the post pass and
            # whole-line stores from the MFMA epilogue
            ldp = (max(P, 1) + 63) // 64 * 64
            out = torch.empty((nchunk, self.D, ldp), dtype=torch.float32,
                              device=x.device)[:, :, :max(P, 0)]
        else:
            self._check_out(out, nchunk, P)
        rc = self.lib.dtx_subspace_ds(self.h, _ptr(x), int(M), int(chunk_stride),
                                      int(nchunk), _ptr(out), int(out.stride(1)))
        self._ck(rc, 'dtx_subspace_ds')
        return out

    def _post_buffers(self, nchunk, D, dev, thr, hist, want_max, cand_cap, packed=False):
        if packed:
            # ONE allocation for everything small the post pass writes -- [histogram | candidate count | maxima |
            # candidate values] as 4-byte words: one fill clears histogram and count, and ONE copy brings
            # histogram, count, maxima and the first candidates' values to the host (_lazy_fetch).  (Apart: two
            # fills and five small copies per scan; each small operation costs its stream 10 - 20 us of bubbles.)
            if hist is not None or thr is None or not want_max:
                raise ValueError('packed results: histogram, maxima and candidates all belong to the call')
            nh, nm = D * HIST_BINS, nchunk * D
            pack = torch.empty((nh + 1 + nm + int(cand_cap),), dtype=torch.int32, device=dev)
            pack[:nh + 1].zero_()
            hist = pack[:nh].view(D, HIST_BINS)
            cnt = pack[nh:nh + 1]
            maxds = pack[nh + 1:nh + 1 + nm].view(torch.float32).view(nchunk, D)
            val = pack[nh + 1 + nm:].view(torch.float32)
            thr_t = thr if isinstance(thr, torch.Tensor) else torch.as_tensor(
                np.broadcast_to(np.asarray(thr, dtype=np.float32), (D,)).copy(), device=dev)
            key = torch.empty((cand_cap,), dtype=torch.int64, device=dev)
            return hist, maxds, thr_t, key, val, cnt, pack
        if hist is None:
            hist = torch.zeros((D, HIST_BINS), dtype=torch.int32, device=dev)
        elif (hist.dtype != torch.int32 or not hist.is_cuda or tuple(hist.shape) != (D, HIST_BINS) or
              not hist.is_contiguous()):
            raise ValueError('hist must be a contiguous int32 CUDA tensor [D=%d, %d]' % (D, HIST_BINS))
        maxds = torch.empty((nchunk, D), dtype=torch.float32, device=dev) if want_max else None
        thr_t = key = val = cnt = None
        if thr is not None:
            if isinstance(thr, torch.Tensor):      # already on the device (a scan loop keeps it there)
                thr_t = thr
            else:
                thr_t = torch.as_tensor(np.broadcast_to(np.asarray(thr, dtype=np.float32), (D,)).copy(),
                                        device=dev)
            key = torch.empty((cand_cap,), dtype=torch.int64, device=dev)
            val = torch.empty((cand_cap,), dtype=torch.float32, device=dev)
            cnt = torch.zeros((1,), dtype=torch.int32, device=dev)
        return hist, maxds, thr_t, key, val, cnt, None

    @_on_ctx_stream
    def chunk_table(self, starts, lengths):
        """The device table of a ragged batch (dtx_subspace_scan_ragged): int64 [nchunk, 2] =
        (offset of the chunk's first sample in the flat buffer, its samples)."""
        tab = np.ascontiguousarray(np.stack([np.asarray(starts, dtype=np.int64),
                                             np.asarray(lengths, dtype=np.int64)], axis=1))
        if tab.ndim != 2 or len(tab) == 0 or (tab < 0).any():
            raise ValueError('chunk table: one non-negative (start, length) pair per chunk')
        return h2d_async(tab, self.device), tab

    @_on_ctx_stream
    def subspace_scan(self, x, M=None, chunk_stride=None, nchunk=None, out=None, thr=None,
                      hist=None, want_max=True, cand_cap=1 << 20, lazy=False, chunk_tab=None, packed=False):
        """subspace_ds + ds_post in one call (dtx_subspace_scan).
        packed=True (with lazy=True, thr given, hist=None): the call owns a zeroed histogram; fetch() returns it
        on the HOST (res['hist'], int32 [D, 400]) with the maxima and candidates -- one copy (_post_buffers).
        Returns (ds, result dict as ds_post).  lazy=True returns (ds, fetch) instead, where
        fetch() synchronises and builds that dict: nothing here then waits for the device, so
        independent scans (other contexts on other streams) can be enqueued behind each other.

        chunk_tab = self.chunk_table(starts, lengths): a ragged batch (dtx_subspace_scan_ragged) --
        chunk b is x.flatten()[start_b : start_b + length_b]; ds has the row length of the longest
        chunk, entries past a chunk's own window count are unspecified, and histogram / maximum /
        candidates do not cover them."""
        if x.dtype != torch.float32 or not x.is_cuda:
            raise ValueError('x must be a float32 CUDA tensor')
        if chunk_tab is not None:
            tab_dev, tab = chunk_tab
            if not x.is_contiguous():
                raise ValueError('a ragged batch reads one flat buffer')
            nchunk, M, chunk_stride = len(tab), int(tab[:, 1].max()), 0
            bad = self.lib.dtx_chunk_table_check(tab.ctypes.data_as(C.c_void_p), nchunk, M, int(x.numel()))
            if bad:
                raise ValueError('flat buffer too small for the chunk table (chunk %d)' % (bad - 1))
        elif M is None:
            if x.dim() == 1:
                x = x[None, :]
            if x.stride(-1) != 1:
                x = x.contiguous()
            nchunk, M = int(x.shape[0]), int(x.shape[1])
            chunk_stride = int(x.stride(0)) if nchunk > 1 else M
        else:
            need = (nchunk - 1) * chunk_stride + M
            if x.numel() < need or not x.is_contiguous():
                raise ValueError('flat buffer too small for the chunk layout')
        P = self.num_windows(M)
        if out is None:
            ldp = (max(P, 1) + 63) // 64 * 64
            out = torch.empty((nchunk, self.D, ldp), dtype=torch.float32,
                              device=x.device)[:, :, :max(P, 0)]
        else:
            self._check_out(out, nchunk, P)
        if packed and not lazy:
            raise ValueError('packed results come through fetch(): lazy=True')
        hist, maxds, thr_t, key, val, cnt, pack = self._post_buffers(nchunk, self.D, x.device, thr, hist,
                                                                     want_max, cand_cap, packed)
        # a stream of its own for the post pass when the caller pipelines (lazy: results fetched later, the next scan
        # enqueued meanwhile); a plain call reads the results at once, the pass stays on the context's stream
        post = self._ensure_post_stream(lazy)
        if post is not None:
            for t in (thr_t, hist, maxds, key, val, cnt, pack, out):     # (from the context stream's pool, used on the post pass's stream)
                if t is not None:
                    t.record_stream(post)
        if chunk_tab is not None:
            rc = self.lib.dtx_subspace_scan_ragged(self.h, _ptr(x), _ptr(tab_dev), int(M), int(nchunk),
                                                   _ptr(out), int(out.stride(1)), _ptr(hist), _ptr(maxds),
                                                   _ptr(thr_t), _ptr(key), _ptr(val), int(cand_cap), _ptr(cnt))
        else:
            rc = self.lib.dtx_subspace_scan(self.h, _ptr(x), int(M), int(chunk_stride), int(nchunk),
                                            _ptr(out), int(out.stride(1)), _ptr(hist), _ptr(maxds),
                                            _ptr(thr_t), _ptr(key), _ptr(val), int(cand_cap), _ptr(cnt))
        self._ck(rc, 'dtx_subspace_scan')
        if lazy:
            return out, self._lazy_fetch(hist, maxds, thr, key, val, cnt, cand_cap, pack=pack)
        return out, self._post_result(hist, maxds, thr, key, val, cnt, cand_cap)

    def _ensure_post_stream(self, on=True):
        """A stream of its own for the scans' post pass (dtx_ctx_set_post_stream): in a pipelined loop -- several
        scans enqueued before their results are fetched -- the pass then runs beside the next scan's preparation.
        Results are complete on THAT stream: _lazy_fetch records its event there.  on=False: back on the context's
        stream (the library first orders that stream behind a pass still running on the other)."""
        if on:
            if getattr(self, '_post', None) is None:
                self._post = torch.cuda.Stream(self.device)
            if not getattr(self, '_post_on', False):
                self._ck(self.lib.dtx_ctx_set_post_stream(self.h, C.c_void_p(self._post.cuda_stream)),
                         'dtx_ctx_set_post_stream')
                self._post_on = True
            return self._post
        if getattr(self, '_post_on', False):
            self._ck(self.lib.dtx_ctx_set_post_stream(self.h, None), 'dtx_ctx_set_post_stream')
            self._post_on = False
        return None

    def _lazy_fetch(self, hist, maxds, thr, key, val, cnt, cand_cap, fast=1 << 14, pack=None):
        """The small results of an enqueued scan copied to pinned host memory on a side stream
        as soon as the scan is done: the returned fetch() waits for exactly that copy, not for
        whatever was enqueued on the context's stream after the scan (the next station's scan)."""
        main = self._post if getattr(self, '_post_on', False) else self._stream     # where the post pass ran
        if main is not self._stream:
            for t in (hist, maxds, key, val, cnt, pack):     # (allocated from the context stream's pool, used on the post pass's)
                if t is not None:
                    t.record_stream(main)
        if pack is not None:
            # packed results: written into page-locked host memory by a kernel on the context's own stream
            # (dtx_export_results) -- no second stream, no copy engine, no event between streams: with several scans in
            # flight the side stream's copies and their barrier packets sat in the hardware queue in front of the next
            # scan's first kernel (~80 us idle behind every post pass, profiles/p7_e2e_timeline.txt)
            nf = min(int(cand_cap), int(fast))
            nh, nm = hist.numel(), maxds.numel()
            hp = torch.empty((nh + 1 + nm + nf,), dtype=torch.int32, pin_memory=True)
            hk = torch.empty((nf,), dtype=torch.int64, pin_memory=True)
            self._ck(self.lib.dtx_export_results(self.h, _ptr(pack), hp.data_ptr(), nh + 1 + nm, nf, nh,
                                                 _ptr(key), hk.data_ptr()), 'dtx_export_results')
            done = torch.cuda.Event()
            done.record(main)
            # (the pinned buffers are written by a kernel torch's host allocator knows nothing about: the last few stay
            # referenced here, so that a fetch() nobody calls cannot hand one back while its kernel is still queued)
            keep = getattr(self, '_export_keep', None)
            if keep is None:
                import collections
                keep = self._export_keep = collections.deque(maxlen=16)
            keep.append((hp, hk, pack, key, thr))
            host = {'key': hk, 'hist': hp[:nh].view(hist.shape), 'cnt': hp[nh:nh + 1],
                    'maxds': hp[nh + 1:nh + 1 + nm].view(torch.float32).view(maxds.shape),
                    'val': hp[nh + 1 + nm:].view(torch.float32)}
            return self._fetcher(done, host, hist, maxds, thr, key, val, cnt, cand_cap)
        if getattr(self, '_side', None) is None:
            self._side = torch.cuda.Stream(self.device)
        side = self._side
        ev = torch.cuda.Event()
        ev.record(main)
        host = {}
        with torch.cuda.stream(side):
            side.wait_event(ev)
            if pack is not None:
                nf = min(int(cand_cap), int(fast))
                nh, nm = hist.numel(), maxds.numel()
                hp = torch.empty((nh + 1 + nm + nf,), dtype=torch.int32, pin_memory=True)
                hp.copy_(pack[:nh + 1 + nm + nf], non_blocking=True)
                host['key'] = torch.empty((nf,), dtype=torch.int64, pin_memory=True)
                host['key'].copy_(key[:nf], non_blocking=True)
                pack.record_stream(side)
                key.record_stream(side)
                host['hist'] = hp[:nh].view(hist.shape)
                host['cnt'] = hp[nh:nh + 1]
                host['maxds'] = hp[nh + 1:nh + 1 + nm].view(torch.float32).view(maxds.shape)
                host['val'] = hp[nh + 1 + nm:].view(torch.float32)
            elif maxds is not None:
                host['maxds'] = torch.empty(maxds.shape, dtype=maxds.dtype, pin_memory=True)
                host['maxds'].copy_(maxds, non_blocking=True)
                maxds.record_stream(side)
            if thr is not None and pack is None:
                nf = min(int(cand_cap), int(fast))
                host['cnt'] = torch.empty((1,), dtype=torch.int32, pin_memory=True)
                host['key'] = torch.empty((nf,), dtype=torch.int64, pin_memory=True)
                host['val'] = torch.empty((nf,), dtype=torch.float32, pin_memory=True)
                host['cnt'].copy_(cnt, non_blocking=True)
                host['key'].copy_(key[:nf], non_blocking=True)
                host['val'].copy_(val[:nf], non_blocking=True)
                for t in (cnt, key, val):
                    t.record_stream(side)
            done = torch.cuda.Event()
            done.record(side)
        return self._fetcher(done, host, hist, maxds, thr, key, val, cnt, cand_cap)

    def _fetcher(self, done, host, hist, maxds, thr, key, val, cnt, cand_cap):
        def fetch():
            done.synchronize()
            res = dict(hist=host.get('hist', hist), maxds=host.get('maxds', maxds), cand=None, ncand=0)
            if thr is not None:
                n = int(host['cnt'][0])
                if n > len(host['key']):          # more candidates than the fast path copies: the plain way
                    self.sync()
                    r = self._post_result(hist, host.get('maxds', maxds), thr, key, val, cnt, cand_cap)
                    r['hist'] = host.get('hist', hist)
                    return r
                res['ncand'] = n
                k = host['key'][:n].numpy().astype(np.uint64)
                v = host['val'][:n].numpy().copy()
                order = np.argsort(k, kind='stable')
                k, v = k[order], v[order]
                res['cand'] = ((k >> np.uint64(48)).astype(np.int64),
                               ((k >> np.uint64(32)) & np.uint64(0xFFFF)).astype(np.int64),
                               (k & np.uint64(0xFFFFFFFF)).astype(np.int64), v)
            return res
        return fetch

    @_on_ctx_stream
    def ds_post(self, ds, thr=None, hist=None, want_max=True, cand_cap=1 << 20):
        """Histogram / max / threshold candidates of a DS tensor [nchunk, D, P].

        Returns dict(hist=[D,400] uint32-as-int64 tensor or None, maxds=[nchunk,D],
        cand=(chunk, det, index, value) numpy arrays sorted by (chunk, det, index),
        ncand=total found)."""
        nchunk, D, P = (int(s) for s in ds.shape)
        hist, maxds, thr_t, key, val, cnt, _ = self._post_buffers(nchunk, D, ds.device, thr, hist,
                                                                  want_max, cand_cap)
        rc = self.lib.dtx_ds_post(self.h, _ptr(ds), P, int(ds.stride(1)), nchunk, D,
                                  _ptr(hist), _ptr(maxds), _ptr(thr_t), _ptr(key), _ptr(val),
                                  int(cand_cap), _ptr(cnt))
        self._ck(rc, 'dtx_ds_post')
        return self._post_result(hist, maxds, thr, key, val, cnt, cand_cap)

    @staticmethod
    def _post_result(hist, maxds, thr, key, val, cnt, cand_cap):
        res = dict(hist=hist, maxds=maxds, cand=None, ncand=0)
        if thr is not None:
            n = int(cnt.item())
            res['ncand'] = n
            m = min(n, cand_cap)
            k = key[:m].cpu().numpy().astype(np.uint64)
            v = val[:m].cpu().numpy()
            order = np.argsort(k, kind='stable')
            k, v = k[order], v[order]
            res['cand'] = ((k >> np.uint64(48)).astype(np.int64),
                           ((k >> np.uint64(32)) & np.uint64(0xFFFF)).astype(np.int64),
                           (k & np.uint64(0xFFFFFFFF)).astype(np.int64), v)
        return res

    @_on_ctx_stream
    def ds_moments(self, ds, lo=-.01, hi=1.0, nbins=400, hist=None, sums=None):
        """FAS pass over a DS tensor [nchunk, D, P]: histogram on
        np.linspace(lo, hi, nbins+1) and the beta-MLE sufficient statistics
        [D, 5] (n, sum x, sum x^2, sum log x, sum log1p(-x)); both accumulate."""
        nchunk, D, P = (int(s) for s in ds.shape)
        if hist is None:
            hist = torch.zeros((D, nbins), dtype=torch.int32, device=ds.device)
        elif (hist.dtype != torch.int32 or not hist.is_cuda or tuple(hist.shape) != (D, nbins) or
              not hist.is_contiguous()):
            raise ValueError('hist must be a contiguous int32 CUDA tensor [D=%d, nbins=%d]' % (D, nbins))
        if sums is None:
            sums = torch.zeros((D, 5), dtype=torch.float64, device=ds.device)
        elif (sums.dtype != torch.float64 or not sums.is_cuda or tuple(sums.shape) != (D, 5) or
              not sums.is_contiguous()):
            raise ValueError('sums must be a contiguous float64 CUDA tensor [D=%d, 5]' % D)
        rc = self.lib.dtx_ds_moments(self.h, _ptr(ds), P, int(ds.stride(1)), nchunk, D,
                                     float(lo), float(hi), int(nbins), _ptr(hist), _ptr(sums))
        self._ck(rc, 'dtx_ds_moments')
        return hist, sums

    @_on_ctx_stream
    def subspace_ds_host(self, x):
        """numpy in, numpy out (one block); includes the PCIe transfers."""
        x = np.ascontiguousarray(x, dtype=np.float32)
        P = self.num_windows(len(x))
        out = np.empty((self.D, max(P, 0)), dtype=np.float32)
        rc = self.lib.dtx_subspace_ds_host(self.h, x.ctypes.data_as(C.c_void_p), len(x),
                                           out.ctypes.data_as(C.c_void_p))
        self._ck(rc, 'dtx_subspace_ds_host')
        return out

    @_on_ctx_stream
    def trigger_stats(self, x, M, chunk_stride, ds, trig, sta_len, lta_len, ewf, eoff, emax, chunk_tab=None):
        return self._trigger_stats(x, M, chunk_stride, ds, trig, sta_len, lta_len, ewf, eoff, emax, chunk_tab)

    def _trigger_stats(self, x, M, chunk_stride, ds, trig, sta_len, lta_len, ewf, eoff, emax, chunk_tab=None):
        """dtx_trigger_stats: STA / LTA means and the magnitude / SNR ingredients at triggers.

        x, M, chunk_stride, ds: the buffers of the subspace_scan call the triggers come from;
        trig: int32 array [ntrig, 5] (chunk, detector, window, STA start or -1, LTA start or -1);
        ewf: [Etot, n] float32 CUDA tensor, eoff: [D+1] int32 CUDA tensor; ewf=None: STA / LTA
        only (columns 0, 1; the rest stays NaN).
        Returns a float64 CUDA tensor [ntrig, 6 + emax] (asynchronous)."""
        trig = np.ascontiguousarray(trig, dtype=np.int32).reshape(-1, 5)
        nt = len(trig)
        out = torch.full((nt, 6 + int(emax)), float('nan'), dtype=torch.float64, device=ds.device)
        if nt == 0:
            return out
        tdev = h2d_async(trig, ds.device)
        nchunk, D, P = (int(v) for v in ds.shape)
        if chunk_tab is not None:       # the triggers of a ragged batch (M = its longest chunk)
            rc = self.lib.dtx_trigger_stats_ragged(self.h, _ptr(x), _ptr(chunk_tab[0]), int(M), _ptr(ds), P,
                                                   int(ds.stride(1)), _ptr(tdev), nt, int(sta_len), int(lta_len),
                                                   _ptr(ewf) if ewf is not None else None,
                                                   _ptr(eoff) if ewf is not None else None, int(emax), _ptr(out))
        else:
            rc = self.lib.dtx_trigger_stats(self.h, _ptr(x), int(M), int(chunk_stride), _ptr(ds), P,
                                            int(ds.stride(1)), _ptr(tdev), nt, int(sta_len), int(lta_len),
                                            _ptr(ewf) if ewf is not None else None,
                                            _ptr(eoff) if ewf is not None else None, int(emax), _ptr(out))
        self._ck(rc, 'dtx_trigger_stats')
        return out

    def trigger_stream(self):
        """The stream the trigger-time kernel runs on when asked to run beside the scans
        (trigger_stats_async): created and registered with the library on first use."""
        if getattr(self, '_trig', None) is None:
            self._trig = torch.cuda.Stream(self.device)
        return self._trig

    def wait_scan_kernel(self, stream):
        """`stream` waits for the end of this context's last scan KERNEL (dtx_stream_wait_scan_kernel)."""
        self._ck(self.lib.dtx_stream_wait_scan_kernel(self.h, C.c_void_p(stream.cuda_stream)),
                 'dtx_stream_wait_scan_kernel')

    def trigger_stats_async(self, x, M, chunk_stride, ds, trig, sta_len, lta_len, ewf, eoff, emax, chunk_tab=None,
                            after=None, behind=None):
        """trigger_stats with its rows on their way to pinned host memory: returns (host tensor, event to wait
        for, device tensor).  The kernel -- a few dozen workgroups of dependent passes -- and the copies run on
        the context's trigger stream: `after` is an event recorded behind the scan that produced ``ds`` (default:
        the stream's tail now); `behind` is the context whose scan is queued NEXT: the kernel then also waits for
        that scan KERNEL to end and runs beside its post pass and the following preparation (small kernels that
        leave room on the CUs) instead of beside a scan kernel, whose workgroups it would keep off every CU it
        occupies."""
        side = self.trigger_stream()
        trig = np.ascontiguousarray(trig, dtype=np.int32).reshape(-1, 5)
        if side is None or len(trig) == 0:
            with self.on_stream():
                out = self.trigger_stats(x, M, chunk_stride, ds, trig, sta_len, lta_len, ewf, eoff, emax,
                                         **({} if chunk_tab is None else {'chunk_tab': chunk_tab}))
                host = torch.empty(out.shape, dtype=out.dtype, pin_memory=True)
                host.copy_(out, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(self._stream)
            return host, ev, out
        if after is None:
            after = torch.cuda.Event()
            after.record(self._stream)
        side.wait_event(after)
        if behind is not None:
            behind.wait_scan_kernel(side)
        rc0 = self.lib.dtx_ctx_set_trigger_stream(self.h, C.c_void_p(side.cuda_stream))
        self._ck(rc0, 'dtx_ctx_set_trigger_stream')
        try:
            with torch.cuda.stream(side):
                out = self._trigger_stats(x, M, chunk_stride, ds, trig, sta_len, lta_len, ewf, eoff, emax, chunk_tab)
                host = torch.empty(out.shape, dtype=out.dtype, pin_memory=True)
                host.copy_(out, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(side)
        finally:
            self._ck(self.lib.dtx_ctx_set_trigger_stream(self.h, None), 'dtx_ctx_set_trigger_stream')
        # buffers of the context stream's pool read by the side stream: not handed out again before it is done
        for t in (x, ds, ewf, eoff, chunk_tab[0] if chunk_tab is not None else None):
            if t is not None and t.is_cuda:
                t.record_stream(side)
        return host, ev, out

    @_on_ctx_stream
    def normcorr(self, t, s):
        """fast_normcorr(t, s) (construct.py:469-483) on the device: t [n], s [m >= n] float32 CUDA
        tensors -> [m - n + 1] float32 (asynchronous)."""
        if t.numel() > s.numel():               # construct.py:473-474: the shorter one is the template
            t, s = s, t
        t = t.to(torch.float32).contiguous()
        s = s.to(torch.float32).contiguous()
        out = torch.empty((s.numel() - t.numel() + 1,), dtype=torch.float32, device=s.device)
        rc = self.lib.dtx_normcorr(self.h, _ptr(t), int(t.numel()), _ptr(s), int(s.numel()), _ptr(out))
        self._ck(rc, 'dtx_normcorr')
        return out

    @_on_ctx_stream
    def stalta_max(self, x, starts, lengths, nsta, nlta, stride=1):
        """max of obspy's classic_sta_lta over each trace x.flat[starts[k] + j * stride], j < lengths[k]
        (fas.py:175-205: the screen of the FAS chunks) -> float64 CUDA tensor [len(starts)] (asynchronous)."""
        if x.dtype != torch.float32 or not x.is_cuda:
            raise ValueError('x must be a float32 CUDA tensor')
        x = x.contiguous()
        tab = np.empty((len(starts), 2), dtype=np.int64)
        tab[:, 0], tab[:, 1] = starts, lengths
        if len(tab) == 0 or tab.min() < 0 or int((tab[:, 0] + (tab[:, 1] - 1).clip(0) * stride).max()) >= x.numel():
            raise ValueError('traces do not fit the buffer')
        out = torch.empty((len(tab),), dtype=torch.float64, device=x.device)
        tdev = h2d_async(tab.reshape(-1), x.device)
        rc = self.lib.dtx_stalta_max(self.h, _ptr(x), _ptr(tdev), len(tab), int(tab[:, 1].max()), int(stride),
                                     int(nsta), int(nlta), _ptr(out))
        self._ck(rc, 'dtx_stalta_max')
        return out

    # ------------------------------------------------------------ xcorr
    @_on_ctx_stream
    def xcorr_allpairs(self, td, Nc, pair_begin=0, pair_end=None):
        """td: [N, n] float32 CUDA tensor of multiplexed events.  Returns
        (cc, lag, subsamp) condensed CUDA tensors of length N(N-1)/2."""
        if td.dtype != torch.float32 or not td.is_cuda or td.dim() != 2:
            raise ValueError('td must be a [N, n] float32 CUDA tensor')
        td = td.contiguous()
        N, n = int(td.shape[0]), int(td.shape[1])
        npair = N * (N - 1) // 2
        if pair_end is None:
            pair_end = npair
        cc = torch.zeros((npair,), dtype=torch.float32, device=td.device)
        lag = torch.zeros((npair,), dtype=torch.int32, device=td.device)
        ss = torch.zeros((npair,), dtype=torch.float32, device=td.device)
        rc = self.lib.dtx_xcorr_allpairs(self.h, _ptr(td), N, n, int(Nc), _ptr(cc), _ptr(lag),
                                         _ptr(ss), int(pair_begin), int(pair_end))
        self._ck(rc, 'dtx_xcorr_allpairs')
        return cc, lag, ss

    # ------------------------------------------------------------ pre-processing
    @_on_ctx_stream
    def preprocess(self, raw, filt=None, sampling_rate=None, detrend=True, out=None):
        """Detrend + band-pass + multiplex on the device (construct.py:990-1030).

        raw : [nchunk, Nc, T] float32 CUDA tensor (channels in st.sort() order), packed or a view with unit
        sample stride -- chunks where they lie in a longer [Nc, Tday] record, overlapping or not
        (dtx_preprocess_planes: no packed copy);
        filt : [freqmin, freqmax, corners, zerophase] as in the reference, or None.
        Returns [nchunk, Nc*T] multiplexed float32."""
        if raw.dtype != torch.float32 or not raw.is_cuda or raw.dim() != 3:
            raise ValueError('raw must be a [nchunk, Nc, T] float32 CUDA tensor')
        nchunk, Nc, T = (int(s) for s in raw.shape)
        if raw.stride(2) != 1 or raw.stride(0) < 0 or (Nc > 1 and raw.stride(1) < T):
            raw = raw.contiguous()
        sos, nsec, zp = None, 0, 0
        if filt is not None:
            sos = np.ascontiguousarray(bandpass_sos(filt[0], filt[1], sampling_rate, filt[2]))
            nsec, zp = len(sos), int(bool(filt[3]))
        if out is None:
            out = torch.empty((nchunk, Nc * T), dtype=torch.float32, device=raw.device)
        rc = self.lib.dtx_preprocess_planes(self.h, _ptr(raw), nchunk, Nc, T, int(raw.stride(0)) if nchunk > 1 else 0,
                                            int(raw.stride(1)) if Nc > 1 else T,
                                            sos.ctypes.data_as(C.c_void_p) if sos is not None else None,
                                            nsec, zp, int(bool(detrend)), _ptr(out), int(out.stride(0)))
        self._ck(rc, 'dtx_preprocess_planes')
        return out

    @_on_ctx_stream
    def ingest_counts(self, counts, frags, Nc, T, want_run=True):
        """dtx_ingest_counts: fragments of raw int32 counts -> zero-filled [Nc, T] float32
        planes (+ the longest run without a zero in any channel, construct.py:1033-1066).

        counts: int32 CUDA tensor, the fragments' samples concatenated; frags: [(channel,
        offset into counts, first sample in the chunk, length), ...].  Returns (planes, run)
        with run a 2-element int64 CUDA tensor (start, length) or None; asynchronous."""
        if counts.dtype != torch.int32 or not counts.is_cuda:
            raise ValueError('counts must be an int32 CUDA tensor')
        fr = np.ascontiguousarray(np.asarray(frags, dtype=np.int64).reshape(-1, 4))
        planes = torch.empty((int(Nc), int(T)), dtype=torch.float32, device=counts.device)
        run = torch.zeros((2,), dtype=torch.int64, device=counts.device) if want_run else None
        rc = self.lib.dtx_ingest_counts(self.h, _ptr(counts), fr.ctypes.data_as(C.c_void_p), len(fr),
                                        int(Nc), int(T), _ptr(planes), _ptr(run))
        self._ck(rc, 'dtx_ingest_counts')
        return planes, run

    @_on_ctx_stream
    def gapfree_run(self, planes, out=None):
        """dtx_gapfree_run on a [Nc, T] float32 CUDA view (columns of a larger plane buffer are
        fine: stride(1) == 1): (start, length) of the longest run without a zero in any channel,
        as a 2-element int64 CUDA tensor (asynchronous)."""
        if planes.dtype != torch.float32 or not planes.is_cuda or planes.dim() != 2 or planes.stride(1) != 1:
            raise ValueError('planes must be a [Nc, T] float32 CUDA tensor with unit sample stride')
        if out is None:
            out = torch.zeros((2,), dtype=torch.int64, device=planes.device)
        rc = self.lib.dtx_gapfree_run(self.h, _ptr(planes), int(planes.shape[0]), int(planes.shape[1]),
                                      int(planes.stride(0)) if planes.shape[0] > 1 else int(planes.shape[1]),
                                      _ptr(out))
        self._ck(rc, 'dtx_gapfree_run')
        return out

    @_on_ctx_stream
    def decimate(self, raw, factor, sampling_rate):
        """Trace.decimate(factor) on the device (construct.py:1016-1017): causal
        Chebyshev-II low-pass, then every factor-th sample.

        raw : [..., T] float32 CUDA tensor of series planes -> [..., ceil(T/factor)]"""
        if raw.dtype != torch.float32 or not raw.is_cuda:
            raise ValueError('raw must be a float32 CUDA tensor')
        factor = int(factor)
        raw = raw.contiguous()
        T = int(raw.shape[-1])
        nser = int(raw.numel() // max(T, 1))
        To = (T + factor - 1) // factor
        out = torch.empty(tuple(raw.shape[:-1]) + (To,), dtype=torch.float32, device=raw.device)
        sos = np.ascontiguousarray(decimate_sos(sampling_rate, factor), dtype=np.float64)
        rc = self.lib.dtx_decimate(self.h, _ptr(raw), nser, T, T, sos.ctypes.data_as(C.c_void_p),
                                   len(sos), factor, _ptr(out), To)
        self._ck(rc, 'dtx_decimate')
        return out

    # ------------------------------------------------------------ timing
    @_on_ctx_stream
    def timer_start(self):
        self._ck(self.lib.dtx_timer_start(self.h), 'dtx_timer_start')

    @_on_ctx_stream
    def timer_stop(self):
        ms = C.c_float()
        self._ck(self.lib.dtx_timer_stop(self.h, C.byref(ms)), 'dtx_timer_stop')
        return float(ms.value)

    def kernel_ms_mean(self, n):
        """Mean duration of the dominant kernel over the last n scan / cross-correlation calls (n <= 64)."""
        ms = C.c_float()
        self._ck(self.lib.dtx_kernel_ms_mean(self.h, int(n), C.byref(ms)), 'dtx_kernel_ms_mean')
        return float(ms.value)

    def kernel_ms_list(self, n):
        """Durations of the dominant kernel of the last n scan / cross-correlation calls, oldest first (n <= 64)."""
        ms = (C.c_float * int(n))()
        self._ck(self.lib.dtx_kernel_ms_list(self.h, int(n), ms), 'dtx_kernel_ms_list')
        return [float(v) for v in ms][::-1]

    def last_kernel_name(self):
        """The dominant kernel the last scan / cross-correlation call dispatched (dtx_last_kernel_name)."""
        return (self.lib.dtx_last_kernel_name(self.h) or b'').decode()

    def repair_counts(self):
        """What the dynamic-range guards of the last scan listed (dtx_repair_counts; waits for the stream):
        {'blocks': listed blocks, 'tiles': column tiles of 128 windows recomputed by the float32 contraction,
        'split': blocks scanned again with outlier samples split off}."""
        c = (C.c_int32 * 4)()
        self._ck(self.lib.dtx_repair_counts(self.h, c), 'dtx_repair_counts')
        return {'blocks': int(c[0]), 'tiles': int(c[1]), 'split': int(c[2])}

    def last_kernel_ms(self):
        ms = C.c_float()
        self._ck(self.lib.dtx_last_kernel_ms(self.h, C.byref(ms)), 'dtx_last_kernel_ms')
        return float(ms.value)
