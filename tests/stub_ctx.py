"""CPU stand-in for detex_amd.hip.HipContext -- TEST INFRASTRUCTURE.  It answers the calls
the host logic makes (createCluster, SubSpace.detex / scan_station) with the CPU oracle, so
that logic -- chunk batching, trigger picking on sparse candidates, trigger-time estimates,
result assembly, multi-process sharding under gloo -- runs in the `-m "not gpu"` suite.  The
product never imports this; on a GPU box the real context (the HIP library) is what runs."""
import numpy as np
import torch

from oracle import detex_oracle as orc


class OracleCtx(object):
    device = torch.device('cpu')
    engine = 'oracle'

    def __init__(self):
        self.calls = []

    # ---- similarity matrix
    def xcorr_allpairs(self, td, Nc, pair_begin=0, pair_end=None):
        td = td.numpy().astype(np.float64)
        N = len(td)
        npairs = N * (N - 1) // 2
        pair_end = npairs if pair_end is None else pair_end
        cc = np.zeros(npairs, np.float32)
        lag = np.zeros(npairs, np.int32)
        ss = np.zeros(npairs, np.float32)
        k = 0
        for b in range(N):
            for c in range(b + 1, N):
                if pair_begin <= k < pair_end:
                    r = orc.ccx2_fft(td[b], td[c], Nc)
                    cc[k], lag[k], ss[k] = r[0], r[1], r[2]
                k += 1
        self.calls.append(('xcorr', pair_begin, pair_end))
        return torch.from_numpy(cc), torch.from_numpy(lag), torch.from_numpy(ss)

    # ---- scan
    def upload_basis(self, U, det_off, Nc):
        self.U = np.asarray(U.cpu().numpy() if isinstance(U, torch.Tensor) else U, dtype=np.float64)
        self.off = [int(v) for v in det_off]
        self.Nc = int(Nc)
        self.n = self.U.shape[1]
        self.D = len(self.off) - 1

    def preprocess(self, raw, filt=None, sampling_rate=None, detrend=True, out=None):
        if filt is None and not detrend:
            res = torch.stack([torch.from_numpy(orc.multiplex(list(p.numpy())).astype(np.float32))
                               for p in raw])
        else:
            res = torch.stack([torch.from_numpy(orc.multiplex(orc.apply_filter(list(p.numpy()), filt, sampling_rate,
                                                                               detrend)).astype(np.float32))
                               for p in raw])
        if out is not None:                   # rows of a wider buffer (the device call writes with a row pitch)
            out[:, :res.shape[1]] = res
            return out
        return res

    def decimate(self, raw, factor, sampling_rate):
        flat = raw.reshape(-1, raw.shape[-1]).numpy()
        out = np.stack(orc.decimate(list(flat), int(factor), sampling_rate)).astype(np.float32)
        return torch.from_numpy(out.reshape(tuple(raw.shape[:-1]) + (out.shape[-1],)))

    def ingest_counts(self, counts, frags, Nc, T, want_run=True):
        cnt = counts.numpy()
        planes = orc.merge_fill([(c, t0, cnt[off:off + m]) for (c, off, t0, m) in frags], Nc, T)
        run = torch.tensor(orc.merge_channels_run(planes), dtype=torch.int64) if want_run else None
        return torch.from_numpy(planes.astype(np.float32)), run

    def gapfree_run(self, planes, out=None):
        self.calls.append(('gapfree', int(planes.shape[1])))
        r = torch.tensor(orc.merge_channels_run(planes.numpy().astype(np.float64)), dtype=torch.int64)
        if out is not None:
            out.copy_(r)
            return out
        return r

    def num_windows(self, M):
        return (int(M) - self.n) // self.Nc + 1

    def _chunks(self, x, M, chunk_stride, nchunk):
        if M is None:
            x = x if x.dim() == 2 else x[None]
            return [x[b].numpy() for b in range(x.shape[0])], int(x.shape[1])
        return [x[b * chunk_stride: b * chunk_stride + M].numpy() for b in range(nchunk)], int(M)

    def chunk_table(self, starts, lengths):
        tab = np.stack([np.asarray(starts, np.int64), np.asarray(lengths, np.int64)], axis=1)
        return torch.from_numpy(tab), tab

    def _post(self, ds, thr, hist, cand_cap, Pb=None):
        nb, D, P = ds.shape
        Pb = np.full(nb, P) if Pb is None else np.asarray(Pb)
        own = np.arange(P)[None, None, :] < Pb[:, None, None]          # a ragged batch: the chunk's own windows
        res = dict(hist=hist, maxds=torch.from_numpy(np.where(own, ds, 0).max(axis=2).astype(np.float32)),
                   cand=None, ncand=0)
        if hist is not None:
            for b in range(nb):
                for d in range(D):
                    hist[d] += torch.from_numpy(orc.ds_histogram(ds[b, d, :Pb[b]]).astype(np.int32))
        if thr is not None:
            thr = np.broadcast_to(np.asarray(thr.numpy() if isinstance(thr, torch.Tensor) else thr, np.float32), (D,))
            b, d, p = np.nonzero((ds >= thr[None, :, None]) & own)
            res['ncand'] = len(b)
            m = min(len(b), cand_cap)
            res['cand'] = (b[:m].astype(np.int64), d[:m].astype(np.int64), p[:m].astype(np.int64),
                           ds[b[:m], d[:m], p[:m]])
        return res

    def subspace_scan(self, x, M=None, chunk_stride=None, nchunk=None, out=None, thr=None, hist=None,
                      want_max=True, cand_cap=1 << 20, lazy=False, chunk_tab=None):
        Pb = None
        if chunk_tab is not None:
            # a ragged batch: rows as long as the longest chunk's, 0.999 where the device leaves them unspecified
            flat = x.reshape(-1).numpy()
            chunks = [flat[a:a + m] for (a, m) in chunk_tab[1]]
            M = int(chunk_tab[1][:, 1].max())
            Pb = [max(0, self.num_windows(len(c))) if len(c) > self.n else 0 for c in chunks]
        else:
            chunks, M = self._chunks(x, M, chunk_stride, nchunk)
        P = self.num_windows(M)
        ds = np.full((len(chunks), self.D, P), 0.999, dtype=np.float32)
        for b, c in enumerate(chunks):
            if len(c) <= self.n:
                continue
            for d in range(self.D):
                v = orc.scrub_ds(orc.mpxds_fft(c.astype(np.float64), self.U[self.off[d]:self.off[d + 1]],
                                               self.Nc))[0]
                ds[b, d, :len(v)] = v
        self.calls.append(('scan', len(chunks), M))
        if hist is None:
            hist = torch.zeros((self.D, 400), dtype=torch.int32)
        res = self._post(ds, thr, hist, cand_cap, Pb)
        return torch.from_numpy(ds), ((lambda: res) if lazy else res)

    def ds_post(self, ds, thr=None, hist=None, want_max=True, cand_cap=1 << 20):
        if hist is None:
            hist = torch.zeros((ds.shape[1], 400), dtype=torch.int32)
        return self._post(ds.numpy(), thr, hist, cand_cap)

    def trigger_stats(self, x, M, chunk_stride, ds, trig, sta_len, lta_len, ewf, eoff, emax, chunk_tab=None):
        trig = np.asarray(trig).reshape(-1, 5)
        out = np.full((len(trig), 6 + emax), np.nan)
        x = x.reshape(-1) if x.dim() == 2 and x.is_contiguous() else x
        eoff = eoff.numpy()
        for r, (b, d, p, slo, llo) in enumerate(trig):
            if chunk_tab is not None:
                a, m = (int(v) for v in chunk_tab[1][b])
                xc = x.reshape(-1)[a:a + m].numpy()
            else:
                xc = (x[b] if x.dim() == 2 else x[b * chunk_stride: b * chunk_stride + M]).numpy()
            ev = ewf[eoff[d]:eoff[d + 1]].numpy() if ewf is not None else np.zeros((0, self.U.shape[1]))
            row = orc.trigger_stats_row(int(p), self.Nc, xc, self.U[self.off[d]:self.off[d + 1]],
                                        ev, ds=ds[b, d].numpy().astype(np.float64),
                                        sta_lo=int(slo), sta_len=sta_len, lta_lo=int(llo), lta_len=lta_len)
            if ewf is None:                       # STA / LTA only, like the device call
                row = row[:2]
            out[r, :len(row)] = row
        return torch.from_numpy(out)

    def normcorr(self, t, s):
        return torch.from_numpy(orc.fast_normcorr(t.numpy().astype(np.float64),
                                                  s.numpy().astype(np.float64)).astype(np.float32))

    def sync(self):
        pass

    def close(self):
        pass
