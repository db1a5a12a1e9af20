#!/usr/bin/env python
"""Stage-1 check for a frequency-domain (overlap-save) scan engine: emulate it in float32
NumPy/SciPy (pocketfft single precision) and compare DS with the float64 closed form at
config-2 sizes (n = 9000, K = 8 per detector), on noise + planted events, with and without
a 1e4-sigma spike in the block."""
import sys, os, time
import numpy as np, scipy.fft
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import detex_oracle as orc
from detex_amd import synth

Nc, L, k, D = 3, 3000, 8, 2
n = Nc * L
T = 60000                      # samples per channel
M = T * Nc
N = 8192
V = N - 3072                   # valid outputs per block
U, det_off = synth.detector_bank(1, D, k, n)
x, plants = synth.continuous_day(3, 1, M, Nc, U, det_off, nplant=6)
x = x[0]
for spike in (0.0, 1e4, 1e6):
    xs = x.copy()
    if spike:
        xs[3 * 20000 + 1] += spike
    ref = np.stack([orc.mpxds_direct(xs.astype(np.float64), U[det_off[d]:det_off[d + 1]].astype(np.float64), Nc)
                    for d in range(D)])
    P = ref.shape[1]
    # float32 overlap-save
    xc = xs.reshape(-1, Nc).T.astype(np.float32)                  # planes [Nc][T]
    m0 = np.float32(xs.mean())
    xc = xc - m0
    Uc = U.reshape(U.shape[0], L, Nc).transpose(0, 2, 1)           # [K][Nc][L]
    Uf = np.conj(scipy.fft.rfft(Uc.astype(np.float32), n=N, axis=-1))   # complex64
    assert Uf.dtype == np.complex64
    S = U.astype(np.float64).sum(axis=1)
    # window stats float64 (as win_stats_kernel)
    w = np.lib.stride_tricks.sliding_window_view(xs.astype(np.float64) - float(m0), n)[::Nc]
    mu = w.mean(axis=1); den = n / (n - 1.0) * ((w - mu[:, None]) ** 2).sum(axis=1)
    ds = np.zeros((D, P), np.float32)
    nb = (P + V - 1) // V
    for b in range(nb):
        seg = np.zeros((Nc, N), np.float32)
        hi = min(T, b * V + N)
        seg[:, :hi - b * V] = xc[:, b * V:hi]
        Xf = scipy.fft.rfft(seg, axis=-1)                          # complex64
        Z = (Uf * Xf[None]).sum(axis=1).astype(np.complex64)       # [K][N/2+1]
        y = scipy.fft.irfft(Z, n=N, axis=-1)[:, :V].astype(np.float32)
        p0, p1 = b * V, min(P, b * V + V)
        yy = (y[:, :p1 - p0] - (S[:, None] * mu[None, p0:p1]).astype(np.float32)) ** 2
        for d in range(D):
            ds[d, p0:p1] = yy[det_off[d]:det_off[d + 1]].sum(axis=0) / den[p0:p1].astype(np.float32)
    err = np.abs(ds - ref)
    big = ref > 0.5
    print('spike %g: max|dDS| %.3g  (at DS>0.5: %.3g, max DS %.4f)  median %.2g  blocks with err>1e-5: %s'
          % (spike, err.max(), err[big].max() if big.any() else 0, ref.max(), np.median(err),
             sorted(set((np.argwhere(err > 1e-5)[:, 1] // V).tolist()))))
