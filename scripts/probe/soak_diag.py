"""Re-run one case of tests/test_gpu_random_sweep.py::test_random_scan under a soak base and print where it differs.
usage: python scripts/probe/soak_diag.py BASE SEED ENGINE(mfma|fft|auto)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from oracle import detex_oracle as orc
from detex_amd.hip import HipContext
base, seed, engine = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
NC_EXTRA = [5, 7, 8, 9, 12]
rs = np.random.RandomState(1000 + seed + 100000 * base)
Nc = int(rs.choice([1, 2, 3, 4, 6]))
if seed >= 24:
    Nc = NC_EXTRA[(seed - 24) % 5]
L = int(rs.randint(4, 260)); n = Nc * L
nd = int(rs.randint(1, 7))
ks = [int(rs.choice([1, 1, 2, 3, 5, 8, 13, 16, 17, 40, 100])) for _ in range(nd)]
if rs.rand() < 0.2:
    ks = [1] * int(rs.randint(200, 450))
off = np.concatenate([[0], np.cumsum(ks)]).astype(np.int32)
U = rs.standard_normal((off[-1], n)); U /= np.linalg.norm(U, axis=1, keepdims=True)
demean = rs.rand() < 0.5
if demean:
    U -= U.mean(axis=1, keepdims=True)
nchunk = int(rs.randint(1, 4)); T = int(rs.randint(L + 1, L + 2500)); M = Nc * T - int(rs.randint(0, Nc))
stride = int(rs.randint(Nc, M + 1) // Nc * Nc)
amp = float(10 ** rs.uniform(-6, 3)); dc = float(rs.choice([0, 0, 5]))
buf = rs.standard_normal((nchunk - 1) * stride + M) * amp + dc
gap = None
if rs.rand() < 0.35 and len(buf) > 6 * n:
    g0 = int(rs.randint(0, len(buf) - 3 * n)); g1 = g0 + int(rs.randint(n, 3 * n)); buf[g0:g1] = dc; gap = (g0, g1)
spike = -1
if rs.rand() < 0.35:
    spike = int(rs.randint(len(buf))); buf[spike] += 1e4 * amp
split = engine == 'mfma' and seed % 2 == 1
eng = {'fft': 'fft', 'auto': 'auto'}.get(engine, 'bf16x3' if split else 'fp32')
print('Nc %d L %d n %d ks %s demean %s nchunk %d M %d stride %d amp %.3g dc %g gap %s spike %d engine %s' %
      (Nc, L, n, ks if len(ks) < 12 else '%d singles' % len(ks), demean, nchunk, M, stride, amp, dc, gap, spike, eng))
ctx = HipContext(0)
ctx.set_engine(eng)
ctx.upload_basis(U.astype(np.float32), off, Nc)
ds = ctx.subspace_ds(torch.from_numpy(buf.astype(np.float32)).cuda(), M=M, chunk_stride=stride, nchunk=nchunk)
ctx.sync(); ds = ds.cpu().numpy()
Uf = U.astype(np.float32).astype(np.float64); bf = buf.astype(np.float32).astype(np.float64)
P = (M - n) // Nc + 1
worst = 0.0
for b in range(nchunk):
    xb = bf[b * stride: b * stride + M]
    win = np.lib.stride_tricks.sliding_window_view(xb, n)[::Nc]
    live = win.std(axis=1) > 0
    for d in range(len(ks)):
        ref = orc.mpxds_direct(xb, Uf[off[d]:off[d + 1]], Nc)
        scale = max(1.0, float(np.max(ref[live])))
        err = np.abs(ds[b, d] - ref); err[~live] = 0
        worst = max(worst, float(err.max() / scale))
        dead_bad = np.flatnonzero(ds[b, d][~live] != 0)
        if err.max() > 1e-5 * scale or len(dead_bad):
            w = int(np.argmax(err))
            print('chunk %d det %d (k=%d): max err %.3g at window %d of %d (got %.8g ref %.8g) scale %.3g; windows over tol %d; '
                  'dead windows non-zero %d; window std %.3g mean %.6g; sample range of window [%d, %d)' %
                  (b, d, ks[d], err.max(), w, P, ds[b, d][w], ref[w], scale, int((err > 1e-5 * scale).sum()),
                   len(dead_bad), win[w].std(), win[w].mean(), b * stride + w * Nc, b * stride + w * Nc + n))
            bad = np.flatnonzero(err > 1e-5 * scale)
            print('   bad windows: first %d last %d' % (bad[0], bad[-1]) if len(bad) else '   none over tol')
print('worst err / scale over all chunks and detectors: %.3g' % worst)
