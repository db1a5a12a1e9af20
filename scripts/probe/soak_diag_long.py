"""As soak_diag.py for test_random_scan_long: python scripts/probe/soak_diag_long.py BASE SEED"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from oracle import detex_oracle as orc
from detex_amd.hip import HipContext
base, seed = int(sys.argv[1]), int(sys.argv[2])
rs = np.random.RandomState(3000 + seed + 100000 * base)
Nc = int([3, 1, 2, 3, 6, 3, 4, 1, 3, 2, 3, 3, 6, 3, 1, 3, 2, 5, 7, 8, 9, 12, 5, 9][seed])
L = int([256, 300, 700, 1100, 1500, 2100, 2600, 3000, 3700, 4200, 5000, 5633, 5634, 7681, 9000, 13000, 16384,
         256, 900, 2100, 3000, 1500, 5634, 9000][seed])
n = Nc * L
ks = [[8, 3], [1, 1, 1, 2], [5], [16, 1], [2, 2], [7], [1], [8] * 3, [4, 1, 1], [3], [9], [2, 5], [4], [2],
      [3, 1, 1], [4], [2], [8, 3], [1, 1, 2], [5], [2, 2], [7], [3], [1, 4]][seed]
off = np.concatenate([[0], np.cumsum(ks)]).astype(np.int32)
U = rs.standard_normal((off[-1], n)); U -= U.mean(axis=1, keepdims=True) * (seed % 2); U /= np.linalg.norm(U, axis=1, keepdims=True)
V = (8192 - L + 1) // 512 * 512 if L <= 5633 else 4096
nchunk = 1 + seed % 2
T = L + int(rs.randint(int(1.2 * V), int(2.7 * V))); M = Nc * T - int(rs.randint(0, Nc))
stride = int(rs.randint(M // 2, M + 1) // Nc * Nc)
amp = float(10 ** rs.uniform(-3, 3)); dc = float(rs.choice([0, 7]))
buf = rs.standard_normal((nchunk - 1) * stride + M) * amp + dc
p_ev = int(rs.randint(0, (M - n) // Nc))
gain = 300.0 if seed % 3 == 0 else 6.0
buf[p_ev * Nc: p_ev * Nc + n] += gain * np.std(buf) * np.sqrt(n) * U[off[1] - 1]
print('Nc %d L %d ks %s V %d nchunk %d M %d stride %d amp %.3g dc %g event at window %d gain %g' % (Nc, L, ks, V, nchunk, M, stride, amp, dc, p_ev, gain))
ctx = HipContext(0)
for eng in ('auto', 'fp32'):
    ctx.set_engine(eng)
    ctx.upload_basis(U.astype(np.float32), off, Nc)
    ds = ctx.subspace_ds(torch.from_numpy(buf.astype(np.float32)).cuda(), M=M, chunk_stride=stride, nchunk=nchunk)
    ctx.sync(); ds = ds.cpu().numpy()
    Uf = U.astype(np.float32).astype(np.float64); bf = buf.astype(np.float32).astype(np.float64)
    for b in range(nchunk):
        xb = bf[b * stride: b * stride + M]
        for d in range(len(ks)):
            ref = orc.mpxds_fft(xb, Uf[off[d]:off[d + 1]], Nc)
            err = np.abs(ds[b, d] - ref); w = int(np.argmax(err))
            print('%s chunk %d det %d (k=%d): max err %.3g at window %d (got %.8g ref %.8g), over 1e-5: %d; err 99.9 pct %.3g; ref max %.4g' %
                  (eng, b, d, ks[d], err.max(), w, ds[b, d][w], ref[w], int((err > 1e-5).sum()), np.percentile(err, 99.9), ref.max()))
