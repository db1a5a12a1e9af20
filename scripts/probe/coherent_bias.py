"""Error of the detection statistic AT a dominant event (300 sigma sqrt(n) times a basis vector: every product of the
window's dot product has the same sign) for the engines, over template shapes: python scripts/probe/coherent_bias.py"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from oracle import detex_oracle as orc
from detex_amd.hip import HipContext
ctx = HipContext(0)
for (Nc, L) in ((3, 1000), (3, 3000), (3, 6000), (12, 1500), (6, 3000), (1, 18000), (2, 9000), (3, 13000), (12, 750), (9, 2000), (4, 4500)):
    rs = np.random.RandomState(Nc * 100000 + L)
    n = Nc * L
    k = 7
    U = rs.standard_normal((k, n)); U /= np.linalg.norm(U, axis=1, keepdims=True)
    off = np.array([0, k], np.int32)
    T = L + 9000
    M = Nc * T
    buf = rs.standard_normal(M) * 140.0 + 7.0
    p_ev = 4000
    buf[p_ev * Nc: p_ev * Nc + n] += 300.0 * np.std(buf) * np.sqrt(n) * U[k - 1]
    Uf = U.astype(np.float32).astype(np.float64); bf = buf.astype(np.float32).astype(np.float64)
    ref = orc.mpxds_fft(bf, Uf, Nc)
    line = 'Nc %2d L %5d n %5d ref %.7f:' % (Nc, L, n, ref[p_ev])
    for eng in ('fp32', 'bf16x3', 'fft', 'auto'):
        ctx.set_engine(eng)
        ctx.upload_basis(U.astype(np.float32), off, Nc)
        ds = ctx.subspace_ds(torch.from_numpy(buf.astype(np.float32)).cuda(), M=M, chunk_stride=M, nchunk=1)
        ctx.sync(); ds = ds.cpu().numpy()[0, 0]
        err = ds - ref
        line += '  %s %+.2e (max |err| elsewhere %.1e)' % (eng, err[p_ev], np.abs(np.delete(err, p_ev)).max())
    print(line)
