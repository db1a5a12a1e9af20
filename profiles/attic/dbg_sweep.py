import os, sys
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from oracle import detex_oracle as orc
from detex_amd.hip import HipContext
ctx = HipContext(0)
for seed in [int(a) for a in sys.argv[1:]]:
    rs = np.random.RandomState(1000 + seed)
    Nc = int(rs.choice([1, 2, 3, 4, 6])); L = int(rs.randint(4, 260)); n = Nc * L
    nd = int(rs.randint(1, 7))
    ks = [int(rs.choice([1, 1, 2, 3, 5, 8, 13, 16, 17, 40, 100])) for _ in range(nd)]
    if rs.rand() < 0.2: ks = [1] * int(rs.randint(200, 450))
    off = np.concatenate([[0], np.cumsum(ks)]).astype(np.int32)
    U = rs.standard_normal((off[-1], n)); U /= np.linalg.norm(U, axis=1, keepdims=True)
    if rs.rand() < 0.5: U -= U.mean(axis=1, keepdims=True)
    nchunk = int(rs.randint(1, 4)); T = int(rs.randint(L + 1, L + 2500)); M = Nc * T - int(rs.randint(0, Nc))
    stride = int(rs.randint(Nc, M + 1) // Nc * Nc)
    amp = float(10 ** rs.uniform(-6, 3)); offs = float(rs.choice([0, 0, 5]))
    buf = rs.standard_normal((nchunk - 1) * stride + M) * amp + offs
    gap = glitch = None
    if rs.rand() < 0.35 and len(buf) > 6 * n:
        g0 = int(rs.randint(0, len(buf) - 3 * n)); gl = int(rs.randint(n, 3 * n)); buf[g0:g0 + gl] = offs; gap = (g0, gl)
    if rs.rand() < 0.35:
        gi = int(rs.randint(len(buf))); buf[gi] += 1e4 * amp; glitch = gi
    eng = 'bf16x3' if seed % 2 else 'fp32'
    print('seed', seed, 'Nc', Nc, 'L', L, 'ks', ks[:8], 'nchunk', nchunk, 'M', M, 'stride', stride, 'amp', amp, 'offs', offs, 'gap', gap, 'glitch', glitch, eng)
    ctx.set_engine(eng); ctx.upload_basis(U.astype(np.float32), off, Nc)
    ds = ctx.subspace_ds(torch.from_numpy(buf.astype(np.float32)).cuda(), M=M, chunk_stride=stride, nchunk=nchunk).cpu().numpy()
    Uf = U.astype(np.float32).astype(np.float64); bf = buf.astype(np.float32).astype(np.float64)
    for b in range(nchunk):
        xb = bf[b * stride: b * stride + M]
        for d in range(min(len(ks), 3)):
            ref = orc.mpxds_direct(xb, Uf[off[d]:off[d + 1]], Nc)
            win = np.lib.stride_tricks.sliding_window_view(xb, n)[::Nc]
            live = win.std(axis=1) > 0
            err = np.abs(ds[b, d] - ref); err[~live] = 0
            j = int(np.argmax(err))
            print('  b', b, 'd', d, 'maxerr %.3e at p=%d ref=%.6g got=%.6g winstd=%.3e winmean=%.3e' % (err[j], j, ref[j], ds[b, d, j], win[j].std(), win[j].mean()),
                  ' dead nonzero:', int(np.sum(ds[b, d][~live] != 0)), 'of', int((~live).sum()))
