import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from detex_amd.hip import HipContext
from detex_amd import synth
Nc, L, D, k, T = 3, 3000, 25, 8, 372000
n = Nc * L
U, off = synth.detector_bank(5, D, k, n)
x, plants = synth.continuous_day(9, 1, Nc * T, Nc, U, off, nplant=40)
xt = torch.from_numpy(x).cuda()
rs = np.random.RandomState(1)
P = T - L + 1
pts = [(b, p, d) for (b, p, d) in plants] + [(0, int(rs.randint(P)), int(rs.randint(D))) for _ in range(40)]
refs = []
for (b, p, d) in pts:
    w = x[b, p * Nc: p * Nc + n].astype(np.float64); mu = w.mean()
    Ud = U[off[d]:off[d + 1]].astype(np.float64)
    refs.append(np.sum((Ud @ w - Ud.sum(axis=1) * mu) ** 2) / (n / (n - 1.0) * np.sum((w - mu) ** 2)))
for eng in ['fp32', 'bf16x3']:
    ctx = HipContext(0); ctx.set_engine(eng); ctx.upload_basis(U, off, Nc)
    ds = ctx.subspace_ds(xt).cpu().numpy()
    err = [abs(ds[b, d, p] - r) for (b, p, d), r in zip(pts, refs)]
    print(eng, 'max err at planted peaks %.2e' % max(err[:len(plants)]), ' noise windows %.2e' % max(err[len(plants):]))
