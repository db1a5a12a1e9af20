"""Accuracy of the scan on a record that rides a LEVEL far above its scatter behind a step in mid-record (no chunk
offset removes it; the reference's pipeline detrends and band-passes before the scan, so such data do not reach it):
max |DS - oracle| before / across / behind the step, per engine.   python scripts/probe/level_step.py"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from detex_amd.hip import HipContext
from detex_amd import synth
from oracle import detex_oracle as orc
rs = np.random.RandomState(3)
Nc, L, T = 3, 400, 30000
n = Nc * L
U = np.concatenate([synth.orthonormal_basis(rs, n, k) for k in (8, 4)]).astype(np.float32)
off = np.array([0, 8, 12], np.int32)
ctx = HipContext(0)
for lvl in (0.0, 200.0, 1e3, 5e3, 3e4):
    x = rs.standard_normal((T, Nc))
    x[14000:] += lvl
    x = x.reshape(-1).astype(np.float32)
    ref = [orc.mpxds_direct(x.astype(np.float64), U[off[d]:off[d + 1]].astype(np.float64), Nc) for d in range(2)]
    for eng in ('auto', 'fp32'):
        ctx.set_engine(eng)
        ctx.upload_basis(U, off, Nc)
        ds = ctx.subspace_ds(torch.from_numpy(x).cuda())[0].cpu().numpy()
        cnt = ctx.repair_counts()
        err = np.max(np.abs(ds - np.stack(ref)), axis=0)
        print('level %7.0f sigma  %-5s  before the step %.1e  across it %.1e  behind it %.1e   (listed %d blocks, %d tiles)'
              % (lvl, eng, err[:14000 - L].max(), err[14000 - L:14001].max(), err[14001:].max(), cnt['blocks'], cnt['tiles']))
