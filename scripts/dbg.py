import sys, numpy as np, torch
sys.path.insert(0, '.')
from detex_amd.hip import HipContext
from detex_amd import synth
Nc, L = 3, 3000; n = Nc*L; M = 372000*Nc; D, k = 25, 8
U, off = synth.detector_bank(1, D, k, n)
ctx = HipContext(0); ctx.upload_basis(U, off, Nc)
x, plants = synth.continuous_day(1, 2, M, Nc, U, off, nplant=6)
print(plants)
ds = ctx.subspace_ds(torch.from_numpy(x).cuda())
res = ctx.ds_post(ds, thr=np.full(D, 0.3, np.float32))
ctx.sync()
print('ncand', res['ncand'], 'max', float(ds.max()))
for (b, p, d) in plants:
    print(b, p, d, float(ds[b, d, p]), float(ds[b, d].max()), int(ds[b, d].argmax()))
print(res['cand'][0][:10], res['cand'][1][:10], res['cand'][2][:10])
print('maxds', res['maxds'].cpu().numpy().max(axis=1)[:5], 'hist sum', int(res['hist'].sum()), 2*25*ds.shape[2])
for Pcut in [4000, 5000, 70000, 200000]:
    sub = ds[:, :, :Pcut].contiguous()
    r = ctx.ds_post(sub, thr=np.full(D, 0.3, np.float32))
    print(Pcut, r['ncand'], int((sub >= 0.3).sum()), int(r['hist'].sum()), float(r['maxds'].max()))
r = ctx.ds_post(ds[:, :3].contiguous(), thr=np.full(3, 0.3, np.float32)); print('D3', r['ncand'])
r = ctx.ds_post(ds[:1].contiguous(), thr=np.full(25, 0.3, np.float32)); print('chunk1', r['ncand'], int((ds[:1] >= 0.3).sum()))
