import os,sys,numpy as np,torch
sys.path.insert(0,os.getcwd())
from detex_amd.hip import HipContext
from detex_amd import synth
Nc,L,D,k,chunks=3,3000,25,8,24
n,M=Nc*L,372000*Nc
ctx=HipContext(0)
U,off=synth.detector_bank(1,D,k,n)
ctx.upload_basis(torch.from_numpy(U).cuda(),off,Nc)
x,_=synth.continuous_day(1,chunks,M,Nc,U,off,nplant=50)
xt=torch.from_numpy(x).cuda()
P=ctx.num_windows(M)
ds=torch.empty((chunks,D,(P+63)//64*64),dtype=torch.float32,device='cuda')[:,:,:P]
thr=np.full(D,0.3,np.float32)
import time
def t(f,nrep=8):
    for _ in range(2): f()
    torch.cuda.synchronize(); t0=time.perf_counter()
    ks=[]
    for _ in range(nrep):
        f(); ks.append(ctx.last_kernel_ms())
    torch.cuda.synchronize(); return 1e3*(time.perf_counter()-t0)/nrep, float(np.mean(ks))
print('two calls  ms/step %.3f kernel %.3f' % t(lambda: (ctx.subspace_ds(xt,out=ds), ctx.ds_post(ds,thr=thr))))
print('fused      ms/step %.3f kernel %.3f' % t(lambda: ctx.subspace_scan(xt,out=ds,thr=thr)))
print('ds only    ms/step %.3f kernel %.3f' % t(lambda: ctx.subspace_ds(xt,out=ds)))
