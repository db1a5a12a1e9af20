import os,sys,time,numpy as np,torch
sys.path.insert(0,os.getcwd())
from detex_amd.hip import HipContext
from detex_amd import synth
Nc,L,D,k,chunks=3,3000,25,8,24
n,M=Nc*L,372000*Nc
U,off=synth.detector_bank(1,D,k,n)
x,_=synth.continuous_day(1,chunks,M,Nc,U,off,nplant=50)
thr=np.full(D,0.3,np.float32)
streams=[torch.cuda.Stream(), torch.cuda.Stream()]
ctxs=[];xts=[];dss=[]
for s in streams:
    with torch.cuda.stream(s):
        c=HipContext(0); c.upload_basis(torch.from_numpy(U).cuda(),off,Nc); ctxs.append(c)
        xts.append(torch.from_numpy(x).cuda())
        P=c.num_windows(M)
        dss.append(torch.empty((chunks,D,(P+63)//64*64),dtype=torch.float32,device='cuda')[:,:,:P])
torch.cuda.synchronize()
def run(nstep, nctx):
    fetch=[]
    torch.cuda.synchronize(); t0=time.perf_counter()
    for i in range(nstep):
        j=i%nctx
        with torch.cuda.stream(streams[j]):
            _,f=ctxs[j].subspace_scan(xts[j],out=dss[j],thr=thr,lazy=True)
        fetch.append(f)
    torch.cuda.synchronize(); dt=time.perf_counter()-t0
    res=[f() for f in fetch]
    return 1e3*dt/nstep, res[-1]['ncand']
for nctx in (1,2,1,2):
    run(4,nctx)
    print('contexts/streams %d: ms/step %.3f  ncand %d' % ((nctx,)+run(12,nctx)))
