import numpy as np
N=8192
def g(a): return ((a>>5)&7) ^ (((a>>6)&3)<<3)
def swz(a): return a ^ g(a)      # in-region (9-bit) slot swizzle
aa=np.arange(512); assert sorted(swz(aa).tolist())==list(range(512))
def dft(u,sg):
    R=len(u); F=np.exp(sg*2j*np.pi*np.outer(np.arange(R),np.arange(R))/R)
    return list(F@np.stack(u))
def conflicts(slots):
    w=1
    for hw in range(len(slots)//32):
        b=slots[hw*32:(hw+1)*32]%32
        w=max(w,np.bincount(b,minlength=32).max())
    return w
def fftX(Z, sg):
    lds=np.zeros(8192,complex)
    l=np.arange(64); worst=1
    out=np.zeros(N,complex)
    for w in range(16):
        base=w*512
        v=[Z[16*(l+64*r)+w] for r in range(8)]
        v=dft(v,sg)                                  # pass 1
        for s in range(8):
            a=8*l+s; worst=max(worst,conflicts(swz(a))); lds[base+swz(a)]=v[s]
        for r in range(8): worst=max(worst,conflicts(swz(l+64*r)))
        v=[lds[base+swz(l+64*r)] for r in range(8)]
        k=l&7; j=(l-k)*8+k                           # pass 2: p=8
        w1=np.exp(sg*2j*np.pi*k/64)
        v=dft([v[s]*w1**s for s in range(8)],sg)
        for s in range(8):
            a=j+8*s; worst=max(worst,conflicts(swz(a))); lds[base+swz(a)]=v[s]
        v=[lds[base+swz(l+64*r)] for r in range(8)]
        w1=np.exp(sg*2j*np.pi*l/512)                 # pass 3: p=64, k=l, j=l
        v=dft([v[s]*w1**s for s in range(8)],sg)     # reg r <-> k2 = l+64 r
        for r in range(8): lds[base+swz(l+64*r)]=v[r]
    # phase B: thread tid (1024): k2 = tid & 511, hb = tid >> 9: outputs k1 = 2q+hb
    tid=np.arange(1024); k2=tid&511; hb=tid>>9
    for q8 in range(16):
        worst=max(worst,conflicts(swz(k2[q8*64:(q8+1)*64])))
    u=[lds[ww*512+swz(k2)]*np.exp(sg*2j*np.pi*ww*k2/N) for ww in range(16)]
    # radix-2 fold then DFT8
    v=[]
    for a in range(8):
        va=np.where(hb==0, u[a]+u[a+8], (u[a]-u[a+8])*np.exp(sg*2j*np.pi*a/16))
        v.append(va)
    X=dft(v,sg)
    for q in range(8):
        out[k2+512*(2*q+hb)]=X[q]
    return out,worst
rs=np.random.RandomState(0)
Z=rs.standard_normal(N)+1j*rs.standard_normal(N)
y,worst=fftX(Z,+1); print(np.abs(y-np.fft.ifft(Z)*N).max(),'worst',worst)
y,worst=fftX(Z,-1); print(np.abs(y-np.fft.fft(Z)).max())
# address formulas
l=np.arange(64); ok=True
L0=l^(l>>5)
for r in range(8):
    Gr=(2*(r&3)) ^ (8*(r&3))
    ok&=np.array_equal(swz(l+64*r), 64*r+(L0^Gr))
# S1: a = 8l+s : a>>5 = l>>2 ; a>>6 = l>>3
B1=(8*l) ^ (((l>>2)&7) ^ (((l>>3)&3)<<3))
for s in range(8): ok&=np.array_equal(swz(8*l+s), B1^s)
# S2: a = 64(l>>3) + (l&7) + 8s
k=l&7
for s in range(8):
    a=64*(l>>3)+k+8*s
    lane=(64*(l>>3)+k) ^ ((2*(l>>3))&7) ^ ((((l>>3))&3)<<3)
    const=(8*s) ^ (s>>2)
    ok&=np.array_equal(swz(a), lane^const)
print('formulas',ok)
