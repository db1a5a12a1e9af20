#!/usr/bin/env python
"""Executable specification of the wave-decimated 8192-point transform of dtx_fft.hip: thread-level
index maps, the XOR swizzle of the LDS image (a bijection; every scatter / gather conflict-free for
ds_*_b64 half-waves on 64 banks) and the per-lane-base + constant address formulas the kernel uses."""
import numpy as np
N=8192
def g(a): return ((a>>5)&15) ^ (((a>>8)&1)<<4)
def swz(a): return a ^ g(a)      # in-region (10-bit) slot swizzle
aa=np.arange(1024); assert sorted(swz(aa).tolist())==list(range(1024))
def dft(u,sg):  # natural order DFT along axis 0
    R=len(u); F=np.exp(sg*2j*np.pi*np.outer(np.arange(R),np.arange(R))/R)
    return list(F@np.stack(u))
def conflicts(slots):  # slots [64] per wave instr (b64): two half-waves
    w=1
    for hw in range(2):
        b=slots[hw*32:(hw+1)*32]%32
        w=max(w,np.bincount(b,minlength=32).max())
    return w
def fftW(Z, sg):
    """Z natural order [8192]; returns natural order transform with kernel e^{sg 2 pi i nk/N}"""
    lds=np.zeros(8192,complex)
    l=np.arange(64)
    worst=1
    out=np.zeros(N,complex)
    # phase A per wave
    for w in range(8):
        base=w*1024
        v=[Z[8*(l+64*r)+w] for r in range(16)]
        # pass 1
        v=dft(v,sg)
        for s in range(16):
            a=16*l+s; worst=max(worst,conflicts(swz(a))); lds[base+swz(a)]=v[s]
        for r in range(16):
            a=l+64*r; worst=max(worst,conflicts(swz(a)))
        v=[lds[base+swz(l+64*r)] for r in range(16)]
        # pass 2: p=16
        k=l&15; j=(l-k)*16+k
        w1=np.exp(sg*2j*np.pi*k/256)
        v=dft([v[s]*w1**s for s in range(16)],sg)
        for s in range(16):
            a=j+16*s; worst=max(worst,conflicts(swz(a))); lds[base+swz(a)]=v[s]
        v=[lds[base+swz(l+64*r)] for r in range(16)]
        # pass 3: radix 4, p=256, 4 butterflies b
        o=[None]*16
        for b in range(4):
            i=l+64*b
            w1=np.exp(sg*2j*np.pi*i/1024)
            u=dft([v[b+4*s]*w1**s for s in range(4)],sg)
            for s in range(4): o[b+4*s]=u[s]
        v=o   # reg r <-> k2 = l+64 r
        for r in range(16):
            a=l+64*r; lds[base+swz(a)]=v[r]
    # barrier; phase B
    tid=np.arange(512)
    for h in range(2):
        k2=tid+512*h
        for q in range(8):   # conflict check over waves of 64 threads
            for ww in range(8):
                worst=max(worst,conflicts(swz(k2[q*64:(q+1)*64])))
        u=[lds[ww*1024+swz(k2)]*np.exp(sg*2j*np.pi*ww*k2/N) for ww in range(8)]
        X=dft(u,sg)
        for k1 in range(8): out[k2+1024*k1]=X[k1]
    return out,worst
rs=np.random.RandomState(0)
Z=rs.standard_normal(N)+1j*rs.standard_normal(N)
y,worst=fftW(Z,+1)
print(np.abs(y-np.fft.ifft(Z)*N).max(), 'worst conflict', worst)
y,worst=fftW(Z,-1)
print(np.abs(y-np.fft.fft(Z)).max())
# address formulas
l=np.arange(64)
L0=l^(l>>5)
ok=True
for r in range(16):
    Gr=(2*(r&7))|(16*((r>>2)&1))
    ok&=np.array_equal(swz(l+64*r), 64*r+(L0^Gr))
# S1: a=16l+s: g = (a>>5)&15 ^ bit8<<4 ; a>>5 = l>>1 ; bit8 = (l>>4)&1
B1=(16*l)^(((l>>1)&15) ^ (((l>>4)&1)<<4))
for s in range(16): ok&=np.array_equal(swz(16*l+s), B1^s)
# S2: a=256(l>>4)+k+16s : a>>5 = 8(l>>4) + (k+16s)>>5 = 8(l>>4)+(s>>1); &15 = (8*((l>>4)&1)) | (s>>1); bit8 = (l>>4)&1
k=l&15
for s in range(16):
    a=256*(l>>4)+k+16*s
    gg=((8*((l>>4)&1)) | (s>>1)) ^ (((l>>4)&1)<<4)
    B2=256*(l>>4)+ (k ^ (8*((l>>4)&1)) ) + 16*((l>>4)&1)   # per-lane: low4 ^ 8*bit, and bit4 ^= bit  (as add since bit4 of k part is 0... careful)
    # a' = a ^ gg ; a = 256(l>>4) + k + 16 s ; write as (per-lane part) ^ (const part)
    lanepart = 256*(l>>4) + k
    lanepart_sw = lanepart ^ ((8*((l>>4)&1)) | (((l>>4)&1)<<4))
    constpart = (16*s) ^ (s>>1)
    ok&=np.array_equal(swz(a), lanepart_sw ^ constpart)
print('formulas', ok)
