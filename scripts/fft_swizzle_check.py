import numpy as np
N=8192; T=512
def g(u): return (u&15) ^ ((u&4)<<2)
def swz(a): return a ^ g(a>>5)
a=np.arange(N); assert sorted(swz(a).tolist())==list(range(N))
tid=np.arange(T)
# per-thread bases
gb = tid ^ g(tid>>5)
B1 = (16*tid) ^ g(tid>>1)
k2=tid&15; c=(tid>>4)&3
B2 = 128*(tid>>4) + (k2 ^ (4*c) ^ (16*(c&1)))
IMM2=lambda s: (s>>1) | 16*(s&1) | 32*(s>>1)
k3=tid&127
B3 = 1024*(tid>>7) + (k3&96) + ((k3&31) ^ (k3>>5))
IMM3=lambda s: 4*(s&3) | 16*(s&1) | 128*s
ok=True
for m in range(16): ok &= np.array_equal(gb+512*m, swz(tid+512*m))
for s in range(16): ok &= np.array_equal(B1 ^ s, swz(16*tid+s))
for h in range(2):
    i=tid+512*h
    for s in range(8):
        kk=i&15; j=(i-kk)*8+kk
        ok &= np.array_equal((B2 ^ IMM2(s)) + 4096*h, swz(j+16*s))
        kk=i&127; j=(i-kk)*8+kk
        ok &= np.array_equal((B3 ^ IMM3(s)) + 4096*h, swz(j+128*s))
print('formulas match swz:', ok)
def conflicts(addr):  # addr: [512] slots (8B each); half-wave of 32 lanes must hit distinct slot%32
    worst=1
    for hw in range(16):
        b=addr[hw*32:(hw+1)*32]%32
        worst=max(worst, np.bincount(b,minlength=32).max())
    return worst
w=1
for m in range(16): w=max(w,conflicts(gb+512*m))
print('gather worst', w)
print('sc1 worst', max(conflicts(B1^s) for s in range(16)))
print('sc2 worst', max(conflicts((B2^IMM2(s))+4096*h) for s in range(8) for h in range(2)))
print('sc3 worst', max(conflicts((B3^IMM3(s))+4096*h) for s in range(8) for h in range(2)))
