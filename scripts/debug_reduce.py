import sys, numpy as np
sys.path.insert(0,'.')
from edge_alignment_amd import capi
rng=np.random.default_rng(0)
V=np.floor(rng.random((32,64))*1000).astype(np.float32)   # integers: exact in fp32
o32,o64,st=capi.selftest_wave_reduce(V)
tot=V.astype(np.float64).sum(axis=1)
print('f64 ok', np.array_equal(o64,tot), 'f32 ok', np.array_equal(o32,tot))
lanes=np.arange(64)
def level(lo,hi,x,maskbit):
    t=lo+lo[lanes^x]; t2=hi+hi[lanes^x]
    return np.where((lanes&maskbit)!=0,t2,t)
a=np.array([level(V[i],V[i+16],15,8) for i in range(16)])
b=np.array([level(a[i],a[i+8],7,4) for i in range(8)])
print('stage a ok', np.array_equal(st[:16],a), 'stage b ok', np.array_equal(st[16:24],b))
if not np.array_equal(st[:16],a):
    bad=np.argwhere(st[:16]!=a); print('a mismatches', len(bad), bad[:10]); i,l=bad[0]; print(st[i,l], a[i,l], V[i,l], V[i,l^15], V[i+16,l], V[i+16,l^15])
def swap16(lo,hi):
    row=lanes//16
    r0=np.where(row%2==0, lo, hi[lanes-16*(row%2)]); r1=np.where(row%2==0, lo[(lanes+16)%64], hi); return r0+r1
def swap32(lo,hi):
    up=lanes>=32
    r0=np.where(up, hi[lanes-32*up], lo); r1=np.where(up, hi, lo[(lanes+32)%64]); return r0+r1
bb=st[16:24]
c=np.array([swap16(bb[i],bb[i+4]) for i in range(4)])
print('stage c ok (given b)', np.array_equal(st[24:28],c))
if not np.array_equal(st[24:28],c):
    bad=np.argwhere(st[24:28]!=c); print('c mismatches', len(bad), bad[:8]); i,l=bad[0]; print('lane',l,'got',st[24+i,l],'exp',c[i,l],'b lo',bb[i,l],bb[i,(l+16)%64],bb[i,(l-16)%64],'b hi',bb[i+4,l],bb[i+4,(l+16)%64],bb[i+4,(l-16)%64])
cc=st[24:28]
d=np.array([swap32(cc[i],cc[i+2]) for i in range(2)])
print('stage d ok (given c)', np.array_equal(st[28:30],d))
if not np.array_equal(st[28:30],d):
    bad=np.argwhere(st[28:30]!=d); print('d mismatches', len(bad), bad[:8]); i,l=bad[0]; print('lane',l,'got',st[28+i,l],'exp',d[i,l],'c lo',cc[i,l],cc[i,(l+32)%64],'c hi',cc[i+2,l],cc[i+2,(l+32)%64])
