import os, sys, ctypes; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from neural_invertible_warp_amd import _lib
_lib.LIB_PATH = os.environ.get('NIW_STAMP_LIB', 'scratch/stampbuild/libniw_hip.so')
from neural_invertible_warp_amd import ops
dev='cuda:0'
SHAPES=[(256,63),(256,256),(256,256),(256,256),(256,319),(256,256),(256,256),(257,256),(128,283),(3,128)]   # nn.Linear [out,in] of the 10 layers
g=torch.Generator().manual_seed(1)
flat=torch.cat([torch.cat([(torch.rand(o*i,generator=g)*2-1)*(6/(o+i))**0.5, (torch.rand(o,generator=g)-0.5)*0.1]) for o,i in SHAPES]).to(dev)
st=ops.FieldState(flat)
N,S=4086,192
center=torch.randn(N,3,device=dev); ray=torch.randn(N,3,device=dev); depth=torch.rand(N,S,device=dev).sort(dim=1).values*4+0.5
params=[]; off=0
for ko,ki in SHAPES:
    for shp in ((ko,ki),(ko,)):
        m=shp[0]*(shp[1] if len(shp)>1 else 1)
        params.append(flat[off:off+m].view(shp).requires_grad_(True)); off+=m
for _ in range(3):
    rgb,sig=ops.field_mlp(st,params,center,ray,depth,[1.0]*10,[1.0]*4,"softplus")
torch.cuda.synchronize()
lib=_lib.load()
cnt=8192*16
buf=(ctypes.c_ulonglong*cnt)()
lib.niw_debug_read_stamps.argtypes=[ctypes.c_void_p, ctypes.c_int]
print('rc', lib.niw_debug_read_stamps(buf, cnt))
a=np.frombuffer(buf,dtype=np.uint64).reshape(8192,16).astype(np.int64)
# stamps: 11=start, 0=after prologue, 1=L0, 2..4=L1-3, 5=L4, 6,7=L5,6, 8=L7, 9=rgb0, 10=rgb1
order=[11,0,1,2,3,4,5,6,7,8,9,10]
names=['prologue','L0','L1','L2','L3','L4','L5','L6','L7','rgb0','rgb1']
mf=[0,256,1024,1024,1024,1280,1024,1024,1152,576,64]
w=a[100:4000]   # waves in the middle of the grid
d=np.diff(w[:,order],axis=1)
med=np.median(d,axis=0)
tot=med.sum()
for n,m,c in zip(names,mf,med):
    print(f"{n:9s} cycles={c:9.0f}  mfma*64={m*64:7d}  overhead={c-m*64:8.0f}  ({100*(c-m*64)/tot:5.2f}% of total)")
print('total', tot, 'mfma total', sum(mf)*64, 'eff', sum(mf)*64/tot)
# in-layer stamps of layer 2: 2 = end of layer 1 (after advance), 12 = entry, 13 = after main loops, 14 = exit, 3 = after advance
seq=[2,12,13,14,3]
dd=np.diff(w[:,seq],axis=1); m2=np.median(dd,axis=0)
print('L2 detail: entry gap %d, main loops %d (mfma %d), exposed epilogue %d, advance %d' % (m2[0], m2[1], 1024*64, m2[2], m2[3]))
