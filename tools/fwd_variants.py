import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neural_invertible_warp_amd import ops
dev='cuda:0'
SHAPES=[(256,63),(256,256),(256,256),(256,256),(256,319),(256,256),(256,256),(257,256),(128,283),(3,128)]   # nn.Linear [out,in] of the 10 layers
g=torch.Generator().manual_seed(1)
flat=torch.cat([torch.cat([(torch.rand(o*i,generator=g)*2-1)*(6/(o+i))**0.5, (torch.rand(o,generator=g)-0.5)*0.1]) for o,i in SHAPES]).to(dev)
st=ops.FieldState(flat)
N,S=4086,192
center=torch.randn(N,3,device=dev); ray=torch.randn(N,3,device=dev); depth=torch.rand(N,S,device=dev).sort(dim=1).values*4+0.5
params=[]
off=0
for ko,ki in SHAPES:
    for shp in ((ko,ki),(ko,)):
        m=shp[0]*(shp[1] if len(shp)>1 else 1)
        params.append(flat[off:off+m].view(shp).requires_grad_(True)); off+=m
w3=[1.0]*10; wv=[1.0]*4
def run(train):
    ops.TIMING.enabled=True; ops.TIMING.reset()
    for _ in range(6):
        if train:
            rgb,sig=ops.field_mlp(st,params,center,ray,depth,w3,wv,"softplus")
        else:
            with torch.no_grad():
                rgb,sig=ops.field_mlp(st,[],center,ray,depth,w3,wv,"softplus")
    torch.cuda.synchronize()
    for k,(n,ms,u) in ops.TIMING.summary().items():
        print(k, n, f"{ms:.3f} ms", f"{u*2*527872/ms/1e9:.1f} TF")
run(True); run(False)
