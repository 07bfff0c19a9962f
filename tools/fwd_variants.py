import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neural_invertible_warp_amd import ops
from oracle import niw_oracle as O
dev='cuda:0'
p=O.make_nerf_params(1)
flat=torch.cat([p[f"{n}.{k}"].reshape(-1) for n,_,_ in O.nerf_layer_shapes() for k in ("weight","bias")]).to(dev)
st=ops.FieldState(flat)
N,S=4086,192
center=torch.randn(N,3,device=dev); ray=torch.randn(N,3,device=dev); depth=torch.rand(N,S,device=dev).sort(dim=1).values*4+0.5
params=[]
off=0
for n,ko,ki in O.nerf_layer_shapes():
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
