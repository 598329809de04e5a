import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import torch
from oracle import idee_oracle as O
from idee_amd.models.classifier.CNN_3D import CNN_3D
V,H,W=6,5,40
torch.manual_seed(V*100+W)
c = CNN_3D(in_var=V, embed_dim=16, dim=16, drop_rate=0.0)
sd = {k: v.clone() for k, v in c.state_dict().items()}
N,T=2,8
x=torch.randn(N,V,16,T,H,W)
for part in ("joint","heads","both"):
    sdl={k:v.clone().requires_grad_() for k,v in sd.items()}
    xo=x.clone().requires_grad_()
    zo,yo=O.classifier(xo,sdl,pre="")
    torch.manual_seed(1)
    rz,ry=torch.randn_like(zo),torch.randn(V,*yo[0].shape)
    if part=="joint": ry=ry*0
    if part=="heads": rz=rz*0
    ((zo*rz).sum()+sum((yo[v]*ry[v]).sum() for v in range(V))).backward()
    cd=CNN_3D(in_var=V, embed_dim=16, dim=16, drop_rate=0.0); cd.load_state_dict(sd); cd=cd.cuda()
    xd=x.cuda().requires_grad_()
    z,y=cd(xd)
    ((z*rz.cuda()).sum()+sum((y[v]*ry[v].cuda()).sum() for v in range(V))).backward()
    err=(xd.grad.cpu()-xo.grad).abs()
    print(part,'max err',err.max().item(),'ref max',xo.grad.abs().max().item())
    bad=err>1e-5+1e-3*xo.grad.abs()
    idx=bad.nonzero()
    print(' nbad',len(idx))
    if len(idx):
        for d,name in enumerate("N V C T H W".split()):
            print('  ',name,torch.bincount(idx[:,d]).tolist())
    named=dict(cd.named_parameters())
    for key,name,i in cd._key_map():
        g=named[name].grad; g=g if i is None else g[i]
        e=(g.cpu()-sdl[key].grad).abs().max().item(); r=sdl[key].grad.abs().max().item()
        if e>1e-4*r+1e-6: print('   param',key,'err',e,'ref',r)
