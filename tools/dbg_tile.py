import math, os, sys
sys.path.insert(0, "h-edit_amd")
import torch, torch.nn.functional as F
from hedit import _lib
lib=_lib.lib(); dev="cuda:0"; Cc=320; M=128
g=torch.Generator().manual_seed(1)
bf=lambda t: t.to(torch.bfloat16).to(dev).contiguous(); f32=lambda t: t.float().to(dev).contiguous()
a=bf(torch.randn(M,Cc,generator=g)); r1=bf(torch.randn(M,Cc,generator=g)*1.5+0.2)
gamma,beta=f32(1+0.1*torch.randn(Cc,generator=g)),f32(0.1*torch.randn(Cc,generator=g))
wo=f32(torch.randn(Cc,Cc,generator=g)/math.sqrt(Cc)); bo=f32(torch.randn(Cc,generator=g)*0.3); wq=f32(torch.randn(Cc,Cc,generator=g)/math.sqrt(Cc))
ws=torch.empty(lib.hedit_k_lin_tile_stream_bytes(1),dtype=torch.uint8,device=dev)
_lib.check(lib.hedit_k_lin_tile_pack(_lib.ptr(wo),_lib.ptr(wq),None,None,1.0,_lib.ptr(ws),None))
mid=torch.zeros(M,Cc,dtype=torch.bfloat16,device=dev); q=torch.zeros(M,Cc,dtype=torch.bfloat16,device=dev)
_lib.check(lib.hedit_k_lin_tile(_lib.ptr(a),Cc,_lib.ptr(r1),Cc,None,0,_lib.ptr(bo),_lib.ptr(gamma),_lib.ptr(beta),1e-5,_lib.ptr(ws),_lib.ptr(mid),Cc,None,0,None,0,_lib.ptr(q),Cc,M,Cc,None))
torch.cuda.synchronize()
bfr=lambda t: t.to(torch.bfloat16).float()
t1=a.float()@bfr(wo).t()+bo+r1.float()
tn=bfr(F.layer_norm(t1,(Cc,),gamma,beta,1e-5))
want=tn@bfr(wq).t()
d=(q.float()-want)
print("mid err", float((mid.float()-t1).norm()/t1.norm()), "q err", float(d.norm()/want.norm()))
print("per 16-col block err:", [round(float(d[:,c:c+16].norm()/want[:,c:c+16].norm()),3) for c in range(0,320,16)])
print("per 16-row block err:", [round(float(d[r:r+16].norm()/want[r:r+16].norm()),3) for r in range(0,128,16)])
# which k-steps are missing?  project the error onto contributions of k-step blocks
for ks in range(10):
    part=tn[:,ks*32:ks*32+32]@bfr(wq)[:,ks*32:ks*32+32].t()
    print(ks, "corr of error with -contribution:", round(float((d*(-part)).sum()/(part.norm()**2)),3), " with contribution of LN-less input:", end=" ")
    part2=bfr(t1)[:,ks*32:ks*32+32]@bfr(wq)[:,ks*32:ks*32+32].t()
    print(round(float((d*part2).sum()/(part2.norm()**2)),3))
