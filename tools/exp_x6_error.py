import sys; sys.path.insert(0,'.')
import torch, numpy as np
from fasterrcnn_amd.models import vgg16 as V
for (M,N,K,relu,clampa) in [(300,256,512,False,True),(300,256,512,False,False),(300,256,512,True,True),(300,256,2048,False,True),(300,1024,256,False,True),(137,256,512,False,True)]:
    gen = torch.Generator().manual_seed(M+N+K)
    a = torch.randn((M,K),generator=gen)
    if clampa: a=a.clamp(min=0)
    w = torch.randn((N,K),generator=gen)*(2.0/K)**0.5
    b = torch.randn((N,),generator=gen)*0.1
    ref = a.double()@w.double().t()+b.double()
    if relu: ref=ref.clamp(min=0)
    ad,wd,bd=a.cuda(),w.cuda(),b.cuda()
    scale=float(ref.abs().max())
    y6t=V.linear_x6t(ad,V.split_rows_x6t(wd,(N+255)//256*256),bd,N,relu)
    npad=(N+127)//128*128
    y6=V.linear_x6(V.split_rows_x6(ad),V.split_rows_x6(wd,rows_out=npad),bd,M,N,K,relu,want="float32")
    wpad=torch.zeros((npad,K),device='cuda'); wpad[:N]=wd
    y32=V.linear(ad,wpad,bd,N,relu)
    e=lambda y: float((y.cpu().double()-ref).abs().max())/scale
    r=lambda y: float(((y.cpu().double()-ref)**2).mean().sqrt())/scale
    print(M,N,K,relu,clampa,"x6t %.3g (rms %.3g)  x6_v1 %.3g (rms %.3g)  f32 %.3g (rms %.3g)"%(e(y6t),r(y6t),e(y6),r(y6),e(y32),r(y32)))
