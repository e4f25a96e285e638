"""tools/exp_f43_numerics.py -- CPU study (numpy, no GPU): would Winograd F(4x4,3x3) be admissible in the f32x3 arithmetic?  One 128 -> 128 channel 3x3 layer on
post-ReLU data against a float64 convolution: float32 input / output transforms, operands rounded to `bits` bits under the block scales the kernels use
(per tile; per position and output channel), products and sums exact-ish (float64, rounded once).  Round 6 result: F(2x2,3x3) at 22 bits
rms 5.0e-7 of max|y|; F(4x4,3x3) at 22 bits 9.6e-6 (19x; worst element 60x) -- its transforms amplify the operand rounding by the 1/24 ... 8 spread of their
constants; with (near) exact operands it is 1.1e-7, i.e. it would need a third fp16 term per operand (six MFMAs per product instead of three), which is
more matrix work than the 2.25 -> 4 multiplication saving returns.  Not a candidate under the held-out admission criterion (K <= 1.4 of the reference's own distance)."""
import numpy as np, torch, torch.nn.functional as F
torch.manual_seed(0)
def wino_mats(m):
    if m==2:
        BT=np.array([[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]],float)
        G=np.array([[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]],float)
        AT=np.array([[1,1,1,0],[0,1,-1,-1]],float)
    else:
        BT=np.array([[4,0,-5,0,1,0],[0,-4,-4,1,1,0],[0,4,-4,-1,1,0],[0,-2,-1,2,1,0],[0,2,-1,-2,1,0],[0,4,0,-5,0,1]],float)
        G=np.array([[1/4,0,0],[-1/6,-1/6,-1/6],[-1/6,1/6,-1/6],[1/24,1/12,1/6],[1/24,-1/12,1/6],[0,0,1]],float)
        AT=np.array([[1,1,1,1,1,0],[0,1,-1,2,-2,0],[0,1,1,4,4,0],[0,1,-1,8,-8,1]],float)
    return BT,G,AT
def round_bits(x, bits, axis_scale):
    # keep `bits` bits relative to the block maximum `axis_scale` (same shape broadcastable): block-scaled two-term fp16 split ~ 22 bits
    e=np.floor(np.log2(np.maximum(axis_scale,1e-300)))
    q=2.0**(e-bits+1)
    return np.round(x/q)*q
def wino_conv(x,w,m,bits=22):
    # x [C,H,W] float32, w [K,C,3,3]; returns y [K,H,W]; transforms in float32, operands block-rounded, products+accumulation float32-ish (f64 acc then f32)
    BT,G,AT=wino_mats(m); a=m+2
    C,H,W=x.shape; K=w.shape[0]
    xp=np.pad(x,((0,0),(1,1+m),(1,1+m))).astype(np.float32)
    th,tw=(H+m-1)//m,(W+m-1)//m
    U=np.einsum('ai,kcij,bj->abkc',G,w.astype(np.float64),G)             # filter transform in f64, then rounded
    U=round_bits(U,bits,np.abs(U).max(axis=3,keepdims=True))                # per (position, k) scale
    y=np.zeros((K,th*m,tw*m),np.float32)
    BT32=BT.astype(np.float32)
    for ty in range(th):
        d=np.stack([xp[:,ty*m:ty*m+a,tx*m:tx*m+a] for tx in range(tw)],0)  # [tw,C,a,a]
        V=np.einsum('ai,tcij->tcaj',BT32,d).astype(np.float32)
        V=np.einsum('tcaj,bj->tcab',V,BT32).astype(np.float32)
        tmax=np.abs(d).max(axis=(1,2,3),keepdims=True)*(4.0 if m==2 else 100.0)    # tile scale bound
        Vr=round_bits(V.astype(np.float64),bits,np.broadcast_to(tmax,V.shape))
        M=np.einsum('abkc,tcab->tkab',U,Vr).astype(np.float32)
        Y=np.einsum('ia,tkab->tkib',AT.astype(np.float32),M).astype(np.float32)
        Y=np.einsum('tkib,jb->tkij',Y,AT.astype(np.float32)).astype(np.float32)
        for tx in range(tw): y[:,ty*m:ty*m+m,tx*m:tx*m+m]=Y[tx]
    return y[:,:H,:W]
C,K,H,W=128,128,48,48
x=torch.relu(torch.randn(C,H,W))*torch.rand(C,1,1)*3
w=torch.randn(K,C,3,3)*np.sqrt(2/(9*C))
truth=F.conv2d(x.double()[None],w.double(),padding=1)[0].numpy()
d32=F.conv2d(x[None],w,padding=1)[0].numpy()
sc=np.abs(truth).max()
print('direct f32 (torch CPU)      max err / max|y| %.3e  rms %.3e'%(np.abs(d32-truth).max()/sc, np.sqrt(((d32-truth)**2).mean())/sc))
for m in (2,4):
    for bits in (22,40):
        y=wino_conv(x.numpy(),w.numpy(),m,bits)
        print('F(%dx%d,3x3) operands %d bits  max err / max|y| %.3e  rms %.3e'%(m,m,bits,np.abs(y-truth).max()/sc, np.sqrt(((y-truth)**2).mean())/sc))
