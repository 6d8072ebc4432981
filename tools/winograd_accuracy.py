"""Element-wise accuracy of a Winograd F(2x2, 3x3) evaluation of the U-Net's 3x3 convs (float32, CPU) against the direct float32 conv,
both measured against float64 relative to each element's own sum |x||w| - the bar of tests/test_unet_gpu.py.  Evidence for DESIGN.md 8(a):
on uniform, decaying and 28-decade-cliff inputs the Winograd form is within 1-2x of the direct one (its transforms only use +-1 and
+-1/2), so halving the MFMA work this way would not by itself cost the float32-class property.   usage: python tools/winograd_accuracy.py"""
import torch, numpy as np, torch.nn.functional as F
torch.manual_seed(0)
def wino_conv(x, w, dtype):
    # x [B,C,H,W], w [Co,C,3,3]; F(2x2,3x3), padding 1, H,W even
    x=x.to(dtype); w=w.to(dtype)
    Bt=torch.tensor([[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]],dtype=dtype)
    G=torch.tensor([[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]],dtype=dtype)
    At=torch.tensor([[1,1,1,0],[0,1,-1,-1]],dtype=dtype)
    U=torch.einsum('ij,ocjk,lk->ocil',G,w,G)          # [Co,C,4,4]
    xp=F.pad(x,(1,1,1,1))
    B,C,H,W=x.shape
    tiles=xp.unfold(2,4,2).unfold(3,4,2)              # [B,C,H/2,W/2,4,4]
    V=torch.einsum('ij,bcyxjk,lk->bcyxil',Bt,tiles,Bt)
    M=torch.einsum('ocil,bcyxil->boyxil',U,V)
    Y=torch.einsum('ij,boyxjk,lk->boyxil',At,M,At)    # [B,Co,H/2,W/2,2,2]
    return Y.permute(0,1,2,4,3,5).reshape(B,-1,H,W)
def ee(y,ref,sc): return float(((y.double()-ref).abs()/sc.clamp_min(1e-300)).max()), float(((y.double()-ref).abs()/sc.clamp_min(1e-300)).pow(2).mean().sqrt())
for C in (48,144):
    x=torch.randn(2,C,64,32)*2; w=torch.randn(C,C,3,3)/np.sqrt(9*C)
    for case in ('uniform','decay','cliff'):
        xm=x.clone()
        if case=='decay': xm[1:]*=torch.logspace(0,-8,64).view(1,1,-1,1)
        if case=='cliff':
            r=torch.zeros(64,dtype=torch.float64); r[:5]=1; r[5:19]=1e-2**torch.arange(1,15,dtype=torch.float64); xm[1:]*=r.float().view(1,1,-1,1)
        ref=F.conv2d(xm.double(),w.double(),padding=1); sc=F.conv2d(xm.abs().double(),w.abs().double(),padding=1)
        d32=F.conv2d(xm,w,padding=1); w32=wino_conv(xm,w,torch.float32); w64=wino_conv(xm,w,torch.float64)
        print(C,case,'direct f32 max/rms',ee(d32,ref,sc),'winograd f32',ee(w32,ref,sc),'winograd f64 sanity',ee(w64,ref,sc)[0])
