import torch, torch.nn.functional as F, time, sys
torch.manual_seed(0)
B=16
def bench(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t=time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.time()-t)/n*1e3
levels=[(48,256,3072),(96,128,1536),(144,64,768),(192,32,384),(240,16,192),(288,8,96)]
for c,T,Fq in levels:
    x=torch.randn(B,c,T,Fq,device='cuda'); w=torch.randn(c,c,3,3,device='cuda')*0.05; b=torch.randn(c,device='cuda')
    fl=2*c*c*9*T*Fq*B/1e12
    t0=bench(lambda: F.conv2d(x,w,None,padding=1))
    t1=bench(lambda: F.relu_(F.conv2d(x,w,b,padding=1)))
    try:
        t2=bench(lambda: torch.ops.aten.miopen_convolution_relu(x,w,b,[1,1],[1,1],[1,1],1))
    except Exception as e:
        t2=float('nan'); print("fused err", str(e)[:100])
    xc=x.to(memory_format=torch.channels_last); wc=w.to(memory_format=torch.channels_last)
    t3=bench(lambda: F.conv2d(xc,wc,None,padding=1))
    print(f"c={c:3d} {T}x{Fq}: conv {t0:7.2f} ms ({fl/t0*1e3:6.1f} TF/s) | conv+bias+relu {t1:7.2f} | miopen fused {t2:7.2f} | channels_last conv {t3:7.2f} ({fl/t3*1e3:6.1f} TF/s)")
    # elementwise pass cost
    te=bench(lambda: x.add_(1.0))
    print(f"      elementwise in-place pass {te:6.2f} ms -> {x.numel()*8/te/1e9:6.2f} TB/s")
    # TDF linear
    wl=torch.randn(Fq//8,Fq,device='cuda')*0.02
    tl=bench(lambda: F.linear(x,wl))
    print(f"      linear F->{Fq//8}: {tl:6.2f} ms ({2*B*c*T*Fq*(Fq//8)/tl/1e9:6.1f} TF/s)")
    del x,xc
