import torch, torch.nn.functional as F, numpy as np
torch.manual_seed(0)
def rel(a,b): return float((a.double().cpu()-b).abs().max()/b.abs().max())
print("allow_tf32 matmul", torch.backends.cuda.matmul.allow_tf32, "cudnn", torch.backends.cudnn.allow_tf32, "prec", torch.get_float32_matmul_precision())
for c,(T,Fq) in [(48,(64,3072)),(96,(64,1536)),(288,(8,96))]:
    x=torch.randn(1,c,T,Fq); w=torch.randn(c,c,3,3)/np.sqrt(9*c); b=torch.randn(c)*0.1
    ref=F.conv2d(x.double(),w.double(),b.double(),padding=1)
    y=F.conv2d(x.cuda(),w.cuda(),b.cuda(),padding=1)
    ycl=F.conv2d(x.cuda().to(memory_format=torch.channels_last),w.cuda().to(memory_format=torch.channels_last),b.cuda(),padding=1)
    print("conv3x3 c",c, rel(y,ref), "channels_last", rel(ycl,ref), "cpu f32", rel(F.conv2d(x,w,b,padding=1),ref))
    # linear on last axis
    wl=torch.randn(Fq//8,Fq)/np.sqrt(Fq)
    refl=F.linear(x.double(),wl.double())
    print("linear f",Fq, rel(F.linear(x.cuda(),wl.cuda()),refl), "cpu f32", rel(F.linear(x,wl),refl))
    wd=torch.randn(c+48,c,2,2)/np.sqrt(4*c)
    print("ds conv", rel(F.conv2d(x.cuda(),wd.cuda(),stride=2), F.conv2d(x.double(),wd.double(),stride=2)))
    wu=torch.randn(c,max(48,c-48),2,2)/np.sqrt(c)
    print("us convT", rel(F.conv_transpose2d(x.cuda(),wu.cuda(),stride=2), F.conv_transpose2d(x.double(),wu.double(),stride=2)))
x=torch.randn(1,4,64,3072); w1=torch.randn(48,4,1,1)
print("1x1", rel(F.conv2d(x.cuda(),w1.cuda()), F.conv2d(x.double(),w1.double())))
