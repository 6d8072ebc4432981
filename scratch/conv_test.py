import sys, time, numpy as np, torch, torch.nn.functional as F
sys.path.insert(0,'/root/repo')
from audio_cut_amd import _native
from audio_cut_amd.separation.conv_pack import pack_conv3x3
hip=_native.Context()
torch.manual_seed(0)
def bench(fn,n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t=time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.time()-t)/n*1e3
for (B,c,H,W) in [(2,48,16,64),(16,48,256,3072),(16,96,128,1536),(16,144,64,768),(16,192,32,384),(16,288,8,96)]:
    x=torch.randn(B,c,H,W,device='cuda')*2; w=torch.randn(c,c,3,3,device='cuda')/np.sqrt(9*c); b=torch.randn(c,device='cuda')*0.1
    pk_,un=pack_conv3x3(w.cpu().numpy()); wp=torch.from_numpy(pk_.view(np.int16)).cuda()
    y=hip.conv3x3_f16x3(x,wp,b,c,un,relu=True)
    ref=F.relu(F.conv2d(x.double(),w.double(),b.double(),padding=1))
    ref32=F.relu(F.conv2d(x,w,b,padding=1))
    pk=float(ref.abs().max())
    e=float((y.double()-ref).abs().max())/pk; e32=float((ref32.double()-ref).abs().max())/pk
    fl=2*c*c*9*H*W*B/1e12
    t=bench(lambda: hip.conv3x3_f16x3(x,wp,b,c,un,relu=True)); tm=bench(lambda: F.conv2d(x,w,None,padding=1))
    print(f"B{B} c{c} {H}x{W}: err f16x3 {e:.2e} (miopen f32 {e32:.2e}) | mine {t:.2f} ms {fl/t*1e3:.1f} TF/s | miopen {tm:.2f} ms {fl/tm*1e3:.1f} TF/s")
