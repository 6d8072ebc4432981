import sys, time, ctypes as C, numpy as np, torch, subprocess
sys.path.insert(0,'/root/repo')
from audio_cut_amd import _native
from audio_cut_amd.separation.conv_pack import pack_conv3x3
hip=_native.Context()
lib=C.CDLL('/root/repo/scratch/exp_conv.so')
P=C.c_void_p
lib.exp_conv.argtypes=[P,P,P,P,P,C.c_int,C.c_int,C.c_int,C.c_int,C.c_int,C.c_float,C.c_int,P,C.c_int]
def bench(fn,n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t=time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.time()-t)/n*1e3
for (B,c,H,W) in [(16,48,256,3072),(16,96,128,1536)]:
    x=torch.randn(B,c,H,W,device='cuda'); w=torch.randn(c,c,3,3)/np.sqrt(9*c); b=torch.randn(c,device='cuda')
    pk,un=pack_conv3x3(w.numpy()); wp=torch.from_numpy(pk.view(np.int16)).cuda()
    out=torch.empty_like(x)
    for mode,name in [(0,'full'),(3,'no output stores'),(6,'LDS reads + MFMA only'),(7,'MFMA only')]:
        t=bench(lambda: lib.exp_conv(hip._h, x.data_ptr(), wp.data_ptr(), b.data_ptr(), out.data_ptr(), B,c,c,H,W, un, 1, None, mode))
        print(f"c{c}: {name:20s} {t:.2f} ms")
